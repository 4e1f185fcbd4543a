"""Data formats either side of the path: the variational-wavefunction file of perform_hci."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLANG = "/opt/rocm/lib/llvm/bin/flang"


def test_wf_filename_matches_fortran_edit_descriptor():
    from sqmc_amd import host as H
    # hci.f90:195-197: write(fmt,'(es7.2e1)') eps_var
    assert H.wf_filename(1e-4) == "wf_eps_var=1.00E-4" and H.wf_filename(2e-3) == "wf_eps_var=2.00E-3"
    assert H.wf_filename(5e-5) == "wf_eps_var=5.00E-5"


def test_wf_file_roundtrip_through_fortran_io(tmp_path):
    """write_wf_var -> a Fortran program using the read/write statements of hci.f90:203-214 /
    606-612 (integer(16) determinants, form='unformatted') -> read_wf_var: byte-identical files,
    and the Fortran side sees the same numbers."""
    from sqmc_amd import host as H
    exe = os.path.join(ROOT, "sqmc_amd", "fortran", "wf_io_check")
    if not os.path.exists(exe):
        if not os.path.exists(FLANG):
            pytest.skip("flang absent")
        subprocess.check_call([FLANG, "-O2", os.path.join(ROOT, "sqmc_amd", "fortran", "wf_io_check.f90"), "-o", exe])
    rs = np.random.RandomState(3)
    n = 2500
    up, dn = rs.randint(1, 2**26, n).astype(np.uint64), rs.randint(1, 2**26, n).astype(np.uint64)
    wts, e = rs.randn(n, 2), np.array([-75.719473642, -75.631097209])
    a, b = str(tmp_path / H.wf_filename(1e-3)), str(tmp_path / "back")
    H.write_wf_var(a, up, dn, wts, e)
    out = subprocess.run([exe, a, b, "2"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    f = out.stdout.split()
    assert int(f[1]) == n and int(f[2]) == int((up.astype(object) % 1000003).sum()) and int(f[3]) == int((dn.astype(object) % 1000003).sum())
    assert abs(float(f[4]) - wts[:, 0].sum()) < 1e-9 and abs(float(f[5]) - (wts[:, 1] ** 2).sum()) < 1e-9 and float(f[6]) == e[1]
    assert open(a, "rb").read() == open(b, "rb").read()
    u2, d2, w2, e2 = H.read_wf_var(b, 2)
    assert np.array_equal(u2, up) and np.array_equal(d2, dn) and np.array_equal(w2, wts) and np.array_equal(e2, e)


def test_heg_madelung_energy_matches_reference_output():
    """src/e2e_tests/heg/o_det_ref:75-76: 'Madelung energy =-10.224153', 'HF energy including Madelung
    = 48.36852150' for 14 electrons at r_s = 0.5 (HF energy 58.592675, :226)."""
    from sqmc_amd import host as H
    h = H.HegHost(3, 0.5, 14, 7, 1.49)
    m = h.madelung_energy()
    assert abs(m - (-10.224153)) < 1e-6 and abs(58.592674968 + m - 48.36852150) < 1e-7


def test_heg_k_point_table_matches_reference_output():
    """src/e2e_tests/heg/o_det_ref:53-72 prints the 19 plane waves inside cutoff 1.49 in the order the
    reference's shell sort leaves them (which fixes every determinant label of the HEG runs); unit
    2 pi / L = 3.2345 for 14 electrons at r_s = 0.5."""
    from sqmc_amd import host as H
    h = H.HegHost(3, 0.5, 14, 7, 1.49)
    ref = [(0, 0, 0), (0, 0, 1), (0, 0, -1), (0, 1, 0), (-1, 0, 0), (0, -1, 0), (1, 0, 0), (1, -1, 0), (1, 0, -1), (0, 1, -1),
           (-1, 0, 1), (0, -1, 1), (-1, 1, 0), (1, 0, 1), (-1, -1, 0), (1, 1, 0), (-1, 0, -1), (0, 1, 1), (0, -1, -1)]
    unit = 2 * np.pi / h.length_cell
    assert abs(unit - 3.2345) < 5e-5 and h.norb == 19
    assert [tuple(int(x) for x in row) for row in np.round(np.asarray(h.k_vectors) / unit)] == ref
    assert (h.hf_up, h.hf_dn) == (127, 127)                                  # 'HF det= 127 127' (:73)
