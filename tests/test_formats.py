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


def test_walk_deck_grammar_and_estimators():
    """run_type `none` decks (read_input, do_walk.f90:222-404 + read_chem, chemistry.f90:119-245): the
    fixture deck parses to the parameters of BASELINE.md's walk smoke test; decks outside the GPU path
    stop with a message; the generation / block estimators (do_walk.f90:2792-2843, 2986-3040) reduce
    to the textbook ratio-of-means error for uncorrelated data."""
    import os
    import numpy as np
    import pytest
    from sqmc_amd.walk_run import parse_walk_deck, WalkStats
    text = open(os.path.join(os.path.dirname(__file__), "golden", "C2_r1.24253_i_walk")).read()
    d = parse_walk_deck(text)
    assert (d["nstep"], d["nblk"], d["nblk_eq"], d["ipr"]) == (100, 4, 2, 0)
    assert (d["w_abs_gen_begin"], d["w_abs_gen_target"], d["mwalk"]) == (100, 10000, 0)
    assert d["proposal_method"] == "uniform2" and d["semistochastic"] and d["size_deterministic"] == 1000
    assert d["irand_seed"][1] == [1346, 5634, 6635, 4361] and d["n_truncate_trial_wf"] == [100] and len(d["orbital_symmetries"]) == 26
    assert parse_walk_deck(text.replace("uniform2 0", "fast_heatbath 0"))["proposal_method"] == "fast_heatbath"      # the library builds its tables (test_library_builds_the_heatbath_tables_itself)
    with pytest.raises(SystemExit, match="proposal_method"):
        parse_walk_deck(text.replace("uniform2 0", "heat_bath 0"))
    assert parse_walk_deck(text.replace("f f 0.5 ", "t f 0.5 "))["hf_to_psit"]            # the transformed projector is on the GPU path (tests/test_gpu_psit.py)
    with pytest.raises(SystemExit, match="run_type"):
        parse_walk_deck(text.replace("none  ", "vmc   "))
    rs = np.random.RandomState(7)
    st = WalkStats()
    num = -75.7 * 100 * (1 + 0.01 * rs.randn(4000)); den = 100 * (1 + 0.02 * rs.randn(4000))
    for b in range(40):
        for i in range(100):
            st.generation(num[b * 100 + i], den[b * 100 + i])
        st.block(num[b * 100:(b + 1) * 100].sum(), den[b * 100:(b + 1) * 100].sum())
    expect = 75.7 * np.sqrt(0.01 ** 2 + 0.02 ** 2) / np.sqrt(4000)
    assert abs(st.e_genabs_err / expect - 1) < 0.05 and abs(st.e_blkabs_err / expect - 1) < 0.35
    assert abs(st.e_genabs_ave + 75.7) < 4 * expect and abs(st.e_blkabs_ave + 75.7) < 4 * expect
    assert 0.4 < st.t_corr < 2.5                       # uncorrelated generations: block and generation errors agree


def test_walk_input_files_roundtrip_through_fortran_io(tmp_path):
    """psit_connections / deterministic-matrix-element files: written by
    sqmc_amd.host, read by a Fortran program with the reference's read statements
    (do_walk.f90:702-741, 898-940: list-directed) and rewritten with the reference's write
    statements (semistoch.f90:86-126, do_walk.f90:970-1010); the rewritten files read back to the
    same data, and the psit file is byte-identical (it is fully format-controlled)."""
    from sqmc_amd import host as H
    exe = os.path.join(ROOT, "sqmc_amd", "fortran", "walk_io_check")
    if not os.path.exists(exe):
        if not os.path.exists(FLANG):
            pytest.skip("flang absent")
        subprocess.check_call([FLANG, "-O2", os.path.join(ROOT, "sqmc_amd", "fortran", "walk_io_check.f90"), "-o", exe])
    rs = np.random.RandomState(5)
    nup = ndn = 4; norb = 26; n = 3000

    def dets(k):
        out = set()
        while len(out) < k:
            out.add((sum(1 << int(o) for o in rs.choice(norb, nup, replace=False)), sum(1 << int(o) for o in rs.choice(norb, ndn, replace=False))))
        a = sorted(out)
        return np.array([x[0] for x in a], np.uint64), np.array([x[1] for x in a], np.uint64)
    cu, cd = dets(n)
    num = rs.randn(n) * 10 ** rs.uniform(-6, 2, n); den = np.where(rs.rand(n) < 0.05, rs.randn(n), 0.0)
    num[0], den[0] = -75.5 * 0.9, 0.9
    num[7] = 1e-12                                            # below the 1e-10 print threshold: dropped
    a, b = str(tmp_path / "psit_connections.out"), str(tmp_path / "psit_back")
    H.write_psit_connections(a, cu[:40], cu, cd, num, den, nup, ndn, norb)
    out = subprocess.run([exe, "psit", a, b, str(nup), str(ndn), str(norb)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr + out.stdout
    assert open(a).read() == open(b).read()
    ru, rd, rn, rden = H.read_psit_connections(b, nup, ndn)
    keep = np.abs(num) > 1e-10
    assert keep.sum() == n - 1 and np.array_equal(ru, cu[keep]) and np.array_equal(rd, cd[keep])
    assert np.allclose(rn, num[keep], rtol=3e-16, atol=6e-16) and np.allclose(rden, den[keep], rtol=3e-16, atol=6e-16)   # f22.15 / f19.15: 15 decimals
    # deterministic space: 500 determinants, random upper-triangular rows
    iu, idn = dets(500)
    counts = rs.randint(1, 30, 500); counts = np.minimum(counts, np.arange(1, 501))
    idx = np.concatenate([[i + 1] + sorted(rs.choice(i, c - 1, replace=False) + 1) if c > 1 else [i + 1] for i, c in enumerate(counts)]).astype(np.int64)
    val = rs.randn(len(idx)) * 10 ** rs.uniform(-8, 1, len(idx))
    a, b = str(tmp_path / "dtm_projector.out"), str(tmp_path / "dtm_back")
    H.write_dtm_elems(a, iu, idn, counts, idx, val, -75.66376534)
    out = subprocess.run([exe, "dtm", a, b, str(nup), str(ndn)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr + out.stdout
    for path in (a, b):                                      # our file, and the one Fortran wrote with the reference's statements
        u2, d2, c2, i2, v2, e2 = H.read_dtm_elems(path, nup, ndn)
        assert np.array_equal(u2, iu) and np.array_equal(d2, idn) and np.array_equal(c2, counts) and np.array_equal(i2, idx)
        assert np.array_equal(v2, val) and e2 == -75.66376534
