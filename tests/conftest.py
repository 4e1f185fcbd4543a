import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
FCIDUMP = os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU validation, excluded from the default CPU run")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def c2_walk(oracle):
    """C2 cc-pVDZ r=1.24253, walk deck conventions (time_sym=f, HF = first 4 orbitals)."""
    return oracle.ChemSystem(FCIDUMP, 8, 4, "d2h", time_sym=False, hf_mode=0)


@pytest.fixture(scope="session")
def c2_hci(oracle):
    """Same molecule with the shipped HCI deck conventions (time_sym=t, z=1, hf_symmetry=1)."""
    return oracle.ChemSystem(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_mode=1, hf_symmetry=1)


@pytest.fixture(scope="session")
def c2_setup(oracle, c2_walk):
    return oracle.setup_walk(c2_walk, 100, 1000, 0.1)


def gpu_ctx_from_oracle(sysm, **kw):
    """Hands the oracle's tables to the HIP library exactly as a Fortran host would."""
    import sqmc_amd
    return sqmc_amd.GpuChem(sysm.norb, sysm.nup, sysm.ndn, sysm.orbsym(), sysm.prod().reshape(-1), sysm.combine_2().reshape(-1),
                            sysm.integrals(), n_group=sysm.s.n_group, time_sym=bool(sysm.s.time_sym), z=sysm.s.z, **kw)


@pytest.fixture(scope="session")
def c2_setup_ts(oracle, c2_hci):
    """walk set-up with time-reversal symmetry (representatives up <= dn), as the shipped decks use it"""
    return oracle.setup_walk(c2_hci, 100, 1000, 0.1)


@pytest.fixture(scope="session")
def heg14(oracle):
    """14 electrons (7 up, 7 dn), r_s = 0.5, cutoff 1.49: the reference's e2e HEG system (19 plane waves)"""
    return oracle.HegSystem(3, 0.5, 14, 7, 1.49)


@pytest.fixture(scope="session")
def heg_setup(oracle, heg14):
    return oracle.setup_walk_heg(heg14, 250, 0.1)


@pytest.fixture(scope="session")
def heg57(oracle):
    """BASELINE.json configs[3]: 14 electrons (7 up, 7 dn), r_s = 1.0, cutoff 2.3 -> 57 plane waves.  Sort keys need 56
    bits, so the library keeps keys and walker indices in two arrays (`pack == 0`) -- no other test system does."""
    return oracle.HegSystem(3, 1.0, 14, 7, 2.3)


@pytest.fixture(scope="session")
def heg57_setup(oracle, heg57):
    return oracle.setup_walk_heg(heg57, 300, 0.1)


@pytest.fixture(scope="session")
def heg14_hci(oracle, heg14):
    """variational stage of the reference's e2e HEG deck (eps_var 1e-3, one state) in the oracle: ~60 s, shared"""
    return oracle.hci_variational(heg14, 1e-3, n_states=1)


@pytest.fixture(scope="session")
def hub44(oracle):
    """BASELINE.json configs[0]: 4x4 Hubbard, U/t = 4, half filling, periodic"""
    return oracle.HubbardSystem(4, 4, True, 8, 8, 1.0, 4.0)


@pytest.fixture(scope="session")
def hub_setup(oracle, hub44):
    return oracle.setup_walk_hubbard(hub44, 500, 0.5, 20)


def gpu_ctx_hub(hsys, **kw):
    import sqmc_amd
    return sqmc_amd.GpuChem.hubbard(hsys.l_x, hsys.l_y, hsys.pbc, hsys.nup, hsys.ndn, hsys.t, hsys.U, **kw)


def gpu_ctx_heg(hsys, **kw):
    import sqmc_amd
    return sqmc_amd.GpuChem.heg(hsys.n_dim, hsys.norb, hsys.nup, hsys.ndn, hsys.length_cell, hsys.k_vectors(), **kw)
