"""ctypes door to the CPU ORACLE (oracle/liboracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The host-side drivers here (HCI iteration, walk population control) restate the
reference's scalar logic: hci.f90:359-520, 865-1040 and do_walk.f90:2171-2184, 2880-2923.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_chem_load.restype = C.c_void_p
        L.orc_chem_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_chem_free.argtypes = [C.c_void_p]
        L.orc_chem_setup_hb.argtypes = [C.c_void_p]
        L.orc_hamiltonian_chem.restype = C.c_double
        L.orc_hamiltonian_chem.argtypes = [C.c_void_p] + [C.c_uint64] * 4 + [C.c_int]
        L.orc_hamiltonian.restype = C.c_double
        L.orc_hamiltonian.argtypes = [C.c_void_p] + [C.c_uint64] * 4
        L.orc_hamiltonian_chem_time_sym.restype = C.c_double
        L.orc_hamiltonian_chem_time_sym.argtypes = [C.c_void_p] + [C.c_uint64] * 4
        L.orc_excitation_level.argtypes = [C.c_uint64] * 4
        L.orc_rannyu.restype = C.c_double
        L.orc_find_connected_dets_chem.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_find_important_connected_dets_chem.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_spmv_sym_upper.argtypes = [C.c_int64] + [C.c_void_p] * 5
        L.orc_build_sparse_ham.restype = C.c_int64
        L.orc_build_sparse_ham.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_off_diagonal_move_chem.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_walk_new.restype = C.c_void_p
        L.orc_walk_new.argtypes = [C.c_int64]
        L.orc_walk_free.argtypes = [C.c_void_p]
        L.orc_walk_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_merge_sort_walkers.argtypes = [C.c_void_p, C.c_int64]
        L.orc_merge_original_with_spawned2.restype = C.c_int64
        L.orc_merge_original_with_spawned2.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_reduce_my_walker.restype = C.c_int64
        L.orc_reduce_my_walker.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_integral_index.restype = C.c_int64
        L.orc_integral_index.argtypes = [C.c_void_p] + [C.c_int] * 4
        _LIB = L
    return _LIB


class Rng(C.Structure):
    _fields_ = [("l", C.c_int * 4)]


ORC_MAXORB, ORC_MAXSYM = 64, 8


class Chem(C.Structure):
    _fields_ = [
        ("norb", C.c_int), ("nelec", C.c_int), ("nup", C.c_int), ("ndn", C.c_int), ("n_core_orb", C.c_int),
        ("time_sym", C.c_int), ("z", C.c_int), ("n_group", C.c_int),
        ("prod", (C.c_int * (ORC_MAXSYM + 1)) * (ORC_MAXSYM + 1)),
        ("orbsym", C.c_int * (ORC_MAXORB + 1)),
        ("orb_order", C.c_int * (ORC_MAXORB + 2)),
        ("orb_order_inv", C.c_int * (ORC_MAXORB + 2)),
        ("combine_2", (C.c_int * (ORC_MAXORB + 2)) * (ORC_MAXORB + 2)),
        ("n_int", C.c_int64), ("integrals", C.POINTER(C.c_double)), ("nuclear", C.c_double),
        ("orbital_energies", C.c_double * (ORC_MAXORB + 1)),
        ("hf_up", C.c_uint64), ("hf_dn", C.c_uint64),
        ("num_orb_by_sym", C.c_int * (ORC_MAXSYM + 1)),
        ("which_orb_by_sym", (C.c_int * (ORC_MAXORB + 1)) * (ORC_MAXSYM + 1)),
        ("n_hb", C.c_int64), ("hb_r", C.POINTER(C.c_int)), ("hb_s", C.POINTER(C.c_int)), ("hb_absH", C.POINTER(C.c_double)),
        ("pq_ind", C.POINTER(C.c_int64)), ("pq_count", C.POINTER(C.c_int)), ("n_pq", C.c_int),
        ("max_double", C.c_double),
    ]


class Walk(C.Structure):
    _fields_ = [
        ("nwalk", C.c_int64), ("mwalk", C.c_int64),
        ("up", C.POINTER(C.c_uint64)), ("dn", C.POINTER(C.c_uint64)), ("wt", C.POINTER(C.c_double)),
        ("imp_distance", C.POINTER(C.c_int8)), ("initiator", C.POINTER(C.c_int8)),
        ("matrix_elements", C.POINTER(C.c_double)), ("e_num_walker", C.POINTER(C.c_double)), ("e_den_walker", C.POINTER(C.c_double)),
        ("n_imp", C.c_int64), ("nnz", C.c_int64),
        ("prj_counts", C.POINTER(C.c_int64)), ("prj_indices", C.POINTER(C.c_int64)), ("prj_values", C.POINTER(C.c_double)),
        ("n_ct", C.c_int64), ("ct_up", C.POINTER(C.c_uint64)), ("ct_dn", C.POINTER(C.c_uint64)),
        ("ct_num", C.POINTER(C.c_double)), ("ct_den", C.POINTER(C.c_double)),
        ("n_perm", C.c_int), ("sign_perm", C.POINTER(C.c_int8)),
        ("rng", Rng), ("n_spawn_draws", C.c_int64),
    ]


class StepParams(C.Structure):
    _fields_ = [("tau", C.c_double), ("e_trial", C.c_double), ("reweight_factor_inv", C.c_double),
                ("r_initiator", C.c_double), ("min_wt", C.c_double), ("always_spawn_cutoff_wt", C.c_double),
                ("initiator_power", C.c_int), ("initiator_min_distance", C.c_int), ("c_t_initiator", C.c_int),
                ("semistochastic", C.c_int), ("reached_w_abs_gen", C.c_int)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ChemSystem:
    """orc_chem handle + numpy views of its tables."""

    def __init__(self, fcidump, nelec, nup, point_group="d2h", time_sym=False, z=1, n_core_orb=0,
                 hf_mode=0, hf_symmetry=1):
        L = lib()
        self.h = L.orc_chem_load(fcidump.encode(), nelec, nup, point_group.encode(), int(time_sym), z,
                                 n_core_orb, hf_mode, hf_symmetry)
        if not self.h:
            raise RuntimeError("orc_chem_load failed for %s" % fcidump)
        self.s = Chem.from_address(self.h)
        self.norb, self.nelec, self.nup, self.ndn = self.s.norb, self.s.nelec, self.s.nup, self.s.ndn
        self.hf_up, self.hf_dn = int(self.s.hf_up), int(self.s.hf_dn)
        self._hb = False

    def close(self):
        if self.h:
            lib().orc_chem_free(self.h)
            self.h = None

    # -- tables as numpy copies (what a Fortran host would hand to sqmc_gpu_init)
    def integrals(self):
        return np.ctypeslib.as_array(self.s.integrals, shape=(self.s.n_int + 1,)).copy()

    def combine_2(self):
        a = np.array([[self.s.combine_2[i][j] for j in range(self.norb + 2)] for i in range(self.norb + 2)], dtype=np.int32)
        return a

    def orbsym(self):
        return np.array([self.s.orbsym[i] for i in range(self.norb + 1)], dtype=np.int32)

    def prod(self):
        return np.array([[self.s.prod[i][j] for j in range(9)] for i in range(9)], dtype=np.int32)

    def setup_hb(self):
        if not self._hb:
            lib().orc_chem_setup_hb(self.h)
            self._hb = True

    def hb_tables(self):
        self.setup_hb()
        n = self.s.n_hb
        r = np.ctypeslib.as_array(self.s.hb_r, shape=(n,)).copy()
        s_ = np.ctypeslib.as_array(self.s.hb_s, shape=(n,)).copy()
        a = np.ctypeslib.as_array(self.s.hb_absH, shape=(n,)).copy()
        pi = np.ctypeslib.as_array(self.s.pq_ind, shape=(self.s.n_pq + 1,)).copy()
        pc = np.ctypeslib.as_array(self.s.pq_count, shape=(self.s.n_pq + 1,)).copy()
        return r, s_, a, pi, pc

    def ham(self, iu, id_, ju, jd):
        return lib().orc_hamiltonian(self.h, iu, id_, ju, jd)

    def ham_chem(self, iu, id_, ju, jd, level):
        return lib().orc_hamiltonian_chem(self.h, iu, id_, ju, jd, level)

    def diag_lowest_highest(self):
        """system_setup_chem, chemistry.f90:401-437"""
        nc = self.s.n_core_orb
        mk = lambda n: (1 << n) - 1
        mu = mk(self.norb) - mk(self.norb + nc - self.nup) + mk(nc)
        md = mk(self.norb) - mk(self.norb + nc - self.ndn) + mk(nc)
        lo = self.ham(self.hf_up, self.hf_dn, self.hf_up, self.hf_dn)
        hi = self.ham_chem(mu, md, mu, md, 0)
        return lo, hi

    def connected(self, up, dn, with_elems=True, cap=400000):
        cu = np.zeros(cap, np.uint64); cd = np.zeros(cap, np.uint64); el = np.zeros(cap)
        n = lib().orc_find_connected_dets_chem(self.h, up, dn, _p(cu), _p(cd), _p(el) if with_elems else None, cap)
        assert n <= cap
        return cu[:n], cd[:n], el[:n]

    def important_connected(self, up, dn, eps, cap=400000):
        self.setup_hb()
        cu = np.zeros(cap, np.uint64); cd = np.zeros(cap, np.uint64); el = np.zeros(cap)
        n = lib().orc_find_important_connected_dets_chem(self.h, up, dn, eps, _p(cu), _p(cd), _p(el), cap)
        assert n <= cap
        return cu[:n], cd[:n], el[:n]

    def build_sparse_ham(self, up, dn):
        """Lower-triangular CSR ('upper triangular' in the reference's naming), 1-based cols."""
        up = np.ascontiguousarray(up, np.uint64); dn = np.ascontiguousarray(dn, np.uint64)
        n = len(up)
        rc = C.c_void_p(); ix = C.c_void_p(); vl = C.c_void_p()
        nnz = lib().orc_build_sparse_ham(self.h, n, _p(up), _p(dn), C.byref(rc), C.byref(ix), C.byref(vl))
        counts = np.ctypeslib.as_array(C.cast(rc, C.POINTER(C.c_int64)), shape=(n,)).copy()
        idx = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_int64)), shape=(nnz,)).copy()
        val = np.ctypeslib.as_array(C.cast(vl, C.POINTER(C.c_double)), shape=(nnz,)).copy()
        for p_ in (rc, ix, vl):
            lib().orc_free(p_)
        return counts, idx, val


def spmv_sym_upper(counts, idx, val, x):
    y = np.zeros_like(x)
    lib().orc_spmv_sym_upper(len(counts), _p(counts), _p(idx), _p(val), _p(np.ascontiguousarray(x)), _p(y))
    return y


def sort_dets(up, dn):
    order = np.lexsort((dn, up))
    return order


def lowest_eigs(counts, idx, val, k=1, v0=None, tol=1e-12):
    """Lowest k eigenpairs of the symmetric matrix stored as lower-tri CSR.  The reference
    uses its own Davidson (more_tools.f90:2018) + LAPACK dsyev; any converged eigensolver
    gives the same pair to round-off, so the oracle uses scipy (dense below 600 rows)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    n = len(counts)
    indptr = np.concatenate(([0], np.cumsum(counts)))
    A = sp.csr_matrix((val, idx - 1, indptr), shape=(n, n))
    d = A.diagonal()
    A = A + A.T - sp.diags(d)
    if n <= 600:
        w, v = np.linalg.eigh(A.toarray())
        return w[:k], v[:, :k]
    w, v = spl.eigsh(A, k=k, which="SA", v0=v0, tol=tol, ncv=max(2 * k + 1, 40))
    o = np.argsort(w)
    return w[o], v[:, o]


def hci_variational(sysm, eps_var, eps_sched=(), n_states=1, max_iters=50, log=None):
    """perform_hci variational loop, hci.f90:359-520 with get_next_det_list 865-1040.
    Returns (dets_up, dets_dn, coeffs[n,n_states], energies, history of ndets)."""
    sysm.setup_hb()
    sched = list(eps_sched) + [eps_var]
    up = np.array([sysm.hf_up], np.uint64); dn = np.array([sysm.hf_dn], np.uint64)
    wts = np.zeros((1, n_states)); wts[0, 0] = 1.0
    energy = np.array([sysm.ham(sysm.hf_up, sysm.hf_dn, sysm.hf_up, sysm.hf_dn)] + [0.0] * (n_states - 1))
    old_energy = energy.copy()
    hist = [1]
    eps = sched[0]
    for it in range(1, max_iters + 1):
        if it <= len(sched):
            eps = sched[it - 1]
        coeffs = np.abs(wts).max(axis=1) if it > 1 else wts[:, 0].copy()
        new = {}
        for i in range(len(up)):
            c = abs(coeffs[i])
            if c == 0.0:
                continue
            cu, cd, _ = sysm.important_connected(int(up[i]), int(dn[i]), eps / c, cap=40000)
            for a, b in zip(cu.tolist(), cd.tolist()):
                new[(a, b)] = 1
        old = set(zip(up.tolist(), dn.tolist()))
        add = sorted(k for k in new if k not in old)          # appended in sorted order (hci.f90:979-991)
        n_old, n_new = len(up), len(up) + len(add)
        if n_new == n_old:
            continue
        if n_new <= int(1.00001 * n_old) and eps == sched[-1]:
            break
        up = np.concatenate((up, np.array([a for a, _ in add], np.uint64)))
        dn = np.concatenate((dn, np.array([b for _, b in add], np.uint64)))
        # diagonalise in sorted order, report in list order
        order = sort_dets(up, dn)
        counts, idx, val = sysm.build_sparse_ham(up[order], dn[order])
        v0 = np.zeros(n_new); v0[np.argsort(order)[:n_old]] = wts[:, 0] if it > 1 else 0.0
        if it == 1 or not np.any(v0):
            v0 = None
        w, v = lowest_eigs(counts, idx, val, k=n_states, v0=v0)
        wts = np.zeros((n_new, n_states)); wts[order, :] = v
        energy = w.copy()
        hist.append(n_new)
        if log:
            log("Iteration %3d eps1=%.1e ndets=%9d energy=%s" % (it, eps, n_new, " ".join("%.9f" % e for e in energy)))
        if np.max(np.abs(energy - old_energy)) < 1e-5 and eps == sched[-1]:
            old_energy = energy.copy()
            break
        old_energy = energy.copy()
    return up, dn, wts, energy, hist
