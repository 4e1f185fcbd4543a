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
    _fields_ = [("l", C.c_int * 4), ("mode", C.c_int), ("pad", C.c_int), ("ctr", C.c_uint64), ("seed", C.c_uint64), ("step", C.c_uint64)]


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
        ("rng", Rng), ("n_spawn_draws", C.c_int64), ("key_norb", C.c_int), ("key_ndn", C.c_int),
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

    def set_active_space(self, core_up, core_dn, virt_up, virt_dn, mode):
        """masks of find_important_connected_dets_chem: mode 0 none, 1 inside the active space only, 2 outside only"""
        lib().orc_set_active_space.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
        lib().orc_set_active_space(self.h, int(core_up), int(core_dn), int(virt_up), int(virt_dn), int(mode))

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


class HeatBath:
    """orc_hb handle: tables of setup_efficient_heatbath (chemistry.f90:872-1230) for a ChemSystem, and the move
    off_diagonal_move_chem_efficient_heatbath (5086-5347) with its proposal-probability function (5431-5549)."""

    def __init__(self, sysm):
        L = lib()
        L.orc_hb_setup.restype = C.c_void_p
        L.orc_hb_setup.argtypes = [C.c_void_p]
        L.orc_off_diagonal_move_chem_heatbath.restype = C.c_int
        L.orc_off_diagonal_move_chem_heatbath.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint64, C.c_uint64] + [C.c_void_p] * 5
        L.orc_hb_proposal_prob.restype = C.c_double
        L.orc_hb_proposal_prob.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint64] * 4 + [C.c_int, C.c_double]
        self.sysm, self.L = sysm, L
        self.h = L.orc_hb_setup(sysm.h)
        if not self.h:
            raise RuntimeError("orc_hb_setup failed")
        self.s = HbTables.from_address(self.h)
        self.unbiased = bool(self.s.unbiased)

    def move(self, rng, tau, up, dn):
        """one proposal; returns [(det_j_up, det_j_dn, weight_j, excite_level), ...] for the 0, 1 or 2 new determinants, and the draws used"""
        ju, jd, w, lev, nd = (C.c_uint64 * 2)(), (C.c_uint64 * 2)(), (C.c_double * 2)(), (C.c_int * 2)(), C.c_int()
        n = self.L.orc_off_diagonal_move_chem_heatbath(self.sysm.h, self.h, C.byref(rng), tau, int(up), int(dn), ju, jd, w, lev, C.byref(nd))
        return [(ju[k], jd[k], w[k], lev[k]) for k in range(2) if lev[k] >= 0 and (k < n or w[k] != 0.0)], n, nd.value

    def proposal_prob(self, iu, id_, ju, jd, level, elem):
        return self.L.orc_hb_proposal_prob(self.sysm.h, self.h, int(iu), int(id_), int(ju), int(jd), int(level), float(elem))

    def fortran_arrays(self):
        """the tables in the reference's own (Fortran, column-major, 1-based -> flat) layout: what a Fortran host would pass through
        sqmc_gpu_set_heatbath_tables with c_loc of the module arrays of chemistry.f90:52-75"""
        s, n = self.s, self.s.norb
        n1 = n + 1
        g = lambda ptr, count, dt: np.ctypeslib.as_array(ptr, shape=(count,)).astype(dt, copy=True)
        one = g(s.one, n1, np.float64)[1:]
        two = g(s.two, (2 * n + 1) ** 2, np.float64).reshape(2 * n + 1, 2 * n + 1)[1:, 1:]
        def t3(ptr, dt):
            a = g(ptr, n1 ** 3, dt).reshape(n1, n1, n1)[1:, 1:, 1:]
            return np.asfortranarray(a).reshape(-1, order="F")            # (i,j,k) column-major
        npairs = s.n_pairs
        hs = g(s.htot_same, (npairs + 1) * n1, np.float64).reshape(npairs + 1, n1)[1:, 1:]
        return dict(norb=n, one=one, two=np.asfortranarray(two).reshape(-1, order="F"),
                    three_same=t3(s.three_same, np.float64), three_opp=t3(s.three_opp, np.float64),
                    j3_same=t3(s.j3_same, np.int32), j3_opp=t3(s.j3_opp, np.int32), q3_same=t3(s.q3_same, np.float64), q3_opp=t3(s.q3_opp, np.float64),
                    size_same=s.size_same, size_opp=s.size_opp,
                    four_same=g(s.four_same, s.size_same + 1, np.float32)[1:], four_opp=g(s.four_opp, s.size_opp + 1, np.float32)[1:],
                    j4_same=g(s.j4_same, s.size_same + 1, np.int32)[1:], j4_opp=g(s.j4_opp, s.size_opp + 1, np.int32)[1:],
                    q4_same=g(s.q4_same, s.size_same + 1, np.float32)[1:], q4_opp=g(s.q4_opp, s.size_opp + 1, np.float32)[1:],
                    htot_same=np.asfortranarray(hs).reshape(-1, order="F"), htot_opp=t3(s.htot_opp, np.float64))

    def close(self):
        if self.h:
            self.L.orc_hb_free.argtypes = [C.c_void_p]
            self.L.orc_hb_free(self.h); self.h = None


class HbTables(C.Structure):
    _fields_ = [("norb", C.c_int), ("nup", C.c_int), ("ndn", C.c_int), ("n_core", C.c_int), ("n_orb_uniq_sym", C.c_int), ("unbiased", C.c_int),
                ("size_same", C.c_int64), ("size_opp", C.c_int64), ("n_pairs", C.c_int64),
                ("one", C.POINTER(C.c_double)), ("two", C.POINTER(C.c_double)),
                ("three_same", C.POINTER(C.c_double)), ("three_opp", C.POINTER(C.c_double)), ("j3_same", C.POINTER(C.c_int)), ("j3_opp", C.POINTER(C.c_int)),
                ("q3_same", C.POINTER(C.c_double)), ("q3_opp", C.POINTER(C.c_double)),
                ("four_same", C.POINTER(C.c_float)), ("four_opp", C.POINTER(C.c_float)), ("j4_same", C.POINTER(C.c_int)), ("j4_opp", C.POINTER(C.c_int)),
                ("q4_same", C.POINTER(C.c_float)), ("q4_opp", C.POINTER(C.c_float)),
                ("htot_same", C.POINTER(C.c_double)), ("htot_opp", C.POINTER(C.c_double))]


def spmv_sym_upper(counts, idx, val, x):
    y = np.zeros_like(x)
    lib().orc_spmv_sym_upper(len(counts), _p(counts), _p(idx), _p(val), _p(np.ascontiguousarray(x)), _p(y))
    return y


def sort_dets(up, dn):
    order = np.lexsort((dn, up))
    return order


def lowest_eigs(counts, idx, val, k=1, v0=None, tol=1e-12):
    """Lowest k eigenpairs of the symmetric matrix stored as lower-tri CSR with a library
    eigensolver (scipy; dense below 600 rows).  Only used where the state asked for is
    unambiguous (tests of the matvec/builder); the HCI and walk set-up paths use davidson_sparse
    below, because a generic solver returns the GLOBAL lowest state while the reference's Davidson
    stays in the symmetry sector of its starting vector (they differ where states of D_inf_h
    symmetry that D2h cannot tell apart cross, e.g. C2 between r = 1.3 and 1.6 A)."""
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


def davidson_sparse(counts, idx, val, n_states=1, initial_vectors=None, epsilon=1.0e-10, trace=None):
    """davidson_sparse, more_tools.f90:2018-2244 (and its one-state twin davidson_sparse_single,
    :3056-3230): diagonally preconditioned Davidson on the upper-triangular sparse storage.
    Start: the given vectors (Gram-Schmidt in order, :2066-2082) or unit vectors on the first
    n_states determinants (:2084-2088); one correction vector per state and sweep,
    (H w - e w)/(e - H_ii) with -1 where the denominator vanishes (:2166-2169), orthogonalised
    against the whole basis; dsyev on the Krylov matrix after every n_states additions; stop when
    the eigenvalues move by less than epsilon (=1e-10, more_tools.f90:73) or the summed squared
    correction norms fall below 1e-12; the basis of 50 vectors per state is recycled from the
    current best vectors (:2142-2160).  Returns (eigenvalues[n_states], vectors[n, n_states])."""
    n = len(counts)
    counts = np.ascontiguousarray(counts, np.int64); idx = np.ascontiguousarray(idx, np.int64); val = np.ascontiguousarray(val, np.float64)
    if n == 1:
        return np.array([val[0]]), np.ones((1, 1))
    mv = lambda x: spmv_sym_upper(counts, idx, val, x)
    iterations = min(n, 50)
    v = np.zeros((n, n_states * iterations)); Hv = np.zeros_like(v)
    if initial_vectors is not None:
        iv = np.asarray(initial_vectors, float).reshape(n, -1)
        for i in range(n_states):
            v[:, i] = iv[:, i] / np.sqrt(np.dot(iv[:, i], iv[:, i]))
            if i > 0:
                for j in range(i):
                    v[:, i] -= np.dot(v[:, i], v[:, j]) * v[:, j]
                v[:, i] /= np.sqrt(np.dot(v[:, i], v[:, i]))
    else:
        for i in range(n_states):
            v[i, i] = 1.0
    starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
    diag = val[starts]
    hk = np.zeros((n_states * iterations, n_states * iterations))
    for i in range(n_states):
        Hv[:, i] = mv(v[:, i])
    low = np.zeros(n_states)
    for i in range(n_states):
        low[i] = np.dot(v[:, i], Hv[:, i]); hk[i, i] = low[i]
        for j in range(i + 1, n_states):
            hk[i, j] = hk[j, i] = np.dot(v[:, i], Hv[:, j])
    w, Hw = v[:, :n_states].copy(), Hv[:, :n_states].copy()
    if trace is not None:
        trace.append(low.copy())                   # 'Iteration, Eigenvalues=  1' (more_tools.f90:2128)
    res = np.ones(n_states)
    niter = min(n, n_states * iterations)
    low_prev = np.full(n_states, np.inf)
    converged = False
    it = n_states
    while it < niter * 10:
        it += 1
        itc = (it - 1) % niter + 1
        if it > niter and itc == 1:
            v[:, :n_states], Hv[:, :n_states] = w, Hw
            for i in range(n_states):
                low[i] = np.dot(v[:, i], Hv[:, i]); hk[i, i] = low[i]
                for j in range(i + 1, n_states):
                    hk[i, j] = hk[j, i] = np.dot(v[:, i], Hv[:, j])
            continue
        i = (itc - 1) % n_states
        den = low[i] - diag
        small = np.abs(den) < 1e-8
        t = (Hw[:, i] - low[i] * w[:, i]) / np.where(small, 1.0, den)
        t[small] = -1.0
        res[i] = np.dot(t, t)
        if res.sum() < 1.0e-12:
            converged = True
        for j in range(itc - 1):
            t -= np.dot(t, v[:, j]) * v[:, j]
        t /= np.sqrt(np.dot(t, t))
        v[:, itc - 1] = t
        Hv[:, itc - 1] = mv(t)
        for j in range(itc):
            hk[j, itc - 1] = hk[itc - 1, j] = np.dot(v[:, j], Hv[:, itc - 1])
        if itc % n_states == 0:
            ev, y = np.linalg.eigh(hk[:itc, :itc])
            low = ev[:n_states].copy()
            w = v[:, :itc] @ y[:, :n_states]
            Hw = Hv[:, :itc] @ y[:, :n_states]
            if np.max(np.abs(low - low_prev)) < epsilon:
                break
            low_prev = low.copy()
            if trace is not None:
                trace.append(low.copy())               # 'Iteration, Eigenvalues=' (more_tools.f90:2219)
            if converged:
                break
    return low, w


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
        # hci.f90:453-476 + iterative_diagonalize 1042-1095: the list keeps its order (old determinants,
        # then the new ones sorted); the sparse H builder wants sorted labels, so rows are permuted
        # back and forth around it.  Starting vectors: the previous eigenvectors padded with zeros;
        # in iteration 1 unit vectors on the first n_states determinants of the list.
        order = sort_dets(up, dn)
        counts, idx, val = sysm.build_sparse_ham(up[order], dn[order])
        start = np.zeros((n_new, n_states))
        if it == 1:
            for i in range(min(n_states, n_new)):
                start[i, i] = 1.0
        else:
            start[:n_old, :] = wts
        w, v = davidson_sparse(counts, idx, val, n_states, initial_vectors=start[order, :])
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


# ------------------------------------------------------------------------------------
# Walk setup + driver (oracle side).  Setup follows one iteration of
# generate_space_iterate (semistoch.f90:145-560): connections of HF -> diagonalise ->
# truncate by |coefficient| at a CSF boundary; C(T) as generate_psi_t_connected_e_loc
# (semistoch.f90:27-133); initial population as do_walk.f90:1245-1366.
# ------------------------------------------------------------------------------------
def _truncate_at_csf(c_sorted, n_keep, eps=1e-10):
    """semistoch.f90:331-345 (orc_truncate_at_csf, sqmc_oracle_setup.c)"""
    L = lib()
    L.orc_truncate_at_csf.restype = C.c_int64
    L.orc_truncate_at_csf.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_double]
    a = np.ascontiguousarray(c_sorted, np.float64)
    return int(L.orc_truncate_at_csf(_p(a), len(a), int(n_keep), float(eps)))


class WalkSetup:
    pass


def _lowest_in_space(counts, idx, val, guess):
    """lowest eigenpair of a (small) sparse symmetric matrix in the upper-triangular storage: dense LAPACK below 3000 rows, as the
    reference's own small cases are handled (real_symmetric_diagonalize, semistoch.f90:1037-1043), Davidson from `guess` above"""
    n = len(counts)
    if n > 3000:
        return davidson_sparse(counts, idx, val, 1, initial_vectors=np.ascontiguousarray(guess).reshape(-1, 1))
    H = np.zeros((n, n))
    row = np.repeat(np.arange(n), np.asarray(counts, np.int64))
    H[row, np.asarray(idx, np.int64) - 1] = val
    H[np.asarray(idx, np.int64) - 1, row] = val
    w, v = np.linalg.eigh(H)
    return w[:1], v[:, :1]


def _reps(cu, cd):
    """unique time-reversal representatives (up <= dn) of a list of determinants, sorted"""
    a, b = np.minimum(cu, cd), np.maximum(cu, cd)
    keys = sorted(set(zip(a.tolist(), b.tolist())))
    return np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint64)


def _take(ptr, n, dtype):
    """copy n items out of a malloc'ed C array and free it"""
    ct = {np.uint64: C.c_uint64, np.float64: C.c_double}[dtype]
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(max(n, 1),))[:n].copy() if n else np.zeros(0, dtype)
    lib().orc_free(ptr)
    return out


def _cut_and_connect(kind, sysm, up, dn, c, e0, n_truncate_trial_wf, size_deterministic, tau, rediagonalize, eigenvector=True):
    """What every system's set-up does once it holds a wave function (up, dn, c) on a sorted determinant list: Psi_T and the
    deterministic space by |c| (cut at CSF boundaries), optional rediagonalisation of Psi_T in its own space (semistoch.f90:575,
    706-712), the projector -tau H on the deterministic space (do_walk.f90:954-962) and the local-energy pieces of C(T)
    (orc_psi_t_connected).  kind: 0 chem, 1 heg, 2 hubbard2."""
    import math
    s = WalkSetup()
    if eigenvector and c[np.argmax(np.abs(c))] < 0:          # an eigenvector's overall sign is the solver's: largest component positive
        c = -c
    by = np.argsort(-np.abs(c), kind="stable")
    up_s, dn_s, c_s = up[by], dn[by], c[by]
    n_t, n_i = _truncate_at_csf(c_s, n_truncate_trial_wf), _truncate_at_csf(c_s, size_deterministic)
    norm = 1.0 / math.sqrt(math.fsum(float(x) * float(x) for x in c_s[:n_t]))
    s.psi_up, s.psi_dn, s.psi_c = up_s[:n_t].copy(), dn_s[:n_t].copy(), c_s[:n_t] * norm
    if rediagonalize:
        o = sort_dets(s.psi_up, s.psi_dn)
        s.psi_up, s.psi_dn = s.psi_up[o], s.psi_dn[o]
        counts, idx, val = sysm.build_sparse_ham(s.psi_up, s.psi_dn)
        wr, vr = _lowest_in_space(counts, idx, val, s.psi_c[o])
        cr = vr[:, 0]
        s.psi_c, s.e_psi_t = (-cr if cr[np.argmax(np.abs(cr))] < 0 else cr), float(wr[0])
    o = sort_dets(up_s[:n_i], dn_s[:n_i])
    s.imp_up, s.imp_dn = up_s[:n_i][o].copy(), dn_s[:n_i][o].copy()
    s.tau, s.e_var = tau, float(e0)
    pc, pi, pv = sysm.build_sparse_ham(s.imp_up, s.imp_dn)
    s.prj_counts, s.prj_indices, s.prj_values = pc, pi, -s.tau * pv
    L = lib()
    L.orc_psi_t_connected.restype = C.c_int64
    L.orc_psi_t_connected.argtypes = [C.c_int, C.c_void_p, C.c_int64] + [C.c_void_p] * 7
    pu, pd, pn, pe = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    psi_up, psi_dn, psi_c = (np.ascontiguousarray(s.psi_up, np.uint64), np.ascontiguousarray(s.psi_dn, np.uint64), np.ascontiguousarray(s.psi_c, np.float64))
    handle = sysm.h
    n_ct = int(L.orc_psi_t_connected(kind, handle, len(psi_c), _p(psi_up), _p(psi_dn), _p(psi_c), C.byref(pu), C.byref(pd), C.byref(pn), C.byref(pe)))
    s.ct_up, s.ct_dn, s.ct_num, s.ct_den = _take(pu, n_ct, np.uint64), _take(pd, n_ct, np.uint64), _take(pn, n_ct, np.float64), _take(pe, n_ct, np.float64)
    s.e_trial0 = float(math.fsum(a * b for a, b in zip(s.ct_num, s.ct_den)) / math.fsum(b * b for b in s.ct_den))
    return s


def setup_walk(sysm, n_truncate_trial_wf=100, size_deterministic=1000, tau_multiplier=0.1, coeffs="eig", rediagonalize=False):
    """Returns Psi_T, C(T), deterministic space + projector (-tau*H), tau.
    coeffs="eig": lowest eigenvector of H in {HF + connections} (the reference's scheme);
    coeffs="pt1": first-order perturbation coefficients H_i0/(H_00-H_ii), which involve no
    eigensolver/BLAS and are therefore bit-reproducible on any machine (golden fixtures)."""
    ts = bool(sysm.s.time_sym)
    cu, cd, el = sysm.connected(sysm.hf_up, sysm.hf_dn, with_elems=(coeffs == "pt1"))
    if ts:      # time-reversal symmetry: work with the representatives up <= dn (chemistry.f90:7346-7386)
        cu, cd = _reps(cu, cd)
        el = np.array([sysm.ham(int(a), int(b), sysm.hf_up, sysm.hf_dn) for a, b in zip(cu, cd)])
    order = sort_dets(cu, cd)
    up, dn = cu[order], cd[order]
    if coeffs == "pt1":
        h00 = sysm.ham(sysm.hf_up, sysm.hf_dn, sysm.hf_up, sysm.hf_dn)
        c = np.array([1.0 if (int(a), int(b)) == (sysm.hf_up, sysm.hf_dn) else h / (h00 - sysm.ham(int(a), int(b), int(a), int(b)))
                      for a, b, h in zip(up, dn, el[order])])
        e0 = h00
    else:
        counts, idx, val = sysm.build_sparse_ham(up, dn)
        w, v = davidson_sparse(counts, idx, val, 1)          # starts on the first (= HF) determinant, more_tools.f90:3113-3114
        c, e0 = v[:, 0], w[0]
    lo, hi = sysm.diag_lowest_highest()
    return _cut_and_connect(0, sysm, up, dn, c, e0, n_truncate_trial_wf, size_deterministic, tau_multiplier / (hi - lo), rediagonalize, eigenvector=(coeffs != "pt1"))


def _initial_population(s, w_abs_gen_begin, r_initiator, initiator_power, psit=None):
    """orc_initial_population (oracle/sqmc_oracle_ctl.c): do_walk.f90:1245-1373 as the text runs -- deterministic-space determinants, then
    Psi_T (or all of C(T) with hf_to_psit), sort, merge_original_with_spawned2 -- instead of a Python reading of its result"""
    L = lib()
    L.orc_initial_population.restype = C.c_int64
    L.orc_initial_population.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                         C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    o = sort_dets(s.psi_up, s.psi_dn)                      # Psi_T in label order (do_walk.f90:1258)
    pu, pd, pc_ = np.ascontiguousarray(s.psi_up[o]), np.ascontiguousarray(s.psi_dn[o]), np.ascontiguousarray(np.asarray(s.psi_c, float)[o])
    iu, idn = np.ascontiguousarray(s.imp_up), np.ascontiguousarray(s.imp_dn)
    cu, cd = np.ascontiguousarray(s.ct_up), np.ascontiguousarray(s.ct_dn)
    cap = len(iu) + len(pu) + (len(cu) if psit is not None else 0) + 16
    h = L.orc_walk_new(cap)
    w = Walk.from_address(h)
    p = StepParams(tau=s.tau, e_trial=0.0, reweight_factor_inv=1.0, r_initiator=r_initiator, min_wt=0.5, always_spawn_cutoff_wt=0.5, initiator_power=initiator_power,
                   initiator_min_distance=0, c_t_initiator=0, semistochastic=1, reached_w_abs_gen=0)
    n = L.orc_initial_population(h, len(iu), _p(iu), _p(idn), len(pu), _p(pu), _p(pd), _p(pc_), float(w_abs_gen_begin), C.byref(p),
                                 1 if psit is not None else 0, len(cu), _p(cu), _p(cd))
    g = lambda ptr: np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    out = dict(up=g(w.up), dn=g(w.dn), wt=g(w.wt), initiator=g(w.initiator), imp_distance=g(w.imp_distance), matrix_elements=g(w.matrix_elements),
               e_num=g(w.e_num_walker), e_den=g(w.e_den_walker))
    signs = np.ctypeslib.as_array(w.sign_perm, shape=(max(w.n_perm, 1),))[:w.n_perm].copy()
    ps = np.zeros(n, np.int8)
    ps[out["initiator"] == 3] = signs                      # sorted order on both sides (do_walk.f90:2593-2594)
    out["perm_sign"] = ps
    L.orc_walk_free(h)
    return out


def initial_walkers(s, w_abs_gen_begin, r_initiator=1.0, initiator_power=0):
    """do_walk.f90:1245-1366 (hf_to_psit = false): deterministic-space dets with weight 0 plus Psi_T dets with weight w_begin*c/sum|c|,
    sorted and merged by merge_original_with_spawned2.  Returns sorted SoA + signs."""
    return _initial_population(s, w_abs_gen_begin, r_initiator, initiator_power)


class OracleWalk:
    """orc_walk handle fed from numpy arrays."""

    def __init__(self, sysm, setup, walkers, mwalk, seed, rng_mode=0, heatbath=None, psit=None, quirks=0, sum_order=1):
        L = lib()
        self.sysm, self.L, self.hb = sysm, L, heatbath          # heatbath: a HeatBath of sysm -> proposal_method fast_heatbath
        self.q = None                                           # psit: a PsitSetup -> the hf_to_psit step (sqmc_oracle_psit.c)
        self.h = L.orc_walk_new(mwalk)
        self.w = Walk.from_address(self.h)
        self._keep = []
        n = len(walkers["up"])
        for name, key in (("up", "up"), ("dn", "dn"), ("wt", "wt"), ("imp_distance", "imp_distance"), ("initiator", "initiator"),
                          ("matrix_elements", "matrix_elements"), ("e_num_walker", "e_num"), ("e_den_walker", "e_den")):
            dst = getattr(self.w, name)
            src = np.ascontiguousarray(walkers[key])
            C.memmove(dst, src.ctypes.data, src.nbytes)
        self.w.nwalk = n
        signs = walkers["perm_sign"][walkers["initiator"] == 3].astype(np.int8)
        self._set_ptr("sign_perm", signs if len(signs) else np.zeros(1, np.int8), C.c_int8)
        self.w.n_perm = len(signs)
        prj = psit if psit is not None else setup          # hf_to_psit: the matrix without its first row and column
        self.w.n_imp = len(prj.prj_counts)
        self.w.nnz = len(prj.prj_values)
        self._set_ptr("prj_counts", prj.prj_counts.astype(np.int64), C.c_int64)
        self._set_ptr("prj_indices", prj.prj_indices.astype(np.int64), C.c_int64)
        self._set_ptr("prj_values", prj.prj_values.astype(np.float64), C.c_double)
        self.w.n_ct = len(setup.ct_up)
        self._set_ptr("ct_up", setup.ct_up, C.c_uint64); self._set_ptr("ct_dn", setup.ct_dn, C.c_uint64)
        self._set_ptr("ct_num", setup.ct_num, C.c_double); self._set_ptr("ct_den", setup.ct_den, C.c_double)
        self.w.key_norb, self.w.key_ndn = int(sysm.norb), int(sysm.ndn)      # COUNTER discipline: rounding draws keyed by the determinant's rank
        sd = (C.c_int * 4)(*seed)
        L.orc_setrn(C.byref(self.w.rng), sd)
        L.orc_rng_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_rng_set_mode(C.byref(self.w.rng), rng_mode)
        if psit is not None:
            L.orc_psit_new.restype = C.c_void_p
            L.orc_psit_new.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
            L.orc_psit_free.argtypes = [C.c_void_p]
            L.orc_psit_set.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.orc_psit_n_out.restype = C.c_int64
            L.orc_psit_n_out.argtypes = [C.c_void_p]
            L.orc_walk_step_psit.argtypes = [C.c_void_p] * 5
            L.orc_walk_step_psit_heg.argtypes = [C.c_void_p] * 5
            lp, cd = np.ascontiguousarray(psit.loc_psit, np.int64), np.ascontiguousarray(psit.cdet, np.float64)
            de, li = np.ascontiguousarray(psit.diag_elems, np.float64), np.ascontiguousarray(psit.loc_imp, np.int64)
            assert n == len(de), "the psit walk starts from the C(T) list"
            self.q = L.orc_psit_new(len(de), len(lp), _p(lp), _p(cd), _p(de), len(li), _p(li))
            L.orc_psit_set(self.q, quirks, sum_order)

    def _set_ptr(self, name, arr, ctype):
        # the C side free()s these in orc_walk_free: hand it malloc'ed copies
        arr = np.ascontiguousarray(arr)
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        p = libc.malloc(max(arr.nbytes, 8))
        C.memmove(p, arr.ctypes.data, arr.nbytes)
        setattr(self.w, name, C.cast(p, C.POINTER(ctype)))

    def scale_projector(self, ratio):
        v = np.ctypeslib.as_array(self.w.prj_values, shape=(self.w.nnz,))
        v *= ratio

    def step(self, params):
        out = np.zeros(16)
        p = StepParams(**params)
        if self.q is not None:
            fn = self.L.orc_walk_step_psit_heg if isinstance(self.sysm, HegSystem) else self.L.orc_walk_step_psit
            return fn(self.sysm.h, self.h, self.q, C.byref(p), _p(out)), out
        if self.hb is not None:
            self.L.orc_walk_step_heatbath.argtypes = [C.c_void_p] * 5
            return self.L.orc_walk_step_heatbath(self.sysm.h, self.hb.h, self.h, C.byref(p), _p(out)), out
        fn = (self.L.orc_walk_step_heg if isinstance(self.sysm, HegSystem) else
              self.L.orc_walk_step_hubbard if isinstance(self.sysm, HubbardSystem) else self.L.orc_walk_step)
        st = fn(self.sysm.h, self.h, C.byref(p), _p(out))
        return st, out

    def walkers(self):
        n = self.w.nwalk
        g = lambda ptr, dt: np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        return dict(up=g(self.w.up, None), dn=g(self.w.dn, None), wt=g(self.w.wt, None), imp_distance=g(self.w.imp_distance, None),
                    initiator=g(self.w.initiator, None), matrix_elements=g(self.w.matrix_elements, None),
                    e_num=g(self.w.e_num_walker, None), e_den=g(self.w.e_den_walker, None))

    def rng_state(self):
        return [self.w.rng.l[i] for i in range(4)]

    def n_outside_ct(self):
        return int(self.L.orc_psit_n_out(self.q))

    def debug_premerge(self, on=True):
        self.L.orc_psit_debug.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_psit_debug(self.q, 1 if on else 0)

    def premerge(self):
        """the list in front of the merge of the last step (debug_premerge on): residents [0, n0) after death/clone and the projection,
        then the sorted spawns"""
        self.L.orc_psit_debug_get.restype = C.c_int64
        self.L.orc_psit_debug_get.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        n0 = C.c_int64()
        z = np.zeros(1, np.uint64)
        n = self.L.orc_psit_debug_get(self.q, 0, C.byref(n0), _p(z), _p(z), _p(z), _p(z), _p(z))
        out = dict(up=np.zeros(n, np.uint64), dn=np.zeros(n, np.uint64), wt=np.zeros(n), imp_distance=np.zeros(n, np.int8), initiator=np.zeros(n, np.int8))
        self.L.orc_psit_debug_get(self.q, n, C.byref(n0), _p(out["up"]), _p(out["dn"]), _p(out["wt"]), _p(out["imp_distance"]), _p(out["initiator"]))
        return out, int(n0.value)

    def close(self):
        if self.q:
            self.L.orc_psit_free(self.q)
            self.q = None
        if self.h:
            self.L.orc_walk_free(self.h)
            self.h = None


class PsitSetup:
    pass


def psit_setup(sysm, s):
    """What hf_to_psit = .true. adds to a walk set-up (reference lines in sqmc_oracle_psit.c):
    the deterministic-space matrix without its first row and column (generate_sparse_ham_*_upper_triangular with hf_to_psit,
    chemistry.f90:7885-7897, 7926-7933: the (1,1) element is kept, as 0), Psi_T in label order with the places of its determinants
    in the walker list (do_walk.f90:1258, 1849-1886), diag_elems (1091-1116), the fixed places of the deterministic space.
    Stops where the reference silently assumes: deterministic space inside C(T), and the first determinant of C(T), of Psi_T and
    of the deterministic space being the same one."""
    q = PsitSetup()
    ct = {(int(a), int(b)): i for i, (a, b) in enumerate(zip(s.ct_up.tolist(), s.ct_dn.tolist()))}
    o = sort_dets(s.psi_up, s.psi_dn)
    pu, pd, q.cdet = s.psi_up[o], s.psi_dn[o], np.array(s.psi_c)[o]
    q.loc_psit = np.array([ct[(int(a), int(b))] for a, b in zip(pu, pd)], np.int64)
    q.loc_imp = np.array([ct.get((int(a), int(b)), -1) for a, b in zip(s.imp_up, s.imp_dn)], np.int64)
    if (q.loc_imp < 0).any():
        raise ValueError("hf_to_psit: the deterministic space is not contained in C(T)")
    if q.loc_psit[0] != 0 or q.loc_imp[0] != 0:
        raise ValueError("hf_to_psit: C(T), Psi_T and the deterministic space do not begin with the same determinant")
    in_imp = np.zeros(len(s.ct_up), bool)
    in_imp[q.loc_imp] = True
    q.in_imp = in_imp
    q.diag_elems = np.array([0.0 if in_imp[i] else sysm.ham(int(a), int(b), int(a), int(b))
                             for i, (a, b) in enumerate(zip(s.ct_up.tolist(), s.ct_dn.tolist()))])
    # rows hold their diagonal first, then the columns j < i (1-based indices): row 1 keeps (1,1) = 0, the others lose column 1
    cnt, idx, val = [], [], []
    k = 0
    for i, c in enumerate(s.prj_counts.tolist()):
        row = [(int(s.prj_indices[k + j]), float(s.prj_values[k + j])) for j in range(c)]
        k += c
        row = [(1, 0.0)] if i == 0 else [(j, v) for (j, v) in row if j != 1]
        cnt.append(len(row)); idx += [j for j, _ in row]; val += [v for _, v in row]
    q.prj_counts, q.prj_indices, q.prj_values = np.array(cnt, np.int64), np.array(idx, np.int64), np.array(val)
    return q


def initial_walkers_psit(s, q, w_abs_gen_begin):
    """do_walk.f90:1245-1373 with hf_to_psit: the deterministic-space determinants (weight 0) and ALL of C(T), where only the first
    one carries weight; sorted and merged (the deterministic-space copy comes first and keeps imp_distance 0, the permanent-initiator
    flag of the first C(T) determinant survives the merge), then every initiator flag but 3 is set to 2 (1367-1373).  With the
    deterministic space inside C(T) the list is C(T) itself."""
    out = _initial_population(s, w_abs_gen_begin, 1.0, 0, psit=q)
    assert np.array_equal(out["up"], s.ct_up) and np.array_equal(out["dn"], s.ct_dn)
    return out


class _PopCtl(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("tau_sav", "tau", "tau_prev", "e_trial", "e_est", "w_target", "w_abs_gen", "r_init_sav", "r_init", "irp", "pop_exp",
                                          "rfi", "rfi_max", "e_num_cum", "e_den_cum")] + [("istep", C.c_int64), ("n_equil", C.c_int64), ("reached", C.c_int), ("pad", C.c_int)]


class PopControl:
    """The scalar logic around the step -- tau ramp do_walk.f90:2171-2184, e_est / e_trial / reweight factor 2880-2923; equilibration =
    the first n_equil_steps steps -- held and advanced by the oracle's C code (oracle/sqmc_oracle_ctl.c: orc_popctl_*), not by a Python
    copy of the product's host logic.  The attributes read and write the C structure."""

    def __init__(self, tau, e_trial, w_target, r_initiator=1.0, initiator_rescale_power=1.0, pop_exp=10.0,
                 rfi_max_multiplier=1.0, n_equil_steps=10**9):
        L = lib()
        L.orc_popctl_init.argtypes = [C.c_void_p] + [C.c_double] * 7 + [C.c_int64]
        L.orc_popctl_pre_step.restype = C.c_double
        L.orc_popctl_pre_step.argtypes = [C.c_void_p, C.c_double]
        L.orc_popctl_post_step.restype = C.c_double
        L.orc_popctl_post_step.argtypes = [C.c_void_p, C.c_void_p]
        object.__setattr__(self, "_L", L)
        object.__setattr__(self, "_c", _PopCtl())
        L.orc_popctl_init(C.byref(self._c), float(tau), float(e_trial), float(w_target), float(r_initiator), float(initiator_rescale_power), float(pop_exp),
                          float(rfi_max_multiplier), int(min(n_equil_steps, 2**62)))

    _names = {f[0] for f in _PopCtl._fields_}

    def __getattr__(self, k):
        if k in PopControl._names:
            return getattr(self._c, k)
        raise AttributeError(k)

    def __setattr__(self, k, v):
        if k in PopControl._names:
            setattr(self._c, k, v)
        else:
            object.__setattr__(self, k, v)

    def pre_step(self, w_abs_gen):
        """returns tau_ratio to apply to the projector before the step (or 1.0)"""
        return float(self._L.orc_popctl_pre_step(C.byref(self._c), float(w_abs_gen)))

    def post_step(self, out):
        """returns tau_ratio to apply to the projector after the step (or 1.0)"""
        o = np.ascontiguousarray(out, np.float64)
        return float(self._L.orc_popctl_post_step(C.byref(self._c), _p(o)))

    def params(self, min_wt=0.5, cutoff=0.5, initiator_power=0, semistochastic=1):
        c = self._c
        return dict(tau=c.tau, e_trial=c.e_trial, reweight_factor_inv=c.rfi, r_initiator=c.r_init, min_wt=min_wt,
                    always_spawn_cutoff_wt=cutoff, initiator_power=initiator_power, initiator_min_distance=0, c_t_initiator=0,
                    semistochastic=semistochastic, reached_w_abs_gen=c.reached)


# ------------------------------------------------------------------------------ HEG
class Heg(C.Structure):
    _fields_ = [("n_dim", C.c_int), ("nelec", C.c_int), ("nup", C.c_int), ("ndn", C.c_int), ("norb", C.c_int), ("n_max", C.c_int),
                ("r_s", C.c_double), ("length_cell", C.c_double), ("k", (C.c_double * 3) * (ORC_MAXORB + 1)), ("krel", (C.c_int * 3) * (ORC_MAXORB + 1))]


class HegSystem:
    """orc_heg handle: 3D/2D electron gas in a plane-wave basis (heg.f90)."""

    def __init__(self, n_dim, r_s, nelec, nup, cutoff_radius):
        L = lib()
        L.orc_heg_new.restype = C.c_void_p
        L.orc_heg_new.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_double]
        L.orc_hamiltonian_heg.restype = C.c_double
        L.orc_hamiltonian_heg.argtypes = [C.c_void_p] + [C.c_uint64] * 4
        L.orc_off_diagonal_move_heg.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_uint64, C.c_uint64] + [C.c_void_p] * 4
        L.orc_connected_heg.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_build_sparse_ham_heg.restype = C.c_int64
        L.orc_build_sparse_ham_heg.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.orc_walk_step_heg.argtypes = [C.c_void_p] * 4
        self.h = L.orc_heg_new(n_dim, r_s, nelec, nup, cutoff_radius)
        if not self.h:
            raise RuntimeError("orc_heg_new failed")
        self.s = Heg.from_address(self.h)
        self.norb, self.nelec, self.nup, self.ndn = self.s.norb, nelec, nup, nelec - nup
        self.hf_up, self.hf_dn = (1 << nup) - 1, (1 << (nelec - nup)) - 1
        self.length_cell, self.n_dim = self.s.length_cell, n_dim

    def k_vectors(self):
        """k_vectors(n_dim? -> 3, norb) as [norb, 3] (row i-1 = orbital i)"""
        return np.array([[self.s.k[i][j] for j in range(3)] for i in range(1, self.norb + 1)])

    def ham(self, iu, id_, ju, jd):
        return lib().orc_hamiltonian_heg(self.h, iu, id_, ju, jd)

    def connected(self, up, dn, with_elems=True, cap=400000):
        cu = np.zeros(cap, np.uint64); cd = np.zeros(cap, np.uint64); el = np.zeros(cap)
        n = lib().orc_connected_heg(self.h, up, dn, _p(cu), _p(cd), _p(el) if with_elems else None, cap)
        assert n <= cap
        return cu[:n], cd[:n], el[:n]

    def setup_hb(self):
        pass

    def important_connected(self, up, dn, eps, cap=400000):
        """find_important_connected_dets_heg, heg.f90:2475-2727: the determinant itself, then every
        momentum-conserving double excitation whose |H| exceeds eps.  The reference walks |H|-sorted
        translation-invariant lists (dtm_hb_same_spin / dtm_hb_opposite_spin, heg.f90:243-640) and
        stops at absH <= eps (:2608, :2629); the set is the same as screening all doubles."""
        cu, cd, el = self.connected(up, dn, with_elems=True, cap=cap)
        keep = np.abs(el) > eps
        keep[0] = True
        el = el.copy(); el[0] = 0.0
        return cu[keep], cd[keep], el[keep]

    def build_sparse_ham(self, up, dn):
        up = np.ascontiguousarray(up, np.uint64); dn = np.ascontiguousarray(dn, np.uint64)
        n = len(up)
        rc = C.c_void_p(); ix = C.c_void_p(); vl = C.c_void_p()
        nnz = lib().orc_build_sparse_ham_heg(self.h, n, _p(up), _p(dn), C.byref(rc), C.byref(ix), C.byref(vl))
        counts = np.ctypeslib.as_array(C.cast(rc, C.POINTER(C.c_int64)), shape=(n,)).copy()
        idx = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_int64)), shape=(nnz,)).copy()
        val = np.ctypeslib.as_array(C.cast(vl, C.POINTER(C.c_double)), shape=(nnz,)).copy()
        for q in (rc, ix, vl):
            lib().orc_free(q)
        return counts, idx, val

    def diag_lowest_highest(self):
        """heg.f90 has no tau_multiplier rule of its own in the walk decks here: tau is chosen from the
        spread between the HF determinant and the determinant with the highest orbitals filled."""
        n = self.norb
        mu = ((1 << n) - 1) ^ ((1 << (n - self.nup)) - 1)
        md = ((1 << n) - 1) ^ ((1 << (n - self.ndn)) - 1)
        return self.ham(self.hf_up, self.hf_dn, self.hf_up, self.hf_dn), self.ham(mu, md, mu, md)


def setup_walk_heg(hsys, size_deterministic=500, tau_multiplier=0.1, n_truncate_trial_wf=1, rediagonalize=False):
    """HEG walk set-up: Psi_T = the largest-|c| determinants of the ground state in {HF + its
    double excitations} (n_truncate_trial_wf = 1: HF alone, the usual choice for a closed shell),
    deterministic space = the size_deterministic largest, C(T) = connections of Psi_T."""
    cu, cd, _ = hsys.connected(hsys.hf_up, hsys.hf_dn, with_elems=False)
    order = sort_dets(cu, cd)
    up, dn = cu[order], cd[order]
    counts, idx, val = hsys.build_sparse_ham(up, dn)
    w, v = davidson_sparse(counts, idx, val, 1)
    lo, hi = hsys.diag_lowest_highest()
    return _cut_and_connect(1, hsys, up, dn, v[:, 0], w[0], n_truncate_trial_wf, size_deterministic, tau_multiplier / (hi - lo), rediagonalize)


class Hub(C.Structure):
    _fields_ = [("l_x", C.c_int), ("l_y", C.c_int), ("pbc", C.c_int), ("nsites", C.c_int), ("nup", C.c_int), ("ndn", C.c_int),
                ("t", C.c_double), ("U", C.c_double)]


def hubbard_start_det(l_x, l_y, nup, ndn):
    """Starting determinant of the harness (not a reference rule: the reference starts hubbard2 walks
    from its Gutzwiller machinery): up electrons on the sites with x+y even first, dn electrons on
    the sites with x+y odd first -- the Neel state at half filling."""
    sites = list(range(l_x * l_y))
    even = [s for s in sites if ((s % l_x) + (s // l_x)) % 2 == 0]
    odd = [s for s in sites if ((s % l_x) + (s // l_x)) % 2 == 1]
    up = sum(1 << s for s in (even + odd)[:nup])
    dn = sum(1 << s for s in (odd + even)[:ndn])
    return up, dn


class HubbardSystem:
    """orc_hub handle: real-space Hubbard model on an l_x by l_y square lattice (hubbard.f90, 'hubbard2')."""

    def __init__(self, l_x, l_y, pbc, nup, ndn, t=1.0, U=4.0):
        L = lib()
        L.orc_hub_new.restype = C.c_void_p
        L.orc_hub_new.argtypes = [C.c_int] * 5 + [C.c_double] * 2
        L.orc_hub_free.argtypes = [C.c_void_p]
        L.orc_get_nbr.argtypes = [C.c_int] * 5
        L.orc_fermionic_phase.argtypes = [C.c_uint64, C.c_int, C.c_int]
        for f in (L.orc_hamiltonian_hubbard, L.orc_hamiltonian_hubbard_checked):
            f.restype = C.c_double
            f.argtypes = [C.c_void_p] + [C.c_uint64] * 4
        L.orc_off_diagonal_move_hubbard.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_uint64, C.c_uint64] + [C.c_void_p] * 4
        L.orc_connected_hubbard.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_build_sparse_ham_hubbard.restype = C.c_int64
        L.orc_build_sparse_ham_hubbard.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.orc_walk_step_hubbard.argtypes = [C.c_void_p] * 4
        self.h = L.orc_hub_new(l_x, l_y, int(bool(pbc)), nup, ndn, t, U)
        if not self.h:
            raise RuntimeError("orc_hub_new failed")
        self.l_x, self.l_y, self.pbc, self.nup, self.ndn, self.t, self.U = l_x, l_y, bool(pbc), nup, ndn, float(t), float(U)
        self.norb, self.nelec = l_x * l_y, nup + ndn
        self.hf_up, self.hf_dn = hubbard_start_det(l_x, l_y, nup, ndn)

    def close(self):
        if self.h:
            lib().orc_hub_free(self.h)
            self.h = None

    def ham(self, iu, id_, ju, jd):
        return lib().orc_hamiltonian_hubbard_checked(self.h, iu, id_, ju, jd)

    def ham_unchecked(self, iu, id_, ju, jd):
        return lib().orc_hamiltonian_hubbard(self.h, iu, id_, ju, jd)

    def connected(self, up, dn, with_elems=True, cap=4 * 128 + 1):
        cu = np.zeros(cap, np.uint64); cd = np.zeros(cap, np.uint64); el = np.zeros(cap)
        n = lib().orc_connected_hubbard(self.h, up, dn, _p(cu), _p(cd), _p(el) if with_elems else None, cap)
        assert n <= cap
        return cu[:n], cd[:n], el[:n]

    def build_sparse_ham(self, up, dn):
        up = np.ascontiguousarray(up, np.uint64); dn = np.ascontiguousarray(dn, np.uint64)
        n = len(up)
        rc = C.c_void_p(); ix = C.c_void_p(); vl = C.c_void_p()
        nnz = lib().orc_build_sparse_ham_hubbard(self.h, n, _p(up), _p(dn), C.byref(rc), C.byref(ix), C.byref(vl))
        counts = np.ctypeslib.as_array(C.cast(rc, C.POINTER(C.c_int64)), shape=(n,)).copy()
        idx = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_int64)), shape=(nnz,)).copy()
        val = np.ctypeslib.as_array(C.cast(vl, C.POINTER(C.c_double)), shape=(nnz,)).copy()
        for q in (rc, ix, vl):
            lib().orc_free(q)
        return counts, idx, val

    def spectral_range_bound(self):
        """U * (largest - smallest possible number of doubly occupied sites) + 4 t per electron:
        an upper bound on the spread of the spectrum; tau = tau_multiplier / this."""
        ns = self.norb
        return self.U * (min(self.nup, self.ndn) - max(0, self.nelec - ns)) + 4.0 * abs(self.t) * self.nelec

    def first_order_space(self, n_levels=2):
        """the start determinant and everything within n_levels hops of it, sorted by (up, dn)"""
        seen = {(self.hf_up, self.hf_dn)}
        frontier = [(self.hf_up, self.hf_dn)]
        for _ in range(n_levels):
            nxt = []
            for (a, b) in frontier:
                cu, cd, _e = self.connected(a, b, with_elems=False)
                for k in zip(cu.tolist(), cd.tolist()):
                    if k not in seen:
                        seen.add(k); nxt.append(k)
            frontier = nxt
        keys = sorted(seen)
        return np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint64)


def setup_walk_hubbard(hsys, size_deterministic=500, tau_multiplier=0.5, n_truncate_trial_wf=20, n_levels=2):
    """hubbard2 walk set-up with a determinant-list trial wavefunction (energy_pieces_hubbard's
    last branch, hubbard.f90:4514-4527): ground state of H in {start det + n_levels hops};
    Psi_T = its n_truncate_trial_wf largest determinants, deterministic space = the
    size_deterministic largest, C(T) = the connections of Psi_T with sum_j H_ij c_j."""
    up, dn = hsys.first_order_space(n_levels)
    counts, idx, val = hsys.build_sparse_ham(up, dn)
    w, v = davidson_sparse(counts, idx, val, 1)
    return _cut_and_connect(2, hsys, up, dn, v[:, 0], w[0], n_truncate_trial_wf, size_deterministic, tau_multiplier / hsys.spectral_range_bound(), False)


def hci_pt2(sysm, up, dn, coeffs, e_var, eps_pt):
    """second_order_pt (hci.f90:1100-1182) with the oracle's generator; Python accumulation, so
    meant for variational spaces of ~10^4 determinants."""
    sysm.setup_hb()
    vset = set(zip(np.asarray(up).tolist(), np.asarray(dn).tolist()))
    acc = {}
    for a, b, c in zip(np.asarray(up).tolist(), np.asarray(dn).tolist(), np.asarray(coeffs).tolist()):
        if c == 0.0:
            continue
        cu, cd, el = sysm.important_connected(a, b, eps_pt / abs(c), cap=60000)
        for p, q, h in zip(cu.tolist(), cd.tolist(), el.tolist()):
            if (p, q) not in vset:
                acc[(p, q)] = acc.get((p, q), 0.0) + h * c
    delta = 0.0
    for (p, q), v in acc.items():
        delta += v * v / (e_var - sysm.ham(p, q, p, q))
    return delta, len(acc)


# ------------------------------------------------------------------------------------
# Semistochastic PT (second_order_pt_alias, hci.f90:1314-1660), oracle side
# ------------------------------------------------------------------------------------
def setup_alias(pdf):
    """setup_alias, more_tools.f90:5603-5662: outcomes split in index order into 'smaller' (K p < 1) and
    'larger', then paired from the END of both lists.  Returns 1-based J and q."""
    K = len(pdf)
    J = np.arange(1, K + 1); q = K * np.asarray(pdf, float)
    smaller = [i + 1 for i in range(K) if q[i] < 1.0]
    larger = [i + 1 for i in range(K) if not (q[i] < 1.0)]
    while smaller and larger:
        small, large = smaller[-1], larger[-1]
        J[small - 1] = large
        q[large - 1] = q[large - 1] + q[small - 1] - 1.0
        if q[large - 1] < 1.0:
            smaller[-1] = large; larger.pop()
        else:
            smaller.pop()
    return J, q


def second_order_pt_alias(sysm, up, dn, coeffs, e_var, eps_pt, eps_pt_big, n_mc, target_error, seed, max_samples=10**6):
    """Per-sample values of the stochastic difference PT(eps_pt) - PT(eps_pt_big): alias draws with the
    rannyu stream set from `seed` (sample_alias, more_tools.f90:5727-5752: one random_int, one rannyu),
    repeats merged (tools.f90:1574-1602), term1/term2 sums over the connections of the sampled
    determinants (hci.f90:1585-1600), Welford statistics (tools.f90:1761-1778).  Python loops: slow,
    meant for a handful of samples."""
    L = lib()
    L.orc_rannyu.restype = C.c_double; L.orc_random_int.restype = C.c_int
    order = sort_dets(up, dn)
    up, dn, c = np.asarray(up)[order], np.asarray(dn)[order], np.asarray(coeffs, float)[order]
    n = len(up)
    prob = np.abs(c) / np.abs(c).sum()
    J, q = setup_alias(prob)
    rng = Rng()
    L.orc_setrn(C.byref(rng), (C.c_int * 4)(*seed)); L.orc_rng_set_mode(C.byref(rng), 0)
    vset = set(zip(up.tolist(), dn.tolist()))
    vals, mean, s_acc, var = [], 0.0, 0.0, float("nan")
    for sample in range(1, max_samples + 1):
        draws = []
        for _ in range(n_mc):
            i = L.orc_random_int(C.byref(rng), n)
            draws.append(i if L.orc_rannyu(C.byref(rng)) < q[i - 1] else int(J[i - 1]))
        ids, counts = np.unique(draws, return_counts=True)
        t1, t2, t1b, t2b = {}, {}, {}, {}
        for i, cnt in zip(ids.tolist(), counts.tolist()):
            ci, wop = c[i - 1], cnt / prob[i - 1]
            cu, cd, el = sysm.important_connected(int(up[i - 1]), int(dn[i - 1]), eps_pt / abs(ci))
            for a, b, hh in zip(cu[1:].tolist(), cd[1:].tolist(), el[1:].tolist()):
                if (a, b) in vset:
                    continue
                x = hh * ci
                t1[(a, b)] = t1.get((a, b), 0.0) + x * wop
                t2[(a, b)] = t2.get((a, b), 0.0) + x * x * ((n_mc - 1) * wop - wop * wop)
                if abs(hh) > eps_pt_big / abs(ci):
                    t1b[(a, b)] = t1b.get((a, b), 0.0) + x * wop
                    t2b[(a, b)] = t2b.get((a, b), 0.0) + x * x * ((n_mc - 1) * wop - wop * wop)
        val = 0.0
        for k in t1:
            val += (t1[k] ** 2 + t2[k] - t1b.get(k, 0.0) ** 2 - t2b.get(k, 0.0)) / (e_var - sysm.ham(k[0], k[1], k[0], k[1]))
        val /= n_mc * float(n_mc - 1)
        vals.append(val)
        old = mean
        mean = mean + (val - mean) / sample
        s_acc = s_acc + (val - mean) * (val - old)
        if sample > 1:
            var = s_acc / (sample - 1) / sample
        if sample >= 10 and var < target_error ** 2:
            break
    return vals, mean, float(np.sqrt(var)) if sample > 1 else float("nan")
