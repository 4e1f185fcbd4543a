"""ctypes binding of include/sqmc_gpu.h (the same entry points the Fortran iso_c_binding
module sqmc_amd/fortran/sqmc_gpu_mod.f90 binds)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
def _lib_path():
    """the library's file: builds with extra flags (SQMC_EXTRA_CFLAGS: the instrumented builds of tools/*_prof.py) get a name of their own, so
    that they can neither overwrite the product's library nor be mistaken for it by the staleness check"""
    extra = os.environ.get("SQMC_EXTRA_CFLAGS", "").split()
    if not extra:
        return os.path.join(_HERE, "libsqmc_gpu.so")
    import hashlib
    return os.path.join(_HERE, "libsqmc_gpu_%s.so" % hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10])


LIB_PATH = _lib_path()
RNG_REPLAY, RNG_COUNTER = 0, 1
_LIB = None

EXPORTS = [
    "sqmc_gpu_set_device", "sqmc_gpu_init_chem", "sqmc_gpu_init_heg", "sqmc_gpu_init_hubbard", "sqmc_gpu_finalize", "sqmc_gpu_last_error", "sqmc_gpu_set_hb_tables", "sqmc_gpu_set_heatbath_tables", "sqmc_gpu_setup_efficient_heatbath", "sqmc_gpu_get_heatbath_tables", "sqmc_gpu_propose_heatbath_batch", "sqmc_gpu_set_projector",
    "sqmc_gpu_scale_projector", "sqmc_gpu_set_ct_table", "sqmc_gpu_set_hf_to_psit", "sqmc_gpu_upload_walkers", "sqmc_gpu_num_walkers",
    "sqmc_gpu_download_walkers", "sqmc_gpu_step", "sqmc_gpu_run", "sqmc_gpu_annihilate", "sqmc_gpu_det_owner", "sqmc_gpu_set_owner_hash", "sqmc_gpu_shard_config",
    "sqmc_gpu_shard_begin", "sqmc_gpu_shard_pack", "sqmc_gpu_shard_finish", "sqmc_gpu_comm_unique_id", "sqmc_gpu_comm_init", "sqmc_gpu_comm_size",
    "sqmc_gpu_shard_step", "sqmc_gpu_shard_run", "sqmc_gpu_shard_time_split", "sqmc_gpu_get_rng", "sqmc_gpu_set_rng", "sqmc_gpu_tail_stats", "sqmc_gpu_slowest_steps", "sqmc_gpu_set_chained_runs", "sqmc_gpu_spmv_prepare", "sqmc_gpu_davidson",
    "sqmc_gpu_spmv_apply", "sqmc_gpu_spmv_free", "sqmc_gpu_build_spmv_plan", "sqmc_gpu_spmv_sym_upper", "sqmc_gpu_hamiltonian_batch",
    "sqmc_gpu_propose_batch", "sqmc_gpu_hamiltonian_chem_batch", "sqmc_gpu_build_sparse_ham", "sqmc_gpu_hci_connections", "sqmc_gpu_hci_connections_slice", "sqmc_gpu_hci_pt2", "sqmc_gpu_hci_set_active_space", "sqmc_gpu_free", "sqmc_gpu_set_timing", "sqmc_gpu_get_timing",
]


class SqmcGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsqmc_gpu status %d: %s" % (code, msg))
        self.code = code


def build_library(force=False):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    csrc = os.path.join(_HERE, "csrc")
    # one translation unit: sqmc_gpu.hip includes every other file of csrc/ textually
    src = [os.path.join(csrc, "sqmc_gpu.hip")] + sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".inc"))) \
        + [os.path.join(_ROOT, "include", "sqmc_gpu.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in src):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value",
           "-I" + os.path.join(_ROOT, "include")] + os.environ.get("SQMC_EXTRA_CFLAGS", "").split() + [src[0], "-o", LIB_PATH]
    subprocess.check_call(cmd)
    return LIB_PATH


class ChemCfg(C.Structure):
    _fields_ = [("norb", C.c_int32), ("nup", C.c_int32), ("ndn", C.c_int32), ("n_core_orb", C.c_int32),
                ("time_sym", C.c_int32), ("z", C.c_int32), ("n_group", C.c_int32),
                ("product_table", C.c_void_p), ("orbital_symmetries", C.c_void_p), ("combine_2", C.c_void_p),
                ("n_integrals", C.c_int64), ("integrals", C.c_void_p), ("rng_mode", C.c_int32),
                ("irand_seed", C.c_int32 * 4), ("mwalk", C.c_int64)]


class HegCfg(C.Structure):
    _fields_ = [("n_dim", C.c_int32), ("norb", C.c_int32), ("nup", C.c_int32), ("ndn", C.c_int32), ("length_cell", C.c_double),
                ("k_vectors", C.c_void_p), ("rng_mode", C.c_int32), ("irand_seed", C.c_int32 * 4), ("mwalk", C.c_int64)]


class HubbardCfg(C.Structure):
    _fields_ = [("l_x", C.c_int32), ("l_y", C.c_int32), ("pbc", C.c_int32), ("nup", C.c_int32), ("ndn", C.c_int32),
                ("t", C.c_double), ("U", C.c_double), ("rng_mode", C.c_int32), ("irand_seed", C.c_int32 * 4), ("mwalk", C.c_int64)]


class StepParams(C.Structure):
    _fields_ = [("tau", C.c_double), ("e_trial", C.c_double), ("reweight_factor_inv", C.c_double), ("r_initiator", C.c_double),
                ("min_wt", C.c_double), ("always_spawn_cutoff_wt", C.c_double), ("initiator_power", C.c_int32),
                ("initiator_min_distance", C.c_int32), ("c_t_initiator", C.c_int32), ("semistochastic", C.c_int32),
                ("reached_w_abs_gen", C.c_int32), ("reserved", C.c_int32)]


class PopCtl(C.Structure):
    _fields_ = [("tau_sav", C.c_double), ("tau", C.c_double), ("tau_prev", C.c_double), ("e_trial", C.c_double), ("e_est", C.c_double),
                ("w_abs_gen_target", C.c_double), ("w_abs_gen", C.c_double), ("r_initiator_sav", C.c_double), ("r_initiator", C.c_double),
                ("initiator_rescale_power", C.c_double), ("population_control_exponent", C.c_double), ("reweight_factor_inv", C.c_double),
                ("reweight_factor_inv_max", C.c_double), ("e_num_cum", C.c_double), ("e_den_cum", C.c_double), ("min_wt", C.c_double),
                ("always_spawn_cutoff_wt", C.c_double), ("reached_w_abs_gen", C.c_int32), ("initiator_power", C.c_int32),
                ("initiator_min_distance", C.c_int32), ("c_t_initiator", C.c_int32), ("semistochastic", C.c_int32), ("reserved", C.c_int32),
                ("istep", C.c_int64), ("n_equil", C.c_int64)]


def load_library():
    """Loads the in-tree HIP library; raises if it is missing (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SqmcGpuError(-2, "sqmc_amd/libsqmc_gpu.so is not built: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.sqmc_gpu_last_error.restype = C.c_char_p
        for name in EXPORTS:
            getattr(L, name)   # AttributeError if the ABI lost a symbol
        L.sqmc_gpu_scale_projector.argtypes = [C.c_void_p, C.c_double]
        L.sqmc_gpu_set_hb_tables.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double]
        L.sqmc_gpu_set_projector.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sqmc_gpu_set_ct_table.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 4
        L.sqmc_gpu_upload_walkers.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 9
        L.sqmc_gpu_download_walkers.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 9
        L.sqmc_gpu_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sqmc_gpu_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.sqmc_gpu_det_owner.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.sqmc_gpu_shard_config.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
        L.sqmc_gpu_shard_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sqmc_gpu_shard_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.sqmc_gpu_shard_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.sqmc_gpu_hamiltonian_batch.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.sqmc_gpu_hamiltonian_chem_batch.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        L.sqmc_gpu_build_sparse_ham.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        L.sqmc_gpu_propose_batch.argtypes = [C.c_void_p, C.c_int64, C.c_double] + [C.c_void_p] * 7
        L.sqmc_gpu_hci_connections.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int] + [C.c_void_p] * 5
        L.sqmc_gpu_spmv_prepare.argtypes = [C.c_int64] + [C.c_void_p] * 4
        L.sqmc_gpu_spmv_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.sqmc_gpu_spmv_free.argtypes = [C.c_void_p]
        L.sqmc_gpu_spmv_sym_upper.argtypes = [C.c_int64] + [C.c_void_p] * 5
        L.sqmc_gpu_free.argtypes = [C.c_void_p]
        L.sqmc_gpu_finalize.argtypes = [C.c_void_p]
        L.sqmc_gpu_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.sqmc_gpu_get_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sqmc_gpu_get_rng.argtypes = [C.c_void_p, C.c_void_p]
        L.sqmc_gpu_set_rng.argtypes = [C.c_void_p, C.c_void_p]
        L.sqmc_gpu_num_walkers.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def set_device(index):
    L = load_library()
    L.sqmc_gpu_set_device.argtypes = [C.c_int]
    _chk(L.sqmc_gpu_set_device(int(index)))


def _chk(code):
    if code != 0:
        raise SqmcGpuError(code, load_library().sqmc_gpu_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class GpuChem:
    """One sqmc_gpu_ctx: chemistry tables + (optionally) the walker arrays in HBM."""

    def __init__(self, norb, nup, ndn, orbsym, product_table, combine_2, integrals, n_group=8, time_sym=False, z=1,
                 n_core_orb=0, rng_mode=RNG_COUNTER, seed=(1346, 5634, 6635, 4361), mwalk=0):
        L = load_library()
        self.L = L
        self._tabs = [np.ascontiguousarray(product_table, np.int32), np.ascontiguousarray(orbsym, np.int32),
                      np.ascontiguousarray(combine_2, np.int32), _f64(integrals)]
        cfg = ChemCfg()
        cfg.norb, cfg.nup, cfg.ndn, cfg.n_core_orb = norb, nup, ndn, n_core_orb
        cfg.time_sym, cfg.z, cfg.n_group = int(time_sym), z, n_group
        cfg.product_table, cfg.orbital_symmetries, cfg.combine_2 = (t.ctypes.data for t in self._tabs[:3])
        cfg.n_integrals, cfg.integrals = len(self._tabs[3]) - 1, self._tabs[3].ctypes.data
        cfg.rng_mode, cfg.mwalk = rng_mode, mwalk
        for i in range(4):
            cfg.irand_seed[i] = seed[i]
        h = C.c_void_p()
        _chk(L.sqmc_gpu_init_chem(C.byref(cfg), C.byref(h)))
        self.h, self.norb, self.mwalk = h, norb, mwalk

    @classmethod
    def heg(cls, n_dim, norb, nup, ndn, length_cell, k_vectors, rng_mode=RNG_COUNTER, seed=(1346, 5634, 6635, 4361), mwalk=0):
        """context for the homogeneous electron gas: k_vectors[norb, n_dim] (orbital-major)"""
        self = cls.__new__(cls)
        self.L = L = load_library()
        kv = _f64(np.asarray(k_vectors)[:, :n_dim])
        self._tabs = [kv]
        cfg = HegCfg()
        cfg.n_dim, cfg.norb, cfg.nup, cfg.ndn, cfg.length_cell = n_dim, norb, nup, ndn, float(length_cell)
        cfg.k_vectors, cfg.rng_mode, cfg.mwalk = kv.ctypes.data, rng_mode, mwalk
        for i in range(4):
            cfg.irand_seed[i] = seed[i]
        h = C.c_void_p()
        _chk(L.sqmc_gpu_init_heg(C.byref(cfg), C.byref(h)))
        self.h, self.norb, self.mwalk = h, norb, mwalk
        return self

    @classmethod
    def hubbard(cls, l_x, l_y, pbc, nup, ndn, t, U, rng_mode=RNG_COUNTER, seed=(1346, 5634, 6635, 4361), mwalk=0):
        """context for the real-space Hubbard model on an l_x by l_y lattice ('hubbard2')"""
        self = cls.__new__(cls)
        self.L = L = load_library()
        self._tabs = []
        cfg = HubbardCfg()
        cfg.l_x, cfg.l_y, cfg.pbc, cfg.nup, cfg.ndn, cfg.t, cfg.U = l_x, l_y, int(bool(pbc)), nup, ndn, float(t), float(U)
        cfg.rng_mode, cfg.mwalk = rng_mode, mwalk
        for i in range(4):
            cfg.irand_seed[i] = seed[i]
        h = C.c_void_p()
        _chk(L.sqmc_gpu_init_hubbard(C.byref(cfg), C.byref(h)))
        self.h, self.norb, self.mwalk = h, l_x * l_y, mwalk
        return self

    def close(self):
        if self.h:
            self.L.sqmc_gpu_finalize(self.h)
            self.h = None

    def set_hb_tables(self, hb_r, hb_s, hb_absH, pq_ind, pq_count, max_double):
        r, s = np.ascontiguousarray(hb_r, np.int32), np.ascontiguousarray(hb_s, np.int32)
        a, pi, pc = _f64(hb_absH), np.ascontiguousarray(pq_ind, np.int64), np.ascontiguousarray(pq_count, np.int32)
        _chk(self.L.sqmc_gpu_set_hb_tables(self.h, len(r), _p(r), _p(s), _p(a), len(pi) - 1, _p(pi), _p(pc), float(max_double)))

    def set_projector(self, counts, indices, values):
        c, i, v = np.ascontiguousarray(counts, np.int64), np.ascontiguousarray(indices, np.int64), _f64(values)
        _chk(self.L.sqmc_gpu_set_projector(self.h, len(c), len(v), _p(c), _p(i), _p(v)))

    def scale_projector(self, ratio):
        _chk(self.L.sqmc_gpu_scale_projector(self.h, float(ratio)))

    def set_ct_table(self, up, dn, num, den):
        u, d, n_, e = _u64(up), _u64(dn), _f64(num), _f64(den)
        _chk(self.L.sqmc_gpu_set_ct_table(self.h, len(u), _p(u), _p(d), _p(n_), _p(e)))

    def set_hf_to_psit(self, psit_ct_index, cdet_psi_t, diag_elems, sum_order=1):
        """hf_to_psit = .true.: after set_projector (matrix without its first row and column) and set_ct_table, before upload_walkers.
        psit_ct_index: 1-based positions of Psi_T's determinants (label order) in the C(T) list."""
        ix, cd, de = np.ascontiguousarray(psit_ct_index, np.int64), _f64(cdet_psi_t), _f64(diag_elems)
        self.L.sqmc_gpu_set_hf_to_psit.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        _chk(self.L.sqmc_gpu_set_hf_to_psit(self.h, len(ix), _p(ix), _p(cd), _p(de), int(sum_order)))

    def upload_walkers(self, w):
        arrs = [_u64(w["up"]), _u64(w["dn"]), _f64(w["wt"]), np.ascontiguousarray(w["imp_distance"], np.int8),
                np.ascontiguousarray(w["initiator"], np.int8), np.ascontiguousarray(w["perm_sign"], np.int8),
                _f64(w["matrix_elements"]), _f64(w["e_num"]), _f64(w["e_den"])]
        _chk(self.L.sqmc_gpu_upload_walkers(self.h, len(arrs[0]), *[_p(a) for a in arrs]))

    def num_walkers(self):
        n = C.c_int64()
        _chk(self.L.sqmc_gpu_num_walkers(self.h, C.byref(n)))
        return n.value

    def download_walkers(self):
        n = self.num_walkers()
        out = dict(up=np.zeros(n, np.uint64), dn=np.zeros(n, np.uint64), wt=np.zeros(n), imp_distance=np.zeros(n, np.int8),
                   initiator=np.zeros(n, np.int8), matrix_elements=np.zeros(n), e_num=np.zeros(n), e_den=np.zeros(n))
        nn = C.c_int64()
        _chk(self.L.sqmc_gpu_download_walkers(self.h, n, C.byref(nn), *[_p(out[k]) for k in
                                              ("up", "dn", "wt", "imp_distance", "initiator", "matrix_elements", "e_num", "e_den")]))
        return out

    def step(self, params):
        p = StepParams(**params)
        out = np.zeros(16)
        code = self.L.sqmc_gpu_step(self.h, C.byref(p), _p(out))
        _chk(code)
        return out

    def annihilate(self, params, spawns):
        """sqmc_gpu_annihilate: caller-supplied spawns (dict up/dn/wt/imp_distance/initiator) through
        sort, merge, rounding and the estimator sums"""
        p = StepParams(**params); out = np.zeros(16)
        arrs = [_u64(spawns["up"]), _u64(spawns["dn"]), _f64(spawns["wt"]), np.ascontiguousarray(spawns["imp_distance"], np.int8),
                np.ascontiguousarray(spawns["initiator"], np.int8)]
        self.L.sqmc_gpu_annihilate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        _chk(self.L.sqmc_gpu_annihilate(self.h, C.byref(p), len(arrs[0]), *[_p(a) for a in arrs], _p(out)))
        return out

    def run(self, pc, nsteps, keep_stats=True):
        """sqmc_gpu_run: nsteps steps with the population control done inside the library."""
        stats = np.zeros((nsteps, 16)) if keep_stats else None
        totals = np.zeros(16)
        _chk(self.L.sqmc_gpu_run(self.h, C.byref(pc), int(nsteps), _p(stats) if keep_stats else None, _p(totals)))
        return stats, totals

    # ---- multi-rank sharding (device pointers come from the caller's collective library)
    def det_owner(self, up, dn, nranks):
        u, d = _u64(up), _u64(dn)
        out = np.zeros(len(u), np.int32)
        _chk(self.L.sqmc_gpu_det_owner(self.h, len(u), _p(u), _p(d), int(nranks), _p(out)))
        return out

    def shard_config(self, rank, nranks, global_rows):
        gr = np.ascontiguousarray(global_rows, np.int32)
        _chk(self.L.sqmc_gpu_shard_config(self.h, int(rank), int(nranks), len(gr), _p(gr) if len(gr) else None))

    def shard_begin(self, params, x_global_ptr):
        p = StepParams(**params); n = C.c_int64()
        _chk(self.L.sqmc_gpu_shard_begin(self.h, C.byref(p), C.c_void_p(x_global_ptr), C.byref(n)))
        return n.value

    def shard_pack(self, params, x_global_ptr, send_ptr, cap_records, nranks):
        p = StepParams(**params); cnt = np.zeros(nranks, np.int64)
        _chk(self.L.sqmc_gpu_shard_pack(self.h, C.byref(p), C.c_void_p(x_global_ptr), C.c_void_p(send_ptr), int(cap_records), _p(cnt)))
        return cnt

    def shard_finish(self, params, recv_ptr, n_recv):
        p = StepParams(**params); out = np.zeros(16)
        _chk(self.L.sqmc_gpu_shard_finish(self.h, C.byref(p), C.c_void_p(recv_ptr), int(n_recv), _p(out)))
        return out

    # ---- the same with the exchanges inside the library (RCCL)
    @staticmethod
    def comm_unique_id():
        L = load_library(); buf = (C.c_uint8 * 128)()
        L.sqmc_gpu_comm_unique_id.argtypes = [C.c_void_p]
        _chk(L.sqmc_gpu_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self.L.sqmc_gpu_comm_init.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_comm_init(self.h, buf))

    def shard_time_split(self, reset=False):
        """host wall clock of the in-library sharded steps since the last reset: ({head, exchange, tail, waiting_for_gpu} in us per step, steps)"""
        us, n = (C.c_double * 4)(), C.c_int64()
        self.L.sqmc_gpu_shard_time_split.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        _chk(self.L.sqmc_gpu_shard_time_split(self.h, us, C.byref(n), 1 if reset else 0))
        k = max(n.value, 1)
        return {"head_us": us[0] / k, "exchange_us": us[1] / k, "tail_us": us[2] / k, "of_which_waiting_for_the_gpu_us": us[3] / k}, n.value

    def tail_stats(self):
        """(steps that took the short-list bucket tail, how many of them were re-run through the radix tail)"""
        a, b = C.c_int64(), C.c_int64()
        self.L.sqmc_gpu_tail_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_tail_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_chained_runs(self, on):
        """sqmc_gpu_set_chained_runs: keep the step pipeline primed across run() calls (block-structured hosts)"""
        self.L.sqmc_gpu_set_chained_runs.argtypes = [C.c_void_p, C.c_int32]
        _chk(self.L.sqmc_gpu_set_chained_runs(self.h, 1 if on else 0))

    def slowest_steps(self):
        """[(microseconds, step index)] of the four slowest steps of the last run() call"""
        us, st = (C.c_double * 4)(), (C.c_int64 * 4)()
        self.L.sqmc_gpu_slowest_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_slowest_steps(self.h, us, st))
        return [(round(us[k], 1), int(st[k])) for k in range(4)]

    def set_owner_hash(self, mode):
        """0: mix of the sort key (default); 1: the reference's djb_hash (mpi_routines.f90:354-379)"""
        self.L.sqmc_gpu_set_owner_hash.argtypes = [C.c_void_p, C.c_int32]
        _chk(self.L.sqmc_gpu_set_owner_hash(self.h, int(mode)))

    def comm_size(self):
        n = C.c_int32()
        self.L.sqmc_gpu_comm_size.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_comm_size(self.h, C.byref(n)))
        return n.value

    def shard_step(self, params):
        p = StepParams(**params); out = np.zeros(16)
        self.L.sqmc_gpu_shard_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_shard_step(self.h, C.byref(p), _p(out)))
        return out

    def shard_run(self, pc, nsteps, keep_stats=True):
        stats = np.zeros((nsteps, 16)) if keep_stats else None
        totals = np.zeros(16)
        self.L.sqmc_gpu_shard_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_shard_run(self.h, C.byref(pc), int(nsteps), _p(stats) if keep_stats else None, _p(totals)))
        return stats, totals

    def rng_state(self):
        s = (C.c_int32 * 4)()
        _chk(self.L.sqmc_gpu_get_rng(self.h, s))
        return list(s)

    def set_timing(self, level=2):
        _chk(self.L.sqmc_gpu_set_timing(self.h, int(level)))

    def timing(self):
        n = C.c_int32(); names = (C.c_char_p * 32)(); ms = (C.c_float * 32)()
        _chk(self.L.sqmc_gpu_get_timing(self.h, C.byref(n), names, ms))
        return [(names[i].decode(), ms[i]) for i in range(n.value)]

    def hamiltonian_batch(self, iu, id_, ju, jd):
        a, b, c, d = _u64(iu), _u64(id_), _u64(ju), _u64(jd)
        h = np.zeros(len(a))
        _chk(self.L.sqmc_gpu_hamiltonian_batch(self.h, len(a), _p(a), _p(b), _p(c), _p(d), _p(h)))
        return h

    def hamiltonian_chem_batch(self, iu, id_, ju, jd):
        a, b, c, d = _u64(iu), _u64(id_), _u64(ju), _u64(jd)
        h = np.zeros(len(a))
        _chk(self.L.sqmc_gpu_hamiltonian_chem_batch(self.h, len(a), _p(a), _p(b), _p(c), _p(d), _p(h)))
        return h

    def build_sparse_ham(self, up, dn):
        u, d = _u64(up), _u64(dn)
        n = len(u)
        nnz = C.c_int64(); prc = C.c_void_p(); pix = C.c_void_p(); pvl = C.c_void_p()
        _chk(self.L.sqmc_gpu_build_sparse_ham(self.h, n, _p(u), _p(d), C.byref(nnz), C.byref(prc), C.byref(pix), C.byref(pvl)))
        k = nnz.value
        rc = np.ctypeslib.as_array(C.cast(prc, C.POINTER(C.c_int64)), shape=(n,)).copy()
        ix = np.ctypeslib.as_array(C.cast(pix, C.POINTER(C.c_int64)), shape=(k,)).copy()
        vl = np.ctypeslib.as_array(C.cast(pvl, C.POINTER(C.c_double)), shape=(k,)).copy()
        for q in (prc, pix, pvl):
            self.L.sqmc_gpu_free(q)
        return rc, ix, vl

    def set_heatbath_tables(self, tabs):
        """tabs: dict with the reference's arrays in the reference's (Fortran, column-major) order -- see sqmc_heatbath_tables"""
        t = HeatbathTables()
        keep = []
        def put(name, arr, dt):
            a = np.ascontiguousarray(arr, dt).reshape(-1); keep.append(a)
            setattr(t, name, a.ctypes.data_as(C.c_void_p))
        t.norb = int(tabs["norb"])
        put("one", tabs["one"], np.float64); put("two", tabs["two"], np.float64)
        put("three_same", tabs["three_same"], np.float64); put("three_opp", tabs["three_opp"], np.float64)
        put("j3_same", tabs["j3_same"], np.int32); put("j3_opp", tabs["j3_opp"], np.int32)
        put("q3_same", tabs["q3_same"], np.float64); put("q3_opp", tabs["q3_opp"], np.float64)
        t.size_same, t.size_opp = int(tabs["size_same"]), int(tabs["size_opp"])
        put("four_same", tabs["four_same"], np.float32); put("four_opp", tabs["four_opp"], np.float32)
        put("j4_same", tabs["j4_same"], np.int32); put("j4_opp", tabs["j4_opp"], np.int32)
        put("q4_same", tabs["q4_same"], np.float32); put("q4_opp", tabs["q4_opp"], np.float32)
        put("htot_same", tabs["htot_same"], np.float64); put("htot_opp", tabs["htot_opp"], np.float64)
        self.L.sqmc_gpu_set_heatbath_tables.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_set_heatbath_tables(self.h, C.byref(t)))

    def setup_efficient_heatbath(self):
        """setup_efficient_heatbath + check_heatbath_unbiased done by the library (proposal_method fast_heatbath from here on if the
        check passes).  Returns is_heatbath_unbiased."""
        ok = C.c_int32()
        self.L.sqmc_gpu_setup_efficient_heatbath.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_setup_efficient_heatbath(self.h, C.byref(ok)))
        return bool(ok.value)

    def heatbath_tables(self):
        """the tables the library built (copies, the reference's layout as in sqmc_heatbath_tables) and the number of orbitals of unique symmetry"""
        t, nu = HeatbathTables(), C.c_int32()
        self.L.sqmc_gpu_get_heatbath_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_get_heatbath_tables(self.h, C.byref(t), C.byref(nu)))
        n = t.norb
        npairs = (n * (n - 1)) // 2 + n

        def take(ptr, count, dt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(count,)).copy()
        out = dict(norb=n, size_same=t.size_same, size_opp=t.size_opp, n_orb_uniq_sym=nu.value,
                   one=take(t.one, n, C.c_double), two=take(t.two, 4 * n * n, C.c_double),
                   three_same=take(t.three_same, n ** 3, C.c_double), three_opp=take(t.three_opp, n ** 3, C.c_double),
                   j3_same=take(t.j3_same, n ** 3, C.c_int32), j3_opp=take(t.j3_opp, n ** 3, C.c_int32),
                   q3_same=take(t.q3_same, n ** 3, C.c_double), q3_opp=take(t.q3_opp, n ** 3, C.c_double),
                   four_same=take(t.four_same, t.size_same, C.c_float), four_opp=take(t.four_opp, t.size_opp, C.c_float),
                   j4_same=take(t.j4_same, t.size_same, C.c_int32), j4_opp=take(t.j4_opp, t.size_opp, C.c_int32),
                   q4_same=take(t.q4_same, t.size_same, C.c_float), q4_opp=take(t.q4_opp, t.size_opp, C.c_float),
                   htot_same=take(t.htot_same, npairs * n, C.c_double), htot_opp=take(t.htot_opp, n ** 3, C.c_double))
        return out

    def propose_heatbath_batch(self, tau, up, dn, seeds):
        u, d = _u64(up), _u64(dn)
        s = np.ascontiguousarray(seeds, np.int32).reshape(-1)
        n = len(u)
        ju, jd, wj, sa = np.zeros(2 * n, np.uint64), np.zeros(2 * n, np.uint64), np.zeros(2 * n), np.zeros(4 * n, np.int32)
        self.L.sqmc_gpu_propose_heatbath_batch.argtypes = [C.c_void_p, C.c_int64, C.c_double] + [C.c_void_p] * 7
        _chk(self.L.sqmc_gpu_propose_heatbath_batch(self.h, n, float(tau), _p(u), _p(d), _p(s), _p(ju), _p(jd), _p(wj), _p(sa)))
        return ju.reshape(n, 2), jd.reshape(n, 2), wj.reshape(n, 2), sa.reshape(n, 4)

    def propose_batch(self, tau, up, dn, seeds):
        u, d = _u64(up), _u64(dn)
        s = np.ascontiguousarray(seeds, np.int32).reshape(-1)
        n = len(u)
        ju, jd, wj, sa = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n), np.zeros(4 * n, np.int32)
        _chk(self.L.sqmc_gpu_propose_batch(self.h, n, float(tau), _p(u), _p(d), _p(s), _p(ju), _p(jd), _p(wj), _p(sa)))
        return ju, jd, wj, sa.reshape(n, 4)

    def hci_connections(self, ref_up, ref_dn, coeffs, eps, diag_mode=0, slice=0, n_slices=1):
        u, d, c = _u64(ref_up), _u64(ref_dn), _f64(coeffs)
        n = C.c_int64(); pu = C.c_void_p(); pd = C.c_void_p(); pn = C.c_void_p(); pe = C.c_void_p()
        self.L.sqmc_gpu_hci_connections_slice.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                                          C.c_int32, C.c_int32] + [C.c_void_p] * 5
        _chk(self.L.sqmc_gpu_hci_connections_slice(self.h, len(u), _p(u), _p(d), _p(c), float(eps), int(diag_mode), int(slice), int(n_slices),
                                                   C.byref(n), C.byref(pu), C.byref(pd), C.byref(pn), C.byref(pe)))
        k = n.value
        if k == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0), np.zeros(0)
        def take(ptr, ct, dt):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(max(k, 1),))[:k].astype(dt, copy=True)
            self.L.sqmc_gpu_free(ptr)
            return a
        return take(pu, C.c_uint64, np.uint64), take(pd, C.c_uint64, np.uint64), take(pn, C.c_double, np.float64), take(pe, C.c_double, np.float64)


def _gpuchem_hci_set_active_space(self, core_up, core_dn, virt_up, virt_dn, mode):
    """masks of the HCI generator: mode 0 none, 1 inside the active space only, 2 outside only"""
    self.L.sqmc_gpu_hci_set_active_space.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32]
    _chk(self.L.sqmc_gpu_hci_set_active_space(self.h, int(core_up), int(core_dn), int(virt_up), int(virt_dn), int(mode)))


def _gpuchem_hci_pt2(self, up, dn, coeffs, e_var, eps_pt, n_slices=1):
    """sqmc_gpu_hci_pt2: (delta_E, number of connected determinants), everything on the device"""
    u, d, c = _u64(up), _u64(dn), _f64(coeffs)
    de, nc = C.c_double(), C.c_int64()
    self.L.sqmc_gpu_hci_pt2.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
    _chk(self.L.sqmc_gpu_hci_pt2(self.h, len(u), _p(u), _p(d), _p(c), float(e_var), float(eps_pt), int(n_slices), C.byref(de), C.byref(nc)))
    return de.value, nc.value


class HeatbathTables(C.Structure):
    """sqmc_heatbath_tables of include/sqmc_gpu.h"""
    _fields_ = [("norb", C.c_int32), ("reserved", C.c_int32), ("one", C.c_void_p), ("two", C.c_void_p), ("three_same", C.c_void_p), ("three_opp", C.c_void_p),
                ("j3_same", C.c_void_p), ("j3_opp", C.c_void_p), ("q3_same", C.c_void_p), ("q3_opp", C.c_void_p), ("size_same", C.c_int64), ("size_opp", C.c_int64),
                ("four_same", C.c_void_p), ("four_opp", C.c_void_p), ("j4_same", C.c_void_p), ("j4_opp", C.c_void_p), ("q4_same", C.c_void_p), ("q4_opp", C.c_void_p),
                ("htot_same", C.c_void_p), ("htot_opp", C.c_void_p)]


class SpmvPlan:
    """sqmc_gpu_spmv_prepare/apply: symmetric matvec of davidson_sparse (more_tools.f90:2115)."""

    def __init__(self, counts, indices, values):
        self.L = load_library()
        c, i, v = np.ascontiguousarray(counts, np.int64), np.ascontiguousarray(indices, np.int64), _f64(values)
        self.n = len(c)
        h = C.c_void_p()
        _chk(self.L.sqmc_gpu_spmv_prepare(self.n, _p(c), _p(i), _p(v), C.byref(h)))
        self.h = h

    @classmethod
    def from_dets(cls, g, up, dn):
        """sqmc_gpu_build_spmv_plan: Hamiltonian of a sorted determinant list built and kept on the GPU.
        Returns (plan, diagonal, stored upper-triangular nonzeros)."""
        self = cls.__new__(cls)
        self.L = g.L
        u, d = _u64(up), _u64(dn)
        self.n = len(u)
        diag = np.zeros(self.n); nnz = C.c_int64(); h = C.c_void_p()
        self.L.sqmc_gpu_build_spmv_plan.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_build_spmv_plan(g.h, self.n, _p(u), _p(d), C.byref(h), _p(diag), C.byref(nnz)))
        self.h = h
        return self, diag, nnz.value

    def apply(self, x):
        x = _f64(x); y = np.zeros(self.n)
        _chk(self.L.sqmc_gpu_spmv_apply(self.h, _p(x), _p(y), 0))
        return y

    def davidson(self, diag, k=1, v0=None, tol=1e-10):
        """sqmc_gpu_davidson: lowest k eigenpairs, basis and products resident on the device.  Returns (eigenvalues[k], vectors[n, k], matvecs)"""
        d = _f64(diag)
        ev = np.zeros(k); X = np.zeros((k, self.n)); nm = C.c_int32()
        s = None if v0 is None else np.ascontiguousarray(np.asarray(v0, float).reshape(self.n, -1)[:, :k].T)       # column-major n x k
        self.L.sqmc_gpu_davidson.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sqmc_gpu_davidson(self.h, _p(d), int(k), None if s is None else _p(s), float(tol), _p(ev), _p(X), C.byref(nm)))
        return ev, np.ascontiguousarray(X.T), nm.value

    def close(self):
        if self.h:
            self.L.sqmc_gpu_spmv_free(self.h)
            self.h = None


GpuChem.hci_pt2 = _gpuchem_hci_pt2
GpuChem.hci_set_active_space = _gpuchem_hci_set_active_space
