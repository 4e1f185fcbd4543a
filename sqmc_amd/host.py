"""Host side of the hot path: the run-once input handling the reference does on the CPU
(FCIDUMP, orbital reordering, heat-bath table, trial wave function, deterministic space)
and the scalar population-control logic of the walk loop.  Everything that touches more
than a few thousand numbers goes through the HIP library; nothing here is a fallback.

Mirrors (file:line relative to the reference's src/):
  read_integrals / sort_integrals        chemistry.f90:538-869, 8921-9022
  compute_orbital_energies               chemistry.f90:9378-9442
  setup_efficient_heatbath (hci part)    chemistry.f90:900-993
  generate_space_iterate (one iteration) semistoch.f90:145-560
  generate_psi_t_connected_e_loc         semistoch.f90:27-133
  initial population                     do_walk.f90:1245-1366
  tau ramp, e_trial, reweight factor     do_walk.f90:2171-2184, 2880-2923
  davidson_sparse                        more_tools.f90:2018-2244
"""
import numpy as np

from ._lib import GpuChem, SpmvPlan, PopCtl, RNG_COUNTER

_D2H = np.array([[1, 2, 3, 4, 5, 6, 7, 8], [2, 1, 4, 3, 6, 5, 8, 7], [3, 4, 1, 2, 7, 8, 5, 6], [4, 3, 2, 1, 8, 7, 6, 5],
                 [5, 6, 7, 8, 1, 2, 3, 4], [6, 5, 8, 7, 2, 1, 4, 3], [7, 8, 5, 6, 3, 4, 1, 2], [8, 7, 6, 5, 4, 3, 2, 1]])
_GROUP_ORDER = {"c1": 1, "cs": 2, "c2v": 4, "c2h": 4, "d2h": 8}


def read_fcidump(path):
    """Header (NORB, ORBSYM) + the (value, p, q, r, s) records."""
    hdr, rows = "", []
    with open(path) as f:
        for line in f:
            hdr += line
            if "&END" in line.upper() or "/" in line:
                break
        data = np.loadtxt(f, ndmin=2)
    import re
    norb = int(re.search(r"NORB\s*=\s*(\d+)", hdr).group(1))
    m = re.search(r"ORBSYM\s*=\s*([\d,\s]+)", hdr)
    orbsym = [int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()][:norb] if m else [1] * norb
    return norb, orbsym, data[:, 0].copy(), data[:, 1:5].astype(np.int64)


def _dets_to_bits(det):
    return [k for k in range(64) if (int(det) >> k) & 1]


class ChemHost:
    """Tables of module chemistry for one FCIDUMP, in the reference's conventions."""

    def __init__(self, fcidump, nelec, nup, point_group="d2h", time_sym=False, z=1, n_core_orb=0, hf_symmetry=None):
        """hf_symmetry: the `&hf_det hf_symmetry=k /` line of the HCI decks -- the starting determinant
        is then found by the reference's descent (needs the GPU for the matrix elements); None keeps
        the first orbitals of the file (the walk decks)."""
        self.norb, orbsym, vals, idx = read_fcidump(fcidump)
        n, n1 = self.norb, self.norb + 1
        self.nelec, self.nup, self.ndn = nelec, nup, nelec - nup
        self.time_sym, self.z, self.n_core_orb = bool(time_sym), z, n_core_orb
        self.n_group = _GROUP_ORDER[point_group]
        self.prod = np.zeros((9, 9), np.int32)
        self.prod[1:self.n_group + 1, 1:self.n_group + 1] = _D2H[:self.n_group, :self.n_group]
        # identity-order combine_2 while reading (chemistry.f90:384-394)
        c2 = np.zeros((n + 2, n + 2), np.int32)
        i, j = np.meshgrid(np.arange(1, n1), np.arange(1, n1), indexing="ij")
        hi, lo = np.maximum(i, j), np.minimum(i, j)
        c2[1:n1, 1:n1] = (hi * (hi - 1)) // 2 + lo
        c2[n1, n1] = (n1 * n) // 2 + n1
        self._c2_id = c2
        self.n_int = self._index(c2, n1, n1, n1, n1)
        ints = np.zeros(self.n_int + 1)
        p = idx.copy(); p[p == 0] = n1
        keep = np.abs(vals) > 1e-9
        ii = self._index(c2, p[:, 0], p[:, 1], p[:, 2], p[:, 3])
        for k in np.nonzero(keep)[0]:          # later records overwrite earlier ones, as in the reference
            ints[ii[k]] = vals[k]
        self.integrals = ints
        self.orbsym_file = np.array([0] + list(orbsym), np.int32)
        # starting determinant: first orbitals (chemistry.f90:700-712)
        hf_up, hf_dn = (1 << self.nup) - 1, (1 << self.ndn) - 1
        if hf_symmetry is not None:
            hf_up, hf_dn = self._auto_hf(hf_up, hf_dn, int(hf_symmetry))
        self._finish(hf_up, hf_dn)

    def _auto_hf(self, hu, hd, hf_symmetry):
        """auto_assign_hci0_occs with a starting determinant (chemistry.f90:10359-10522): move to the
        lowest-diagonal determinant of total symmetry hf_symmetry among the current determinant and
        all its single and double excitations (first one wins a tie, in the generation order of
        find_connected_dets_chem), until nothing lower is found.  File orbital order."""
        self.orbsym, self.combine_2 = self.orbsym_file.copy(), self._c2_id
        g = self.gpu()
        try:
            du = dd = 0
            while (hu, hd) != (du, dd):
                if du != 0:
                    hu, hd = du, dd
                cand = [(a, b) for a, b in self._connected_in_order(hu, hd, hf_symmetry)
                        if not (self.time_sym and self.z < 0 and a == b)]
                cu = np.array([a for a, _ in cand], np.uint64); cd = np.array([b for _, b in cand], np.uint64)
                e = g.hamiltonian_batch(cu, cd, cu, cd)
                k = int(np.argmin(e))                     # first minimum
                du, dd = cand[k]
        finally:
            g.close()
        return hu, hd

    def _det_sym(self, up, dn):
        sym = 1
        for det in (up, dn):
            for k in _dets_to_bits(det):
                sym = int(self.prod[sym, self.orbsym[k + 1]])
        return sym

    def _connected_in_order(self, up, dn, sym_filter=0):
        """find_connected_dets_chem, chemistry.f90:6471-6815, in its own order: the determinant, up-up,
        dn-dn and up-dn doubles, up singles, dn singles.  sym_filter = 0: excitations allowed by the
        irreps of the orbitals involved (the matrix element may still vanish); > 0: every excitation
        whose determinant has that total symmetry."""
        n, nc, os_, pr = self.norb, self.n_core_orb, self.orbsym, self.prod
        fu = [i for i in range(n) if (up >> i) & 1]; eu = [i for i in range(n) if not (up >> i) & 1]
        fd = [i for i in range(n) if (dn >> i) & 1]; ed = [i for i in range(n) if not (dn >> i) & 1]
        sy = lambda o: int(os_[o + 1])
        ok = (lambda a, b, pair: self._det_sym(a, b) == sym_filter) if sym_filter else (lambda a, b, pair: pair)
        out = [(up, dn)]
        for occ, emp, is_up in ((fu, eu, True), (fd, ed, False)):
            det = up if is_up else dn
            for a in range(nc, len(occ) - 1):
                for b in range(a + 1, len(occ)):
                    ps = pr[sy(occ[a]), sy(occ[b])]
                    base = det & ~(1 << occ[a]) & ~(1 << occ[b])
                    for k in range(len(emp) - 1):
                        for l in range(k + 1, len(emp)):
                            t = base | (1 << emp[k]) | (1 << emp[l])
                            d2 = (t, dn) if is_up else (up, t)
                            if ok(d2[0], d2[1], ps == pr[sy(emp[k]), sy(emp[l])]):
                                out.append(d2)
        for a in range(nc, len(fu)):
            for b in range(nc, len(fd)):
                ps = pr[sy(fu[a]), sy(fd[b])]
                for k in eu:
                    tu = (up & ~(1 << fu[a])) | (1 << k)
                    for l in ed:
                        td = (dn & ~(1 << fd[b])) | (1 << l)
                        if ok(tu, td, ps == pr[sy(k), sy(l)]):
                            out.append((tu, td))
        for a in range(nc, len(fu)):
            for k in eu:
                t = (up & ~(1 << fu[a])) | (1 << k)
                if ok(t, dn, sy(fu[a]) == sy(k)):
                    out.append((t, dn))
        for a in range(nc, len(fd)):
            for k in ed:
                t = (dn & ~(1 << fd[a])) | (1 << k)
                if ok(up, t, sy(fd[a]) == sy(k)):
                    out.append((up, t))
        return out

    @staticmethod
    def _index(c2, i, j, k, l):
        a, b = c2[i, j].astype(np.int64) if hasattr(i, "__len__") else int(c2[i, j]), c2[k, l].astype(np.int64) if hasattr(k, "__len__") else int(c2[k, l])
        hi, lo = np.maximum(a, b), np.minimum(a, b)
        return (hi * (hi - 1)) // 2 + lo

    def _iv(self, c2, p, q, r, s):
        return self.integrals[self._index(c2, p, q, r, s)]

    def _orbital_energies(self, hf_up, hf_dn):
        n, n1, c2 = self.norb, self.norb + 1, self._c2_id
        oe = np.zeros(n + 1)
        for i in range(1, n + 1):
            ex = di = 0.0
            for j in range(1, n + 1):
                if j != i and (hf_up >> (j - 1)) & 1: ex = ex - self._iv(c2, i, j, j, i)
                if j != i and (hf_dn >> (j - 1)) & 1: ex = ex - self._iv(c2, i, j, j, i)
            for j in range(1, n + 1):
                if j != i and (hf_up >> (j - 1)) & 1: di = di + self._iv(c2, i, i, j, j)
            for j in range(1, n + 1):
                if (hf_dn >> (j - 1)) & 1: di = di + self._iv(c2, i, i, j, j)
            for j in range(1, n + 1):
                if j != i and (hf_dn >> (j - 1)) & 1: di = di + self._iv(c2, i, i, j, j)
            for j in range(1, n + 1):
                if (hf_up >> (j - 1)) & 1: di = di + self._iv(c2, i, i, j, j)
            oe[i] = self._iv(c2, i, i, n1, n1) + .5 * (ex + di)
        return oe

    def _finish(self, hf_up, hf_dn):
        """sort_integrals: orbital order by (occupied first, then orbital energy)."""
        n, n1 = self.norb, self.norb + 1
        oe = self._orbital_energies(hf_up, hf_dn)
        tmp = oe.copy()
        for i in range(1, n + 1):
            if (hf_up >> (i - 1)) & 1: tmp[i] -= 1e9
            if (hf_dn >> (i - 1)) & 1: tmp[i] -= 1e9
        order, inv = np.zeros(n + 2, np.int32), np.zeros(n + 2, np.int32)
        for i in range(1, n + 1):
            j = 1 + int(np.argmin(tmp[1:]))
            order[i], inv[j], tmp[j] = j, i, 1e99
        order[n1] = inv[n1] = n1
        self.orb_order, self.orb_order_inv = order, inv
        self.orbsym = np.zeros(n + 1, np.int32); self.orbsym[1:] = self.orbsym_file[order[1:n1]]
        self.orbital_energies = np.zeros(n + 1); self.orbital_energies[1:] = oe[order[1:n1]]
        nu = sum(1 << (int(inv[k + 1]) - 1) for k in _dets_to_bits(hf_up))
        nd = sum(1 << (int(inv[k + 1]) - 1) for k in _dets_to_bits(hf_dn))
        if self.time_sym and nd < nu:
            nu, nd = nd, nu
        self.hf_up, self.hf_dn = nu, nd
        a = order[1:n1][:, None].astype(np.int64); b = order[1:n1][None, :].astype(np.int64)
        hi, lo = np.maximum(a, b), np.minimum(a, b)
        c2 = np.zeros((n + 2, n + 2), np.int32)
        c2[1:n1, 1:n1] = (hi * (hi - 1)) // 2 + lo
        c2[n1, n1] = (n1 * n) // 2 + n1
        self.combine_2 = c2

    def gpu(self, **kw):
        return GpuChem(self.norb, self.nup, self.ndn, self.orbsym, self.prod.reshape(-1), self.combine_2.reshape(-1), self.integrals,
                       n_group=self.n_group, time_sym=self.time_sym, z=self.z, n_core_orb=self.n_core_orb, **kw)

    def connected_all(self, up, dn):
        """Every symmetry-allowed single and double excitation of one determinant plus itself, sorted
        and unique (with time-reversal symmetry: the representatives up <= dn).  Membership is decided
        by the irreps alone; the matrix element may be zero, which the heat-bath lists would skip."""
        out = self._connected_in_order(up, dn)
        if self.time_sym:
            out = [(min(a, b), max(a, b)) for a, b in out]
        keys = sorted(set(out))
        return np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint64)

    def diag_lowest_highest(self, g):
        nc, n = self.n_core_orb, self.norb
        mk = lambda k: (1 << k) - 1
        mu = mk(n) - mk(n + nc - self.nup) + mk(nc)
        md = mk(n) - mk(n + nc - self.ndn) + mk(nc)
        lo = g.hamiltonian_batch([self.hf_up], [self.hf_dn], [self.hf_up], [self.hf_dn])[0]
        hi = g.hamiltonian_chem_batch([mu], [md], [mu], [md])[0]
        return float(lo), float(hi)

    def setup_walk(self, g, n_truncate_trial_wf=100, size_deterministic=1000, tau_multiplier=0.1, rediagonalize=False):
        """Psi_T, C(T), deterministic space for a walk of this molecule on context g"""
        if not hasattr(self, "hb"):
            self.hb_tables(g)
        g.set_hb_tables(*self.hb)
        return setup_walk(self, g, n_truncate_trial_wf, size_deterministic, tau_multiplier, rediagonalize)

    def hb_tables(self, g):
        """dtm_hb: for every electron-pair class (p,q) the (r,s,|H|) list, |H| descending.
        Matrix elements come from the GPU (hamiltonian_chem on two-electron determinants,
        like double_excitation_matrix_element_no_ref)."""
        n = self.norb
        c2i = lambda i, j: (max(i, j) * (max(i, j) - 1)) // 2 + min(i, j)
        n_pq = c2i(n, 2 * n)
        by_sym = {sy: [int(o) for o in np.nonzero(self.orbsym[1:] == sy)[0] + 1] for sy in range(1, self.n_group + 1)}
        classes, cand_cls, cand_r, cand_s, dets = [], [], [], [], []
        for opposite in (0, 1):
            for p in range(1, n + 1):
                for q in (range(n + p, 2 * n + 1) if opposite else range(p + 1, n + 1)):
                    sym_q = self.prod[self.orbsym[p], self.orbsym[q - n if opposite else q]]
                    ci = len(classes); classes.append(c2i(p, q))
                    for r in range(1, n + 1):
                        for s_ in by_sym[int(self.prod[sym_q, self.orbsym[r]])]:
                            if not opposite and s_ < r:
                                continue
                            ss = s_ + n if opposite else s_
                            if len({p, q, r, ss}) < 4:
                                continue
                            cand_cls.append(ci); cand_r.append(r); cand_s.append(ss)
                            if opposite:
                                dets.append((1 << (p - 1), 1 << (q - n - 1), 1 << (r - 1), 1 << (ss - n - 1)))
                            else:
                                dets.append(((1 << (p - 1)) | (1 << (q - 1)), 0, (1 << (r - 1)) | (1 << (s_ - 1)), 0))
        D = np.array(dets, np.uint64)
        h = np.abs(g.hamiltonian_chem_batch(D[:, 0], D[:, 1], D[:, 2], D[:, 3]))
        cls, rr, ss = np.array(cand_cls), np.array(cand_r, np.int32), np.array(cand_s, np.int32)
        nz = h != 0.0
        cls, rr, ss, h = cls[nz], rr[nz], ss[nz], h[nz]
        order = np.lexsort((np.arange(len(h)), -h, cls))       # class, |H| descending, generation order on ties
        cls, rr, ss, h = cls[order], rr[order], ss[order], h[order]
        per = np.bincount(cls, minlength=len(classes))
        start = np.concatenate(([0], np.cumsum(per)))
        pq_ind, pq_count = np.zeros(n_pq + 1, np.int64), np.zeros(n_pq + 1, np.int32)
        for ci, e in enumerate(classes):
            pq_ind[e], pq_count[e] = start[ci] + 1, per[ci]
        self.hb = (rr, ss, h, pq_ind, pq_count, float(h.max()))
        return self.hb


# ----------------------------------------------------------------------------- Davidson
def davidson_lowest(plan, diag, k=1, v0=None, tol=1e-10):
    """Lowest k eigenpairs of the plan's matrix: sqmc_gpu_davidson (csrc/davidson.inc), the reference's own iteration
    (davidson_sparse, more_tools.f90:2018-2244) with basis, products and correction vectors resident on the device."""
    w, X, _ = plan.davidson(diag, k=k, v0=v0, tol=tol)
    return w, X


def _key64(up, dn):
    """(up, dn) as one 64-bit sort key when both strings fit 32 bits (up to 32 orbitals), else None"""
    up, dn = np.asarray(up, np.uint64), np.asarray(dn, np.uint64)
    if len(up) and (int(up.max()) >> 32 or int(dn.max()) >> 32):
        return None
    return (up << np.uint64(32)) | dn


def sort_dets(up, dn):
    k = _key64(up, dn)
    return np.lexsort((dn, up)) if k is None else np.argsort(k, kind="stable")


def _dets_in(au, ad, bu, bd):
    """mask over the determinants (au, ad): which of them occur in the list (bu, bd); repeats allowed on both sides"""
    ka, kb = _key64(au, ad), _key64(bu, bd)
    if ka is not None and kb is not None:
        return np.isin(ka, kb)
    n = len(au)
    u = np.concatenate((au, bu)); d = np.concatenate((ad, bd))
    o = np.lexsort((d, u))
    us, ds = u[o], d[o]
    new = np.ones(len(o), bool)
    new[1:] = (us[1:] != us[:-1]) | (ds[1:] != ds[:-1])
    gid = np.empty(len(o), np.int64)
    gid[o] = np.cumsum(new)                          # one integer per distinct determinant
    return np.isin(gid[:n], gid[n:])



def lowest_state(g, up, dn, k=1, v0=None):
    """Sparse H on the GPU (all-pairs builder) + Davidson with the GPU matvec."""
    counts, idx, val = g.build_sparse_ham(up, dn)
    starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
    diag = val[starts]
    plan = SpmvPlan(counts, idx, val)
    try:
        w, X = davidson_lowest(plan, diag, k=k, v0=v0)
    finally:
        plan.close()
    return w, X, (counts, idx, val)


def rediagonalize_in_own_space(g, up, dn, guess):
    """Lowest eigenpair of H among the determinants (up, dn) (label order): the 'Finally, rediagonalize' of generate_space_iterate
    (semistoch.f90:575, 706-712).  A trial wave function has tens to thousands of determinants: LAPACK on the dense matrix below
    4000 (real_symmetric_diagonalize is what the reference itself uses for its small cases, semistoch.f90:1037-1043), the GPU
    Davidson from `guess` above."""
    n = len(up)
    if n > 4000:
        w, X, _ = lowest_state(g, up, dn, v0=np.asarray(guess, float).reshape(-1, 1))
        return float(w[0]), X[:, 0]
    counts, idx, val = g.build_sparse_ham(up, dn)
    A = np.zeros((n, n))
    r = np.repeat(np.arange(n), counts)
    A[r, idx - 1] = val
    A[idx - 1, r] = val
    w, X = np.linalg.eigh(A)
    return float(w[0]), X[:, 0]


def _truncate_at_csf(c_sorted, n_keep, eps=1e-10):
    prev = 0.0
    for i, v in enumerate(c_sorted):
        if abs(abs(prev) - abs(v)) > eps:
            prev = v
            if i + 1 > n_keep:
                return i
    return len(c_sorted)


class WalkSetup:
    pass


def setup_walk(host, g, n_truncate_trial_wf=100, size_deterministic=1000, tau_multiplier=0.1, rediagonalize=False):
    """Psi_T, C(T) and the deterministic space from one connect-diagonalise-truncate pass.
    rediagonalize: the trial wave function is the lowest eigenvector of H among its own determinants, as generate_space_iterate
    leaves it (semistoch.f90:575, 706-712) -- what hf_to_psit needs; without it the truncated vector is only renormalised."""
    s = WalkSetup()
    tiny = 1e-300
    # the first-order space of the HF determinant by symmetry alone (zero matrix elements included:
    # such determinants still couple to the rest of the space), as the reference builds it
    up, dn = host.connected_all(host.hf_up, host.hf_dn)                            # sorted, unique
    w, X, _ = lowest_state(g, up, dn)
    c = X[:, 0]
    if c[np.argmax(np.abs(c))] < 0:
        c = -c
    by = np.argsort(-np.abs(c), kind="stable")
    up_s, dn_s, c_s = up[by], dn[by], c[by]
    n_t, n_i = _truncate_at_csf(c_s, n_truncate_trial_wf), _truncate_at_csf(c_s, size_deterministic)
    s.psi_up, s.psi_dn = up_s[:n_t].copy(), dn_s[:n_t].copy()
    s.psi_c = c_s[:n_t] / np.sqrt(np.dot(c_s[:n_t], c_s[:n_t]))
    if rediagonalize and n_t > 1:
        ol = sort_dets(s.psi_up, s.psi_dn)
        s.psi_up, s.psi_dn = s.psi_up[ol], s.psi_dn[ol]
        e_t, cr = rediagonalize_in_own_space(g, s.psi_up, s.psi_dn, s.psi_c[ol])
        s.psi_c, s.e_psi_t = (-cr if cr[np.argmax(np.abs(cr))] < 0 else cr), e_t
    o = sort_dets(up_s[:n_i], dn_s[:n_i])
    s.imp_up, s.imp_dn = up_s[:n_i][o].copy(), dn_s[:n_i][o].copy()
    lo, hi = host.diag_lowest_highest(g)
    s.tau, s.e_var = tau_multiplier / (hi - lo), float(w[0])
    pc, pi, pv = g.build_sparse_ham(s.imp_up, s.imp_dn)
    s.prj_counts, s.prj_indices, s.prj_values = pc, pi, -s.tau * pv
    s.ct_up, s.ct_dn, s.ct_num, s.ct_den = g.hci_connections(s.psi_up, s.psi_dn, s.psi_c, tiny, diag_mode=1)
    s.e_trial0 = float(np.dot(s.ct_num, s.ct_den) / np.dot(s.ct_den, s.ct_den))
    return s


def initial_walkers(s, w_abs_gen_begin, r_initiator=1.0, initiator_power=0):
    """do_walk.f90:1245-1366: deterministic-space dets (weight 0, initiator 2, imp_distance 0)
    + Psi_T dets (weight w_begin*c/sum|c|, permanent initiators where |c| ~ max|c|), equal
    determinants combined, sorted by (up,dn)."""
    cmax, csum = np.max(np.abs(s.psi_c)), np.sum(np.abs(s.psi_c))
    recs = {(int(a), int(b)): [0.0, 2, 0, 0] for a, b in zip(s.imp_up, s.imp_dn)}
    scale = min(w_abs_gen_begin * cmax / csum, 1.0)
    for a, b, c in zip(s.psi_up.tolist(), s.psi_dn.tolist(), s.psi_c.tolist()):
        wt, perm = (w_abs_gen_begin * c / csum) / scale, abs(abs(c) - cmax) < 1e-3
        r = recs.get((a, b))
        if r is None:
            recs[(a, b)] = [wt, 3 if perm else 2, 0 if perm else 1, int(np.sign(c)) if perm else 0]
        else:
            r[0] += wt
            if perm:
                r[1], r[3] = 3, int(np.sign(c))
    keys = sorted(recs)
    n = len(keys)
    out = dict(up=np.array([k[0] for k in keys], np.uint64), dn=np.array([k[1] for k in keys], np.uint64),
               wt=np.array([recs[k][0] for k in keys]), initiator=np.array([recs[k][1] for k in keys], np.int8),
               imp_distance=np.array([recs[k][2] for k in keys], np.int8), perm_sign=np.array([recs[k][3] for k in keys], np.int8),
               matrix_elements=np.full(n, 1e51), e_num=np.full(n, 1e51), e_den=np.full(n, 1e51))
    d, ini, aw = out["imp_distance"].astype(int), out["initiator"], np.abs(out["wt"])
    thr = r_initiator * np.where(initiator_power == 0, 1.0, np.maximum(d, 0).astype(float) ** initiator_power)
    ini[(ini == 2) & (aw <= thr) & (d > 0)] = 1
    keep = ~((out["wt"] == 0) & (out["imp_distance"] >= 1))
    return {k: v[keep] for k, v in out.items()}


def _slots_in(au, ad, bu, bd):
    """positions of the determinants (au, ad) in the list (bu, bd); -1 where absent"""
    where = {k: i for i, k in enumerate(zip(bu.tolist(), bd.tolist()))}
    return np.array([where.get(k, -1) for k in zip(au.tolist(), ad.tolist())], np.int64)


def psit_tables(g, s):
    """The tables of the step variant hf_to_psit = .true. from a walk set-up whose Psi_T is an eigenvector among its own determinants:
    Psi_T in label order with its places in the C(T) list (do_walk.f90:1258, 1849-1886), diag_elems (1091-1116), and the
    deterministic-space matrix as generate_sparse_ham_*_upper_triangular builds it with hf_to_psit (chemistry.f90:7885-7897,
    7926-7933): no first row and column, the (1,1) element stored as 0.  Returns (psit_ct_index 1-based, cdet, diag_elems, counts,
    indices, values); raises where the reference's assumptions do not hold."""
    ol = sort_dets(s.psi_up, s.psi_dn)
    loc_psit = _slots_in(s.psi_up[ol], s.psi_dn[ol], s.ct_up, s.ct_dn)
    loc_imp = _slots_in(s.imp_up, s.imp_dn, s.ct_up, s.ct_dn)
    if (loc_imp < 0).any():
        raise ValueError("hf_to_psit: the deterministic space is not contained in C(T)")
    if loc_psit[0] != 0 or loc_imp[0] != 0:
        raise ValueError("hf_to_psit: C(T), Psi_T and the deterministic space do not begin with the same determinant")
    outside = np.ones(len(s.ct_up), bool)
    outside[loc_imp] = False
    diag = np.zeros(len(s.ct_up))
    diag[outside] = g.hamiltonian_batch(s.ct_up[outside], s.ct_dn[outside], s.ct_up[outside], s.ct_dn[outside])
    # rows: diagonal first, then the columns below it (1-based).  Row 1 shrinks to its diagonal, set to 0; every other row loses column 1.
    counts, idx, val = np.asarray(s.prj_counts, np.int64), np.asarray(s.prj_indices, np.int64), np.asarray(s.prj_values, float)
    row = np.repeat(np.arange(len(counts)), counts)
    keep = ((row == 0) & (idx == 1)) | ((row != 0) & (idx != 1))
    val = np.where((row == 0) & (idx == 1), 0.0, val)
    new_counts = np.bincount(row[keep], minlength=len(counts)).astype(np.int64)
    return loc_psit + 1, np.asarray(s.psi_c, float)[ol], diag, new_counts, idx[keep], val[keep], ~outside


def initial_walkers_psit(s, cdet, in_imp, w_abs_gen_begin):
    """do_walk.f90:1245-1373 with hf_to_psit: the list is C(T); only its first determinant -- the first state -- carries weight
    (w_abs_gen_begin, rescaled as at 1332-1334); imp_distance 0 inside the deterministic space, -2 elsewhere; initiator 2, or 3 on the
    first state when |c_1| is the largest coefficient (1276-1292)."""
    n = len(s.ct_up)
    ac = np.abs(cdet)
    wt = np.zeros(n)
    wt[0] = w_abs_gen_begin / min(w_abs_gen_begin * ac.max() / ac.sum(), 1.0)
    ini, psign = np.full(n, 2, np.int8), np.zeros(n, np.int8)
    if abs(ac[0] - ac.max()) < 1e-3:
        ini[0], psign[0] = 3, (1 if cdet[0] > 0 else -1)
        wt[0] = wt[0] if wt[0] * psign[0] >= 1.0 else float(psign[0])
    return dict(up=s.ct_up.copy(), dn=s.ct_dn.copy(), wt=wt, initiator=ini, imp_distance=np.where(in_imp, 0, -2).astype(np.int8),
                perm_sign=psign, matrix_elements=np.full(n, 1e51), e_num=np.full(n, 1e51), e_den=np.full(n, 1e51))


class PopControl:
    """Scalars around sqmc_gpu_step: tau/r_initiator ramp until the target population is first
    reached (do_walk.f90:2175-2184, 2913-2923), e_est / e_trial / reweight_factor_inv
    (2880-2901).  The first n_equil_steps steps count as equilibration."""

    def __init__(self, tau, e_trial, w_target, r_initiator=1.0, initiator_rescale_power=1.0, pop_exp=10.0, rfi_max_multiplier=1.0,
                 n_equil_steps=10**9):
        self.tau_sav = self.tau = self.tau_prev = tau
        self.e_trial = self.e_est = e_trial
        self.w_target, self.r_init_sav, self.r_init, self.irp, self.pop_exp = w_target, r_initiator, r_initiator, initiator_rescale_power, pop_exp
        self.rfi, self.rfi_max, self.reached = 1.0, 1.0 + rfi_max_multiplier * tau, 0
        self.n_equil, self.istep, self.e_num_cum, self.e_den_cum, self.w_abs_gen = n_equil_steps, 0, 0.0, 0.0, None

    def pre_step(self, w_abs_gen):
        if self.reached != 0:
            return 1.0
        f = 1.0 + np.log(self.w_target / w_abs_gen)
        self.tau = self.tau_sav * f
        self.r_init = self.r_init_sav * f ** self.irp
        return self.tau / self.tau_prev

    def post_step(self, out):
        self.istep += 1
        w_abs_gen, e_den_gen, e_num_gen = out[1], out[2], out[3]
        self.e_num_cum += e_num_gen * np.sign(e_den_gen) if e_den_gen != 0 else 0.0
        self.e_den_cum += abs(e_den_gen)
        if self.e_den_cum != 0:
            self.e_est = self.e_num_cum / self.e_den_cum
        pw = min(1.0, self.tau * self.pop_exp)
        if self.istep <= self.n_equil:
            d = self.e_est - self.e_trial
            self.e_trial = self.e_trial + np.sign(d) * min(abs(d), 1.0)
            self.rfi = min(2.0, max(0.5, (self.w_target / w_abs_gen) ** pw))
        else:
            self.rfi = min(2.0, max(0.5, (1.0 / (1.0 + self.tau * (self.e_trial - self.e_est))) * (self.w_target / w_abs_gen) ** pw))
        self.rfi = min(self.rfi, self.rfi_max)
        ratio = 1.0
        if self.reached == 0 and w_abs_gen >= self.w_target:
            self.reached, ratio = 2, self.tau_sav / self.tau
            self.tau, self.r_init = self.tau_sav, self.r_init_sav
        self.tau_prev, self.w_abs_gen = self.tau, w_abs_gen
        return ratio

    def params(self, min_wt=0.5, cutoff=0.5, initiator_power=0, semistochastic=1):
        return dict(tau=self.tau, e_trial=self.e_trial, reweight_factor_inv=self.rfi, r_initiator=self.r_init, min_wt=min_wt,
                    always_spawn_cutoff_wt=cutoff, initiator_power=initiator_power, initiator_min_distance=0, c_t_initiator=0,
                    semistochastic=semistochastic, reached_w_abs_gen=self.reached)

    def to_c(self, w_abs_gen, min_wt=0.5, cutoff=0.5, initiator_power=0, semistochastic=1):
        """the same state as a sqmc_popctl for sqmc_gpu_run"""
        pc = PopCtl()
        pc.tau_sav, pc.tau, pc.tau_prev, pc.e_trial, pc.e_est = self.tau_sav, self.tau, self.tau_prev, self.e_trial, self.e_est
        pc.w_abs_gen_target, pc.w_abs_gen = self.w_target, w_abs_gen
        pc.r_initiator_sav, pc.r_initiator, pc.initiator_rescale_power = self.r_init_sav, self.r_init, self.irp
        pc.population_control_exponent, pc.reweight_factor_inv, pc.reweight_factor_inv_max = self.pop_exp, self.rfi, self.rfi_max
        pc.e_num_cum, pc.e_den_cum, pc.min_wt, pc.always_spawn_cutoff_wt = self.e_num_cum, self.e_den_cum, min_wt, cutoff
        pc.reached_w_abs_gen, pc.initiator_power, pc.initiator_min_distance, pc.c_t_initiator = self.reached, initiator_power, 0, 0
        pc.semistochastic, pc.istep, pc.n_equil = semistochastic, self.istep, min(self.n_equil, 2**62)
        return pc

    def from_c(self, pc):
        self.tau, self.tau_prev, self.e_trial, self.e_est = pc.tau, pc.tau_prev, pc.e_trial, pc.e_est
        self.r_init, self.rfi, self.e_num_cum, self.e_den_cum = pc.r_initiator, pc.reweight_factor_inv, pc.e_num_cum, pc.e_den_cum
        self.reached, self.istep, self.w_abs_gen = pc.reached_w_abs_gen, pc.istep, pc.w_abs_gen


class GpuWalk:
    """A C2-style semistochastic walk resident on one GPU."""

    def __init__(self, host, w_target, w_begin=None, mwalk=None, n_truncate_trial_wf=100, size_deterministic=1000, tau_multiplier=0.1,
                 e_trial=None, seed=(1346, 5634, 6635, 4361), rng_mode=RNG_COUNTER, min_wt=0.5, hf_to_psit=False, sum_order=1, proposal="uniform"):
        self.host = host
        w_begin = w_begin if w_begin is not None else w_target
        mwalk = mwalk or int(max(4 * (w_target / min_wt + size_deterministic), 200000))
        self.g = host.gpu(rng_mode=rng_mode, seed=seed, mwalk=mwalk)
        if proposal == "heatbath":          # proposal_method fast_heatbath: the library builds the tables and refuses what the reference refuses
            if not self.g.setup_efficient_heatbath():
                self.g.close()
                raise ValueError("Heatbath may be biased for this system!")
        if hf_to_psit:
            self.setup = s = host.setup_walk(self.g, n_truncate_trial_wf, size_deterministic, tau_multiplier, rediagonalize=True)
            ix, cdet, diag, pc_, pi_, pv_, in_imp = psit_tables(self.g, s)
            need = int(3 * (w_target / min_wt + len(s.ct_up)))          # do_walk.f90:653-655
            if mwalk < need:
                self.g.close()
                self.g = host.gpu(rng_mode=rng_mode, seed=seed, mwalk=need)
                if hasattr(host, "hb"):
                    self.g.set_hb_tables(*host.hb)
            self.g.set_projector(pc_, pi_, pv_)
            self.g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
            self.g.set_hf_to_psit(ix, cdet, diag, sum_order)
            wk = initial_walkers_psit(s, cdet, in_imp, w_begin)
        else:
            self.setup = s = host.setup_walk(self.g, n_truncate_trial_wf, size_deterministic, tau_multiplier)
            self.g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
            self.g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
            wk = initial_walkers(s, w_begin)
        self.g.upload_walkers(wk)
        self.pc = PopControl(s.tau, e_trial if e_trial is not None else s.e_trial0, w_target)
        self.w_abs = float(np.abs(wk["wt"]).sum())
        self.min_wt = min_wt
        self.last = None

    def step(self):
        r = self.pc.pre_step(self.w_abs)
        if r != 1.0:
            self.g.scale_projector(r)
        out = self.g.step(self.pc.params(min_wt=self.min_wt))
        r = self.pc.post_step(out)
        if r != 1.0:
            self.g.scale_projector(r)
        self.w_abs, self.last = out[1], out
        return out

    def run(self, nsteps, keep_stats=True):
        """nsteps steps inside the library (sqmc_gpu_run): no Python between steps"""
        pc = self.pc.to_c(self.w_abs, min_wt=self.min_wt)
        stats, totals = self.g.run(pc, nsteps, keep_stats)
        self.pc.from_c(pc)
        self.w_abs = pc.w_abs_gen
        return stats, totals

    def close(self):
        self.g.close()


# ------------------------------------------------------------------------ multi-rank glue
def rank_seed(seed, rank):
    """Seed 2 of input line 1 is offset by the rank (do_walk.f90:234: irand_seed(:,2)+rank);
    rannyu forces the last limb odd, so ranks step by 2 to stay distinct."""
    s = list(seed)
    s[3] = (s[3] + 2 * rank) % 10000
    return tuple(s)


def allreduce_step_sums(out, device=None):
    """The MPI_Allreduce of do_walk.f90:2778 on the 7 per-step sums (w_gen, w_abs_gen,
    e_den_gen, e_num_gen, w_perm_initiator_gen, nwalk, w_abs_gen_imp): torch.distributed SUM
    (RCCL when the group backend is nccl -- the 56 bytes then travel as a device tensor --
    gloo on CPU).  Returns the reduced copy; entries 7..15 stay rank-local.  No-op without an
    initialised process group."""
    import torch
    import torch.distributed as dist
    res = np.array(out, dtype=np.float64, copy=True)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return res
    t = torch.from_numpy(res[:7].copy())
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    res[:7] = t.cpu().numpy()
    return res


# ------------------------------------------------------------------------------- HCI
def hci_variational(host, g, eps_var, eps_sched=(), n_states=1, max_iters=50, log=None, davidson_tol=1e-10):
    """Variational stage of perform_hci (hci.f90:359-520) with get_next_det_list
    (hci.f90:865-1040): every piece that scales with the number of determinants runs on the
    GPU -- eps-screened connection generation + dedup (sqmc_gpu_hci_connections), sparse
    Hamiltonian (sqmc_gpu_build_sparse_ham), Davidson matvec (sqmc_gpu_spmv_apply).
    Returns (up, dn, coeffs[n, n_states], energies, history of ndets)."""
    sched = list(eps_sched) + [eps_var]
    up = np.array([host.hf_up], np.uint64); dn = np.array([host.hf_dn], np.uint64)
    wts = np.zeros((1, n_states)); wts[0, 0] = 1.0
    energy = np.zeros(n_states)
    energy[0] = g.hamiltonian_batch(up, dn, up, dn)[0]
    old_energy = energy.copy()
    hist = [1]
    eps = sched[0]
    for it in range(1, max_iters + 1):
        if it <= len(sched):
            eps = sched[it - 1]
        coeffs = np.abs(wts).max(axis=1) if it > 1 else wts[:, 0].copy()
        cu, cd, _, _ = g.hci_connections(up, dn, coeffs, eps)              # sorted, unique, includes the old list
        # append the new determinants behind the old list in sorted order (hci.f90:979-991)
        is_new = ~_dets_in(cu, cd, up, dn)
        n_old, n_new = len(up), len(up) + int(is_new.sum())
        if n_new == n_old:
            continue
        if n_new <= int(1.00001 * n_old) and eps == sched[-1]:
            break
        up = np.concatenate((up, cu[is_new])); dn = np.concatenate((dn, cd[is_new]))
        order = sort_dets(up, dn)
        v0 = None
        if it > 1:
            v0 = np.zeros((n_new, n_states)); v0[np.argsort(order)[:n_old], :] = wts
        plan, diag, nnz = SpmvPlan.from_dets(g, up[order], dn[order])      # H built, expanded and kept on the GPU
        try:
            w, X = davidson_lowest(plan, diag, k=n_states, v0=v0, tol=davidson_tol)
        finally:
            plan.close()
        wts = np.zeros((n_new, n_states)); wts[order, :] = X
        energy = np.array(w)
        hist.append(n_new)
        if log:
            log("Iteration %3d eps1=%.1e ndets=%9d nnz=%10d energy=%s" % (it, eps, n_new, nnz, " ".join("%.9f" % e for e in energy)))
        if np.max(np.abs(energy - old_energy)) < 1e-5 and eps == sched[-1]:
            break
        old_energy = energy.copy()
    return up, dn, wts, energy, hist


# --------------------------------------------------------------------- sharded walk
def _backend():
    import torch.distributed as dist
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def _allreduce_dev(t):
    """SUM all-reduce of a device tensor: RCCL directly, or staged through the host for gloo."""
    import torch.distributed as dist
    if _backend() is None or dist.get_world_size() == 1:
        return
    if _backend() == "nccl":
        dist.all_reduce(t)
    else:
        h = t.cpu(); dist.all_reduce(h); t.copy_(h)


def exchange_records(send, send_counts, recv):
    """Personalised all-to-all of 32-byte spawn records (rows of 4 int64), as mpi_sendnewwalks does
    with MPI_Allgather(counts) + MPI_ALLTOALLV (mpi_routines.f90:2522-2622).  send: [cap,4] tensor
    whose first sum(send_counts) rows are grouped by destination rank; returns the number of
    rows received (rank order, sender's order inside a rank) now at the head of recv."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if _backend() else 1
    sc = [int(x) for x in send_counts]
    ns = sum(sc)
    if world == 1:
        recv[:ns].copy_(send[:ns])
        return ns
    cnt = torch.tensor(sc, dtype=torch.int64)
    rcnt = torch.empty(world, dtype=torch.int64)
    if _backend() == "nccl":
        cd, rd = cnt.to(send.device), rcnt.to(send.device)
        dist.all_to_all_single(rd, cd)
        rcnt = rd.cpu()
    else:
        dist.all_to_all_single(rcnt, cnt)
    rc = [int(x) for x in rcnt]
    nr = sum(rc)
    if nr > recv.shape[0]:
        raise RuntimeError("receive buffer too small: %d > %d records" % (nr, recv.shape[0]))
    if _backend() == "nccl":
        dist.all_to_all_single(recv[:nr], send[:ns], output_split_sizes=rc, input_split_sizes=sc)
    else:
        hs, hr = send[:ns].cpu(), torch.empty((nr, 4), dtype=torch.int64)
        dist.all_to_all_single(hr, hs, output_split_sizes=rc, input_split_sizes=sc)
        recv[:nr].copy_(hr)
    return nr


class ShardedWalk:
    """One rank of a walk whose determinants are sharded over ranks by hash ownership; RCCL (or
    gloo, for tests) moves the deterministic-space weights, the spawned walkers and the seven
    estimator sums, exactly the three exchanges of the reference's MPI walk."""

    def __init__(self, host, w_target, rank, world, w_begin=None, mwalk=None, n_truncate_trial_wf=100, size_deterministic=1000,
                 tau_multiplier=0.1, e_trial=None, seed=(1346, 5634, 6635, 4361), min_wt=0.5, device_index=0, n_equil_steps=10**9, owner_hash=0,
                 semistochastic=True):
        import torch
        self.rank, self.world, self.min_wt = rank, world, min_wt
        self.semi = 1 if semistochastic else 0          # 0: semistochastic = f, no deterministic space; join_walker2 is local to a rank (do_walk.f90:2475)
        w_begin = w_begin if w_begin is not None else w_target
        per_rank = w_target / world
        mwalk = mwalk or int(max(6 * (per_rank / min_wt + size_deterministic), 200000))
        self.g = g = host.gpu(rng_mode=RNG_COUNTER, seed=rank_seed(seed, rank), mwalk=mwalk)
        if owner_hash:
            g.set_owner_hash(owner_hash)       # 1: the reference's get_det_owner (djb_hash), mpi_routines.f90:354-445
        self.setup = s = host.setup_walk(g, n_truncate_trial_wf, size_deterministic, tau_multiplier)
        if self.semi:
            g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
        g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
        wk = initial_walkers(s, w_begin)
        if not self.semi:                              # a purely stochastic population
            wk["imp_distance"] = np.where(wk["imp_distance"] == 0, 1, wk["imp_distance"]).astype(np.int8)
            keep = ~((wk["wt"] == 0) & (wk["initiator"] < 3))
            wk = {k: v[keep] for k, v in wk.items()}
        own = g.det_owner(wk["up"], wk["dn"], world) == rank
        mine = {k: v[own] for k, v in wk.items()}
        # global row of every deterministic-space walker this rank owns (both lists sorted by (up,dn))
        imp_index = {(int(a), int(b)): i for i, (a, b) in enumerate(zip(s.imp_up, s.imp_dn))}
        rows = [imp_index[(int(a), int(b))] for a, b, d in zip(mine["up"], mine["dn"], mine["imp_distance"]) if d == 0]
        g.shard_config(rank, world, rows)
        g.upload_walkers(mine)
        dev = torch.device("cuda", device_index)
        self.dev = dev
        self.xg = torch.zeros(max(len(s.imp_up), 1), dtype=torch.float64, device=dev)
        cap = mwalk
        self.send = torch.empty((cap, 4), dtype=torch.int64, device=dev)
        self.recv = torch.empty((cap, 4), dtype=torch.int64, device=dev)
        self.cap = cap
        self.pc = PopControl(s.tau, e_trial if e_trial is not None else s.e_trial0, w_target, n_equil_steps=n_equil_steps)
        self.w_abs = float(np.abs(wk["wt"]).sum())          # global
        self.n_imp_global = len(s.imp_up) if self.semi else 0
        self.in_library = False

    def attach_rccl(self):
        """Give the library its own RCCL communicator (sqmc_gpu_comm_init) so that run() issues
        the three exchanges of a step itself.  The unique id travels over the caller's
        torch.distributed group (any backend) -- MPI_Bcast in the reference's build."""
        import torch.distributed as dist
        box = [self.g.comm_unique_id() if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        self.g.comm_init(box[0])
        self.in_library = True

    def run(self, nsteps, keep_stats=True):
        """nsteps sharded steps inside the library (sqmc_gpu_shard_run); needs attach_rccl()"""
        if not self.in_library:
            raise RuntimeError("ShardedWalk.run needs attach_rccl(); use step() for the caller-driven exchange")
        pc = self.pc.to_c(self.w_abs, min_wt=self.min_wt, semistochastic=self.semi)
        stats, totals = self.g.shard_run(pc, nsteps, keep_stats)
        self.pc.from_c(pc)
        self.w_abs = pc.w_abs_gen
        return stats, totals

    def step(self):
        import torch
        r = self.pc.pre_step(self.w_abs)
        if r != 1.0 and self.semi:
            self.g.scale_projector(r)
        prm = self.pc.params(min_wt=self.min_wt, semistochastic=self.semi)
        if self.in_library:
            out = self.g.shard_step(prm)
            r = self.pc.post_step(out)
            if r != 1.0 and self.semi:
                self.g.scale_projector(r)
            self.w_abs, self.last_local = out[1], out
            return out
        self.g.shard_begin(prm, self.xg.data_ptr())
        _allreduce_dev(self.xg)
        counts = self.g.shard_pack(prm, self.xg.data_ptr(), self.send.data_ptr(), self.cap, self.world)
        nr = exchange_records(self.send, counts, self.recv)
        torch.cuda.synchronize()
        local = self.g.shard_finish(prm, self.recv.data_ptr(), nr)
        out = allreduce_step_sums(local, device=self.dev)
        r = self.pc.post_step(out)
        if r != 1.0 and self.semi:
            self.g.scale_projector(r)
        self.w_abs, self.last_local = out[1], local
        return out

    def close(self):
        self.g.close()



# ------------------------------------------------------------------------------ HEG host
class HegHost:
    """Homogeneous electron gas in a plane-wave basis: the tables of read_heg / system_setup_heg /
    generate_k_vectors (heg.f90:119-215, 643-749) and the walk set-up on the GPU path."""

    def __init__(self, n_dim, r_s, nelec, nup, cutoff_radius):
        import math
        eps, pi = 1.0e-15, 4.0 * math.atan(1.0)
        self.n_dim, self.r_s, self.nelec, self.nup, self.ndn = n_dim, r_s, nelec, nup, nelec - nup
        density = 1.0 / (pi * (r_s * r_s)) if n_dim == 2 else 3.0 / (4.0 * pi * (r_s * r_s * r_s))
        self.length_cell = L = math.pow(nelec / density, 1.0 / n_dim)
        n_max = int(cutoff_radius + eps)
        vals = [2 * pi / L * i for i in range(-n_max, n_max + 1)]
        import itertools
        kv = [list(t) for t in itertools.product(vals, repeat=n_dim)]            # last index fastest, heg.f90:676-697
        n = len(kv)
        sq = lambda v: sum_left(x * x for x in v)
        inc = n // 2                                                             # shell_sort_real_rank2, generic_sort.f90:554-591
        while inc > 0:
            for i in range(inc, n):
                j, t = i, kv[i]
                while j >= inc:
                    if sq(kv[j - inc]) <= sq(t):
                        break
                    kv[j] = kv[j - inc]
                    j -= inc
                kv[j] = t
            inc = 1 if inc == 2 else inc * 5 // 11
        norb = 0
        for v in kv:
            if math.sqrt(sq(v)) > 2 * pi / L * cutoff_radius + eps:
                break
            norb += 1
        if norb > 64:
            raise ValueError("more than 64 plane waves need two-word determinants (later round)")
        self.norb = norb
        self.k_vectors = np.zeros((norb, 3)); self.k_vectors[:, :n_dim] = np.array(kv[:norb])
        self.k_rel = np.rint(self.k_vectors * L / (2 * pi)).astype(np.int64)
        self.hf_up, self.hf_dn = (1 << nup) - 1, (1 << self.ndn) - 1

    def gpu(self, **kw):
        return GpuChem.heg(self.n_dim, self.norb, self.nup, self.ndn, self.length_cell, self.k_vectors, **kw)

    def madelung_energy(self):
        """madelung_energy, heg.f90:2828-2906 (3D): Ewald self-interaction of the periodic images with
        kappa = 10/L (the real-space erfc term is negligible), reciprocal sum over the cube of g-vectors
        up to the first n_max whose term drops below 1e-10, times nelec/2.  Added to HCI totals as
        'Total energy (includ. Madelung)'."""
        import math
        if self.n_dim != 3:
            raise ValueError("Madelung energy is only implemented for 3d (heg.f90:2844)")
        L, pi = self.length_cell, 4.0 * math.atan(1.0)
        kappa = 10.0 / L
        n_max = 1
        while True:
            g_max = 2 * pi * n_max / L
            if 4 * pi / L ** 3 * math.exp(-(g_max / (2 * kappa)) ** 2) / g_max ** 2 < 1e-10:
                break
            n_max += 1
        vals = [2 * pi / L * i for i in range(-n_max, n_max + 1)]
        e = 0.0
        for a in vals:
            for b in vals:
                for c_ in vals:
                    g2 = sum_left((a * a, b * b, c_ * c_))
                    if g2 < 1e-10:
                        continue
                    e = e + math.exp(-g2 / (2 * kappa) ** 2) / g2
        e = e * 4 * pi / L ** 3
        e = e - pi / L ** 3 / kappa ** 2 - 2 * kappa / pi ** 0.5
        return e * self.nelec / 2.0

    def connected(self, up, dn):
        """the determinant + all momentum-conserving double excitations (unique, unsorted)"""
        n, kr = self.norb, self.k_rel
        lut = {tuple(kr[i]): i for i in range(n)}
        occ = [(o, 0) for o in range(n) if (up >> o) & 1] + [(o, 1) for o in range(n) if (dn >> o) & 1]
        out = {(up, dn)}
        for a in range(len(occ)):
            for b in range(a + 1, len(occ)):
                (p, sp), (q, sq_) = occ[a], occ[b]
                for r in range(n):
                    det_r = dn if sp else up
                    if (det_r >> r) & 1:
                        continue
                    s_ = lut.get(tuple(kr[p] + kr[q] - kr[r]))
                    if s_ is None:
                        continue
                    det_s = dn if sq_ else up
                    if (det_s >> s_) & 1 or (sp == sq_ and s_ == r):
                        continue
                    nu, nd = up, dn
                    if sp: nd &= ~(1 << p)
                    else: nu &= ~(1 << p)
                    if sq_: nd &= ~(1 << q)
                    else: nu &= ~(1 << q)
                    if sp: nd |= (1 << r)
                    else: nu |= (1 << r)
                    if sq_: nd |= (1 << s_)
                    else: nu |= (1 << s_)
                    out.add((nu, nd))
        keys = sorted(out)
        return np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint64)

    def setup_walk(self, g, n_truncate_trial_wf=1, size_deterministic=500, tau_multiplier=0.1, rediagonalize=False):
        s = WalkSetup()
        up, dn = self.connected(self.hf_up, self.hf_dn)
        w, X, _ = lowest_state(g, up, dn)
        c = X[:, 0]
        if c[np.argmax(np.abs(c))] < 0:
            c = -c
        by = np.argsort(-np.abs(c), kind="stable")
        up_s, dn_s, c_s = up[by], dn[by], c[by]
        n_t, n_i = _truncate_at_csf(c_s, n_truncate_trial_wf), _truncate_at_csf(c_s, size_deterministic)
        s.psi_up, s.psi_dn = up_s[:n_t].copy(), dn_s[:n_t].copy()
        s.psi_c = c_s[:n_t] / np.sqrt(np.dot(c_s[:n_t], c_s[:n_t]))
        if rediagonalize and n_t > 1:          # semistoch.f90:575, 706-712 (see setup_walk)
            ol = sort_dets(s.psi_up, s.psi_dn)
            s.psi_up, s.psi_dn = s.psi_up[ol], s.psi_dn[ol]
            e_t, cr = rediagonalize_in_own_space(g, s.psi_up, s.psi_dn, s.psi_c[ol])
            s.psi_c, s.e_psi_t = (-cr if cr[np.argmax(np.abs(cr))] < 0 else cr), e_t
        o = sort_dets(up_s[:n_i], dn_s[:n_i])
        s.imp_up, s.imp_dn = up_s[:n_i][o].copy(), dn_s[:n_i][o].copy()
        n = self.norb
        mu = ((1 << n) - 1) ^ ((1 << (n - self.nup)) - 1); md = ((1 << n) - 1) ^ ((1 << (n - self.ndn)) - 1)
        d = g.hamiltonian_batch([self.hf_up, mu], [self.hf_dn, md], [self.hf_up, mu], [self.hf_dn, md])
        s.tau, s.e_var = tau_multiplier / float(d[1] - d[0]), float(w[0])
        pc, pi_, pv = g.build_sparse_ham(s.imp_up, s.imp_dn)
        s.prj_counts, s.prj_indices, s.prj_values = pc, pi_, -s.tau * pv
        acc = {}
        psi_index = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(s.psi_up, s.psi_dn))}
        for j in range(n_t):
            cu, cd = self.connected(int(s.psi_up[j]), int(s.psi_dn[j]))
            h = g.hamiltonian_batch(cu, cd, np.full(len(cu), s.psi_up[j]), np.full(len(cu), s.psi_dn[j]))
            for a, b, v in zip(cu.tolist(), cd.tolist(), h.tolist()):
                acc[(a, b)] = acc.get((a, b), 0.0) + v * s.psi_c[j]
        keys = sorted(acc)
        s.ct_up = np.array([k[0] for k in keys], np.uint64); s.ct_dn = np.array([k[1] for k in keys], np.uint64)
        s.ct_num = np.array([acc[k] for k in keys])
        s.ct_den = np.array([s.psi_c[psi_index[k]] if k in psi_index else 0.0 for k in keys])
        s.e_trial0 = float(np.dot(s.ct_num, s.ct_den) / np.dot(s.ct_den, s.ct_den))
        return s


class HubbardHost:
    """Real-space Hubbard model on an l_x by l_y square lattice (hamiltonian_type 'hubbard2'): the
    scalars of read_hubbard (hubbard.f90:138-382) and the walk set-up on the GPU path with a
    determinant-list trial wavefunction (the last branch of energy_pieces_hubbard, 4514-4527); the
    Gutzwiller / Slater trial functions of hubbard.f90 are outside this path."""

    def __init__(self, l_x, l_y, pbc, nup, ndn, t=1.0, U=4.0):
        self.l_x, self.l_y, self.pbc, self.nup, self.ndn, self.t, self.U = l_x, l_y, bool(pbc), nup, ndn, float(t), float(U)
        self.norb, self.nelec = l_x * l_y, nup + ndn
        sites = list(range(self.norb))
        even = [q for q in sites if ((q % l_x) + (q // l_x)) % 2 == 0]
        odd = [q for q in sites if ((q % l_x) + (q // l_x)) % 2 == 1]
        # start determinant of the harness: the Neel state at half filling (up on x+y even first, dn on x+y odd first)
        self.hf_up = sum(1 << q for q in (even + odd)[:nup])
        self.hf_dn = sum(1 << q for q in (odd + even)[:ndn])
        self.nbr = [[self._get_nbr(site, k) for k in (1, 2, 3, 4)] for site in range(1, self.norb + 1)]

    def _get_nbr(self, site, nbr_type):
        """get_nbr, more_tools.f90:223-355 (1 LEFT, 2 RIGHT, 3 UP, 4 DOWN); 0 when not allowed"""
        lx, ly = self.l_x, self.l_y
        y1 = (site - 1) // lx + 1
        x1 = site - (y1 - 1) * lx
        x2, y2, ok = x1, y1, True
        if nbr_type in (1, 2):
            x2 = x1 - 1 if nbr_type == 1 else x1 + 1
            if not self.pbc:
                ok = 0 < x2 <= lx
            else:
                x2 = lx if x2 == 0 else (1 if x2 == lx + 1 else x2)
                ok = x2 != x1
        else:
            y2 = y1 + 1 if nbr_type == 3 else y1 - 1
            if not self.pbc:
                ok = 0 < y2 <= ly
            else:
                y2 = ly if y2 == 0 else (1 if y2 == ly + 1 else y2)
                ok = y2 != y1
        return (y2 - 1) * lx + x2 if ok else 0

    def gpu(self, **kw):
        return GpuChem.hubbard(self.l_x, self.l_y, self.pbc, self.nup, self.ndn, self.t, self.U, **kw)

    def connected(self, up, dn):
        """the determinant and its nearest-neighbour hops (find_connected_dets_hubbard, hubbard.f90:5306-5457), unique, sorted"""
        out = {(up, dn)}
        for site in range(1, self.norb + 1):
            for is_up in (True, False):
                cfg = up if is_up else dn
                if not (cfg >> (site - 1)) & 1:
                    continue
                for nb in self.nbr[site - 1]:
                    if nb and not (cfg >> (nb - 1)) & 1:
                        nc = (cfg & ~(1 << (site - 1))) | (1 << (nb - 1))
                        out.add((nc, dn) if is_up else (up, nc))
        keys = sorted(out)
        return np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint64)

    def spectral_range_bound(self):
        return self.U * (min(self.nup, self.ndn) - max(0, self.nelec - self.norb)) + 4.0 * abs(self.t) * self.nelec

    def first_order_space(self, n_levels=2):
        seen = {(self.hf_up, self.hf_dn)}
        frontier = [(self.hf_up, self.hf_dn)]
        for _ in range(n_levels):
            nxt = []
            for (a, b) in frontier:
                cu, cd = self.connected(a, b)
                for k in zip(cu.tolist(), cd.tolist()):
                    if k not in seen:
                        seen.add(k); nxt.append(k)
            frontier = nxt
        keys = sorted(seen)
        return np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint64)

    def setup_walk(self, g, n_truncate_trial_wf=20, size_deterministic=500, tau_multiplier=0.5, n_levels=2):
        s = WalkSetup()
        up, dn = self.first_order_space(n_levels)
        w, X, _ = lowest_state(g, up, dn)
        c = X[:, 0]
        if c[np.argmax(np.abs(c))] < 0:
            c = -c
        by = np.argsort(-np.abs(c), kind="stable")
        up_s, dn_s, c_s = up[by], dn[by], c[by]
        n_t, n_i = _truncate_at_csf(c_s, n_truncate_trial_wf), _truncate_at_csf(c_s, size_deterministic)
        s.psi_up, s.psi_dn = up_s[:n_t].copy(), dn_s[:n_t].copy()
        s.psi_c = c_s[:n_t] / np.sqrt(np.dot(c_s[:n_t], c_s[:n_t]))
        o = sort_dets(up_s[:n_i], dn_s[:n_i])
        s.imp_up, s.imp_dn = up_s[:n_i][o].copy(), dn_s[:n_i][o].copy()
        s.tau, s.e_var = tau_multiplier / self.spectral_range_bound(), float(w[0])
        pc, pi_, pv = g.build_sparse_ham(s.imp_up, s.imp_dn)
        s.prj_counts, s.prj_indices, s.prj_values = pc, pi_, -s.tau * pv
        acc = {}
        psi_index = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(s.psi_up, s.psi_dn))}
        for j in range(n_t):
            cu, cd = self.connected(int(s.psi_up[j]), int(s.psi_dn[j]))
            h = g.hamiltonian_batch(cu, cd, np.full(len(cu), s.psi_up[j]), np.full(len(cu), s.psi_dn[j]))
            for a, b, v in zip(cu.tolist(), cd.tolist(), h.tolist()):
                acc[(a, b)] = acc.get((a, b), 0.0) + v * s.psi_c[j]
        keys = sorted(acc)
        s.ct_up = np.array([k[0] for k in keys], np.uint64); s.ct_dn = np.array([k[1] for k in keys], np.uint64)
        s.ct_num = np.array([acc[k] for k in keys])
        s.ct_den = np.array([s.psi_c[psi_index[k]] if k in psi_index else 0.0 for k in keys])
        s.e_trial0 = float(np.dot(s.ct_num, s.ct_den) / np.dot(s.ct_den, s.ct_den))
        return s


def sum_left(it):
    """left-to-right floating sum (Fortran SUM over a tiny array)"""
    t = 0.0
    for x in it:
        t = t + x
    return t


def hci_pt2(host, g, up, dn, coeffs, e_var, eps_pt, n_slices=1):
    """Deterministic Epstein-Nesbet second-order correction, second_order_pt (hci.f90:1100-1182):
    delta_E = sum over determinants a outside the variational space of
    (sum_i H_ai c_i)^2 / (E_var - H_aa), the inner sum screened by |H_ai c_i| >= eps_pt.
    One library call (sqmc_gpu_hci_pt2): connections, their dedup-sum, the membership search, the
    diagonal elements and the reduction all stay on the device.  n_slices > 1 does the connected
    space in that many slices of the determinant-key range (exact: every connected determinant
    lives in one slice), for spaces whose connections do not fit one call.
    Returns (delta_E, number of connected determinants)."""
    return g.hci_pt2(up, dn, coeffs, e_var, eps_pt, n_slices)


def hci_pt2_by_doors(host, g, up, dn, coeffs, e_var, eps_pt, n_slices=1):
    """The same sum assembled on the host from the batch doors (connections, then H_aa); kept as
    the cross-check of sqmc_gpu_hci_pt2 in the tests."""
    up, dn = np.ascontiguousarray(up, np.uint64), np.ascontiguousarray(dn, np.uint64)
    delta, n_conn = 0.0, 0
    for sl in range(n_slices):
        cu, cd, num, den = g.hci_connections(up, dn, coeffs, eps_pt, slice=sl, n_slices=n_slices)
        if len(cu) == 0:
            continue
        out = ~_dets_in(cu, cd, up, dn)        # membership in the variational space (binary_search at hci.f90:1159)
        h_aa = g.hamiltonian_batch(cu[out], cd[out], cu[out], cd[out])
        delta += float(np.sum(num[out] ** 2 / (e_var - h_aa)))
        n_conn += len(cu)
    return delta, n_conn


def time_symmetrized_to_dets(up, dn, coeffs, z=1):
    """convert_time_symmetrized_to_dets (hci.f90:4365-4564): each time-reversal-symmetrised
    combination with up != dn becomes the two determinants (up,dn) and (dn,up) with
    coefficients c/sqrt2 and z*c/sqrt2; the reference leaves time symmetry this way before PT."""
    up, dn, c = np.asarray(up, np.uint64), np.asarray(dn, np.uint64), np.asarray(coeffs, float)
    inv_sqrt2 = 1.0 / np.sqrt(2.0)
    same = up == dn
    ou = np.concatenate((up[same], up[~same], dn[~same]))
    od = np.concatenate((dn[same], dn[~same], up[~same]))
    oc = np.concatenate((c[same], inv_sqrt2 * c[~same], z * inv_sqrt2 * c[~same]))
    order = sort_dets(ou, od)
    return ou[order], od[order], oc[order]


def hci_pt2_determinant_basis(host, up, dn, coeffs, e_var, eps_pt, n_slices=1):
    """do_pt as the reference runs it for a time-symmetric variational stage: back to the
    determinant basis, time_sym off from then on (hci.f90:648-659), then second_order_pt."""
    import copy
    if not host.time_sym:
        g = host.gpu()
        g.set_hb_tables(*host.hb_tables(g))
        try:
            return hci_pt2(host, g, up, dn, coeffs, e_var, eps_pt, n_slices)
        finally:
            g.close()
    plain = copy.copy(host)
    plain.time_sym = False
    du, dd, dc = time_symmetrized_to_dets(up, dn, coeffs, host.z)
    g = plain.gpu()
    try:
        g.set_hb_tables(*plain.hb_tables(g))
        return hci_pt2(plain, g, du, dd, dc, e_var, eps_pt, n_slices)
    finally:
        g.close()


# ------------------------------------------------------------------------- Fortran host deck
def dump_walk_deck(path, host, setup, walkers, w_target, e_trial, seed=(1346, 5634, 6635, 4361), mwalk=400000, rng_mode=RNG_COUNTER):
    """Everything a compiled host needs to start the same walk through the C ABI, as one
    little-endian stream file (read by sqmc_amd/fortran/example_walk.f90 with access='stream'):
    an int64 header, then the tables in the order of the header.  This is test plumbing for the
    Fortran binding: the reference builds these tables itself."""
    prod, osym, c2 = (np.ascontiguousarray(a, np.int32).reshape(-1) for a in (host.prod, host.orbsym, host.combine_2))
    ints = _np_f64(host.integrals)
    cnt, idx, val = np.ascontiguousarray(setup.prj_counts, np.int64), np.ascontiguousarray(setup.prj_indices, np.int64), _np_f64(setup.prj_values)
    n = len(walkers["up"])
    hdr = np.array([0x73716D63, host.norb, host.nup, host.ndn, host.n_core_orb, int(host.time_sym), host.z, host.n_group, len(ints) - 1,
                    rng_mode, seed[0], seed[1], seed[2], seed[3], mwalk, len(cnt), len(idx), len(setup.ct_up), n, len(prod), len(osym), len(c2)], np.int64)
    scal = np.array([setup.tau, e_trial, w_target, float(np.abs(walkers["wt"]).sum())], np.float64)
    with open(path, "wb") as f:
        for a in (hdr, scal, prod, osym, c2, ints, cnt, idx, val,
                  np.ascontiguousarray(setup.ct_up, np.uint64), np.ascontiguousarray(setup.ct_dn, np.uint64), _np_f64(setup.ct_num), _np_f64(setup.ct_den),
                  np.ascontiguousarray(walkers["up"], np.uint64), np.ascontiguousarray(walkers["dn"], np.uint64), _np_f64(walkers["wt"]),
                  np.ascontiguousarray(walkers["imp_distance"], np.int8), np.ascontiguousarray(walkers["initiator"], np.int8),
                  np.ascontiguousarray(walkers["perm_sign"], np.int8), _np_f64(walkers["matrix_elements"]), _np_f64(walkers["e_num"]),
                  _np_f64(walkers["e_den"])):
            f.write(a.tobytes())


def _np_f64(a):
    return np.ascontiguousarray(a, np.float64)


# ------------------------------------------------------------- HCI wavefunction files
def wf_filename(eps_var):
    """hci.f90:195-197: write(fmt,'(es7.2e1)') eps_var ; filename = 'wf_eps_var=' // fmt"""
    m, e = ("%.2e" % eps_var).split("e")
    return "wf_eps_var=%sE%s%d" % (m, "-" if int(e) < 0 else "+", abs(int(e)))


def _frecord(f, payload):
    f.write(np.int32(len(payload)).tobytes()); f.write(payload); f.write(np.int32(len(payload)).tobytes())


def write_wf_var(path, up, dn, wts, energy):
    """The variational wavefunction dump of perform_hci (hci.f90:602-625; read back at :199-216):
    Fortran sequential unformatted records  ndets (default integer) | dets_up(1:n) | dets_dn(1:n) |
    wts(1:n,1:n_states) | energy(1:n_states), determinants as the reference's 128-bit integers
    (little endian: low word, high word = 0 for norb <= 64), weights column-major."""
    up, dn = np.ascontiguousarray(up, np.uint64), np.ascontiguousarray(dn, np.uint64)
    wts = np.asarray(wts, np.float64).reshape(len(up), -1)
    d128 = lambda a: np.stack([a, np.zeros_like(a)], axis=1).tobytes()
    with open(path, "wb") as f:
        _frecord(f, np.int32(len(up)).tobytes())
        _frecord(f, d128(up)); _frecord(f, d128(dn))
        _frecord(f, np.asfortranarray(wts).tobytes(order="F"))
        _frecord(f, np.asarray(energy, np.float64).tobytes())


def read_wf_var(path, n_states=1):
    """-> (up, dn, wts[n, n_states], energy[n_states]) from a file written by the reference or by write_wf_var"""
    raw = open(path, "rb").read()
    recs, o = [], 0
    while o < len(raw):
        n = int(np.frombuffer(raw, np.int32, 1, o)[0])
        recs.append(raw[o + 4:o + 4 + n])
        if int(np.frombuffer(raw, np.int32, 1, o + 4 + n)[0]) != n:
            raise ValueError("not a Fortran sequential unformatted file: " + path)
        o += n + 8
    nd = int(np.frombuffer(recs[0], np.int32)[0])
    u = np.frombuffer(recs[1], np.uint64).reshape(nd, 2); d = np.frombuffer(recs[2], np.uint64).reshape(nd, 2)
    if u[:, 1].any() or d[:, 1].any():
        raise ValueError("determinants beyond 64 orbitals are not supported by this host")
    wts = np.frombuffer(recs[3], np.float64).reshape(n_states, nd).T.copy()
    return u[:, 0].copy(), d[:, 0].copy(), wts, np.frombuffer(recs[4], np.float64).copy()


def _orbs(det, n_core_orb=0):
    """1-based occupied orbitals above the core, renumbered from the first non-core orbital"""
    d, out = int(det) >> n_core_orb, []
    while d:
        low = d & -d
        out.append(low.bit_length())
        d ^= low
    return out


def _det_from_orbs(orbs, n_core_orb=0):
    d = (1 << n_core_orb) - 1
    for o in orbs:
        d |= 1 << (int(o) + n_core_orb - 1)
    return d


def _f(x, w, dec):
    """Fortran fw.d (a leading zero before the point may be dropped when the field is full)"""
    t = "%*.*f" % (w, dec, x)
    if len(t) > w and t.lstrip("-").startswith("0."):
        t = t.replace("0.", ".", 1)
    return t if len(t) <= w else "*" * w


def write_psit_connections(path, psi_up, ct_up, ct_dn, ct_num, ct_den, nup, ndn, norb, n_core_orb=0):
    """psit_con_out_file, semistoch.f90:86-126: C(T) as text -- header `(i8,i12,2i4,i6,i3,f15.8,...)`,
    a title line, then per determinant with |e_loc_num| > 1e-10 the occupied orbitals (core removed)
    `(i3, (nel-1)i4)` and `f22.15, f19.15` numerator and denominator.  A reference run with
    use_psit_con_in reads it back (do_walk.f90:702-741)."""
    keep = np.abs(ct_num) > 1e-10
    with open(path, "w") as f:
        f.write("%8d%12d%4d%4d%6d%3d%s ndet_psi_t, ndet_connections_nonzero, nup-n_core_orb, ndn-n_core_orb, norb, 0, E_T\n"
                % (len(psi_up), int(keep.sum()), nup - n_core_orb, ndn - n_core_orb, norb, 0, _f(ct_num[0] / ct_den[0] if ct_den[0] != 0 else 0.0, 15, 8)))
        f.write("orb_up          orb_dn           e_loc_num            e_loc_den\n")
        for u, d_, a, b in zip(ct_up[keep].tolist(), ct_dn[keep].tolist(), ct_num[keep].tolist(), ct_den[keep].tolist()):
            o = _orbs(u, n_core_orb) + _orbs(d_, n_core_orb)
            f.write("%3d" % o[0] + "".join("%4d" % v for v in o[1:]) + _f(a, 22, 15) + _f(b, 19, 15) + "\n")


def read_psit_connections(path, nup, ndn, n_core_orb=0):
    """the reader of do_walk.f90:702-741 (list-directed): returns (ct_up, ct_dn, ct_num, ct_den), sorted by (up, dn);
    the determinants with |e_loc_den| > 1e-12 are Psi_T with that coefficient"""
    with open(path) as f:
        head = f.readline().split()
        n_con = int(head[1])
        f.readline()
        rows = [f.readline().replace(",", " ").split() for _ in range(n_con)]
    nu, nd = nup - n_core_orb, ndn - n_core_orb
    up = np.array([_det_from_orbs(r[:nu], n_core_orb) for r in rows], np.uint64)
    dn = np.array([_det_from_orbs(r[nu:nu + nd], n_core_orb) for r in rows], np.uint64)
    num = np.array([float(r[nu + nd].lower().replace("d", "e")) for r in rows])
    den = np.array([float(r[nu + nd + 1].lower().replace("d", "e")) for r in rows])
    o = sort_dets(up, dn)
    return up[o], dn[o], num[o], den[o]


def write_dtm_elems(path, imp_up, imp_dn, counts, indices, h_values, dtm_energy, n_core_orb=0):
    """dtm_elems_out_file, do_walk.f90:970-1010: the deterministic space and its Hamiltonian (NOT yet
    multiplied by -tau) in the upper-triangular row format of the projector: `n_imp nnz energy`,
    the row counts, one line `i orbitals...` per determinant, one line `index value` per stored element."""
    with open(path, "w") as f:
        f.write(" %d %d %s number of deterministic dets, number of nonzero deterministic Hamiltonian elements, ground state energy within deterministic space\n"
                % (len(imp_up), len(h_values), repr(float(dtm_energy))))
        f.write("".join("%8d" % c for c in counts) + "\n")
        for i, (u, d_) in enumerate(zip(np.asarray(imp_up).tolist(), np.asarray(imp_dn).tolist())):
            f.write(" %d " % (i + 1) + " ".join("%d" % v for v in _orbs(u, n_core_orb) + _orbs(d_, n_core_orb)) + "\n")
        for ix, v in zip(np.asarray(indices).tolist(), np.asarray(h_values).tolist()):
            f.write(" %d %s\n" % (ix, repr(float(v))))


def read_dtm_elems(path, nup, ndn, n_core_orb=0):
    """the reader of do_walk.f90:898-940: (imp_up, imp_dn, counts, indices, H values, energy); multiply the
    values by -tau for sqmc_gpu_set_projector as the reference does at :943"""
    body = open(path).read().replace(",", " ").split()
    n_imp, nnz, e = int(body[0]), int(body[1]), float(body[2].lower().replace("d", "e"))
    pos = 3
    while not body[pos].lstrip("-").isdigit():       # the describing text of the first record (a compiler may wrap it over two lines)
        pos += 1
    body = body[pos:]
    counts = np.array(body[:n_imp], np.int64)
    pos = n_imp
    nu, nd = nup - n_core_orb, ndn - n_core_orb
    up, dn = np.zeros(n_imp, np.uint64), np.zeros(n_imp, np.uint64)
    for _ in range(n_imp):
        ind = int(body[pos]) - 1
        up[ind] = _det_from_orbs(body[pos + 1:pos + 1 + nu], n_core_orb)
        dn[ind] = _det_from_orbs(body[pos + 1 + nu:pos + 1 + nu + nd], n_core_orb)
        pos += 1 + nu + nd
    idx = np.array(body[pos:pos + 2 * nnz:2], np.int64)
    val = np.array([float(t.lower().replace("d", "e")) for t in body[pos + 1:pos + 2 * nnz:2]])
    return up, dn, counts, idx, val, e


def dump_hci_deck(path, host, hb, eps_var, eps_sched=(), n_states=1):
    """Tables + heat-bath lists + run parameters for a compiled HCI host (example_hci.f90), one
    little-endian stream file: int64 header, float64 scalars, then the arrays in header order."""
    prod, osym, c2 = (np.ascontiguousarray(a, np.int32).reshape(-1) for a in (host.prod, host.orbsym, host.combine_2))
    ints = _np_f64(host.integrals)
    hb_r, hb_s, hb_a, pq_ind, pq_count, max_double = hb
    sched = np.array(list(eps_sched) + [eps_var], np.float64)
    hdr = np.array([0x68636930, host.norb, host.nup, host.ndn, host.n_core_orb, int(host.time_sym), host.z, host.n_group, len(ints) - 1,
                    len(prod), len(osym), len(c2), len(hb_r), len(pq_ind), len(sched), n_states, host.hf_up, host.hf_dn], np.int64)
    assert len(pq_count) == len(pq_ind)
    with open(path, "wb") as f:
        for a in (hdr, np.array([max_double], np.float64), sched, prod, osym, c2, ints, np.ascontiguousarray(hb_r, np.int32),
                  np.ascontiguousarray(hb_s, np.int32), _np_f64(hb_a), np.ascontiguousarray(pq_ind, np.int64), np.ascontiguousarray(pq_count, np.int32)):
            f.write(a.tobytes())


# --------------------------------------------------------------- semistochastic PT (hci.f90:1314-1660)
class Rannyu:
    """The host's copy of the reference's generator (rannyu.f90:11-87, tools.f90:129-147): 48-bit LCG
    x <- x * 11^13 mod 2^48 seeded from four limbs (last one forced odd), r = x / 2^48,
    random_int(n) = int(n r) + 1.  Stream 1 of the input line drives the alias sampling of the
    stochastic PT (do_walk.f90:229-238)."""
    M, MASK = 34522712143931, (1 << 48) - 1

    def __init__(self, seed):
        l = list(seed); l[3] = 2 * (l[3] // 2) + 1
        self.x = (l[0] * (1 << 36) + l[1] * (1 << 24) + l[2] * (1 << 12) + l[3]) & self.MASK

    def rannyu(self):
        self.x = (self.x * self.M) & self.MASK
        return self.x * 3.552713678800500929355621337890625e-15

    def random_int(self, n):
        return int(n * self.rannyu()) + 1


def setup_alias(pdf):
    """setup_alias, more_tools.f90:5603-5662 (Walker's alias tables; outcomes are split in index order
    and paired from the END of the two lists, which fixes J and q to the last bit).  1-based J."""
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


def hci_pt2_stochastic(host, g, up, dn, coeffs, e_var, eps_pt, eps_pt_big, n_mc, target_error, seed=(2726, 5165, 6543, 6524),
                       max_samples=10**6, log=None):
    """second_order_pt_alias, hci.f90:1314-1660 (one rank): the PT correction at eps_pt as the
    deterministic correction at eps_pt_big (hci_pt2) plus a stochastic estimate of the difference.
    Each sample draws n_mc variational determinants with probability |c_i| / sum|c| (alias method,
    one random_int and one rannyu per draw), merges repeats (w_i copies), generates their connections
    on the GPU and accumulates, for every connected determinant k outside the variational space,
        term1 = sum_i H_ki c_i w_i/p_i        term2 = sum_i (H_ki c_i)^2 ((n_mc-1) w_i/p_i - (w_i/p_i)^2)
    over the connections with |H_ki c_i| above eps_pt, and the same above eps_pt_big; the sample's value
    is sum_k (term1^2 + term2 - term1_big^2 - term2_big) / (E_var - H_kk) / (n_mc (n_mc-1)).  Welford mean
    and variance; stops after >= 10 samples once the error bar is below target_error.
    The determinant list must be sorted by (up,dn) (hci.f90:1373-1380).
    Returns dict(pt_big, pt_diff, pt_diff_std_dev, samples=[per-sample values], n_connected_big)."""
    up, dn, c = np.ascontiguousarray(up, np.uint64), np.ascontiguousarray(dn, np.uint64), np.asarray(coeffs, float)
    order = sort_dets(up, dn)
    up, dn, c = up[order], dn[order], c[order]
    n = len(up)
    pt_big, n_big = hci_pt2(host, g, up, dn, c, e_var, eps_pt_big)
    prob = np.abs(c) / np.abs(c).sum()
    J, q = setup_alias(prob)
    rng = Rannyu(seed)
    mean = s_acc = var = 0.0
    values = []
    for sample in range(1, max_samples + 1):
        draws = np.empty(n_mc, np.int64)
        for k in range(n_mc):
            i = rng.random_int(n)
            draws[k] = i if rng.rannyu() < q[i - 1] else J[i - 1]
        ids, counts = np.unique(draws, return_counts=True)                 # sort_and_merge_count_repeats, tools.f90:1574-1602
        ci, wop = c[ids - 1], counts / prob[ids - 1]
        cu, cd, x, src = g.hci_connections(up[ids - 1], dn[ids - 1], ci, eps_pt, diag_mode=2)
        src = src.astype(np.int64)
        own = (cu == up[ids - 1][src]) & (cd == dn[ids - 1][src])           # the self slot of every reference determinant
        keep = ~own & ~_dets_in(cu, cd, up, dn)                             # connected determinants outside the variational space
        cu, cd, x, src = cu[keep], cd[keep], x[keep], src[keep]
        w1 = wop[src]
        a1, a2 = x * w1, x * x * ((n_mc - 1) * w1 - w1 * w1)
        big = np.abs(x) > eps_pt_big
        o = np.lexsort((cd, cu))
        cu, cd, a1, a2, big = cu[o], cd[o], a1[o], a2[o], big[o]
        head = np.ones(len(cu), bool); head[1:] = (cu[1:] != cu[:-1]) | (cd[1:] != cd[:-1])
        st = np.nonzero(head)[0]
        t1, t2 = np.add.reduceat(a1, st), np.add.reduceat(a2, st)
        t1b, t2b = np.add.reduceat(np.where(big, a1, 0.0), st), np.add.reduceat(np.where(big, a2, 0.0), st)
        h_kk = g.hamiltonian_batch(cu[st], cd[st], cu[st], cd[st])
        val = float(np.sum((t1 * t1 + t2 - t1b * t1b - t2b) / (e_var - h_kk))) / (n_mc * float(n_mc - 1))
        values.append(val)
        old = mean
        mean = mean + (val - mean) / sample                                 # welford, tools.f90:1761-1778
        s_acc = s_acc + (val - mean) * (val - old)
        var = s_acc / (sample - 1) / sample if sample > 1 else float("nan")
        if log:
            log("Sample, E_2pt_now, E_2pt estimate, total energy=%6d%15.9f%12.8f%15.8f +-%12.8f" % (sample, val, mean, e_var + pt_big + mean, np.sqrt(var) if sample > 1 else float("nan")))
        if sample >= 10 and var < target_error ** 2:
            break
    return dict(pt_big=pt_big, pt_diff=mean, pt_diff_std_dev=float(np.sqrt(var)), samples=values, n_connected_big=n_big)
