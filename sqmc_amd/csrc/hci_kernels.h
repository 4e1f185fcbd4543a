// hci_kernels.h -- Heat-bath CI connection generation kernels
// Textually included by sqmc_gpu.hip (one translation unit: the kernels share the ChemTab LDS
// image, the walker SoA types and the launch helpers defined there); not a standalone header.

#define HEG_LUT_MAX 729            // (2*4+1)^3: plane-wave indices up to +-4 per direction
// ================================================================ HCI connections
// find_important_connected_dets_chem, chemistry.f90:6819-7159: one thread per reference
// determinant; pass 0 counts, pass 1 writes at the scanned offsets.  Emits (up, dn,
// H_ij*c_j, e_mix_den) with the reference determinant itself in slot 0.
__global__ void __launch_bounds__(TPB) k_hci_gen(ChemDev dev, const u64 *__restrict__ rup, const u64 *__restrict__ rdn, const double *__restrict__ coef,
                                                 double eps_var, int diag_mode, long long n_ref, int pass, u64 *__restrict__ counts,
                                                 const u64 *__restrict__ offs, u64 *__restrict__ ou, u64 *__restrict__ od,
                                                 double *__restrict__ onum, double *__restrict__ oden, u64 key_lo, u64 key_hi, ActiveSpace as) {
  __shared__ ChemTab t;
  __shared__ unsigned char s_lut[HEG_LUT_MAX];        // plane wave (kx,ky,kz) -> orbital id, 0 = not in the basis (find_orb_id, heg.f90:752-771)
  stage_tab(&t, dev.tab, dev.tab_words);
  if (t.sys_type == 1) {
    const int W = 2 * t.heg_nmax + 1;
    for (int k = threadIdx.x; k < W * W * W; k += TPB) s_lut[k] = 0;
    __syncthreads();
    for (int o = 1 + threadIdx.x; o <= t.norb; o += TPB)
      s_lut[((t.krel[o][0] + t.heg_nmax) * W + (t.krel[o][1] + t.heg_nmax)) * W + (t.krel[o][2] + t.heg_nmax)] = (unsigned char)o;
    __syncthreads();
  }
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n_ref) return;
  const double c = coef[i];
  if (c == 0.0) { if (!pass) counts[i] = 0; return; }
  const double eps = eps_var / fabs(c);
  const u64 up = rup[i], dn = rdn[i];
  const int n = t.norb;
  const double sqrt2 = sqrt(2.0), sqrt2inv = 1.0 / sqrt2;
  u64 cnt = 0; const u64 base = pass ? offs[i] : 0;
  // diag_mode 2 ("raw", for the semistochastic PT): e_mix_den carries the index of the reference determinant instead
  // [key_lo, key_hi): only connections whose determinant key falls in this slice are kept -- the PT stage
  // of a large space is done in slices of the connected space, each with exact sums (the role of
  // n_energy_batch, hci.f90:642); the full range keeps everything without computing keys
  const bool sliced = !(key_lo == 0 && key_hi == ~0ull);
#define EMIT(U, D, M, DEN) do { bool in_ = true; if (sliced) { const u64 kk_ = det_key(dev, (U), (D)); in_ = (kk_ >= key_lo && kk_ < key_hi); } \
    if (in_) { if (pass) { ou[base + cnt] = (U); od[base + cnt] = (D); onum[base + cnt] = (M) * c; oden[base + cnt] = (diag_mode == 2) ? (double)i : (DEN); } cnt++; } } while (0)
  { double hd = (diag_mode == 1) ? h_any(t, dev.integrals, up, dn, up, dn) : 0.0; EMIT(up, dn, hd, c); }
  if (t.sys_type == 1) {
    // find_important_connected_dets_heg, heg.f90:2475-2727: no single excitations (momentum); every
    // double p,q -> r,s with k_p + k_q = k_r + k_s whose |H| exceeds eps/|c|.  The reference walks
    // |H|-sorted translation-invariant lists and stops at absH <= eps (:2608, :2629); here each
    // candidate's element is evaluated and screened -- the same set, 91 pairs x norb holes for 14
    // electrons.  Same-spin pairs take r < s (:2618).
    const int nm = t.heg_nmax, W = 2 * nm + 1;
    for (int cls = 0; cls < 3; cls++) {
      const u64 A = (cls == 1) ? dn : up, B = (cls == 0) ? up : dn;     // strings of the first / second electron
      for (u64 ea = A; ea; ea &= ea - 1) {
        const int pa = ctz64(ea) + 1;
        for (u64 eb = (cls == 2) ? B : (ea & (ea - 1)); eb; eb &= eb - 1) {
          const int qb = ctz64(eb) + 1;
          const int sx = t.krel[pa][0] + t.krel[qb][0], sy = t.krel[pa][1] + t.krel[qb][1], sz = t.krel[pa][2] + t.krel[qb][2];
          for (u64 hr = t.orb_mask & ~A; hr; hr &= hr - 1) {
            const int r = ctz64(hr) + 1;
            const int kx = sx - t.krel[r][0], ky = sy - t.krel[r][1], kz = sz - t.krel[r][2];
            if (kx < -nm || kx > nm || ky < -nm || ky > nm || kz < -nm || kz > nm) continue;
            const int s_ = s_lut[((kx + nm) * W + (ky + nm)) * W + (kz + nm)];
            if (!s_) continue;
            if (cls != 2 && s_ <= r) continue;
            if ((B >> (s_ - 1)) & 1) continue;
            u64 nu = up, nd = dn;
            if (cls == 0) nu = (up & ~bit64(pa - 1) & ~bit64(qb - 1)) | bit64(r - 1) | bit64(s_ - 1);
            else if (cls == 1) nd = (dn & ~bit64(pa - 1) & ~bit64(qb - 1)) | bit64(r - 1) | bit64(s_ - 1);
            else { nu = (up & ~bit64(pa - 1)) | bit64(r - 1); nd = (dn & ~bit64(qb - 1)) | bit64(s_ - 1); }
            const double mel = h_heg(t, up, dn, nu, nd);
            if (!(fabs(mel) > eps)) continue;
            EMIT(nu, nd, mel, 0.0);
          }
        }
      }
    }
    if (!pass) counts[i] = cnt;
    return;
  }
  // singles
  for (int sp = 0; sp < 2; sp++) {
    const u64 occ = sp ? dn : up;
    for (u64 e = occ; e; e &= e - 1) {
      const int pe = ctz64(e) + 1;
      for (u64 h = t.sym_mask[t.orbsym[pe]] & ~occ; h; h &= h - 1) {
        const int r = ctz64(h) + 1;
        u64 nu = up, nd = dn;
        if (!sp) nu = (up & ~bit64(pe - 1)) | bit64(r - 1); else nd = (dn & ~bit64(pe - 1)) | bit64(r - 1);
        if (active_space_skip(as, nu, nd)) continue;
        if (t.time_sym) { if (nu == nd && t.z < 0) continue; if (up == nd && dn == nu) continue; }
        double mel = h_single(t, dev.integrals, up, dn, nu, nd);
        if (fabs(mel) < eps) continue;
        if (t.time_sym) {
          if (up == dn && nu != nd) mel = sqrt2inv * mel;
          if (nu == nd && up != dn) mel = sqrt2 * mel;
          if (nu > nd) { u64 x = nu; nu = nd; nd = x; mel = t.z * mel; }
        }
        EMIT(nu, nd, mel, 0.0);
      }
    }
  }
  if (!(eps > dev.max_double)) {
    // occupied pairs: up-up, dn-dn, up-dn (chemistry.f90:7000-7021)
    for (int cls = 0; cls < 3; cls++) {
      const u64 A = (cls == 1) ? dn : up, B = (cls == 0) ? up : dn;
      for (u64 ea = A; ea; ea &= ea - 1) {
        const int pa = ctz64(ea) + 1;
        for (u64 eb = (cls == 2) ? B : (ea & (ea - 1)); eb; eb &= eb - 1) {
          const int qb = ctz64(eb) + 1;
          int p = pa + (cls == 1 ? n : 0), q = qb + (cls == 0 ? 0 : n);
          int p2 = p, q2 = q;
          const bool both_dn = (cls == 1), swapped = (cls == 2 && p > q - n);
          if (both_dn) { p2 = p - n; q2 = q - n; }
          if (swapped) { p2 = q - n; q2 = p + n; }
          const long long e = (p2 > q2) ? ((long long)p2 * (p2 - 1)) / 2 + q2 : ((long long)q2 * (q2 - 1)) / 2 + p2;
          const long long k0 = dev.pq_ind[e] - 1; const int kc = dev.pq_count[e];
          for (int hh = 0; hh < kc; hh++) {
            if (dev.hb_absH[k0 + hh] <= eps) break;
            int r = dev.hb_r[k0 + hh], s = dev.hb_s[k0 + hh];
            if (both_dn) { r += n; s += n; }
            if (swapped) { int rt = s - n; s = r + n; r = rt; }
            if (r <= n ? ((up >> (r - 1)) & 1) : ((dn >> (r - n - 1)) & 1)) continue;
            if (s <= n ? ((up >> (s - 1)) & 1) : ((dn >> (s - n - 1)) & 1)) continue;
            u64 nu = up, nd = dn;
            if (p <= n) nu &= ~bit64(p - 1); else nd &= ~bit64(p - n - 1);
            if (q <= n) nu &= ~bit64(q - 1); else nd &= ~bit64(q - n - 1);
            if (r <= n) nu |= bit64(r - 1); else nd |= bit64(r - n - 1);
            if (s <= n) nu |= bit64(s - 1); else nd |= bit64(s - n - 1);
            if (active_space_skip(as, nu, nd)) continue;
            if (t.time_sym) { if (nu == nd && t.z < 0) continue; if (up == nd && dn == nu) continue; }
            double mel = 0.0;
            if (pass) {
              mel = h_double(t, dev.integrals, up, dn, nu, nd);
              if (t.time_sym) {
                if (up == dn && nu != nd) mel = sqrt2inv * mel;
                if (nu == nd && up != dn) mel = sqrt2 * mel;
              }
            }
            if (t.time_sym && nu > nd) { u64 x = nu; nu = nd; nd = x; mel = t.z * mel; }
            EMIT(nu, nd, mel, 0.0);
          }
        }
      }
    }
  }
#undef EMIT
  if (!pass) counts[i] = cnt;
}
// dedup of the sorted connection list: sums e_mix_num / e_mix_den of equal determinants
// left to right (merge_original_with_spawned3, tools.f90:577-660)
__global__ void __launch_bounds__(TPB) k_hci_heads(const u64 *__restrict__ skey, u64 *__restrict__ flags, long long n) {
  long long j = (long long)blockIdx.x * TPB + threadIdx.x;
  if (j < n) flags[j] = (j == 0 || skey[j] != skey[j - 1]) ? 1ull : 0ull;
}
__global__ void __launch_bounds__(TPB) k_hci_dedup(const u64 *__restrict__ skey, const u32 *__restrict__ perm, const u64 *__restrict__ flags,
                                                   const u64 *__restrict__ pos, const u64 *__restrict__ iu, const u64 *__restrict__ id,
                                                   const double *__restrict__ inum, const double *__restrict__ iden,
                                                   u64 *__restrict__ ou, u64 *__restrict__ od, double *__restrict__ onum, double *__restrict__ oden, long long n) {
  long long j = (long long)blockIdx.x * TPB + threadIdx.x;
  if (j >= n || !flags[j]) return;
  const u64 key = skey[j]; u32 t = perm[j];
  double a = inum[t], b = iden[t];
  for (long long jj = j + 1; jj < n && skey[jj] == key; jj++) { a = a + inum[perm[jj]]; b = b + iden[perm[jj]]; }
  const u64 o = pos[j];
  ou[o] = iu[t]; od[o] = id[t]; onum[o] = a; oden[o] = b;
}
