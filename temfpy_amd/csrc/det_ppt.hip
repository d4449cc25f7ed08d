// Gathered determinants of a whole charge sector from ONE pivoted exchange of its matrix.
//
// Reference: slater.py:828-869 computes every minor det(M[rows(a)][:, cols(b)]) of a sector from
// scratch; det_reduced.hip shares one Gauss-Jordan elimination per bra row-set a.  Here the sharing is
// two-sided.  Pivoting the n x n block M* = M[a*, b*] out of the sb x sk sector matrix exchanges the
// roles of the variables y_{a*} and x_{b*} in y = M x (Tucker's principal pivot transform): the same
// graph {(x, M x)} is then described by a matrix G, and every Pluecker coordinate of the graph - every
// minor - can be read from either description:
//
//     det M[a, b] = (+-) det M* . det G[R, C],
//     R = (a \ a*)  u  { pivot rows of the columns b* \ b },   C = (b \ b*)  u  { pivot columns of the rows a* \ a }.
//
// The small determinant has order d = |a \ a*| + |b \ b*| - the number of orbitals in which the pair
// differs from the pivot configuration, 0..4 for almost every pair of a Slater -> MPS sweep - and G is
// computed once per workgroup by n exchange steps with FULL pivoting over the whole sector matrix
// (rows of the leading bra set and columns of the leading ket set preferred by a factor ~3 in
// magnitude), so it is as stable as a completely pivoted LU.  The per-bra-set eliminations, 85 % of the
// time of det_reduced.hip (measured), disappear.
//
// The sign is the product of five shuffle / sorting signs of the Pluecker correspondence; all of them
// reduce to O(d) mask operations and table look-ups (derivation and a brute-force check against
// numpy.linalg.det in tools/prototype_ppt_minors.py).  One lane owns one (a, b) pair: masks -> R, C, sign
// -> closed-form determinant for d <= 4; the few larger ones are queued in LDS and evaluated eight at a
// time by 8-lane groups (det_group, det_common.hpp).  The kernel is not specialised on n: one launch
// covers all sectors of all sites.
#include <stdlib.h>

#include "det_common.hpp"

namespace tmf {

namespace {

__device__ inline uint64_t below(int i) { return (i >= 64) ? ~0ull : ((1ull << i) - 1ull); }
__device__ inline uint64_t above(uint64_t m, int i) { return (i >= 63) ? 0ull : (m >> (i + 1)); }
// The same on 32-bit masks: almost every sector of a sweep has at most 32 rows and columns, and the kernel is bound by its
// VALU instruction count (PMC: 32 k instructions per wavefront, ~1000 per pair, most of them 64-bit shifts / popcounts /
// find-first-sets of the sign bookkeeping, each of which is 2-4 instructions on 32-bit ALUs).
template <typename M>
struct mk;
template <>
struct mk<uint64_t> {
  static constexpr int W = 64;
  __device__ static inline uint64_t below(int i) { return (i >= 64) ? ~0ull : ((1ull << i) - 1ull); }
  __device__ static inline uint64_t above(uint64_t m, int i) { return (i >= 63) ? 0ull : (m >> (i + 1)); }
  __device__ static inline int popc(uint64_t m) { return __popcll(m); }
  __device__ static inline int ffs(uint64_t m) { return __ffsll((unsigned long long)m); }
};
template <>
struct mk<uint32_t> {
  static constexpr int W = 32;
  __device__ static inline uint32_t below(int i) { return (i >= 32) ? ~0u : ((1u << i) - 1u); }
  __device__ static inline uint32_t above(uint32_t m, int i) { return (i >= 31) ? 0u : (m >> (i + 1)); }
  __device__ static inline int popc(uint32_t m) { return __popc(m); }
  __device__ static inline int ffs(uint32_t m) { return __ffs((int)m); }
};

template <typename T, int DD, int G>
__device__ __attribute__((noinline)) T small_det_group(const T* __restrict__ Gm, const int ld, uint64_t rmask, uint64_t cmask, const int d,
                                    const int c, T* __restrict__ scratch) {
  T a[DD];
  int col = 0;
  {
    uint64_t cm = cmask;
    for (int i = 0; i < c && i < d; ++i) cm &= cm - 1;
    col = (c < d) ? __ffsll((unsigned long long)cm) - 1 : 0;
  }
#pragma unroll
  for (int r = 0; r < DD; ++r) {
    T v = (r == c) ? sc<T>::one() : sc<T>::zero();
    if (r < d) {
      const int row = __ffsll((unsigned long long)rmask) - 1;
      rmask &= rmask - 1;
      v = (c < d) ? Gm[row + col * ld] : sc<T>::zero();
    } else if (c < d) {
      v = sc<T>::zero();
    }
    a[r] = v;
  }
  return det_group<T, DD, G>(a, c, scratch);
}

// Determinant of the n x n matrix A (LDS, column-major, leading dimension n, n <= 16) by Gaussian
// elimination with partial pivoting, the whole wavefront cooperating.  Only for the (never observed)
// pairs that differ from the pivot configuration in more than 8 orbitals: small code, few registers.
template <typename T>
__device__ inline T wave_lds_det(T* __restrict__ A, const int n, const int lane) {
  T det = sc<T>::one();
  for (int k = 0; k < n; ++k) {
    // pivot row of column k
    float best = -1.0f;
    int brow = k;
    for (int r = k + lane; r < n; r += 64) {
      const float v = (float)sc<T>::abs2(A[r + k * n]);
      if (v > best) best = v, brow = r;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int orow = __shfl_xor(brow, o);
      if (ob > best || (ob == best && orow < brow)) best = ob, brow = orow;
    }
    if (!(best > 0.0f)) return sc<T>::zero();
    __builtin_amdgcn_wave_barrier();
    if (brow != k) {
      for (int c = lane; c < n; c += 64) {
        const T t = A[k + c * n];
        A[k + c * n] = A[brow + c * n];
        A[brow + c * n] = t;
      }
      det = sc<T>::neg(det);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const T p = A[k + k * n];
    det = sc<T>::mul(det, p);
    const T pinv = sc<T>::inv_fast(p);
    const int m = n - 1 - k;
    for (int e = lane; e < m * m; e += 64) {
      const int r = k + 1 + e % m, c = k + 1 + e / m;
      A[r + c * n] = sc<T>::fms(A[r + c * n], sc<T>::mul(A[r + k * n], pinv), A[k + c * n]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  return det;
}

}  // namespace

// dynamic LDS: [ G : sb*sk T ][ kmask : nsk u64 ][ amask : na u64 ][ per wave: scratch max(264, n*n) T | queue 72 u32 ]
template <typename T, typename M>
__device__ __attribute__((always_inline)) inline void ppt_det_body(const tmf_det_desc& d, const float boost, unsigned char* __restrict__ smem,
                                                                    unsigned long long* __restrict__ dbg) {
  // diagnostic build only (TMF_PPT_STAMPS=1, tools/ppt_probe.py): cycles of the phases of a workgroup, summed over the launch
  unsigned long long t_s[5] = {0, 0, 0, 0, 0};
  auto stamp = [&](int i) {
    if (dbg) t_s[i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  __shared__ uint8_t row_of[64], col_of[64], invr[64], prow_seq[64], pcol_seq[64];
  __shared__ uint64_t s_PA, s_PB;
  __shared__ int s_singular, s_csector;
  __shared__ double s_prod[2];

  const int sb = d.sb, sk = d.sk, n = d.n, nsk = d.nsk, na = d.a1 - d.a0;
  T* Gm = reinterpret_cast<T*>(smem);                       // Gm[r + c * sb]
  size_t off = ((size_t)sb * sk * sizeof(T) + 15) & ~(size_t)15;
  M* kmask = reinterpret_cast<M*>(smem + off);          // (8-byte slots in both mask widths: the host sizes the LDS for u64)
  off += (size_t)nsk * 8;
  M* amask = reinterpret_cast<M*>(smem + off);
  off += (size_t)na * 8;
  off = (off + 15) & ~(size_t)15;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int SCR = (n * n > 8 * 33) ? n * n : 8 * 33;   // 8 groups x 33, or one n x n minor (fallback path)
  const size_t per_wave = (size_t)SCR * sizeof(T) + 72 * 4;
  T* scr_w = reinterpret_cast<T*>(smem + off + (size_t)wave * per_wave);
  uint32_t* queue = reinterpret_cast<uint32_t*>(scr_w + SCR);

  const T* __restrict__ S = reinterpret_cast<const T*>(d.S);
  for (int e = tid; e < sb * sk; e += 256) Gm[e] = S[(size_t)(e % sb) + (size_t)(e / sb) * d.lds];
  const uint8_t* __restrict__ gk = reinterpret_cast<const uint8_t*>(d.ket_idx);
  const uint8_t* __restrict__ gb = reinterpret_cast<const uint8_t*>(d.bra_idx);
  for (int b = tid; b < nsk; b += 256) {
    M m = 0;
    for (int u = 0; u < n; ++u) m |= M(1) << gk[(size_t)b * n + u];
    kmask[b] = m;
  }
  for (int a = tid; a < na; a += 256) {
    M m = 0;
    for (int u = 0; u < n; ++u) m |= M(1) << gb[(size_t)(d.a0 + a) * n + u];
    amask[a] = m;
  }
  // preferred pivots: rows of the sector's leading bra set, columns of its leading ket set
  M pref_r = 0, pref_c = 0;
  for (int u = 0; u < n; ++u) {
    pref_r |= M(1) << gb[u];
    pref_c |= M(1) << gk[u];
  }
  if (tid == 0) {
    s_PA = 0, s_PB = 0, s_singular = 0;
    s_prod[0] = 1.0, s_prod[1] = 0.0;
  }
  __syncthreads();

  stamp(1);
  // ---------------- n exchange steps with full pivoting over the sector matrix -----------------------
  // The search for the next pivot rides on the rank-1 update of the current step (every thread takes the maximum over the
  // elements it has just updated), the wavefront maxima go through DPP row operations instead of the LDS crossbar, and
  // the keys are double buffered: two barriers per step instead of five (the exchange was 20 - 30 % of the kernel).
  auto wave_max_key = [](unsigned k) -> unsigned {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x111, 0xf, 0xf, false), k = o > k ? o : k;   // row_shr:1
    o = (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x112, 0xf, 0xf, false), k = o > k ? o : k;   // row_shr:2
    o = (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x114, 0xf, 0xf, false), k = o > k ? o : k;   // row_shr:4
    o = (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x118, 0xf, 0xf, false), k = o > k ? o : k;   // row_shr:8
    o = (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x142, 0xf, 0xf, false), k = o > k ? o : k;   // row_bcast:15
    o = (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x143, 0xf, 0xf, false), k = o > k ? o : k;   // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
  };
  // 20 bits of magnitude (sign bit clear), 12 bits of element index (sb * sk <= 4096): ties -> larger index
  auto key_of = [&](const T v_, const int e, const int r, const int c) -> unsigned {
    float v = (float)sc<T>::abs2(v_);
    if ((pref_r >> r) & M(1)) v *= boost;
    if ((pref_c >> c) & M(1)) v *= boost;
    const unsigned mag = __float_as_uint(v) & ~4095u;
    return mag ? (mag | (unsigned)e) : 0u;
  };
  __shared__ unsigned red_key2[2][4];
  const int step_r = 256 % sb, step_c = 256 / sb;
  M usedR = 0, usedC = 0;   // uniform copies
  {
    unsigned key = 0u;
    for (int e = tid; e < sb * sk; e += 256) {
      const unsigned k = key_of(Gm[e], e, e % sb, e / sb);
      key = k > key ? k : key;
    }
    key = wave_max_key(key);
    if (lane == 0) red_key2[0][wave] = key;
    __syncthreads();
  }
  for (int t = 0; t < n; ++t) {
    unsigned key = red_key2[t & 1][0];
    for (int w = 1; w < 4; ++w) key = red_key2[t & 1][w] > key ? red_key2[t & 1][w] : key;
    if (key == 0u) {  // nothing left to pivot on: rank(M) < n, every minor of order n vanishes
      if (tid == 0) s_singular = 1;
      __syncthreads();
      break;
    }
    const int pe = (int)(key & 4095u), pr = pe % sb, pc = pe / sb;
    const T p = Gm[pe];
    const T pinv = sc<T>::inv_fast(p);
    usedR |= M(1) << pr, usedC |= M(1) << pc;
    // rank-1 part on the elements outside the pivot row and column, and the search among what is still free
    unsigned nkey = 0u;
    int r = tid % sb, c = tid / sb;            // (row, column) of element e, advanced without divisions
    for (int e = tid; e < sb * sk; e += 256, r += step_r, c += step_c) {
      if (r >= sb) r -= sb, ++c;
      if (r == pr || c == pc) continue;
      const T f = sc<T>::mul(Gm[r + pc * sb], pinv);
      const T v = sc<T>::fms(Gm[e], f, Gm[pr + c * sb]);
      Gm[e] = v;
      if (!(((usedR >> r) | (usedC >> c)) & M(1))) {
        const unsigned k = key_of(v, e, r, c);
        nkey = k > nkey ? k : nkey;
      }
    }
    nkey = wave_max_key(nkey);
    if (lane == 0) red_key2[(t + 1) & 1][wave] = nkey;
    __syncthreads();   // the pivot row and column have been read by everybody
    for (int e = tid; e < sb + sk; e += 256) {
      if (e < sb) {        // pivot column: G[i, pc] = M[i, pc] / p
        if (e != pr) Gm[e + pc * sb] = sc<T>::mul(Gm[e + pc * sb], pinv);
      } else {             // pivot row: G[pr, j] = -M[pr, j] / p
        const int c = e - sb;
        if (c != pc) Gm[pr + c * sb] = sc<T>::neg(sc<T>::mul(Gm[pr + c * sb], pinv));
      }
    }
    if (tid == 0) {
      Gm[pe] = pinv;
      row_of[pc] = (uint8_t)pr, col_of[pr] = (uint8_t)pc;
      prow_seq[t] = (uint8_t)pr, pcol_seq[t] = (uint8_t)pc;
      s_PA |= 1ull << pr, s_PB |= 1ull << pc;
      const T q = sc<T>::mul(sc<T>::from2(s_prod[0], s_prod[1]), p);
      s_prod[0] = sc<T>::real(q), s_prod[1] = sc<T>::imag(q);
    }
    __syncthreads();
  }
  stamp(2);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  if (s_singular) {
    for (int64_t e = tid; e < (int64_t)na * nsk; e += 256) out[(size_t)d.a0 * nsk + e] = sc<T>::zero();
    return;
  }
  const M PA = (M)s_PA, PB = (M)s_PB;
  const M NPB = mk<M>::below(sk) & ~PB;
  // tables: invr[r] = inversions of the sequence col_of over the pivot rows (ascending) that involve r;
  // sign of det M*_sorted relative to the product of the pivots = parity of the two pivot sequences
  if (tid < 64) {
    const int r = tid;
    int v = 0;
    if (r < mk<M>::W && ((PA >> r) & M(1))) {
      const int cr = col_of[r];
      for (M m = PA & mk<M>::below(r); m; m &= m - 1) v += col_of[mk<M>::ffs(m) - 1] > cr;
      for (M m = mk<M>::above(PA, r); m; m &= m - 1) v += col_of[mk<M>::ffs(m) + r] < cr;
    }
    invr[r] = (uint8_t)v;
  }
  if (tid == 64) {
    int inv = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) inv += (prow_seq[i] > prow_seq[j]) + (pcol_seq[i] > pcol_seq[j]);
    s_csector = inv & 1;
  }
  __syncthreads();
  // Everything in the sign and the masks of a pair that depends on the ket set only is computed ONCE per ket set (32-bit
  // masks; kmask[b] is replaced by bin = its columns outside the pivot set, the rows of its leaving pivot columns go into
  // the spare half of the 8-byte mask slots, the parity of the ket-only terms into a bit array).  What is left per pair
  // are two cross terms whose loops run over the bits of the BRA set, i.e. are uniform in the wavefront:
  //   #{(x, r1): x in A.in, r1 in Rb, x < r1}  +  #{(y, c0): y in A.cm, c0 in bin, y < c0}.
  __shared__ uint32_t parbits[64];
  constexpr bool FAST = sizeof(M) == 4;
  const bool fast = FAST && nsk <= 2048;
  M* rbm = reinterpret_cast<M*>(reinterpret_cast<unsigned char*>(kmask) + (size_t)nsk * 4);   // (FAST only)
  if (fast) {
    if (tid < 64) parbits[tid] = 0u;
    __syncthreads();
    for (int b = tid; b < nsk; b += 256) {
      const M bm = kmask[b];
      const M bin = bm & ~PB, bout = PB & ~bm;
      int par = 0;
      M Rb = 0, seen = 0;
      for (M m = bout; m; m &= m - 1) {
        const int c1 = mk<M>::ffs(m) - 1, r1 = row_of[c1];
        par += c1 + mk<M>::popc(mk<M>::above(NPB, c1)) + mk<M>::popc(mk<M>::above(bin, c1)) + mk<M>::popc(mk<M>::above(seen, r1));
        seen |= M(1) << r1;
        Rb |= M(1) << r1;
      }
      for (M m = bin; m; m &= m - 1) {
        const int c0 = mk<M>::ffs(m) - 1;
        par += c0 + mk<M>::popc(PB & mk<M>::below(c0)) + c0;          // T1, I_X0Y1 (ket part), and c0's share of csum (T5)
      }
      kmask[b] = bin;
      rbm[b] = Rb;
      if (par & 1) atomicOr(&parbits[b >> 5], 1u << (b & 31));
    }
    __syncthreads();
  }
  const T scale = *reinterpret_cast<const T*>(d.scale);
  const T pref_fac = sc<T>::mul(scale, sc<T>::from2(s_prod[0], s_prod[1]));
  const int csec = s_csector;
  const int grp = lane >> 3, c8 = lane & 7;
  T* scr_g = scr_w + grp * 33;

  // everything that depends on the bra set only (uniform in the wavefront)
  struct ASide {
    M in, out_, cm;   // rows entering, pivot rows leaving, pivot columns of the leaving rows
    int da, par;
  };
  auto a_side = [&](const M am) {
    ASide s;
    s.in = am & ~PA, s.out_ = PA & ~am, s.cm = 0;
    s.da = mk<M>::popc(s.in);
    int par = 0;
    M seen = 0;
    for (M m = s.out_; m; m &= m - 1) {
      const int r = mk<M>::ffs(m) - 1, cr = col_of[r];
      par += mk<M>::popc(mk<M>::above(NPB, cr)) + invr[r] + mk<M>::popc(mk<M>::above(seen, cr));   // I_X0Y1, I_Y1Y1 (two parts)
      seen |= M(1) << cr;
      s.cm |= M(1) << cr;
    }
    for (M m = s.in; m; m &= m - 1) {
      const int r0 = mk<M>::ffs(m) - 1;
      par += mk<M>::popc(mk<M>::above(PA, r0)) + mk<M>::popc(mk<M>::above(s.out_, r0));            // I_Y0Y1
    }
    s.par = par;
    return s;
  };
  // the pair: masks of the small determinant and the parity of the sign
  auto pair = [&](const ASide& A, const M bm, M& Rm, M& Cm) {
    const M bin = bm & ~PB, bout = PB & ~bm;
    const int db = mk<M>::popc(bin);
    int par = csec + A.par + db * (n - A.da);                                  // sector, bra part, I_X1Y1
    Rm = A.in;
    M seen = 0;
    for (M m = bout; m; m &= m - 1) {
      const int c1 = mk<M>::ffs(m) - 1, r1 = row_of[c1];
      par += c1 + mk<M>::popc(mk<M>::above(NPB, c1)) + mk<M>::popc(mk<M>::above(bin, c1))          // T1, I_X0X1
             + mk<M>::popc(mk<M>::above(seen, r1)) + mk<M>::popc(A.in & mk<M>::below(r1));         // I_X1X1, I_X1Y0
      seen |= M(1) << r1;
      Rm |= M(1) << r1;
    }
    for (M m = bin; m; m &= m - 1) {
      const int c0 = mk<M>::ffs(m) - 1;
      par += c0 + mk<M>::popc(PB & mk<M>::below(c0)) + mk<M>::popc(A.cm & mk<M>::below(c0));       // T1, I_X0Y1 (two parts)
    }
    Cm = bin | A.cm;
    const int dd = mk<M>::popc(Cm);
    int csum = 0;
    for (M m = Cm; m; m &= m - 1) csum += mk<M>::ffs(m) - 1;
    par += dd * (sk - 1) + csum + ((dd * (dd - 1)) >> 1);                       // T5
    return par;
  };

  // bra-only extras of the fast path: csum over A.cm (T5) is folded into the bra parity
  auto pair_fast = [&](const ASide& A, const int acm_sum, const int b, M& Rm, M& Cm) {
    const M bin = kmask[b], Rb = rbm[b];
    const int db = mk<M>::popc(bin), dd = db + A.da;
    int par = csec + A.par + acm_sum + (int)((parbits[b >> 5] >> (b & 31)) & 1u) + db * (n - A.da);
    for (M m = A.in; m; m &= m - 1) par += mk<M>::popc(mk<M>::above(Rb, mk<M>::ffs(m) - 1));     // I_X1Y0
    for (M m = A.cm; m; m &= m - 1) par += mk<M>::popc(mk<M>::above(bin, mk<M>::ffs(m) - 1));    // I_X0Y1 (cross part)
    par += dd * (sk - 1) + ((dd * (dd - 1)) >> 1);                                                // T5 without csum
    Rm = A.in | Rb;
    Cm = bin | A.cm;
    return par;
  };
  auto cm_sum = [&](const ASide& A) {
    int v = 0;
    for (M m = A.cm; m; m &= m - 1) v += mk<M>::ffs(m) - 1;
    return v;
  };

  auto slow_batch = [&](const uint32_t item, const bool live) {
    const int al = live ? (int)(item >> 16) & 0x7fff : 0, b = live ? (int)(item & 0xffffu) : 0;
    const ASide A = a_side(amask[al]);
    M Rm, Cm;
    const int par = fast ? pair_fast(A, cm_sum(A), b, Rm, Cm) : pair(A, kmask[b], Rm, Cm);
    const int dd = live ? mk<M>::popc(Cm) : 0;
    unsigned dmax = (unsigned)dd;
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned other = (unsigned)__shfl_xor((int)dmax, o);
      dmax = other > dmax ? other : dmax;
    }
    T det = sc<T>::one();
    bool direct = false;
    if (dmax <= 8u) {
      det = small_det_group<T, 8, 8>(Gm, sb, (uint64_t)Rm, (uint64_t)Cm, dd, c8, scr_g);
    } else {
      // (never seen in a sweep: more than 8 exchanged orbitals.)  The 8 pairs are evaluated one after the
      // other as full n x n minors of the ORIGINAL matrix, gathered from global memory into the
      // wavefront's LDS scratch (n <= 32): no exchange identity, no sign bookkeeping.
      direct = true;
      for (int g = 0; g < 8; ++g) {
        const uint32_t it = (uint32_t)__shfl((int)item, g * 8);
        const int lv = __shfl((int)live, g * 8);
        const int ag = (int)(it >> 16) & 0x7fff, bg = (int)(it & 0xffffu);
        const uint8_t* ra = gb + (size_t)(d.a0 + ag) * n;
        const uint8_t* cb = gk + (size_t)bg * n;
        for (int e = lane; e < n * n; e += 64) scr_w[e] = S[(size_t)ra[e % n] + (size_t)cb[e / n] * d.lds];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const T dt = wave_lds_det<T>(scr_w, n, lane);
        if (grp == g && lv) det = dt;
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (live && c8 == 0) {
      T v = direct ? sc<T>::mul(scale, det) : sc<T>::mul(pref_fac, det);
      if (!direct && (par & 1)) v = sc<T>::neg(v);
      out[(size_t)(d.a0 + al) * nsk + b] = v;
    }
  };

  // ------------------------------------------------------------------------------------------------------------------
  // Pair phase, 32-bit masks: pairs grouped by the ORDER d of their small determinant.
  // d = da + db with da = rows of the bra set outside the pivot rows, db = columns of the ket set outside the pivot
  // columns.  Bra sets sorted by da and ket sets by db (counting sorts in LDS) make d uniform in a wavefront: a unit of
  // work is a block of up to 8 bra sets of one da, processed ket class by ket class with the 64 lanes spread over the
  // (bra, ket) pairs of the class.  The determinant is then ONE straight-line closed form per wavefront instead of a
  // four-way divergent branch, orders 5 are closed forms too (3 % of the pairs of the benchmark took 45 % of the pair
  // phase on the queued 8-lane path: measured with the stamps below), and a unit writes all kets of its rows within a
  // few microseconds (the partial lines merge in L2).  d >= 6 (0.03 %) keeps the queue.
  // ------------------------------------------------------------------------------------------------------------------
  if constexpr (FAST) {
    if (fast) {
      __shared__ int s_kcnt[33], s_koff[34], s_acnt[33], s_aoff[34], s_uoff[34], s_ticket;   // s_uoff: units (blocks of 8 bra sets) before class da
      uint32_t* s_ain = reinterpret_cast<uint32_t*>(smem + off + 4 * per_wave);
      uint32_t* s_acm = s_ain + na;
      uint16_t* s_ainfo = reinterpret_cast<uint16_t*>(s_acm + na);   // da | parity << 8
      uint16_t* s_aord = s_ainfo + na;
      uint16_t* s_kord = s_aord + na;
      uint8_t* s_kdb = reinterpret_cast<uint8_t*>(s_kord + nsk);
      if (tid < 33) s_kcnt[tid] = 0, s_acnt[tid] = 0;
      if (tid == 0) s_ticket = 0;
      __syncthreads();
      for (int a = tid; a < na; a += 256) {
        const ASide A = a_side(amask[a]);
        s_ain[a] = A.in, s_acm[a] = A.cm;
        s_ainfo[a] = (uint16_t)(A.da | (((csec + A.par + cm_sum(A)) & 1) << 8));
        atomicAdd(&s_acnt[A.da], 1);
      }
      for (int b = tid; b < nsk; b += 256) {
        const int db = __popc(kmask[b]);
        s_kdb[b] = (uint8_t)db;
        atomicAdd(&s_kcnt[db], 1);
      }
      __syncthreads();
      if (wave == 0) {   // exclusive prefix sums over the 33 classes, one lane per class
        const int c = lane;
        const int kc = c < 33 ? s_kcnt[c] : 0, ac = c < 33 ? s_acnt[c] : 0, uc = (ac + 7) >> 3;
        int ks = kc, as = ac, us = uc;
        for (int o = 1; o < 64; o <<= 1) {
          const int k2 = __shfl_up(ks, o), a2 = __shfl_up(as, o), u2 = __shfl_up(us, o);
          if (lane >= o) ks += k2, as += a2, us += u2;
        }
        if (c < 33) s_koff[c] = ks - kc, s_aoff[c] = as - ac, s_uoff[c] = us - uc;
        if (c == 32) s_koff[33] = ks, s_aoff[33] = as, s_uoff[33] = us;
      }
      __syncthreads();
      // stable scatter into the class order: wavefront 0 the kets, wavefront 1 the bra sets
      if (wave < 2) {
        const int cnt = wave == 0 ? nsk : na;
        for (int c = 0; c < 33; ++c) {
          const int total = wave == 0 ? s_kcnt[c] : s_acnt[c];
          if (total == 0) continue;
          int pos = (wave == 0 ? s_koff[c] : s_aoff[c]);
          for (int i0 = 0; i0 < cnt; i0 += 64) {
            const int i = i0 + lane;
            const int cls = i < cnt ? (wave == 0 ? (int)s_kdb[i] : (int)(s_ainfo[i] & 0xff)) : -1;
            const uint64_t mk_ = __ballot(cls == c);
            if (cls == c) (wave == 0 ? s_kord : s_aord)[pos + __popcll(mk_ & below(lane))] = (uint16_t)i;
            pos += __popcll(mk_);
          }
        }
      }
      __syncthreads();
      stamp(3);

      // closed forms on G, one lane per pair, order uniform in the wavefront
      auto ld = [&](int i, int j) -> T { return Gm[i + j]; };
      auto det2 = [&](const int* iv, const int* jv) -> T {
        const T m00 = ld(iv[0], jv[0]), m10 = ld(iv[1], jv[0]), m01 = ld(iv[0], jv[1]), m11 = ld(iv[1], jv[1]);
        return sc<T>::fms(sc<T>::mul(m00, m11), m01, m10);
      };
      auto det3 = [&](const int* iv, const int* jv) -> T {
        const T m00 = ld(iv[0], jv[0]), m10 = ld(iv[1], jv[0]), m20 = ld(iv[2], jv[0]);
        const T m01 = ld(iv[0], jv[1]), m11 = ld(iv[1], jv[1]), m21 = ld(iv[2], jv[1]);
        const T m02 = ld(iv[0], jv[2]), m12 = ld(iv[1], jv[2]), m22 = ld(iv[2], jv[2]);
        const T c0 = sc<T>::fms(sc<T>::mul(m11, m22), m12, m21);
        const T c1 = sc<T>::fms(sc<T>::mul(m10, m22), m12, m20);
        const T c2 = sc<T>::fms(sc<T>::mul(m10, m21), m11, m20);
        return sc<T>::fmac(sc<T>::fms(sc<T>::mul(m00, c0), m01, c1), m02, c2);
      };
      auto det4 = [&](const int* iv, const int* jv) -> T {      // Laplace expansion along the first two columns
        T p01, p02, p03, p12, p13, p23;
        {
          const T a0 = ld(iv[0], jv[0]), a1 = ld(iv[1], jv[0]), a2 = ld(iv[2], jv[0]), a3 = ld(iv[3], jv[0]);
          const T b0_ = ld(iv[0], jv[1]), b1 = ld(iv[1], jv[1]), b2 = ld(iv[2], jv[1]), b3 = ld(iv[3], jv[1]);
          p01 = sc<T>::fms(sc<T>::mul(a0, b1), b0_, a1);
          p02 = sc<T>::fms(sc<T>::mul(a0, b2), b0_, a2);
          p03 = sc<T>::fms(sc<T>::mul(a0, b3), b0_, a3);
          p12 = sc<T>::fms(sc<T>::mul(a1, b2), b1, a2);
          p13 = sc<T>::fms(sc<T>::mul(a1, b3), b1, a3);
          p23 = sc<T>::fms(sc<T>::mul(a2, b3), b2, a3);
        }
        const T c0 = ld(iv[0], jv[2]), c1 = ld(iv[1], jv[2]), c2 = ld(iv[2], jv[2]), c3 = ld(iv[3], jv[2]);
        const T e0 = ld(iv[0], jv[3]), e1 = ld(iv[1], jv[3]), e2 = ld(iv[2], jv[3]), e3 = ld(iv[3], jv[3]);
        T det = sc<T>::mul(p01, sc<T>::fms(sc<T>::mul(c2, e3), e2, c3));
        det = sc<T>::fms(det, p02, sc<T>::fms(sc<T>::mul(c1, e3), e1, c3));
        det = sc<T>::fmac(det, p03, sc<T>::fms(sc<T>::mul(c1, e2), e1, c2));
        det = sc<T>::fmac(det, p12, sc<T>::fms(sc<T>::mul(c0, e3), e0, c3));
        det = sc<T>::fms(det, p13, sc<T>::fms(sc<T>::mul(c0, e2), e0, c2));
        det = sc<T>::fmac(det, p23, sc<T>::fms(sc<T>::mul(c0, e1), e0, c1));
        return det;
      };

      int qn = 0;
      for (;;) {
        int u = 0;
        if (lane == 0) u = atomicAdd(&s_ticket, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= s_uoff[33]) break;
        int da = 0;
        while (s_uoff[da + 1] <= u) ++da;
        da = __builtin_amdgcn_readfirstlane(da);
        const int i0 = 8 * (u - __builtin_amdgcn_readfirstlane(s_uoff[da]));
        const int rows = __builtin_amdgcn_readfirstlane(min(8, s_acnt[da] - i0)), abase = __builtin_amdgcn_readfirstlane(s_aoff[da]) + i0;
        for (int db = 0; db < 33; ++db) {
          const int cb = __builtin_amdgcn_readfirstlane(s_kcnt[db]);
          if (cb == 0) continue;
          const int dd = da + db, kbase = __builtin_amdgcn_readfirstlane(s_koff[db]), npair = rows * cb;
          const float rcb = 1.0f / (float)cb;
          const int base_par = db * (n - da) + dd * (sk - 1) + ((dd * (dd - 1)) >> 1);
          for (int p0 = 0; p0 < npair; p0 += 64) {
            const int pidx = p0 + lane;
            const bool live = pidx < npair;
            int r = (int)((float)pidx * rcb);                 // pidx / cb for pidx < 2^14: one correction step
            r -= (r * cb > pidx);
            r += ((r + 1) * cb <= pidx);
            const int j = pidx - r * cb;
            const int al = live ? (int)s_aord[abase + r] : (int)s_aord[abase];
            const int b = live ? (int)s_kord[kbase + j] : (int)s_kord[kbase];
            const M ain = s_ain[al], acm = s_acm[al], bin = kmask[b], Rb = rbm[b];
            int par = base_par + (s_ainfo[al] >> 8) + (int)((parbits[b >> 5] >> (b & 31)) & 1u);
            {
              M m1 = ain, m2 = acm;
              for (int t = 0; t < da; ++t) {                   // (trip count uniform in the wavefront)
                par += mk<M>::popc(mk<M>::above(Rb, mk<M>::ffs(m1) - 1));
                par += mk<M>::popc(mk<M>::above(bin, mk<M>::ffs(m2) - 1));
                m1 &= m1 - 1, m2 &= m2 - 1;
              }
            }
            M Rm = ain | Rb, Cm = bin | acm;
            if (dd > 5) {                                      // the queued path of the general kernel
              const uint64_t sm = __ballot(live);
              if (live) queue[qn + __popcll(sm & below(lane))] = ((uint32_t)al << 16) | (uint32_t)b;
              qn += __popcll(sm);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              while (qn >= 8) {
                qn -= 8;
                slow_batch(queue[qn + grp], true);
              }
              continue;
            }
            int iv[5], jv[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
              iv[t] = 0, jv[t] = 0;
              if (t < dd) {
                iv[t] = mk<M>::ffs(Rm) - 1;
                jv[t] = (mk<M>::ffs(Cm) - 1) * sb;
                Rm &= Rm - 1, Cm &= Cm - 1;
              }
            }
            T det = sc<T>::one();
            switch (dd) {
              case 0: break;
              case 1: det = ld(iv[0], jv[0]); break;
              case 2: det = det2(iv, jv); break;
              case 3: det = det3(iv, jv); break;
              case 4: det = det4(iv, jv); break;
              default: {   // 5: expansion along the first column, five minors of order 4
                det = sc<T>::zero();
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                  int ir[4];
#pragma unroll
                  for (int t = 0; t < 4; ++t) ir[t] = iv[t + (t >= i)];
                  const T mi = sc<T>::mul(ld(iv[i], jv[0]), det4(ir, jv + 1));
                  det = (i & 1) ? sc<T>::sub(det, mi) : sc<T>::add(det, mi);
                }
              }
            }
            if (live) {
              T v = sc<T>::mul(pref_fac, det);
              if (par & 1) v = sc<T>::neg(v);
              out[(size_t)(d.a0 + al) * nsk + b] = v;
            }
          }
        }
      }
      if (qn > 0) slow_batch(queue[grp < qn ? grp : 0], grp < qn);
      if (dbg && lane == 0) {
        stamp(4);
        for (int i = 0; i < 4; ++i) atomicAdd(&dbg[4 * wave + i], t_s[i + 1] - t_s[i]);
        if (wave == 0) atomicAdd(&dbg[16], 1ull), atomicAdd(&dbg[17], (unsigned long long)na * nsk), atomicAdd(&dbg[18], (unsigned long long)n);
      }
      return;
    }
  }
  stamp(3);
  unsigned long long acc[4] = {0, 0, 0, 0}, hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;   // diagnostics
  int qn = 0;  // pairs waiting for the slow path (uniform in the wavefront)
  for (int al = wave; al < na; al += 4) {
    if (dbg) tq = __builtin_amdgcn_s_memtime();
    const ASide A = a_side(amask[al]);
    const int acm = cm_sum(A);
    T* __restrict__ orow = out + (size_t)(d.a0 + al) * nsk;
    if (dbg) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      acc[0] += t - tq, tq = t;
    }
    for (int b0 = 0; b0 < nsk; b0 += 64) {
      const int b = b0 + lane;
      const bool live = b < nsk;
      M Rm = 0, Cm = 0;
      int par = 0, dd = 0;
      if (live) {
        par = fast ? pair_fast(A, acm, b, Rm, Cm) : pair(A, kmask[b], Rm, Cm);
        dd = mk<M>::popc(Cm);
      }
      if (dbg) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        acc[1] += t - tq, tq = t;
        for (int k = 0; k < 8; ++k) hist[k] += __popcll(__ballot(live && (k < 7 ? dd == k : dd >= 7)));
      }
      // (order 5 by the same closed form as the classed pair phase of the 32-bit kernel: the two kernels must do the same
      // arithmetic pair by pair - which of them runs depends on the widest sector of the LAUNCH, and a rank's shard has to
      // reproduce the unsharded conversion bit for bit, tests/soak/soak_shards.py seed 60313)
      const bool slow = live && dd > (FAST ? 4 : 5);   // (the 32-bit kernel comes here only for sectors of more than 2048 ket sets, which the drivers give to the 64-bit kernel: keeping the order-5 form out of it keeps its registers at three wavefronts per SIMD)
      if (live && !slow) {
        int iv[5], jv[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          iv[t] = 0, jv[t] = 0;
          if (t < dd) {
            iv[t] = mk<M>::ffs(Rm) - 1;
            jv[t] = (mk<M>::ffs(Cm) - 1) * sb;
            Rm &= Rm - 1, Cm &= Cm - 1;
          }
        }
        T det = sc<T>::one();
        if (dd == 1) {
          det = Gm[iv[0] + jv[0]];
        } else if (dd == 2) {
          const T m00 = Gm[iv[0] + jv[0]], m10 = Gm[iv[1] + jv[0]], m01 = Gm[iv[0] + jv[1]], m11 = Gm[iv[1] + jv[1]];
          det = sc<T>::fms(sc<T>::mul(m00, m11), m01, m10);
        } else if (dd == 3) {
          const T m00 = Gm[iv[0] + jv[0]], m10 = Gm[iv[1] + jv[0]], m20 = Gm[iv[2] + jv[0]];
          const T m01 = Gm[iv[0] + jv[1]], m11 = Gm[iv[1] + jv[1]], m21 = Gm[iv[2] + jv[1]];
          const T m02 = Gm[iv[0] + jv[2]], m12 = Gm[iv[1] + jv[2]], m22 = Gm[iv[2] + jv[2]];
          const T c0 = sc<T>::fms(sc<T>::mul(m11, m22), m12, m21);
          const T c1 = sc<T>::fms(sc<T>::mul(m10, m22), m12, m20);
          const T c2 = sc<T>::fms(sc<T>::mul(m10, m21), m11, m20);
          det = sc<T>::fmac(sc<T>::fms(sc<T>::mul(m00, c0), m01, c1), m02, c2);
        } else if (dd == 4) {
          // Laplace expansion along the first two columns: sum over row pairs (p < q) of
          // (-1)^(p+q+1) |M[pq; 01]| |M[rs; 23]|  with {r, s} the complementary rows
          // (the six 2 x 2 minors of the first two columns are formed before the last two columns are
          // loaded: half the live registers of loading all 16 entries first)
          T p01, p02, p03, p12, p13, p23;
          {
            const T a0 = Gm[iv[0] + jv[0]], a1 = Gm[iv[1] + jv[0]], a2 = Gm[iv[2] + jv[0]], a3 = Gm[iv[3] + jv[0]];
            const T b0_ = Gm[iv[0] + jv[1]], b1 = Gm[iv[1] + jv[1]], b2 = Gm[iv[2] + jv[1]], b3 = Gm[iv[3] + jv[1]];
            p01 = sc<T>::fms(sc<T>::mul(a0, b1), b0_, a1);
            p02 = sc<T>::fms(sc<T>::mul(a0, b2), b0_, a2);
            p03 = sc<T>::fms(sc<T>::mul(a0, b3), b0_, a3);
            p12 = sc<T>::fms(sc<T>::mul(a1, b2), b1, a2);
            p13 = sc<T>::fms(sc<T>::mul(a1, b3), b1, a3);
            p23 = sc<T>::fms(sc<T>::mul(a2, b3), b2, a3);
          }
          const T c0 = Gm[iv[0] + jv[2]], c1 = Gm[iv[1] + jv[2]], c2 = Gm[iv[2] + jv[2]], c3 = Gm[iv[3] + jv[2]];
          const T e0 = Gm[iv[0] + jv[3]], e1 = Gm[iv[1] + jv[3]], e2 = Gm[iv[2] + jv[3]], e3 = Gm[iv[3] + jv[3]];
          det = sc<T>::mul(p01, sc<T>::fms(sc<T>::mul(c2, e3), e2, c3));
          det = sc<T>::fms(det, p02, sc<T>::fms(sc<T>::mul(c1, e3), e1, c3));
          det = sc<T>::fmac(det, p03, sc<T>::fms(sc<T>::mul(c1, e2), e1, c2));
          det = sc<T>::fmac(det, p12, sc<T>::fms(sc<T>::mul(c0, e3), e0, c3));
          det = sc<T>::fms(det, p13, sc<T>::fms(sc<T>::mul(c0, e2), e0, c2));
          det = sc<T>::fmac(det, p23, sc<T>::fms(sc<T>::mul(c0, e1), e0, c1));
        } else if (!FAST && dd == 5) {      // expansion along the first column, five minors of order 4 (as in the classed phase)
          auto det4g = [&](const int* ir, const int* jc) -> T {
            T p01, p02, p03, p12, p13, p23;
            {
              const T a0 = Gm[ir[0] + jc[0]], a1 = Gm[ir[1] + jc[0]], a2 = Gm[ir[2] + jc[0]], a3 = Gm[ir[3] + jc[0]];
              const T b0_ = Gm[ir[0] + jc[1]], b1 = Gm[ir[1] + jc[1]], b2 = Gm[ir[2] + jc[1]], b3 = Gm[ir[3] + jc[1]];
              p01 = sc<T>::fms(sc<T>::mul(a0, b1), b0_, a1);
              p02 = sc<T>::fms(sc<T>::mul(a0, b2), b0_, a2);
              p03 = sc<T>::fms(sc<T>::mul(a0, b3), b0_, a3);
              p12 = sc<T>::fms(sc<T>::mul(a1, b2), b1, a2);
              p13 = sc<T>::fms(sc<T>::mul(a1, b3), b1, a3);
              p23 = sc<T>::fms(sc<T>::mul(a2, b3), b2, a3);
            }
            const T c0 = Gm[ir[0] + jc[2]], c1 = Gm[ir[1] + jc[2]], c2 = Gm[ir[2] + jc[2]], c3 = Gm[ir[3] + jc[2]];
            const T e0 = Gm[ir[0] + jc[3]], e1 = Gm[ir[1] + jc[3]], e2 = Gm[ir[2] + jc[3]], e3 = Gm[ir[3] + jc[3]];
            T d4 = sc<T>::mul(p01, sc<T>::fms(sc<T>::mul(c2, e3), e2, c3));
            d4 = sc<T>::fms(d4, p02, sc<T>::fms(sc<T>::mul(c1, e3), e1, c3));
            d4 = sc<T>::fmac(d4, p03, sc<T>::fms(sc<T>::mul(c1, e2), e1, c2));
            d4 = sc<T>::fmac(d4, p12, sc<T>::fms(sc<T>::mul(c0, e3), e0, c3));
            d4 = sc<T>::fms(d4, p13, sc<T>::fms(sc<T>::mul(c0, e2), e0, c2));
            d4 = sc<T>::fmac(d4, p23, sc<T>::fms(sc<T>::mul(c0, e1), e0, c1));
            return d4;
          };
          det = sc<T>::zero();
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            int ir[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) ir[t] = iv[t + (t >= i)];
            const T mi = sc<T>::mul(Gm[iv[i] + jv[0]], det4g(ir, jv + 1));      // (the same expression as in the classed phase)
            det = (i & 1) ? sc<T>::sub(det, mi) : sc<T>::add(det, mi);
          }
        }
        T v = sc<T>::mul(pref_fac, det);
        if (par & 1) v = sc<T>::neg(v);
        orow[b] = v;
      }
      if (dbg) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        acc[2] += t - tq, tq = t;
      }
      const uint64_t sm = __ballot(slow);
      if (sm) {
        if (slow) queue[qn + __popcll(sm & below(lane))] = ((uint32_t)al << 16) | (uint32_t)b;
        qn += __popcll(sm);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        while (qn >= 8) {
          qn -= 8;
          slow_batch(queue[qn + grp], true);
        }
      }
      if (dbg) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        acc[3] += t - tq, tq = t;
      }
    }
  }
  if (qn > 0) slow_batch(queue[grp < qn ? grp : 0], grp < qn);
  if (dbg && lane == 0) {
    stamp(4);
    for (int i = 0; i < 4; ++i) atomicAdd(&dbg[19 + i], acc[i]);
    for (int i = 0; i < 8; ++i) atomicAdd(&dbg[23 + i], hist[i]);
    for (int i = 0; i < 4; ++i) atomicAdd(&dbg[4 * wave + i], t_s[i + 1] - t_s[i]);
    if (wave == 0) atomicAdd(&dbg[16], 1ull), atomicAdd(&dbg[17], (unsigned long long)na * nsk), atomicAdd(&dbg[18], (unsigned long long)n);
  }
}


// Two kernels, not one kernel with a branch: the 64-bit body needs 131 VGPRs (3 wavefronts per SIMD), the 32-bit one fits 4.
template <typename T, typename M>
__global__ __launch_bounds__(256) void ppt_det_kernel(const tmf_det_desc* __restrict__ desc, const float boost,
                                                      unsigned long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_det_desc d = desc[blockIdx.x];
  ppt_det_body<T, M>(d, boost, smem, dbg);
}

}  // namespace tmf

// Diagnostic counters (TMF_PPT_STAMPS=1): 19 words on the device, read and cleared by tmf_det_ppt_stamps.
static unsigned long long* ppt_stamps(bool force) {
  static unsigned long long* p = nullptr;
  static bool on = getenv("TMF_PPT_STAMPS") != nullptr;
  if ((on || force) && !p) {
    if (hipMalloc((void**)&p, 32 * 8) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, 32 * 8);
  }
  return on ? p : nullptr;
}

extern "C" int tmf_det_ppt_stamps(uint64_t* out32) {
  unsigned long long* p = ppt_stamps(false);
  if (!p) {
    tmf::set_error("tmf_det_ppt_stamps: set TMF_PPT_STAMPS=1 before the first launch");
    return TMF_E_ARG;
  }
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out32, p, 32 * 8, hipMemcpyDeviceToHost) != hipSuccess) return TMF_E_HIP;
  (void)hipMemset(p, 0, 32 * 8);
  return TMF_OK;
}

extern "C" int tmf_det_ppt_batched_w(int dtype, const tmf_det_desc* d_desc, int ntiles, int lds_bytes, int mask_bits, void* stream) {
  using namespace tmf;
  if (ntiles <= 0) return TMF_OK;
  if (lds_bytes < 0 || lds_bytes > 150 * 1024) {
    set_error("tmf_det_ppt_batched: lds_bytes %d exceeds the dynamic LDS budget (150 KiB)", lds_bytes);
    return TMF_E_LIMIT;
  }
  if (mask_bits != 32 && mask_bits != 64) {
    set_error("tmf_det_ppt_batched_w: mask_bits %d (32 or 64)", mask_bits);
    return TMF_E_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)ppt_det_kernel<cd, uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)ppt_det_kernel<double, uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)ppt_det_kernel<cd, uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)ppt_det_kernel<double, uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  static float boost = getenv("TMF_PPT_BOOST") ? (float)atof(getenv("TMF_PPT_BOOST")) : 10.0f;  // experiment knob
  const dim3 g(ntiles), b(256);
  unsigned long long* dbg = ppt_stamps(false);
  if (dtype == TMF_C128 && mask_bits == 32) hipLaunchKernelGGL((ppt_det_kernel<cd, uint32_t>), g, b, lds_bytes, s, d_desc, boost, dbg);
  else if (dtype == TMF_C128) hipLaunchKernelGGL((ppt_det_kernel<cd, uint64_t>), g, b, lds_bytes, s, d_desc, boost, dbg);
  else if (dtype == TMF_F64 && mask_bits == 32) hipLaunchKernelGGL((ppt_det_kernel<double, uint32_t>), g, b, lds_bytes, s, d_desc, boost, dbg);
  else if (dtype == TMF_F64) hipLaunchKernelGGL((ppt_det_kernel<double, uint64_t>), g, b, lds_bytes, s, d_desc, boost, dbg);
  else {
    set_error("tmf_det_ppt_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_det_ppt_batched launch");
}

extern "C" int tmf_det_ppt_batched(int dtype, const tmf_det_desc* d_desc, int ntiles, int lds_bytes, void* stream) {
  return tmf_det_ppt_batched_w(dtype, d_desc, ntiles, lds_bytes, 64, stream);
}
