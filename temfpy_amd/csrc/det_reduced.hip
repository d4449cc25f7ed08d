// Gathered determinants through a shared reduction per bra row-set ("reduced minors").
//
// Reference: slater.py:828-869 computes every minor det(M[rows(a)][:, cols(b)]) of a charge
// sector from scratch.  All minors of one bra row-set a are column subsets of the same
// n x sk matrix M_a, and the ket sets b of a sector differ from each other in a few columns
// only.  One Gauss-Jordan elimination of M_a with FULL pivoting (columns of the sector's
// leading ket set preferred) gives R = G M_a with unit columns e_{row(c)} on the n pivot
// columns c in P_a, and
//
//      det(M_a[:, b]) = prod(pivots) * sgn(sigma) * det(R[rows(P_a \ b), b \ P_a]),
//
// a determinant of order d = |b \ P_a| (measured: d <= 2 for 96 % of the minors of a
// Slater -> MPS sweep, n ~ 12).  It is the same Schur-complement identity the reference uses
// once for the always-occupied orbitals (slater.py:905-911), applied per bra row-set; full
// pivoting makes it as stable as the pivoted LU it replaces (no fallback path needed).
//
// One wavefront owns one bra row-set at a time: lane = column of M_a (sk <= 64) during the
// elimination, result kept in the wavefront's LDS.  The ket sets are then swept with ONE LANE
// PER PAIR: d, the row/column lists and the permutation sign come from 64-bit column masks and
// two small per-row-set tables in O(d), and the order-d determinant (d <= 3) is a closed form.
// Pairs with d > 3 (a few per cent) are queued in LDS and evaluated eight at a time by 8-lane
// groups with det_group (det_gather.hip).
#include "det_common.hpp"

namespace tmf {

__device__ inline unsigned wave_max_u32(unsigned k) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0xB1, 0xF, 0xF, false);
  k = o > k ? o : k;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x4E, 0xF, 0xF, false);
  k = o > k ? o : k;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x124, 0xF, 0xF, false);
  k = o > k ? o : k;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x128, 0xF, 0xF, false);
  k = o > k ? o : k;
  o = (unsigned)__shfl_xor((int)k, 16);
  k = o > k ? o : k;
  o = (unsigned)__shfl_xor((int)k, 32);
  k = o > k ? o : k;
  return k;
}

__device__ inline int nth_set_bit(uint64_t m, int k) {  // position of the k-th (0-based) set bit
  for (int i = 0; i < k; ++i) m &= m - 1;
  return __ffsll((unsigned long long)m) - 1;
}

// order-DD determinant of the d x d block (d <= DD), padded with the identity
template <typename T, int DD, int G>
__device__ inline T minor_det(const T* __restrict__ Na, const int NS, const uint64_t jmask, const unsigned imask,
                              const int d, const int c, T* __restrict__ scratch) {
  T a[DD];
  const int col = (c < d) ? nth_set_bit(jmask, c) : 0;
  unsigned im = imask;
#pragma unroll
  for (int r = 0; r < DD; ++r) {
    T v = (r == c) ? sc<T>::one() : sc<T>::zero();
    if (r < d) {
      const int row = __ffs(im) - 1;
      im &= im - 1;
      if (c < d) v = Na[row + col * NS];
      else v = sc<T>::zero();
    } else if (c < d) {
      v = sc<T>::zero();
    }
    a[r] = v;
  }
  return det_group<T, DD, G>(a, c, scratch);
}

// LDS (dynamic): [ M : sb*sk T ][ ket idx : nsk*N u8 ][ ket masks : nsk u64 ][ bra idx : na*N u8 ]
//                [ per wave: R (N|1)*sk T | scratch 8*(33) T | rowsbelow 64 u32 | queue 72 u16 | rowof, invc 64 u8 ]
template <typename T, int N>
__global__ __launch_bounds__(256) void reduced_det_kernel(const tmf_det_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_det_desc d = desc[blockIdx.x];
  constexpr int NS = N | 1;
  constexpr int SCR = 8 * 33;  // 8 groups x (32 + 1) elements: enough for every det_group order
  const int na = d.a1 - d.a0, sk = d.sk, nsk = d.nsk;
  T* Ms = reinterpret_cast<T*>(smem);
  size_t off = ((size_t)d.sb * sk * sizeof(T) + 15) & ~(size_t)15;
  uint8_t* kidx = smem + off;
  off += ((size_t)nsk * N + 15) & ~(size_t)15;
  uint64_t* kmask = reinterpret_cast<uint64_t*>(smem + off);
  off += ((size_t)nsk * 8 + 15) & ~(size_t)15;
  uint8_t* bidx = smem + off;
  off += ((size_t)na * N + 15) & ~(size_t)15;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per_wave = ((size_t)NS * sk + SCR) * sizeof(T) + 576;
  unsigned char* wb = smem + off + (size_t)wave * per_wave;
  T* Na = reinterpret_cast<T*>(wb);
  T* scr_w = Na + (size_t)NS * sk;
  unsigned* rowsbelow = reinterpret_cast<unsigned*>(scr_w + SCR);        // 64 x u32
  uint16_t* queue = reinterpret_cast<uint16_t*>(rowsbelow + 64);         // 72 x u16 (<= 7 + 64 waiting pairs)
  uint8_t* rowof = reinterpret_cast<uint8_t*>(queue + 72);               // 64 x u8
  uint8_t* invc = rowof + 64;                                            // 64 x u8

  const T* __restrict__ S = reinterpret_cast<const T*>(d.S);
  for (int e = threadIdx.x; e < d.sb * sk; e += 256) Ms[e] = S[(size_t)(e % d.sb) + (size_t)(e / d.sb) * d.lds];
  const uint8_t* __restrict__ gk = reinterpret_cast<const uint8_t*>(d.ket_idx);
  const uint8_t* __restrict__ gb = reinterpret_cast<const uint8_t*>(d.bra_idx) + (size_t)d.a0 * N;
  for (int e = threadIdx.x; e < nsk * N; e += 256) kidx[e] = gk[e];
  for (int e = threadIdx.x; e < na * N; e += 256) bidx[e] = gb[e];
  __syncthreads();
  for (int b = threadIdx.x; b < nsk; b += 256) {
    uint64_t m = 0;
    for (int u = 0; u < N; ++u) m |= 1ull << kidx[b * N + u];
    kmask[b] = m;
  }
  __syncthreads();
  const uint64_t pref = kmask[0];  // columns of the sector's leading ket configuration
  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const int grp = lane >> 3, c8 = lane & 7;
  T* scr_g = scr_w + grp * 33;

  for (int al = wave; al < na; al += 4) {
    // ---------------- Gauss-Jordan with full pivoting, lane = column ---------------------------
    const uint8_t* rows = bidx + al * N;
    T a[N];
#pragma unroll
    for (int r = 0; r < N; ++r) a[r] = (lane < sk) ? Ms[rows[r] + lane * d.sb] : sc<T>::zero();
    unsigned usedrows = 0u;
    bool colused = false;
    int myrow = 0;
    T detg = sc<T>::one();
    bool singular = false;
    const float boost = ((pref >> lane) & 1ull) ? 100.0f : 1.0f;  // |.|^2 x 100: prefer within a factor 10
#pragma unroll
    for (int t = 0; t < N; ++t) {
      int best = 0;
      float bv = -1.0f;
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const float v = (float)sc<T>::abs2(a[r]);
        if (!((usedrows >> r) & 1u) && v > bv) {
          bv = v;
          best = r;
        }
      }
      unsigned key = 0u;
      if (!colused && lane < sk && bv > 0.0f)
        key = ((__float_as_uint(bv * boost) >> 1) & ~63u) | (unsigned)lane | 0x80000000u;
      key = wave_max_u32(key);
      if (key == 0u) {  // no non-zero pivot left: rank(M_a) < N, every minor is zero
        singular = true;
        break;
      }
      const int pcol = (int)(key & 63u);
      const int prow = __shfl(best, pcol);
      if (lane == pcol) {
#pragma unroll
        for (int r = 0; r < N; ++r) scr_w[r] = a[r];
        colused = true;
        myrow = prow;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const T piv = scr_w[prow];
      detg = sc<T>::mul(detg, piv);
      const T pinv = sc<T>::inv_fast(piv);
      T arow = sc<T>::zero();
#pragma unroll
      for (int r = 0; r < N; ++r) arow = sel(r == prow, a[r], arow);
      const T ars = sc<T>::mul(arow, pinv);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const T tmp = sc<T>::fms(a[r], scr_w[r], ars);
        a[r] = sel(r == prow, ars, tmp);
      }
      usedrows |= 1u << prow;
      __builtin_amdgcn_wave_barrier();
    }
    T* __restrict__ orow = out + (size_t)(d.a0 + al) * nsk;
    if (singular) {
      for (int b = lane; b < nsk; b += 64) orow[b] = sc<T>::zero();
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    if (lane < sk) {
#pragma unroll
      for (int r = 0; r < N; ++r) Na[r + lane * NS] = a[r];
    }
    rowof[lane] = colused ? (uint8_t)myrow : (uint8_t)0xFF;
    const uint64_t pa = __ballot(colused);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const T pref_fac = sc<T>::mul(scale, detg);

    // ---------------- per row-set tables for the permutation sign -------------------------------
    // sigma maps position u of the (ascending) ket set b to a row: a kept pivot column goes to its
    // pivot row, the new columns (ascending) to the rows of the removed pivot columns (ascending).
    // Its inversion parity splits into  inv(P_a) + sum_{c removed} invc[c] + inv(removed pairs)
    // + sum_{new column j -> row i} cross(j, i)   (mod 2), every term O(1) from these tables:
    //   rowsbelow[j] = rows of the pivot columns c < j,   invc[c] = inversions of P_a that involve c
    const unsigned mybit = colused ? (1u << myrow) : 0u;
    unsigned incl = mybit;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned y = (unsigned)__shfl_up((int)incl, o);
      if (lane >= o) incl |= y;
    }
    const unsigned below = incl & ~mybit;  // pivot rows are distinct: removing the own bit = exclusive scan
    constexpr unsigned allrows = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);
    int myinv = 0;
    if (colused) myinv = __popcll((uint64_t)below >> (myrow + 1)) + __popc(allrows & ~incl & ((1u << myrow) - 1u));
    int invsum = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) invsum += __popcll(__ballot((myinv >> k) & 1)) << k;
    const int par_pp = (invsum >> 1) & 1;
    rowsbelow[lane] = below;
    invc[lane] = (uint8_t)myinv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---------------- slow path: 8 lanes per pair, any number of exchanged columns ---------------
    auto slow_batch = [&](const int b, const bool live) {
      const uint64_t bm = live ? kmask[b] : pa;
      const uint64_t jmask = bm & ~pa;          // columns of b outside the pivot set
      uint64_t rem = pa & ~bm;                  // pivot columns missing in b
      const int dd = __popcll(jmask);
      unsigned imask = 0u;                      // rows of the missing pivot columns
      while (rem) {
        const int cc = __ffsll((unsigned long long)rem) - 1;
        rem &= rem - 1;
        imask |= 1u << rowof[cc];
      }
      uint64_t seen = 0ull;  // 64-bit: the shift below reaches 32 for r = 31
      unsigned irest = imask;
      int inv = 0;
      const uint8_t* kb = kidx + (live ? b : 0) * N;
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const int cu = kb[u];
        int r;
        if ((pa >> cu) & 1ull) {
          r = rowof[cu];
        } else {
          r = __ffs(irest) - 1;
          irest &= irest - 1;
        }
        inv += __popcll(seen >> (r + 1));
        seen |= 1ull << r;
      }
      const unsigned dmax = wave_max_u32(live ? (unsigned)dd : 0u);
      T det = sc<T>::one();
      if (dmax <= 4u) {
        det = minor_det<T, 4, 8>(Na, NS, jmask, imask, live ? dd : 0, c8, scr_g);
      } else if (dmax <= 8u) {
        det = minor_det<T, 8, 8>(Na, NS, jmask, imask, live ? dd : 0, c8, scr_g);
      } else {
        // rare: more than 8 exchanged columns.  The 8 pairs are handled one after the other by
        // groups of 16 / 32 lanes (lanes 0..15 / 0..31 of the wavefront carry the pair).
        for (int g = 0; g < 8; ++g) {
          const uint64_t jm = __shfl((unsigned long long)jmask, g * 8);
          const unsigned im = (unsigned)__shfl((int)imask, g * 8);
          const int dg = __shfl(live ? dd : 0, g * 8);
          T dt;
          if constexpr (N > 16) {
            dt = minor_det<T, 32, 32>(Na, NS, jm, im, dg, lane & 31, scr_w + (lane >> 5) * 33);
          } else if constexpr (N > 8) {
            dt = minor_det<T, 16, 16>(Na, NS, jm, im, dg, lane & 15, scr_w + (lane >> 4) * 33);
          } else {
            dt = minor_det<T, 8, 8>(Na, NS, jm, im, dg, lane & 7, scr_w + (lane >> 3) * 33);
          }
          dt = shfl_t<T>(dt, 0, 64);
          if (grp == g) det = dt;
        }
      }
      if (live && c8 == 0) {
        T v = sc<T>::mul(pref_fac, det);
        if (inv & 1) v = sc<T>::neg(v);
        orow[b] = v;
      }
    };

    // ---------------- sweep the ket sets: one lane per pair while <= 3 columns are exchanged ------
    int qn = 0;  // pairs waiting for the slow path (uniform)
    for (int b0 = 0; b0 < nsk; b0 += 64) {
      const int b = b0 + lane;
      const bool live = b < nsk;
      const uint64_t bm = live ? kmask[b] : pa;
      const uint64_t jmask = bm & ~pa;
      uint64_t rem = pa & ~bm;
      const int dd = __popcll(jmask);
      const bool slow = dd > 3;
      if (live && !slow) {
        unsigned imask = 0u;
        int par = par_pp;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if (rem) {
            const int cc = __ffsll((unsigned long long)rem) - 1;
            rem &= rem - 1;
            const int r = rowof[cc];
            par += invc[cc] + __popcll((uint64_t)imask >> (r + 1));
            imask |= 1u << r;
          }
        }
        const unsigned krows = allrows & ~imask;
        int iv[3], jv[3];
        unsigned im = imask;
        uint64_t jm = jmask;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          iv[t] = 0;
          jv[t] = 0;
          if (t < dd) {
            const int i = __ffs(im) - 1, j = __ffsll((unsigned long long)jm) - 1;
            im &= im - 1;
            jm &= jm - 1;
            const unsigned rb = rowsbelow[j];
            par += __popc(rb & krows & ~((2u << i) - 1u)) + __popc(~rb & krows & ((1u << i) - 1u));
            iv[t] = i;
            jv[t] = j * NS;
          }
        }
        T det = sc<T>::one();
        if (dd == 1) {
          det = Na[iv[0] + jv[0]];
        } else if (dd == 2) {
          const T m00 = Na[iv[0] + jv[0]], m10 = Na[iv[1] + jv[0]], m01 = Na[iv[0] + jv[1]], m11 = Na[iv[1] + jv[1]];
          det = sc<T>::fms(sc<T>::mul(m00, m11), m01, m10);
        } else if (dd == 3) {
          const T m00 = Na[iv[0] + jv[0]], m10 = Na[iv[1] + jv[0]], m20 = Na[iv[2] + jv[0]];
          const T m01 = Na[iv[0] + jv[1]], m11 = Na[iv[1] + jv[1]], m21 = Na[iv[2] + jv[1]];
          const T m02 = Na[iv[0] + jv[2]], m12 = Na[iv[1] + jv[2]], m22 = Na[iv[2] + jv[2]];
          const T c0 = sc<T>::fms(sc<T>::mul(m11, m22), m12, m21);
          const T c1 = sc<T>::fms(sc<T>::mul(m10, m22), m12, m20);
          const T c2 = sc<T>::fms(sc<T>::mul(m10, m21), m11, m20);
          det = sc<T>::fmac(sc<T>::fms(sc<T>::mul(m00, c0), m01, c1), m02, c2);
        }
        T v = sc<T>::mul(pref_fac, det);
        if (par & 1) v = sc<T>::neg(v);
        orow[b] = v;
      }
      const uint64_t sm = __ballot(live && slow);
      if (sm) {
        if (live && slow) queue[qn + __popcll(sm & ((1ull << lane) - 1ull))] = (uint16_t)b;
        qn += __popcll(sm);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        while (qn >= 8) {
          qn -= 8;
          slow_batch((int)queue[qn + grp], true);
        }
      }
    }
    if (qn > 0) slow_batch((int)queue[grp < qn ? grp : 0], grp < qn);
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T, int N>
static void launch_reduced(dim3 g, int lds, hipStream_t s, const tmf_det_desc* d) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)reduced_det_kernel<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((reduced_det_kernel<T, N>), g, dim3(256), lds, s, d);
}

template <typename T>
static int launch_red(int n, const tmf_det_desc* d, int nt, int lds, hipStream_t s) {
  dim3 g(nt);
  switch (n) {
#define TMF_CASE(N) case N: launch_reduced<T, N>(g, lds, s, d); break;
    TMF_CASE(1) TMF_CASE(2) TMF_CASE(3) TMF_CASE(4) TMF_CASE(5) TMF_CASE(6) TMF_CASE(7) TMF_CASE(8)
    TMF_CASE(9) TMF_CASE(10) TMF_CASE(11) TMF_CASE(12) TMF_CASE(13) TMF_CASE(14) TMF_CASE(15) TMF_CASE(16)
    TMF_CASE(17) TMF_CASE(18) TMF_CASE(19) TMF_CASE(20) TMF_CASE(21) TMF_CASE(22) TMF_CASE(23) TMF_CASE(24)
    TMF_CASE(25) TMF_CASE(26) TMF_CASE(27) TMF_CASE(28) TMF_CASE(29) TMF_CASE(30) TMF_CASE(31) TMF_CASE(32)
#undef TMF_CASE
    default:
      set_error("tmf_det_reduced_batched: order must be 1..32, got %d", n);
      return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_det_reduced_batched launch");
}

}  // namespace tmf

extern "C" int tmf_det_reduced_batched(int dtype, int order, const tmf_det_desc* d_desc, int ntiles, int lds_bytes,
                                       void* stream) {
  if (ntiles <= 0) return TMF_OK;
  if (lds_bytes < 0 || lds_bytes > 160 * 1024) {
    tmf::set_error("tmf_det_reduced_batched: lds_bytes %d exceeds the 160 KiB LDS of a CU", lds_bytes);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128) return tmf::launch_red<tmf::cd>(order, d_desc, ntiles, lds_bytes, s);
  if (dtype == TMF_F64) return tmf::launch_red<double>(order, d_desc, ntiles, lds_bytes, s);
  tmf::set_error("tmf_det_reduced_batched: bad dtype %d", dtype);
  return TMF_E_ARG;
}
