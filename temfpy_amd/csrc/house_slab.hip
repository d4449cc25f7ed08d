// Householder QR of tall slabs (n x c, c <= 64) with the working panel resident in LDS; one workgroup per slab,
// thousands of slabs per launch.
//
// Replaces the blocked Gram-Schmidt (bcgs.hip: 3 projection passes + LDS panel kernel, ~60 launches per call)
// for the two 64-column range-finder QRs of every entanglement cut (the orthonormal bases inside the replacement
// of numpy.linalg.eigh, slater.py:347): these slabs are numerically rank deficient (singular values down to
// 1e-17), which Gram-Schmidt only survives with three passes and a per-column rank decision, while Householder
// reflectors are orthogonal whatever the rank.
//
// Unblocked global-memory Householder would re-read the trailing matrix once per column (140 GB per conversion
// over all cuts: HBM bound).  Here a panel of 16 columns is held ON CHIP, one column per wavefront, in registers
// (element i of a lane = row lane + 64 i; up to 1024 complex / 2048 real rows), and only reflectors go through LDS:
//   phase 1, per panel: load it, apply the reflectors of all earlier panels (their vectors stream in from global
//            memory, one element per thread, prefetched one reflector ahead), factor the panel, store it;
//   phase 2, per panel of Q: start from unit columns, apply the reflectors backwards, store to the scratch Q;
//            finally Q is copied over A.
// A reflector application is a lane-strided dot product with the LDS vector + a DPP wave reduction + update in registers.
// (First version with the panel in LDS: 5 LDS accesses per element and reflector, LDS-bandwidth bound at 5.4 ms
// per launch - no faster than the Gram-Schmidt it replaces.)  HBM traffic: 7 x the slab (PMC: 1.9 GB per launch for
// 0.27 GB of slabs with Q, the earlier reflectors are re-read by every later panel), 0.8 TB/s: not the bound.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace tmf {

template <typename T>
__device__ inline T wsum(T v) {      // DPP path: the butterfly through ds_bpermute cost 24 dependent LDS-crossbar hops per complex sum
  return wave_sum64(v);
}

constexpr int VB = 8;   // reflectors per LDS block

template <typename T, int RMAX>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, (RMAX * sizeof(T) <= 128) ? 8 : 4)))
void house_slab_kernel(const tmf_slab_desc* __restrict__ desc, int w, unsigned long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  // diagnostics (TMF_SLAB_STAMPS=1): cycles per phase of wavefront 1, summed over the launch
  unsigned long long tq = dbg ? __builtin_amdgcn_s_memtime() : 0ull, tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto lap = [&](int i) {
    if (dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[i] += now - tq, tq = now;
    }
  };
  const tmf_slab_desc d = desc[blockIdx.x];
  const int n = d.n, c = d.c;
  if (n <= 0 || c <= 0) return;
  const int NT = blockDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = n < c ? n : c;
  T* VBLK = reinterpret_cast<T*>(smem);        // VB reflectors, VBLK[r + j * n]; slot 0 doubles as the current one
  T* taus = VBLK + (size_t)VB * n;             // K scalars
  T* __restrict__ A = reinterpret_cast<T*>(d.A);
  T* __restrict__ Q = reinterpret_cast<T*>(d.Q);
  T* __restrict__ R = reinterpret_cast<T*>(d.R);
  const size_t lda = d.lda, ldq = d.ldq;

  // Each wavefront owns one panel column and keeps it in registers: element i of `col` is row lane + 64 i.
  // (With the panel in LDS the kernel was bound by LDS bandwidth: 5 accesses per element and reflector.)
  T col[RMAX];
  // col <- (I - f v v^H) col, v with support on rows >= k
  auto apply = [&](int k, T f, const T* __restrict__ v) {
    T dot = sc<T>::zero();
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r >= k && r < n) dot = sc<T>::fmacc(dot, v[r], col[i]);
    }
    dot = sc<T>::mul(f, wsum<T>(dot));
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r >= k && r < n) col[i] = sc<T>::fms(col[i], dot, v[r]);
    }
  };
  // reflectors [kb, kb + nb) from global memory into the LDS block (unit diagonal, zeros above)
  auto load_block = [&](int kb, int nb) {
    for (int e = tid; e < n * nb; e += NT) {
      const int r = e % n, j = e / n, k = kb + j;
      VBLK[e] = (r == k) ? sc<T>::one() : (r > k ? A[r + (size_t)k * lda] : sc<T>::zero());
    }
  };

  // ---------------- phase 1: factor panel by panel (w columns, one per wavefront) ----------------
  for (int p0 = 0; p0 < c; p0 += w) {
    const int wp = (c - p0 < w) ? c - p0 : w;
    const bool mine = wave < wp;
    T* __restrict__ a = A + (size_t)(p0 + (mine ? wave : 0)) * lda;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      col[i] = (mine && r < n) ? a[r] : sc<T>::zero();
    }
    lap(0);
    const int kprev = p0 < K ? p0 : K;
    for (int kb = 0; kb < kprev; kb += VB) {      // reflectors of the earlier panels, VB at a time
      const int nb = (kprev - kb < VB) ? kprev - kb : VB;
      __syncthreads();
      load_block(kb, nb);
      __syncthreads();
      lap(1);
      if (mine)
        for (int j = 0; j < nb; ++j) apply(kb + j, sc<T>::conj(taus[kb + j]), VBLK + (size_t)j * n);
      lap(2);
    }
    __syncthreads();
    // The wavefront of column jj builds reflector k = p0 + jj from its registers into an LDS slot; the others
    // apply it.  Look-ahead: the wavefront of column jj + 1 builds the next reflector right after its own update,
    // into the other slot, while the rest is still applying the current one - one barrier per reflector.
    auto build = [&](int k, T* __restrict__ slot) {
      double s_ = 0.0;
      T al = sc<T>::zero();
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r > k && r < n) s_ += sc<T>::abs2(col[i]);
        if (r == k) al = col[i];
      }
      s_ = wave_sum64(s_);
      const T alpha = wsum<T>(al);
      T tau = sc<T>::zero(), scal = sc<T>::zero(), beta = alpha;
      // (a column whose squared length underflows - the rounding noise of rounding noise, reached after a dozen steps past
      // the rank of an exactly rank-deficient block - is a zero column: no reflector.  Without the guard b_ = 0 and
      // tau = 0 / 0 poisoned R: tests/soak/soak_gutzwiller.py, pinned in tests/test_gpu_kernels.py.)
      if ((s_ > 0.0 || sc<T>::imag(alpha) != 0.0) && sc<T>::abs2(alpha) + s_ > 1e-290) {
        double b_ = sqrt(sc<T>::abs2(alpha) + s_);
        if (sc<T>::real(alpha) > 0.0) b_ = -b_;
        beta = sc<T>::from_real(b_);
        tau = sc<T>::scale(sc<T>::sub(beta, alpha), 1.0 / b_);
        scal = sc<T>::inv(sc<T>::sub(alpha, beta));
      }
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r >= k && r < n) {
          const T v = (r == k) ? sc<T>::one() : sc<T>::mul(col[i], scal);
          slot[r] = v;
          col[i] = (r == k) ? beta : v;
        }
      }
      if (lane == 0) taus[k] = tau;
    };
    if (wave == 0 && p0 < K) build(p0, VBLK);
    __syncthreads();
    for (int jj = 0; jj < wp; ++jj) {
      const int k = p0 + jj;
      if (k >= K) break;
      T* cur = VBLK + (size_t)(jj & 1) * n;
      if (mine && wave > jj) {
        apply(k, sc<T>::conj(taus[k]), cur);
        if (wave == jj + 1 && k + 1 < K) build(k + 1, VBLK + (size_t)((jj + 1) & 1) * n);
      }
      __syncthreads();
    }
    lap(3);
    if (mine) {
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r < n) a[r] = col[i];
      }
    }
    __syncthreads();
    lap(4);
  }
  // ---------------- R (c x c, zero rows beyond K), optionally as R^H ----------------
  if (R)
    for (int e = tid; e < c * c; e += NT) {
      const int r = e % c, cc = e / c;
      const T v = (r <= cc && r < K) ? A[r + (size_t)cc * lda] : sc<T>::zero();
      if (d.flags & 1) R[cc + (size_t)r * d.ldr] = sc<T>::conj(v);
      else R[r + (size_t)cc * d.ldr] = v;
    }
  if (d.flags & 12) {           // 4: only R wanted, no Q at all; 8: Q later (tmf_house_form_q_batched): A is left holding
    if (d.flags & 8)            // the reflectors, their scalars go to the caller's buffer
      for (int e = tid; e < c; e += NT) Q[e] = e < K ? taus[e] : sc<T>::zero();
    if (dbg && tid == 64) {
      for (int i = 0; i < 8; ++i) atomicAdd(&dbg[i], tacc[i]);
      atomicAdd(&dbg[8], 1ull), atomicAdd(&dbg[9], (unsigned long long)n);
    }
    return;
  }
  lap(5);
  // ---------------- phase 2: thin Q, panel by panel, into the scratch ----------------
  for (int p0 = 0; p0 < c; p0 += w) {
    const int wp = (c - p0 < w) ? c - p0 : w;
    const bool mine = wave < wp;
    const int jc = p0 + wave;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      col[i] = (mine && r == jc && r < K) ? sc<T>::one() : sc<T>::zero();
    }
    int ktop = p0 + wp;             // reflectors [0, ktop) act on these columns, applied from the top down
    if (ktop > K) ktop = K;
    for (int kb_end = ktop; kb_end > 0; kb_end -= VB) {
      const int kb = (kb_end - VB > 0) ? kb_end - VB : 0;
      const int nb = kb_end - kb;
      __syncthreads();
      load_block(kb, nb);
      __syncthreads();
      lap(1);
      if (mine)
        for (int j = nb - 1; j >= 0; --j)
          if (kb + j <= jc) apply(kb + j, taus[kb + j], VBLK + (size_t)j * n);      // H_k e_j = e_j for k > j
      lap(6);
    }
    if (mine) {
      T* __restrict__ q = Q + (size_t)jc * ldq;
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r < n) q[r] = col[i];
      }
    }
  }
  __syncthreads();
  // ---------------- Q over A ----------------
  if (!(d.flags & 2))
    for (int e = tid; e < n * c; e += NT) {
      const int r = e % n, j = e / n;
      A[r + (size_t)j * lda] = Q[r + (size_t)j * ldq];
    }
}

// Thin Q of slabs factored with flags & 8, in place: A holds the reflectors below the diagonal, `Q` their K scalars.  The
// arithmetic per column is that of phase 2 above; the panels are taken from the LAST to the first, since the columns of panel
// p only need the reflectors of the panels <= p (H_k e_j = e_j for k > j) and may then overwrite their own.
// For callers whose factorisations form a dependent chain (the canonicalisation sweeps of gutzwiller.py: 2 x 512 of them with
// <= 5 workgroups each) while nothing in the chain needs Q: one launch over all slabs of the chain afterwards fills the device.
template <typename T, int RMAX>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, (RMAX * sizeof(T) <= 128) ? 8 : 4)))
void house_formq_kernel(const tmf_slab_desc* __restrict__ desc, int w) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_slab_desc d = desc[blockIdx.x];
  const int n = d.n, c = d.c;
  if (n <= 0 || c <= 0) return;
  const int NT = blockDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = n < c ? n : c;
  T* VBLK = reinterpret_cast<T*>(smem);
  T* taus = VBLK + (size_t)VB * n;
  T* __restrict__ A = reinterpret_cast<T*>(d.A);
  const T* __restrict__ tau_g = reinterpret_cast<const T*>(d.Q);
  const size_t lda = d.lda;
  for (int e = tid; e < K; e += NT) taus[e] = tau_g[e];
  T col[RMAX];
  auto apply = [&](int k, T f, const T* __restrict__ v) {
    T dot = sc<T>::zero();
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r >= k && r < n) dot = sc<T>::fmacc(dot, v[r], col[i]);
    }
    dot = sc<T>::mul(f, wsum<T>(dot));
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r >= k && r < n) col[i] = sc<T>::fms(col[i], dot, v[r]);
    }
  };
  const int npan = (c + w - 1) / w;
  for (int pi = npan - 1; pi >= 0; --pi) {
    const int p0 = pi * w;
    const int wp = (c - p0 < w) ? c - p0 : w;
    const bool mine = wave < wp;
    const int jc = p0 + wave;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      col[i] = (mine && r == jc && r < K) ? sc<T>::one() : sc<T>::zero();
    }
    int ktop = p0 + wp;
    if (ktop > K) ktop = K;
    for (int kb_end = ktop; kb_end > 0; kb_end -= VB) {
      const int kb = (kb_end - VB > 0) ? kb_end - VB : 0;
      const int nb = kb_end - kb;
      __syncthreads();
      for (int e = tid; e < n * nb; e += NT) {
        const int r = e % n, j = e / n, k = kb + j;
        VBLK[e] = (r == k) ? sc<T>::one() : (r > k ? A[r + (size_t)k * lda] : sc<T>::zero());
      }
      __syncthreads();
      if (mine)
        for (int j = nb - 1; j >= 0; --j)
          if (kb + j <= jc) apply(kb + j, taus[kb + j], VBLK + (size_t)j * n);
    }
    __syncthreads();            // the reflectors held by this panel's columns have been read by everybody
    if (mine) {
      T* __restrict__ q = A + (size_t)jc * lda;
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r < n) q[r] = col[i];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same factorisation for blocks whose columns ALL fit the registers of one workgroup (n <= 64 RMAX rows, c <= NW CPW
// columns): wavefront w holds columns w, w + NW, w + 2 NW, ... for the whole factorisation, so there are no panels, no
// reflector blocks to stream back in, and one barrier per column: the wavefront that owns column k + 1 applies reflector
// k to it first and builds reflector k + 1 into the other LDS slot while the rest is still applying reflector k.  The dot
// products of a wavefront's columns with the reflector are formed together and their wave reductions interleaved.
// Made for the charge blocks of the Gutzwiller canonicalisation sweeps (~220 x 110, two dependent QRs per site:
// gutzwiller.py:266 / :471, `npc.qr` inside TeNPy's canonical_form_finite), where the panel kernel above took 0.6 - 0.9 ms
// per launch for at most five blocks: 5 - 7 us per column, 92 % of the projection's GPU time.
// (A layout with a column in 16 lanes and four columns reduced by one DPP sequence was tried: per step it is bound by the
// 4x redundant LDS reads of the reflector and by the owner's chain, 271 us against 235 us at 220 x 110.)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int RMAX, int CPW, int NW>
__global__ __launch_bounds__(64 * NW) void house_reg_kernel(const tmf_slab_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_slab_desc d = desc[blockIdx.x];
  const int n = d.n, c = d.c;
  if (n <= 0 || c <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = n < c ? n : c;
  constexpr int NS = 64 * RMAX;
  T* slots = reinterpret_cast<T*>(smem);       // two reflectors of 64 RMAX entries (zeros above the diagonal, 1 on it)
  T* taus = slots + 2 * NS;                    // K scalars
  T* __restrict__ A = reinterpret_cast<T*>(d.A);
  T* __restrict__ Q = reinterpret_cast<T*>(d.Q);
  T* __restrict__ R = reinterpret_cast<T*>(d.R);
  const size_t lda = d.lda, ldq = d.ldq;

  // col[q][i]: row lane + 64 i of column wave + NW q; rows >= n and columns >= c hold zeros throughout
  T col[CPW][RMAX];
#pragma unroll
  for (int q = 0; q < CPW; ++q) {
    const int j = wave + NW * q;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      col[q][i] = (j < c && r < n) ? A[r + (size_t)j * lda] : sc<T>::zero();
    }
    __builtin_amdgcn_sched_barrier(0);      // (one column's addresses at a time: hoisted together they spill the columns)
  }
  // Reflector k from register column QQ of this wavefront (the column's entries above row k are final R entries):
  // v into `slot` (all 64 RMAX entries), tau into LDS.  One wave reduction (the squared length below the diagonal); the
  // diagonal entry comes by v_readlane, 1 / b and 1 / (alpha - beta) by v_rcp_f64 + Newton steps.
  auto build = [&](const int k, auto qq_tag, T* __restrict__ slot) {
    constexpr int QQ = decltype(qq_tag)::value;
    double s_ = 0.0;
    T al = sc<T>::zero();
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r > k) s_ += sc<T>::abs2(col[QQ][i]);
      if ((k >> 6) == i) al = sc<T>::from2(readlane_d(sc<T>::real(col[QQ][i]), k & 63), sc<T>::cplx ? readlane_d(sc<T>::imag(col[QQ][i]), k & 63) : 0.0);
    }
    s_ = wave_sum64(s_);
    const T alpha = al;
    T tau = sc<T>::zero(), scal = sc<T>::zero(), beta = alpha;
    if ((s_ > 0.0 || sc<T>::imag(alpha) != 0.0) && sc<T>::abs2(alpha) + s_ > 1e-290) {    // (see the panel kernel)
      const double nn = sc<T>::abs2(alpha) + s_;
      double rs = __builtin_amdgcn_rsq(nn);                 // 1 / sqrt(nn), two Newton steps
      rs = rs * fma(-0.5 * nn * rs, rs, 1.5);
      rs = rs * fma(-0.5 * nn * rs, rs, 1.5);
      double b_ = nn * rs, binv = rs;
      if (sc<T>::real(alpha) > 0.0) b_ = -b_, binv = -binv;
      beta = sc<T>::from_real(b_);
      tau = sc<T>::scale(sc<T>::sub(beta, alpha), binv);
      scal = sc<T>::inv_fast(sc<T>::sub(alpha, beta));
    }
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      T v = sc<T>::zero();
      if (r == k) v = sc<T>::one();
      else if (r > k) v = sc<T>::mul(col[QQ][i], scal);
      slot[r] = v;
      if (r == k) col[QQ][i] = beta;
      else if (r > k) col[QQ][i] = v;
    }
    if (lane == 0) taus[k] = tau;
  };
  // (I - f v v^H) on register column QQ
  auto apply_one = [&](const T f, const T* __restrict__ v, auto qq_tag) {
    constexpr int QQ = decltype(qq_tag)::value;
    T dot = sc<T>::zero();
#pragma unroll
    for (int i = 0; i < RMAX; ++i) dot = sc<T>::fmacc(dot, v[lane + 64 * i], col[QQ][i]);
    dot = sc<T>::mul(f, wave_sum64(dot));
#pragma unroll
    for (int i = 0; i < RMAX; ++i) col[QQ][i] = sc<T>::fms(col[QQ][i], dot, v[lane + 64 * i]);
  };
  // ... on every register column of this wavefront whose index lies in [lo, c), except column `skip`; four columns at a
  // time: their dot products are formed together and the wave reductions interleaved (the DPP latencies overlap)
  auto apply_range = [&](const T f, const T* __restrict__ v, const int lo, const int skip) {
    T vr[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) vr[i] = v[lane + 64 * i];
#pragma unroll
    for (int q0 = 0; q0 < CPW; q0 += 4) {
      if (wave + NW * (q0 + 3 < CPW ? q0 + 3 : CPW - 1) < lo) continue;      // (uniform: nothing of this group is left)
      T dot[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dot[u] = sc<T>::zero();
        if (q0 + u < CPW) {
#pragma unroll
          for (int i = 0; i < RMAX; ++i) dot[u] = sc<T>::fmacc(dot[u], vr[i], col[q0 + u][i]);
        }
      }
      if constexpr (!sc<T>::cplx && CPW >= 3) {      // all four sums by one folded reduction
        if (q0 + 2 < CPW) {
          wave_sum64x4(dot[0], dot[1], dot[2], dot[3]);
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q0 + u < CPW) {
              const int j = wave + NW * (q0 + u);
              dot[u] = (j >= lo && j != skip) ? f * dot[u] : 0.0;
            }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q0 + u < CPW) {
              const int j = wave + NW * (q0 + u);
              dot[u] = (j >= lo && j != skip) ? sc<T>::mul(f, wave_sum64(dot[u])) : sc<T>::zero();
            }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q0 + u < CPW) {
            const int j = wave + NW * (q0 + u);
            dot[u] = (j >= lo && j != skip) ? sc<T>::mul(f, wave_sum64(dot[u])) : sc<T>::zero();     // (condition uniform in the wavefront)
          }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + u < CPW) {
#pragma unroll
          for (int i = 0; i < RMAX; ++i) col[q0 + u][i] = sc<T>::fms(col[q0 + u][i], dot[u], vr[i]);
        }
    }
  };
  // compile-time dispatch on the register column
  auto with_q = [&](const int q, auto&& fn) {
#pragma unroll
    for (int qq = 0; qq < CPW; ++qq)
      if (qq == q) {
        switch (qq) {   // (integral_constant per case keeps the register index static)
#define TMF_Q(N_) case N_: if constexpr (N_ < CPW) fn(std::integral_constant<int, N_>{}); break;
          TMF_Q(0) TMF_Q(1) TMF_Q(2) TMF_Q(3) TMF_Q(4) TMF_Q(5) TMF_Q(6) TMF_Q(7) TMF_Q(8) TMF_Q(9) TMF_Q(10) TMF_Q(11)
          TMF_Q(12) TMF_Q(13) TMF_Q(14) TMF_Q(15) TMF_Q(16) TMF_Q(17) TMF_Q(18) TMF_Q(19)
#undef TMF_Q
        }
      }
  };

  // ---------------- phase 1 ----------------
  if (wave == 0) build(0, std::integral_constant<int, 0>{}, slots);
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    const T* cur = slots + (size_t)(k & 1) * NS;
    const T f = sc<T>::conj(taus[k]);
    const int k1 = k + 1, q1 = k1 / NW;
    const bool next_owner = (k1 % NW) == wave && k1 < c;
    if (next_owner) {      // column k + 1 first, then its reflector into the other slot while the others still apply this one
      with_q(q1, [&](auto tag) {
        apply_one(f, cur, tag);
        if (k1 < K) build(k1, tag, slots + (size_t)(k1 & 1) * NS);
      });
    }
    apply_range(f, cur, k1, next_owner ? k1 : -1);
    __syncthreads();
  }
  // ---------------- R (c x c, zero rows beyond K), optionally as R^H: straight from the registers ----------------
  if (R) {
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
      const int j = wave + NW * q;
      if (j >= c) continue;
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r >= c) continue;
        const T v = (r <= j && r < K) ? col[q][i] : sc<T>::zero();
        if (d.flags & 1) R[j + (size_t)r * d.ldr] = sc<T>::conj(v);
        else R[r + (size_t)j * d.ldr] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (d.flags & 4) return;      // only R wanted
  // the reflectors go to A (below the diagonal), where phase 2 reads them one per step
#pragma unroll
  for (int q = 0; q < CPW; ++q) {
    const int j = wave + NW * q;
    if (j >= c) continue;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r < n) A[r + (size_t)j * lda] = col[q][i];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (d.flags & 8) {            // Q later (tmf_house_form_q_batched): the scalars of the reflectors to the caller's buffer
    for (int e = tid; e < c; e += 64 * NW) Q[e] = e < K ? taus[e] : sc<T>::zero();
    return;
  }
  __syncthreads();
  // ---------------- phase 2: thin Q = H_0 ... H_{K-1} [1; 0], reflectors applied from the last to the first ----------------
#pragma unroll
  for (int q = 0; q < CPW; ++q) {
    const int j = wave + NW * q;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) col[q][i] = (j < c && j < K && lane + 64 * i == j) ? sc<T>::one() : sc<T>::zero();
  }
  auto load_v = [&](const int k, T* __restrict__ slot) {
    for (int r = tid; r < NS; r += 64 * NW) slot[r] = (r == k) ? sc<T>::one() : ((r > k && r < n) ? A[r + (size_t)k * lda] : sc<T>::zero());
  };
  if (K > 0) load_v(K - 1, slots + (size_t)((K - 1) & 1) * NS);
  __syncthreads();
  for (int k = K - 1; k >= 0; --k) {
    if (k > 0) load_v(k - 1, slots + (size_t)((k - 1) & 1) * NS);
    apply_range(taus[k], slots + (size_t)(k & 1) * NS, k, -1);          // H_k e_j = e_j for j < k
    __syncthreads();
  }
  T* __restrict__ dst = (d.flags & 2) ? Q : A;
  const size_t ldd = (d.flags & 2) ? ldq : lda;
#pragma unroll
  for (int q = 0; q < CPW; ++q) {
    const int j = wave + NW * q;
    if (j >= c) continue;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r < n) dst[r + (size_t)j * ldd] = col[q][i];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The register form for blocks with more columns than one workgroup holds: panels of NW CPW columns, each factored as above;
// the reflectors of the earlier panels are read back from A one per step (their entry for step k + 2 is on its way from
// memory while reflector k is applied) and applied to the CPW columns of a wavefront together.  320 x 160 blocks of the
// Gutzwiller sweeps (two panels of 80 columns): against the panel kernel above (16 columns per panel, one per wavefront,
// 90 reflector blocks loaded behind two barriers each).  Phase 2 takes the panels from the last to the first, in place.
// (Kept apart from the one-panel kernel: with the panel loop around it that one needs 27 more registers and spills.)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int RMAX, int CPW, int NW, bool STAMPS>
__global__ __launch_bounds__(64 * NW) void house_regp_kernel(const tmf_slab_desc* __restrict__ desc, unsigned long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_slab_desc d = desc[blockIdx.x];
  // diagnostics (TMF_SLAB_STAMPS=1): clock of wavefront 1 per part, summed over the launch's blocks of more than 256 rows
  if (!STAMPS || d.n <= 256) dbg = nullptr;
  unsigned long long tq = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull, tacc[STAMPS ? 8 : 1] = {0};
  auto lap = [&](int i) {
    if constexpr (STAMPS) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[i] += now - tq, tq = now;
    }
  };
  const int n = d.n, c = d.c;
  if (n <= 0 || c <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = n < c ? n : c;
  constexpr int NS = 64 * RMAX;
  constexpr int PW = NW * CPW;                 // columns per panel; blocks of up to PW columns have one panel
  static_assert(NS <= 64 * NW, "one thread per reflector entry");
  T* slots = reinterpret_cast<T*>(smem);       // two reflectors of 64 RMAX entries (zeros above the diagonal, 1 on it)
  T* taus = slots + 2 * NS;                    // K scalars
  T* __restrict__ A = reinterpret_cast<T*>(d.A);
  T* __restrict__ Q = reinterpret_cast<T*>(d.Q);
  T* __restrict__ R = reinterpret_cast<T*>(d.R);
  const size_t lda = d.lda, ldq = d.ldq;
  const int npan = (c + PW - 1) / PW;
  int p0 = 0;                                  // first column of the panel in the registers

  // col[q][i]: row lane + 64 i of column p0 + wave + NW q; rows >= n and columns >= c hold zeros throughout
  T col[CPW][RMAX];
  // Reflector k from register column QQ of this wavefront (the column's entries above row k are final R entries):
  // v into `slot` (all 64 RMAX entries), tau into LDS.  One wave reduction (the squared length below the diagonal); the
  // diagonal entry comes by v_readlane, 1 / b and 1 / (alpha - beta) by v_rcp_f64 + Newton steps.
  auto build = [&](const int k, auto qq_tag, T* __restrict__ slot) {
    constexpr int QQ = decltype(qq_tag)::value;
    double s_ = 0.0;
    T al = sc<T>::zero();
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      if (r > k) s_ += sc<T>::abs2(col[QQ][i]);
      if ((k >> 6) == i) al = sc<T>::from2(readlane_d(sc<T>::real(col[QQ][i]), k & 63), sc<T>::cplx ? readlane_d(sc<T>::imag(col[QQ][i]), k & 63) : 0.0);
    }
    s_ = wave_sum64(s_);
    const T alpha = al;
    T tau = sc<T>::zero(), scal = sc<T>::zero(), beta = alpha;
    if ((s_ > 0.0 || sc<T>::imag(alpha) != 0.0) && sc<T>::abs2(alpha) + s_ > 1e-290) {    // (see the panel kernel)
      const double nn = sc<T>::abs2(alpha) + s_;
      double rs = __builtin_amdgcn_rsq(nn);                 // 1 / sqrt(nn), two Newton steps
      rs = rs * fma(-0.5 * nn * rs, rs, 1.5);
      rs = rs * fma(-0.5 * nn * rs, rs, 1.5);
      double b_ = nn * rs, binv = rs;
      if (sc<T>::real(alpha) > 0.0) b_ = -b_, binv = -binv;
      beta = sc<T>::from_real(b_);
      tau = sc<T>::scale(sc<T>::sub(beta, alpha), binv);
      scal = sc<T>::inv_fast(sc<T>::sub(alpha, beta));
    }
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int r = lane + 64 * i;
      T v = sc<T>::zero();
      if (r == k) v = sc<T>::one();
      else if (r > k) v = sc<T>::mul(col[QQ][i], scal);
      slot[r] = v;
      if (r == k) col[QQ][i] = beta;
      else if (r > k) col[QQ][i] = v;
    }
    if (lane == 0) taus[k] = tau;
  };
  // (I - f v v^H) on register column QQ
  auto apply_one = [&](const T f, const T* __restrict__ v, auto qq_tag) {
    constexpr int QQ = decltype(qq_tag)::value;
    T dot = sc<T>::zero();
#pragma unroll
    for (int i = 0; i < RMAX; ++i) dot = sc<T>::fmacc(dot, v[lane + 64 * i], col[QQ][i]);
    dot = sc<T>::mul(f, wave_sum64(dot));
#pragma unroll
    for (int i = 0; i < RMAX; ++i) col[QQ][i] = sc<T>::fms(col[QQ][i], dot, v[lane + 64 * i]);
  };
  // ... on every register column of this wavefront whose index lies in [lo, c), except column `skip`; four columns at a
  // time: their dot products are formed together and the wave reductions interleaved (the DPP latencies overlap)
  auto apply_range = [&](const T f, const T* __restrict__ v, const int lo, const int skip) {
    T vr[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) vr[i] = v[lane + 64 * i];
#pragma unroll
    for (int q0 = 0; q0 < CPW; q0 += 4) {
      if (p0 + wave + NW * (q0 + 3 < CPW ? q0 + 3 : CPW - 1) < lo) continue;      // (uniform: nothing of this group is left)
      T dot[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dot[u] = sc<T>::zero();
        if (q0 + u < CPW) {
#pragma unroll
          for (int i = 0; i < RMAX; ++i) dot[u] = sc<T>::fmacc(dot[u], vr[i], col[q0 + u][i]);
        }
      }
      if constexpr (!sc<T>::cplx && CPW >= 3) {      // all four sums by one folded reduction
        if (q0 + 2 < CPW) {
          wave_sum64x4(dot[0], dot[1], dot[2], dot[3]);
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q0 + u < CPW) {
              const int j = p0 + wave + NW * (q0 + u);
              dot[u] = (j >= lo && j != skip) ? f * dot[u] : 0.0;
            }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q0 + u < CPW) {
              const int j = p0 + wave + NW * (q0 + u);
              dot[u] = (j >= lo && j != skip) ? sc<T>::mul(f, wave_sum64(dot[u])) : sc<T>::zero();
            }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q0 + u < CPW) {
            const int j = p0 + wave + NW * (q0 + u);
            dot[u] = (j >= lo && j != skip) ? sc<T>::mul(f, wave_sum64(dot[u])) : sc<T>::zero();     // (condition uniform in the wavefront)
          }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + u < CPW) {
#pragma unroll
          for (int i = 0; i < RMAX; ++i) col[q0 + u][i] = sc<T>::fms(col[q0 + u][i], dot[u], vr[i]);
        }
    }
  };
  // compile-time dispatch on the register column
  auto with_q = [&](const int q, auto&& fn) {
#pragma unroll
    for (int qq = 0; qq < CPW; ++qq)
      if (qq == q) {
        switch (qq) {   // (integral_constant per case keeps the register index static)
#define TMF_Q(N_) case N_: if constexpr (N_ < CPW) fn(std::integral_constant<int, N_>{}); break;
          TMF_Q(0) TMF_Q(1) TMF_Q(2) TMF_Q(3) TMF_Q(4) TMF_Q(5) TMF_Q(6) TMF_Q(7) TMF_Q(8) TMF_Q(9) TMF_Q(10) TMF_Q(11)
          TMF_Q(12) TMF_Q(13) TMF_Q(14) TMF_Q(15) TMF_Q(16) TMF_Q(17) TMF_Q(18) TMF_Q(19)
#undef TMF_Q
        }
      }
  };
  // this thread's entry of reflector k as A holds it (the vector has 64 RMAX entries, one per thread of the first wavefronts)
  auto fetch = [&](const int k) -> T {
    const int r = tid;
    if (r >= NS) return sc<T>::zero();
    return (r == k) ? sc<T>::one() : ((r > k && r < n) ? A[r + (size_t)k * lda] : sc<T>::zero());
  };
  auto store_cols = [&](T* __restrict__ dst, const size_t ldd) {
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
      const int j = p0 + wave + NW * q;
      if (j >= c) continue;
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r < n) dst[r + (size_t)j * ldd] = col[q][i];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---------------- phase 1, panel by panel (one panel when c <= PW) ----------------
  for (int pi = 0; pi < npan; ++pi) {
    p0 = pi * PW;
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
      const int j = p0 + wave + NW * q;
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        col[q][i] = (j < c && r < n) ? A[r + (size_t)j * lda] : sc<T>::zero();
      }
      __builtin_amdgcn_sched_barrier(0);      // (one column's addresses at a time: hoisted together they spill the columns)
    }
    lap(0);
    // the reflectors of the earlier panels, read back from A one per step; the entry of reflector k + 2 is on its way from
    // memory while reflector k is applied
    const int kprev = p0 < K ? p0 : K;
    if (kprev > 0) {
      __syncthreads();                         // (everybody is done with the slots; the earlier panels are in memory)
      if (tid < NS) slots[tid] = fetch(0);
      T pre = kprev > 1 ? fetch(1) : sc<T>::zero();
      __syncthreads();
      for (int k = 0; k < kprev; ++k) {
        apply_range(sc<T>::conj(taus[k]), slots + (size_t)(k & 1) * NS, p0, -1);
        if (k + 1 < kprev && tid < NS) slots[(size_t)((k + 1) & 1) * NS + tid] = pre;
        if (k + 2 < kprev) pre = fetch(k + 2);
        __syncthreads();
      }
    }
    lap(1);
    const int kend = (p0 + PW < K) ? p0 + PW : K;
    if (wave == 0 && p0 < K) build(p0, std::integral_constant<int, 0>{}, slots + (size_t)(p0 & 1) * NS);
    __syncthreads();
    // One barrier per column; everybody waits for the chain of the wavefront that owns column k + 1 (reflector k on it, then
    // reflector k + 1 from it: two dependent wave reductions, 1 900 of the 3 700 ticks of a step; its other columns 800).
    // Tried without gain: those other columns one step later (three slots; the compiler then spills 44 B per lane), a higher
    // issue priority for the chain (s_setprio).
    for (int k = p0; k < kend; ++k) {
      const T* cur = slots + (size_t)(k & 1) * NS;
      const T f = sc<T>::conj(taus[k]);
      const int k1 = k + 1, kk1 = k1 - p0, q1 = kk1 / NW;
      const bool next_owner = (kk1 % NW) == wave && kk1 < PW && k1 < c;
      lap(2);
      if (next_owner) {      // column k + 1 first, then its reflector into the other slot while the others still apply this one
        with_q(q1, [&](auto tag) {
          apply_one(f, cur, tag);
          if (k1 < K) build(k1, tag, slots + (size_t)(k1 & 1) * NS);
        });
        lap(3);
      }
      apply_range(f, cur, k1, next_owner ? k1 : -1);
      lap(next_owner ? 4 : 5);
      __syncthreads();
      lap(6);
    }
    // R (c x c, zero rows beyond K), optionally as R^H: straight from the registers
    if (R) {
#pragma unroll
      for (int q = 0; q < CPW; ++q) {
        const int j = p0 + wave + NW * q;
        if (j >= c) continue;
#pragma unroll
        for (int i = 0; i < RMAX; ++i) {
          const int r = lane + 64 * i;
          if (r >= c) continue;
          const T v = (r <= j && r < K) ? col[q][i] : sc<T>::zero();
          if (d.flags & 1) R[j + (size_t)r * d.ldr] = sc<T>::conj(v);
          else R[r + (size_t)j * d.ldr] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the reflectors go to A (below the diagonal): the later panels and phase 2 read them there
    if (!(d.flags & 4) || npan > 1) store_cols(A, lda);
    lap(7);
  }
  if constexpr (STAMPS)
    if (dbg && tid == 64) {
      for (int i = 0; i < 8; ++i) atomicAdd(&dbg[i], tacc[i]);
      atomicAdd(&dbg[8], 1ull), atomicAdd(&dbg[9], (unsigned long long)n);
    }
  if (d.flags & 4) return;      // only R wanted
  if (d.flags & 8) {            // Q later (tmf_house_form_q_batched): the scalars of the reflectors to the caller's buffer
    __syncthreads();
    for (int e = tid; e < c; e += 64 * NW) Q[e] = e < K ? taus[e] : sc<T>::zero();
    return;
  }
  // ---------------- phase 2: thin Q = H_0 ... H_{K-1} [1; 0], reflectors applied from the last to the first; the panels
  // from the last to the first as well: panel p needs the reflectors of the panels <= p only and may then overwrite its own
  for (int pi = npan - 1; pi >= 0; --pi) {
    p0 = pi * PW;
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
      const int j = p0 + wave + NW * q;
#pragma unroll
      for (int i = 0; i < RMAX; ++i) col[q][i] = (j < c && j < K && lane + 64 * i == j) ? sc<T>::one() : sc<T>::zero();
    }
    const int ktop = (p0 + PW < K) ? p0 + PW : K;
    __syncthreads();            // (the stores of phase 1 / of the previous panel are done, the slots are free)
    if (ktop > 0) {
      if (tid < NS) slots[(size_t)((ktop - 1) & 1) * NS + tid] = fetch(ktop - 1);
      T pre = ktop > 1 ? fetch(ktop - 2) : sc<T>::zero();
      __syncthreads();
      for (int k = ktop - 1; k >= 0; --k) {
        apply_range(taus[k], slots + (size_t)(k & 1) * NS, k, -1);          // H_k e_j = e_j for j < k
        if (k > 0 && tid < NS) slots[(size_t)((k - 1) & 1) * NS + tid] = pre;
        if (k > 1) pre = fetch(k - 2);
        __syncthreads();
      }
    }
    if (d.flags & 2) store_cols(Q, ldq);
    else store_cols(A, lda);
  }
}

}  // namespace tmf

static unsigned long long* slab_stamps() {
  static unsigned long long* p = nullptr;
  static bool on = getenv("TMF_SLAB_STAMPS") != nullptr;
  if (on && !p && hipMalloc((void**)&p, 16 * 8) == hipSuccess) (void)hipMemset(p, 0, 16 * 8);
  return on ? p : nullptr;
}
extern "C" int tmf_house_slab_stamps(uint64_t* out16) {
  unsigned long long* p = slab_stamps();
  if (!p) return TMF_E_ARG;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out16, p, 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) return TMF_E_HIP;
  (void)hipMemset(p, 0, 16 * 8);
  return TMF_OK;
}

static int house_slab_launch(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream, bool use_reg);

// The panel kernel only: its arithmetic per slab does not depend on what else is in the launch (a rank's shard must
// reproduce the unsharded conversion bit for bit; found by tests/soak/soak_shards.py seed 60313 when the
// register form below was chosen by the launch's largest slab).
extern "C" int tmf_house_slab_batched(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream) {
  return house_slab_launch(dtype, d_desc, nprob, max_n, max_c, stream, false);
}
// The same factorisation with every column in registers when the LAUNCH allows it (real, <= 256 x 128), else the panel
// kernel: for callers without that requirement (the canonicalisation sweeps of gutzwiller.py).  TMF_SLAB_REG=0: never.
extern "C" int tmf_house_qr_regs_batched(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream) {
  static const bool allow = !(getenv("TMF_SLAB_REG") && atoi(getenv("TMF_SLAB_REG")) == 0);
  return house_slab_launch(dtype, d_desc, nprob, max_n, max_c, stream, allow);
}

extern "C" int tmf_house_form_q_batched(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const int w = 16;
  if (max_n <= 0 || max_c <= 0 || max_n > ((dtype == TMF_C128) ? 1024 : 2048)) {
    set_error("tmf_house_form_q_batched: %d rows not in 1..%d", max_n, (dtype == TMF_C128) ? 1024 : 2048);
    return TMF_E_LIMIT;
  }
  const size_t lds = ((size_t)max_n * VB + (size_t)max_c + 4) * elem + 64;
  if (lds > 150 * 1024) {
    set_error("tmf_house_form_q_batched: %d rows need %zu B of LDS", max_n, lds);
    return TMF_E_LIMIT;
  }
  static bool attr_done = false;
  if (!attr_done) {
#define TMF_FQ_ATTR(T, RM) (void)hipFuncSetAttribute((const void*)house_formq_kernel<T, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)
    TMF_FQ_ATTR(cd, 4); TMF_FQ_ATTR(cd, 8); TMF_FQ_ATTR(cd, 16); TMF_FQ_ATTR(double, 4); TMF_FQ_ATTR(double, 8);
    TMF_FQ_ATTR(double, 16); TMF_FQ_ATTR(double, 32);
#undef TMF_FQ_ATTR
    attr_done = true;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 g(nprob), b(64 * w);
#define TMF_FQ_LAUNCH(T, RM) hipLaunchKernelGGL((house_formq_kernel<T, RM>), g, b, lds, s, d_desc, w)
  if (dtype == TMF_C128) {
    if (max_n <= 256) TMF_FQ_LAUNCH(cd, 4);
    else if (max_n <= 512) TMF_FQ_LAUNCH(cd, 8);
    else TMF_FQ_LAUNCH(cd, 16);
  } else if (dtype == TMF_F64) {
    if (max_n <= 256) TMF_FQ_LAUNCH(double, 4);
    else if (max_n <= 512) TMF_FQ_LAUNCH(double, 8);
    else if (max_n <= 1024) TMF_FQ_LAUNCH(double, 16);
    else TMF_FQ_LAUNCH(double, 32);
  } else {
    set_error("tmf_house_form_q_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
#undef TMF_FQ_LAUNCH
  return check_hip(hipGetLastError(), "tmf_house_form_q_batched");
}

static int house_slab_launch(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream, bool use_reg) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const int w = 16;                                       // columns per panel = wavefronts per workgroup
  if (max_n <= 0 || max_c <= 0 || max_n > ((dtype == TMF_C128) ? 1024 : 2048)) {
    set_error("tmf_house_slab_batched: %d rows not in 1..%d", max_n, (dtype == TMF_C128) ? 1024 : 2048);
    return TMF_E_LIMIT;
  }
  const size_t lds = ((size_t)max_n * VB + (size_t)max_c + 4) * elem + 64;   // VB >= 2 slots for the look-ahead
  if (lds > 150 * 1024) {
    set_error("tmf_house_slab_batched: %d rows need %zu B of LDS", max_n, lds);
    return TMF_E_LIMIT;
  }
  static bool attr_done = false;
  if (!attr_done) {
#define TMF_SLAB_ATTR(T, RM) (void)hipFuncSetAttribute((const void*)house_slab_kernel<T, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)
    TMF_SLAB_ATTR(cd, 4); TMF_SLAB_ATTR(cd, 8); TMF_SLAB_ATTR(cd, 16); TMF_SLAB_ATTR(double, 4); TMF_SLAB_ATTR(double, 8);
    TMF_SLAB_ATTR(double, 16); TMF_SLAB_ATTR(double, 32);
#undef TMF_SLAB_ATTR
    attr_done = true;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // blocks whose columns all fit the registers of one workgroup: no panels, one barrier per column (house_reg_kernel)
  if (use_reg) {
#define TMF_REG_TRY(T, RM, CW, NW_)                                                                                         \
    if (max_n <= 64 * RM && max_c <= NW_ * CW) {                                                                              \
      const size_t lds_r = ((size_t)2 * 64 * RM + (size_t)max_c + 4) * elem + 64;                                             \
      hipLaunchKernelGGL((house_reg_kernel<T, RM, CW, NW_>), dim3(nprob), dim3(64 * NW_), lds_r, s, d_desc);                  \
      return check_hip(hipGetLastError(), "tmf_house_slab_batched (register form)");                                         \
    }
    // (16 wavefronts leave 128 registers per lane: RMAX x CPW <= 32 doubles next to ~60 registers of bookkeeping; forms with
    // 8 wavefronts and twice the columns per wavefront spilled hundreds of registers, complex columns were put into scratch
    // memory by the compiler: larger and complex blocks stay on the panel kernel)
    if (dtype == TMF_F64) {
      TMF_REG_TRY(double, 2, 8, 16) TMF_REG_TRY(double, 4, 7, 16) TMF_REG_TRY(double, 4, 8, 16)
      // larger blocks: panels of 80 / 64 columns in registers, the reflectors of the earlier panels read back one per step
      static const bool panels = !(getenv("TMF_SLAB_REG_PANELS") && atoi(getenv("TMF_SLAB_REG_PANELS")) == 0);
      if (panels && max_n <= 320) {
        const size_t lds_r = ((size_t)2 * 320 + (size_t)max_c + 4) * elem + 64;
        if (slab_stamps()) hipLaunchKernelGGL((house_regp_kernel<double, 5, 5, 16, true>), dim3(nprob), dim3(1024), lds_r, s, d_desc, slab_stamps());
        else hipLaunchKernelGGL((house_regp_kernel<double, 5, 5, 16, false>), dim3(nprob), dim3(1024), lds_r, s, d_desc, nullptr);
        return check_hip(hipGetLastError(), "tmf_house_slab_batched (register form, panels)");
      }
      if (panels && max_n <= 512) {
        const size_t lds_r = ((size_t)2 * 512 + (size_t)max_c + 4) * elem + 64;
        hipLaunchKernelGGL((house_regp_kernel<double, 8, 2, 16, false>), dim3(nprob), dim3(1024), lds_r, s, d_desc, nullptr);
        return check_hip(hipGetLastError(), "tmf_house_slab_batched (register form, panels)");
      }
    }
#undef TMF_REG_TRY
  }
  const dim3 g(nprob), b(64 * w);
  unsigned long long* dbg = slab_stamps();
#define TMF_SLAB_LAUNCH(T, RM) hipLaunchKernelGGL((house_slab_kernel<T, RM>), g, b, lds, s, d_desc, w, dbg)
  if (dtype == TMF_C128) {
    if (max_n <= 256) TMF_SLAB_LAUNCH(cd, 4);
    else if (max_n <= 512) TMF_SLAB_LAUNCH(cd, 8);
    else TMF_SLAB_LAUNCH(cd, 16);
  } else if (dtype == TMF_F64) {
    if (max_n <= 256) TMF_SLAB_LAUNCH(double, 4);
    else if (max_n <= 512) TMF_SLAB_LAUNCH(double, 8);
    else if (max_n <= 1024) TMF_SLAB_LAUNCH(double, 16);
    else TMF_SLAB_LAUNCH(double, 32);
  } else {
    set_error("tmf_house_slab_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
#undef TMF_SLAB_LAUNCH
  return check_hip(hipGetLastError(), "tmf_house_slab_batched");
}
