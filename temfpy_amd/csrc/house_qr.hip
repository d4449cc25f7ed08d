// Batched Householder QR, one 1024-thread workgroup per matrix:  A (m x n) = Q R.
//
// The canonicalisation sweeps of an MPS (TeNPy `MPS.canonical_form_finite`, called by the reference at
// gutzwiller.py:266 / :471: `npc.qr` on the way right, `npc.svd` on the way left) factor one (p chi_l) x chi_r
// or (p chi_r) x chi_l matrix per charge block and site.  After a Gutzwiller projection those matrices are
// EXACTLY rank deficient (half of the Schmidt rank is projected away) with graded columns, which is where
// Gram-Schmidt variants lose orthogonality or need a rank decision (measured with the blocked Gram-Schmidt
// of bcgs.hip: normalised rounding-noise columns, isometry of the result off by 0.3).  Householder
// reflectors are orthogonal to machine precision whatever the rank, so Q needs no rank decision at all;
// the rank shows up as tiny rows of R and is decided later by the singular values.
//
// Layout: the matrix stays in global memory (a 280 x 140 block is 313 KB: L2, not LDS), the current reflector
// lives in LDS; in step k the 16 wavefronts each take trailing columns (lane-strided dot product with the
// reflector, 6 shuffle steps, rank-1 update; the column stays in registers between the two when it has at most
// 512 rows).  Two barriers per step: the wavefront that updates column k+1 also computes its norm and the next
// reflector's scalars (look-ahead), so no separate norm pass is needed.  Phase 2 forms the thin Q in place
// (LAPACK org2r order) unless flags & 2 says that only R is wanted.
// For m < n (more columns than rows: rank <= m) Q is m x m padded with zero columns and R is padded with
// zero rows, so that shapes stay fixed for the caller.
#include "common.hpp"

namespace tmf {

template <typename T>
__device__ inline T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v = sc<T>::add(v, shfl_xor_t<T>(v, o, 64));
  return v;
}

// tau, 1 / (alpha - beta), beta of the reflector that maps (alpha, x) with |x|^2 = xn2 onto (beta, 0)  [LAPACK larfg]
template <typename T>
__device__ inline void reflector(T alpha, double xn2, T& tau, T& scal, T& beta) {
  tau = sc<T>::zero();
  scal = sc<T>::zero();
  beta = alpha;
  // (a column whose squared length underflows is a zero column: no reflector; b = 0 would give tau = 0 / 0, see house_slab.hip)
  if ((xn2 > 0.0 || sc<T>::imag(alpha) != 0.0) && sc<T>::abs2(alpha) + xn2 > 1e-290) {
    double b = sqrt(sc<T>::abs2(alpha) + xn2);
    if (sc<T>::real(alpha) > 0.0) b = -b;  // beta = -sign(Re alpha) |(alpha, x)|
    beta = sc<T>::from_real(b);
    tau = sc<T>::scale(sc<T>::sub(beta, alpha), 1.0 / b);
    scal = sc<T>::inv(sc<T>::sub(alpha, beta));
  }
}

constexpr int RC = 8;  // rows per lane kept in registers (64 * RC = 512 rows)

// a_c -= f * (v^H a_c) * v over rows [k, m); with LOOK returns through `nxt` the squared norm of the updated
// rows > k + 1 (the next reflector's |x|^2 when c == k + 1) and through `head` the updated element of row k + 1.
template <typename T, bool LOOK>
__device__ inline void apply_reflector(T* __restrict__ a, const T* __restrict__ vs, int k, int m, T f, int lane,
                                       double& nxt, T& head) {
  const int nrow = m - k;
  T dot = sc<T>::zero();
  double s = 0.0;
  T h = sc<T>::zero();
  if (nrow <= 64 * RC) {
    T reg[RC];
#pragma unroll
    for (int i = 0; i < RC; ++i) {
      const int r = k + lane + 64 * i;
      reg[i] = (r < m) ? a[r] : sc<T>::zero();
      if (r < m) dot = sc<T>::fmacc(dot, vs[r], reg[i]);
    }
    dot = sc<T>::mul(f, wave_sum<T>(dot));
#pragma unroll
    for (int i = 0; i < RC; ++i) {
      const int r = k + lane + 64 * i;
      if (r < m) {
        const T v = sc<T>::fms(reg[i], dot, vs[r]);
        a[r] = v;
        if (LOOK) {
          if (r > k + 1) s += sc<T>::abs2(v);
          if (r == k + 1) h = v;
        }
      }
    }
  } else {
    for (int r = k + lane; r < m; r += 64) dot = sc<T>::fmacc(dot, vs[r], a[r]);
    dot = sc<T>::mul(f, wave_sum<T>(dot));
    for (int r = k + lane; r < m; r += 64) {
      const T v = sc<T>::fms(a[r], dot, vs[r]);
      a[r] = v;
      if (LOOK) {
        if (r > k + 1) s += sc<T>::abs2(v);
        if (r == k + 1) h = v;
      }
    }
  }
  if (LOOK) {
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    nxt = s;
    head = shfl_t<T>(h, 1, 64);   // row k + 1 belongs to lane 1
  }
}

// Two columns at once (rows in registers): twice the loads in flight per wavefront, which is what the step time
// is made of (each wavefront walks ~(n - k) / 16 columns whose loads would otherwise be exposed one by one).
template <typename T>
__device__ inline void apply_reflector2(T* __restrict__ a0, T* __restrict__ a1, const T* __restrict__ vs, int k, int m,
                                        T f, int lane) {
  T r0[RC], r1[RC];
  T d0 = sc<T>::zero(), d1 = sc<T>::zero();
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    const int r = k + lane + 64 * i;
    const bool in = r < m;
    r0[i] = in ? a0[r] : sc<T>::zero();
    r1[i] = in ? a1[r] : sc<T>::zero();
  }
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    const int r = k + lane + 64 * i;
    if (r < m) {
      const T v = vs[r];
      d0 = sc<T>::fmacc(d0, v, r0[i]);
      d1 = sc<T>::fmacc(d1, v, r1[i]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    d0 = sc<T>::add(d0, shfl_xor_t<T>(d0, o, 64));
    d1 = sc<T>::add(d1, shfl_xor_t<T>(d1, o, 64));
  }
  d0 = sc<T>::mul(f, d0);
  d1 = sc<T>::mul(f, d1);
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    const int r = k + lane + 64 * i;
    if (r < m) {
      const T v = vs[r];
      a0[r] = sc<T>::fms(r0[i], d0, v);
      a1[r] = sc<T>::fms(r1[i], d1, v);
    }
  }
}

// all trailing columns [c_first, c_end) of this wavefront (stride NW), pairwise when the rows fit the registers
template <typename T>
__device__ inline void apply_trailing(T* __restrict__ A, size_t lda, const T* __restrict__ vs, int k, int m, T f, int lane,
                                      int c_first, int c_end, int stride) {
  int c = c_first;
  if (m - k <= 64 * RC)
    for (; c + stride < c_end; c += 2 * stride)
      apply_reflector2<T>(A + (size_t)c * lda, A + (size_t)(c + stride) * lda, vs, k, m, f, lane);
  for (; c < c_end; c += stride) {
    double nxt;
    T head;
    apply_reflector<T, false>(A + (size_t)c * lda, vs, k, m, f, lane, nxt, head);
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void house_qr_kernel(const tmf_qr_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_qr_desc d = desc[blockIdx.x];
  const int m = d.m, n = d.n;
  if (m <= 0 || n <= 0) return;
  constexpr int NT = 1024, NW = NT / 64;
  T* vs = reinterpret_cast<T*>(smem);                 // reflector, m elements
  T* taus = vs + m;                                   // K reflector scalars
  const int K = m < n ? m : n;
  double* red = reinterpret_cast<double*>(taus + K);  // NW partial sums
  T* par = reinterpret_cast<T*>(red + NW);            // [0] = 1 / (alpha - beta), [1] = beta of the CURRENT step
  T* __restrict__ A = reinterpret_cast<T*>(d.A);
  T* __restrict__ R = reinterpret_cast<T*>(d.R);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t lda = d.lda;

  // ---- prologue: scalars of the first reflector -------------------------------------------------------
  {
    double s = 0.0;
    for (int r = 1 + tid; r < m; r += NT) s += sc<T>::abs2(A[r]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
      double xn2 = 0.0;
      for (int i = 0; i < NW; ++i) xn2 += red[i];
      T tau, scal, beta;
      reflector<T>(A[0], xn2, tau, scal, beta);
      taus[0] = tau;
      par[0] = scal;
      par[1] = beta;
    }
    __syncthreads();
  }
  // ---- phase 1: A = H_0 .. H_{K-1} R, reflectors stored below the diagonal -------------------------------
  for (int k = 0; k < K; ++k) {
    const T tau = taus[k], scal = par[0], beta = par[1];
    for (int r = k + tid; r < m; r += NT) {
      const T v = (r == k) ? sc<T>::one() : sc<T>::mul(A[r + k * lda], scal);
      vs[r] = v;
      A[r + k * lda] = (r == k) ? beta : v;
    }
    __syncthreads();
    // trailing columns: a_c -= conj(tau) (v^H a_c) v ; wavefront 0 looks ahead for column k + 1
    const T ctau = sc<T>::conj(tau);
    int c0 = k + 1 + wave;
    if (wave == 0 && c0 < n) {
      double nxt;
      T head;
      apply_reflector<T, true>(A + (size_t)c0 * lda, vs, k, m, ctau, lane, nxt, head);
      if (lane == 0 && k + 1 < K) {
        T t2, s2, b2;
        reflector<T>(head, nxt, t2, s2, b2);
        taus[k + 1] = t2;
        par[0] = s2;
        par[1] = b2;
      }
      c0 += NW;
    }
    apply_trailing<T>(A, lda, vs, k, m, ctau, lane, c0, n, NW);
    __syncthreads();
  }
  // ---- R (n x n, zero rows beyond K), optionally as R^H ----------------------------------------------
  if (R)
    for (int e = tid; e < n * n; e += NT) {
      const int r = e % n, c = e / n;
      const T v = (r <= c && r < K) ? A[r + (size_t)c * lda] : sc<T>::zero();
      if (d.flags & 1) R[c + (size_t)r * d.ldr] = sc<T>::conj(v);
      else R[r + (size_t)c * d.ldr] = v;
    }
  if (d.flags & 2) return;   // only R wanted: A is left holding the reflectors
  __syncthreads();
  // ---- phase 2: thin Q in place (columns >= K become zero) -------------------------------------------
  for (int c = K + wave; c < n; c += NW)
    for (int r = lane; r < m; r += 64) A[r + (size_t)c * lda] = sc<T>::zero();
  for (int k = K - 1; k >= 0; --k) {
    const T tau = taus[k];
    for (int r = k + tid; r < m; r += NT) vs[r] = (r == k) ? sc<T>::one() : A[r + k * lda];
    __syncthreads();
    // Q[k:, k+1:K] = H_k Q[k:, k+1:K] = Q - tau v (v^H Q)
    apply_trailing<T>(A, lda, vs, k, m, tau, lane, k + 1 + wave, K, NW);
    // column k itself: H_k e_k = e_k - tau v
    for (int r = tid; r < m; r += NT) {
      T v = sc<T>::zero();
      if (r == k) v = sc<T>::sub(sc<T>::one(), tau);
      else if (r > k) v = sc<T>::neg(sc<T>::mul(tau, vs[r]));
      A[r + k * lda] = v;
    }
    __syncthreads();
  }
}

}  // namespace tmf

extern "C" int tmf_house_qr_batched(int dtype, const tmf_qr_desc* d_desc, int nprob, int max_m, int max_n, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const size_t lds = ((size_t)max_m + (size_t)(max_m < max_n ? max_m : max_n) + 2) * elem + 16 * 8 + 64;
  if (max_m <= 0 || max_n <= 0 || lds > 150 * 1024) {
    set_error("tmf_house_qr_batched: %d x %d does not fit the LDS staging (%zu B)", max_m, max_n, lds);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)house_qr_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)house_qr_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_done = true;
  }
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(house_qr_kernel<cd>, dim3(nprob), dim3(1024), lds, s, d_desc);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(house_qr_kernel<double>, dim3(nprob), dim3(1024), lds, s, d_desc);
  else {
    set_error("tmf_house_qr_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_house_qr_batched");
}
