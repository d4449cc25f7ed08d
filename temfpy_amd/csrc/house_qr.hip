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
// Layout: the matrix stays in global memory (a 260 x 130 block is 270 KB: L2, not LDS), the current reflector
// lives in LDS; in step k the 16 wavefronts each take trailing columns (lane-strided dot product with the
// reflector, 6 shuffle steps, rank-1 update).  Phase 2 forms the thin Q in place (LAPACK org2r order).
// For m < n (more columns than rows: rank <= m) Q is m x m padded with zero columns and R is padded with
// zero rows, so that shapes stay fixed for the caller.
#include "common.hpp"

namespace tmf {

template <typename T>
__device__ inline T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v = sc<T>::add(v, shfl_xor_t<T>(v, o, 64));
  return v;
}

template <typename T>
__global__ __launch_bounds__(1024) void house_qr_kernel(const tmf_qr_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_qr_desc d = desc[blockIdx.x];
  const int m = d.m, n = d.n;
  if (m <= 0 || n <= 0) return;
  T* vs = reinterpret_cast<T*>(smem);                 // reflector, m elements
  T* taus = vs + m;                                   // K reflector scalars
  const int K = m < n ? m : n;
  double* red = reinterpret_cast<double*>(taus + K);  // 8 partial sums + scalars
  T* par = reinterpret_cast<T*>(red + 16);            // [0] = 1 / (alpha - beta), [1] = beta
  T* __restrict__ A = reinterpret_cast<T*>(d.A);
  T* __restrict__ R = reinterpret_cast<T*>(d.R);
  constexpr int NT = 1024, NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t lda = d.lda;

  // ---- phase 1: A = H_0 .. H_{K-1} R, reflectors stored below the diagonal -------------------------
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int r = k + tid; r < m; r += NT) {
      const T x = A[r + k * lda];
      vs[r] = x;
      if (r > k) s += sc<T>::abs2(x);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
      double xn2 = 0.0;
      for (int i = 0; i < NW; ++i) xn2 += red[i];
      const T alpha = vs[k];
      T tau = sc<T>::zero(), scal = sc<T>::zero(), beta = alpha;
      if (xn2 > 0.0 || sc<T>::imag(alpha) != 0.0) {
        double b = sqrt(sc<T>::abs2(alpha) + xn2);
        if (sc<T>::real(alpha) > 0.0) b = -b;              // beta = -sign(Re alpha) |x|
        beta = sc<T>::from_real(b);
        // tau = (beta - alpha) / beta ;  v = x / (alpha - beta)
        tau = sc<T>::scale(sc<T>::sub(beta, alpha), 1.0 / b);
        scal = sc<T>::inv(sc<T>::sub(alpha, beta));
      }
      taus[k] = tau;
      par[0] = scal;
      par[1] = beta;
    }
    __syncthreads();
    const T tau = taus[k], scal = par[0];
    for (int r = k + tid; r < m; r += NT) {
      const T v = (r == k) ? sc<T>::one() : sc<T>::mul(vs[r], scal);
      vs[r] = v;
      A[r + k * lda] = (r == k) ? par[1] : v;
    }
    __syncthreads();
    // trailing columns: a_c -= conj(tau) (v^H a_c) v
    const T ctau = sc<T>::conj(tau);
    for (int c = k + 1 + wave; c < n; c += NW) {
      T* __restrict__ a = A + (size_t)c * lda;
      T dot = sc<T>::zero();
      for (int r = k + lane; r < m; r += 64) dot = sc<T>::fmacc(dot, vs[r], a[r]);
      dot = sc<T>::mul(ctau, wave_sum<T>(dot));
      for (int r = k + lane; r < m; r += 64) a[r] = sc<T>::fms(a[r], dot, vs[r]);
    }
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
  __syncthreads();
  // ---- phase 2: thin Q in place (columns >= K become zero) -------------------------------------------
  for (int c = K + wave; c < n; c += NW)
    for (int r = lane; r < m; r += 64) A[r + (size_t)c * lda] = sc<T>::zero();
  for (int k = K - 1; k >= 0; --k) {
    const T tau = taus[k];
    for (int r = k + tid; r < m; r += NT) vs[r] = (r == k) ? sc<T>::one() : A[r + k * lda];
    __syncthreads();
    // Q[k:, k+1:K] = H_k Q[k:, k+1:K] = Q - tau v (v^H Q)
    for (int c = k + 1 + wave; c < K; c += NW) {
      T* __restrict__ a = A + (size_t)c * lda;
      T dot = sc<T>::zero();
      for (int r = k + lane; r < m; r += 64) dot = sc<T>::fmacc(dot, vs[r], a[r]);
      dot = sc<T>::mul(tau, wave_sum<T>(dot));
      for (int r = k + lane; r < m; r += 64) a[r] = sc<T>::fms(a[r], dot, vs[r]);
    }
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
