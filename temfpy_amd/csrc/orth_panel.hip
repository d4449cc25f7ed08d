// Panel orthonormalisation: classical Gram-Schmidt applied twice ("twice is enough"),
// one 256-thread workgroup per problem, the n x w panel resident in LDS.
//
// It is the LDS-staged panel step of the blocked QR (BCGS2) that orthonormalises orbital
// slabs: the host driver first projects the panel against all previous panels with two
// MFMA GEMM passes (gemm.hip) and then calls this kernel.  Together they produce the
// orthonormal orbital bases that the reference obtains from numpy.linalg.eigh
// (slater.py:347).  A column whose norm is exactly zero stays zero (an exactly
// rank-deficient slab, e.g. a product-state cut); downstream it carries singular value 0.
// Inside the panel every column is re-orthogonalised THREE times: range-finder slabs are
// numerically rank deficient, and for a column that is rounding noise after the first pass
// "twice is enough" fails (measured: 6e-4 loss of orthogonality on spinful chains); the third
// pass leaves the noise component outside span(Q) and removes the inside one to eps.
//
// Mapping: 16-lane group g computes the coefficient <q_g, v> (lanes stride the rows, one
// ds_read_b128 per element, 4 shuffle steps to reduce); the update v -= Q c is one row per
// thread.  LDS image is column-major so that consecutive lanes touch consecutive rows.
#include "common.hpp"

namespace tmf {

constexpr int WMAX = 16;

template <typename T>
__global__ __launch_bounds__(256) void orth_panel_kernel(const tmf_panel_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_panel_desc d = desc[blockIdx.x];
  const int n = d.n, w = d.w;
  if (n <= 0 || w <= 0) return;
  T* P = reinterpret_cast<T*>(smem);            // P[c * n + r]
  T* coef = P + (size_t)n * w;                  // WMAX coefficients
  double* red = reinterpret_cast<double*>(coef + WMAX);  // 4 partial norms
  // A residual below DROP * (original column norm) is rounding noise: the column is numerically
  // dependent on the previous ones (rank-deficient slab, e.g. exactly decoupled spin species).
  // Normalising it would break orthogonality, so it becomes an exact zero column.
  const double* __restrict__ norms = reinterpret_cast<const double*>(d.norms);
  const double DROP = 1e-14;

  T* __restrict__ A = reinterpret_cast<T*>(d.A);
  const int tid = threadIdx.x;
  for (int e = tid; e < n * w; e += 256) {
    const int r = e % n, c = e / n;
    P[e] = A[(size_t)r + (size_t)c * d.lda];
  }
  __syncthreads();

  const int grp = tid >> 4, gl = tid & 15;
  const int lane = tid & 63, wave = tid >> 6;
  for (int j = 0; j < w; ++j) {
    T* v = P + (size_t)j * n;
    for (int pass = 0; pass < 3 && j > 0; ++pass) {
      // coefficients c_g = <q_g, v>, g < j
      if (grp < j) {
        const T* q = P + (size_t)grp * n;
        // four independent accumulators: with one workgroup per CU (the panel fills the LDS) nothing else
        // hides the LDS latency of the two dependent reads per row (measured: the kernel was 3x slower
        // than its instruction count)
        T acc = sc<T>::zero(), acc1 = sc<T>::zero(), acc2 = sc<T>::zero(), acc3 = sc<T>::zero();
        int r = gl;
        for (; r + 48 < n; r += 64) {
          acc = sc<T>::fmacc(acc, q[r], v[r]);
          acc1 = sc<T>::fmacc(acc1, q[r + 16], v[r + 16]);
          acc2 = sc<T>::fmacc(acc2, q[r + 32], v[r + 32]);
          acc3 = sc<T>::fmacc(acc3, q[r + 48], v[r + 48]);
        }
        for (; r < n; r += 16) acc = sc<T>::fmacc(acc, q[r], v[r]);
        acc = sc<T>::add(sc<T>::add(acc, acc1), sc<T>::add(acc2, acc3));
        for (int o = 8; o > 0; o >>= 1) acc = sc<T>::add(acc, shfl_xor_t<T>(acc, o, 16));
        if (gl == 0) coef[grp] = acc;
      }
      __syncthreads();
      for (int r = tid; r < n; r += 256) {
        T x = v[r], x1 = sc<T>::zero();
        int g = 0;
        for (; g + 1 < j; g += 2) {   // two chains: the reads of consecutive columns overlap
          x = sc<T>::fms(x, P[(size_t)g * n + r], coef[g]);
          x1 = sc<T>::fms(x1, P[(size_t)(g + 1) * n + r], coef[g + 1]);
        }
        if (g < j) x = sc<T>::fms(x, P[(size_t)g * n + r], coef[g]);
        v[r] = sc<T>::add(x, x1);
      }
      __syncthreads();
    }
    // norm and scale
    double s = 0.0;
    for (int r = tid; r < n; r += 256) s += sc<T>::abs2(v[r]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const double nrm2 = red[0] + red[1] + red[2] + red[3];
    const double lim = norms ? DROP * norms[j] : 0.0;
    const double f = (nrm2 > 0.0 && nrm2 >= lim * lim) ? 1.0 / sqrt(nrm2) : 0.0;
    for (int r = tid; r < n; r += 256) v[r] = sc<T>::scale(v[r], f);
    __syncthreads();
  }
  for (int e = tid; e < n * w; e += 256) {
    const int r = e % n, c = e / n;
    A[(size_t)r + (size_t)c * d.lda] = P[e];
  }
}

}  // namespace tmf

extern "C" int tmf_orth_panel_batched(int dtype, const tmf_panel_desc* d_desc, int nprob, int max_n, int max_w,
                                      void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  if (max_w > WMAX || max_w <= 0) {
    set_error("tmf_orth_panel_batched: panel width %d not in 1..%d", max_w, WMAX);
    return TMF_E_ARG;
  }
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const size_t lds = ((size_t)max_n * max_w + WMAX) * elem + 64;
  if (lds > 160 * 1024) {
    set_error("tmf_orth_panel_batched: panel %d x %d needs %zu B of LDS (> 160 KiB); use a narrower panel", max_n,
              max_w, lds);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)orth_panel_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)orth_panel_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_done = true;
  }
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(orth_panel_kernel<cd>, dim3(nprob), dim3(256), lds, s, d_desc);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(orth_panel_kernel<double>, dim3(nprob), dim3(256), lds, s, d_desc);
  else {
    set_error("tmf_orth_panel_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_orth_panel_batched launch");
}
