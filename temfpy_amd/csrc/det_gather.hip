// Batched gathered determinants -- the hot kernel of the sweep.
//
// Reference: slater.py:828-869 (`_tensor_block`): for one charge sector it materialises the
// 4-D gather O[a, b, :, :] = M[rows(a)][:, cols(b)] and calls numpy.linalg.det on the
// batch (LAPACK zgetrf one matrix at a time): 90 % of the reference's wall time.
//
// Here nothing is materialised in HBM: the (<= 64 KiB) Schur-complement matrix M of a site
// is staged in LDS once per workgroup, index lists sit next to it, and every group of G
// lanes (G = 8, 16 or 32 >= n) owns one determinant: lane c holds column c of the n x n
// minor in registers, LU with partial pivoting runs with the pivot column broadcast by
// wavefront shuffles, and the only HBM traffic is the 8/16-byte result per determinant.
// Work is vector-fp64 + LDS bound, not MFMA shaped: n ~ 8..19.
#include "common.hpp"

namespace tmf {

__device__ inline double sel(bool m, double a, double b) { return m ? a : b; }
__device__ inline cd sel(bool m, cd a, cd b) { return make_cd(m ? a.x : b.x, m ? a.y : b.y); }

// One determinant per G-lane group, column c of the minor in registers of lane c.
template <typename T, int NMAX, int G>
__device__ inline T det_group(T (&a)[NMAX], const int n, const int c) {
  T det = sc<T>::one();
#pragma unroll
  for (int j = 0; j < NMAX; ++j) {
    if (j < n) {  // wave-uniform: every determinant of a sector has the same order
      // pivot search in column j (every lane scans its own column; lane j's answer counts)
      int best = j;
      double bv = sc<T>::abs2(a[j]);
#pragma unroll
      for (int r = j + 1; r < NMAX; ++r) {
        const double v = sc<T>::abs2(a[r]);
        if (r < n && v > bv) {
          bv = v;
          best = r;
        }
      }
      const int piv = __shfl(best, j, G);
      // row swap j <-> piv in every column, as selects (a runtime-indexed register array
      // would be demoted to scratch: cdna_hip_programming.md section 5.4 rule 20)
      {
        const T aj = a[j];
        T nj = aj;
#pragma unroll
        for (int r = j + 1; r < NMAX; ++r) {
          const bool m = (r == piv);
          const T ar = a[r];
          nj = sel(m, ar, nj);
          a[r] = sel(m, aj, ar);
        }
        a[j] = nj;
      }
      const T p = shfl_t<T>(a[j], j, G);
      det = sc<T>::mul(det, p);
      if (piv != j) det = sc<T>::neg(det);
      const T pinv = sc<T>::abs2(p) > 0.0 ? sc<T>::inv(p) : sc<T>::zero();
      const T pc = sc<T>::mul(pinv, a[j]);  // pivot-row entry of this column / pivot
#pragma unroll
      for (int r = j + 1; r < NMAX; ++r) {
        if (r < n) {
          const T l = shfl_t<T>(a[r], j, G);  // column-j entry below the pivot
          a[r] = sc<T>::fms(a[r], l, pc);
        }
      }
    }
  }
  return det;
}

// LDS layout (dynamic): [ M : sb*sk T ][ ket idx : nsk*n u8 ][ bra idx : (a1-a0)*n u8 ]
template <typename T, int NMAX, int G>
__global__ __launch_bounds__(256) void det_kernel(const tmf_det_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_det_desc d = desc[blockIdx.x];
  const int n = d.n;
  const int na = d.a1 - d.a0;
  T* Ms = reinterpret_cast<T*>(smem);
  const size_t mbytes = ((size_t)d.sb * d.sk * sizeof(T) + 15) & ~(size_t)15;
  uint8_t* kidx = smem + mbytes;
  uint8_t* bidx = kidx + (((size_t)d.nsk * n + 15) & ~(size_t)15);

  const T* __restrict__ S = reinterpret_cast<const T*>(d.S);
  for (int e = threadIdx.x; e < d.sb * d.sk; e += 256) {
    const int r = e % d.sb, c = e / d.sb;
    Ms[e] = S[(size_t)r + (size_t)c * d.lds];
  }
  const uint8_t* __restrict__ gk = reinterpret_cast<const uint8_t*>(d.ket_idx);
  const uint8_t* __restrict__ gb = reinterpret_cast<const uint8_t*>(d.bra_idx) + (size_t)d.a0 * n;
  for (int e = threadIdx.x; e < d.nsk * n; e += 256) kidx[e] = gk[e];
  for (int e = threadIdx.x; e < na * n; e += 256) bidx[e] = gb[e];
  __syncthreads();

  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  constexpr int NG = 256 / G;
  const int grp = threadIdx.x / G, c = threadIdx.x % G;
  const int npairs = na * d.nsk;
  // all groups of a wave iterate the same number of times (shuffles need the whole wave)
  const int iters = (npairs + NG - 1) / NG;
  for (int it = 0; it < iters; ++it) {
    const int p = it * NG + grp;
    const bool live = p < npairs;
    const int al = live ? p / d.nsk : 0, b = live ? p % d.nsk : 0;
    T a[NMAX];
    const int col = (c < n) ? kidx[b * n + c] : 0;
#pragma unroll
    for (int r = 0; r < NMAX; ++r) {
      T v = sc<T>::zero();
      if (r < n && c < n) v = Ms[bidx[al * n + r] + col * d.sb];
      a[r] = v;
    }
    const T det = det_group<T, NMAX, G>(a, n, c);
    if (live && c == 0) out[(size_t)(d.a0 + al) * d.nsk + b] = sc<T>::mul(scale, det);
  }
}

// Fallback for 32 < n <= 64: one determinant per wave, minor held in LDS.
// LDS: [ M ][ ket idx ][ bra idx ][ scratch n*n T ]
template <typename T>
__global__ __launch_bounds__(64) void det_lds_kernel(const tmf_det_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_det_desc d = desc[blockIdx.x];
  const int n = d.n, na = d.a1 - d.a0;
  T* Ms = reinterpret_cast<T*>(smem);
  const size_t mbytes = ((size_t)d.sb * d.sk * sizeof(T) + 15) & ~(size_t)15;
  uint8_t* kidx = smem + mbytes;
  uint8_t* bidx = kidx + (((size_t)d.nsk * n + 15) & ~(size_t)15);
  T* W = reinterpret_cast<T*>(bidx + (((size_t)na * n + 15) & ~(size_t)15));
  const T* __restrict__ S = reinterpret_cast<const T*>(d.S);
  for (int e = threadIdx.x; e < d.sb * d.sk; e += 64) Ms[e] = S[(size_t)(e % d.sb) + (size_t)(e / d.sb) * d.lds];
  const uint8_t* __restrict__ gk = reinterpret_cast<const uint8_t*>(d.ket_idx);
  const uint8_t* __restrict__ gb = reinterpret_cast<const uint8_t*>(d.bra_idx) + (size_t)d.a0 * n;
  for (int e = threadIdx.x; e < d.nsk * n; e += 64) kidx[e] = gk[e];
  for (int e = threadIdx.x; e < na * n; e += 64) bidx[e] = gb[e];
  __syncthreads();
  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const int c = threadIdx.x;
  for (int p = 0; p < na * d.nsk; ++p) {
    const int al = p / d.nsk, b = p % d.nsk;
    if (c < n) {
      const int col = kidx[b * n + c];
      for (int r = 0; r < n; ++r) W[r + c * n] = Ms[bidx[al * n + r] + col * d.sb];
    }
    __syncthreads();
    T det = sc<T>::one();
    for (int j = 0; j < n; ++j) {
      int best = j;
      if (c == j) {
        double bv = sc<T>::abs2(W[j + j * n]);
        for (int r = j + 1; r < n; ++r) {
          const double v = sc<T>::abs2(W[r + j * n]);
          if (v > bv) bv = v, best = r;
        }
      }
      const int piv = __shfl(best, j, 64);
      if (c < n && piv != j) {
        const T t = W[j + c * n];
        W[j + c * n] = W[piv + c * n];
        W[piv + c * n] = t;
      }
      __syncthreads();
      const T p_ = W[j + j * n];
      det = sc<T>::mul(det, p_);
      if (piv != j) det = sc<T>::neg(det);
      const T pinv = sc<T>::abs2(p_) > 0.0 ? sc<T>::inv(p_) : sc<T>::zero();
      if (c > j && c < n) {
        const T pc = sc<T>::mul(pinv, W[j + c * n]);
        for (int r = j + 1; r < n; ++r) W[r + c * n] = sc<T>::fms(W[r + c * n], W[r + j * n], pc);
      }
      __syncthreads();
    }
    if (c == 0) out[(size_t)(d.a0 + al) * d.nsk + b] = sc<T>::mul(scale, det);
    __syncthreads();
  }
}

template <typename T>
static int launch_det(int n_class, const tmf_det_desc* d, int nt, int lds, hipStream_t s) {
  dim3 g(nt);
  switch (n_class) {
    case 8:
      hipLaunchKernelGGL((det_kernel<T, 8, 8>), g, dim3(256), lds, s, d);
      break;
    case 16:
      hipLaunchKernelGGL((det_kernel<T, 16, 16>), g, dim3(256), lds, s, d);
      break;
    case 32:
      hipLaunchKernelGGL((det_kernel<T, 32, 32>), g, dim3(256), lds, s, d);
      break;
    case 64:
      hipLaunchKernelGGL((det_lds_kernel<T>), g, dim3(64), lds, s, d);
      break;
    default:
      set_error("tmf_det_gather_batched: n_class must be 8, 16, 32 or 64, got %d", n_class);
      return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_det_gather_batched launch");
}

}  // namespace tmf

extern "C" int tmf_det_gather_batched(int dtype, int n_class, const tmf_det_desc* d_desc, int ntiles, int lds_bytes,
                                      void* stream) {
  if (ntiles <= 0) return TMF_OK;
  if (lds_bytes < 0 || lds_bytes > 160 * 1024) {
    tmf::set_error("tmf_det_gather_batched: lds_bytes %d exceeds the 160 KiB LDS of a CU", lds_bytes);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {  // allow > 64 KiB of dynamic LDS
    using namespace tmf;
    const int big = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)det_kernel<cd, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)det_kernel<cd, 16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)det_kernel<cd, 32, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)det_lds_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)det_kernel<double, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)det_kernel<double, 16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)det_kernel<double, 32, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)det_lds_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    attr_done = true;
  }
  if (dtype == TMF_C128) return tmf::launch_det<tmf::cd>(n_class, d_desc, ntiles, lds_bytes, s);
  if (dtype == TMF_F64) return tmf::launch_det<double>(n_class, d_desc, ntiles, lds_bytes, s);
  tmf::set_error("tmf_det_gather_batched: bad dtype %d", dtype);
  return TMF_E_ARG;
}
