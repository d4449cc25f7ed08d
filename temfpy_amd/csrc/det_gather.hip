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
#include "det_common.hpp"

namespace tmf {

// LDS layout (dynamic):
//   [ M : sb*sk T ][ ket idx : nsk*N u8 ][ bra idx : (a1-a0)*N u8 ]
//   [ per wave: Ma = M[rows(a), :]  (N|1)*sk T ][ per group: scratch N+1 T ]
// Each wavefront takes one bra row-set a at a time, gathers its N rows of M once, and its
// 64/G groups then sweep the ket sets b; the only HBM traffic is one result per determinant.
// Column stride of Ma is odd (N|1) and the scratch stride is N+1 so that the 16-lane phases
// of ds_read_b128 fall on distinct banks.
template <typename T, int N, int G>
__global__ __launch_bounds__(256) void det_kernel(const tmf_det_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_det_desc d = desc[blockIdx.x];
  constexpr int NS = N | 1;
  const int na = d.a1 - d.a0;
  const int sk = d.sk;
  T* Ms = reinterpret_cast<T*>(smem);
  const size_t mbytes = ((size_t)d.sb * sk * sizeof(T) + 15) & ~(size_t)15;
  uint8_t* kidx = smem + mbytes;
  uint8_t* bidx = kidx + (((size_t)d.nsk * N + 15) & ~(size_t)15);
  T* wbase = reinterpret_cast<T*>(bidx + (((size_t)na * N + 15) & ~(size_t)15));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int GPW = 64 / G;  // groups per wavefront
  T* Ma = wbase + (size_t)wave * ((size_t)NS * sk + GPW * (N + 1));
  T* scratch = Ma + (size_t)NS * sk + (size_t)(lane / G) * (N + 1);

  const T* __restrict__ S = reinterpret_cast<const T*>(d.S);
  for (int e = threadIdx.x; e < d.sb * sk; e += 256) {
    const int r = e % d.sb, c = e / d.sb;
    Ms[e] = S[(size_t)r + (size_t)c * d.lds];
  }
  const uint8_t* __restrict__ gk = reinterpret_cast<const uint8_t*>(d.ket_idx);
  const uint8_t* __restrict__ gb = reinterpret_cast<const uint8_t*>(d.bra_idx) + (size_t)d.a0 * N;
  for (int e = threadIdx.x; e < d.nsk * N; e += 256) kidx[e] = gk[e];
  for (int e = threadIdx.x; e < na * N; e += 256) bidx[e] = gb[e];
  __syncthreads();

  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const int grp = lane / G, c = lane % G;
  for (int al = wave; al < na; al += 4) {
    // gather the N rows of this bra set: Ma[r + col*NS] = M[rows_a[r], col]
    const uint8_t* rows = bidx + al * N;
    for (int e = lane; e < N * sk; e += 64) {
      const int r = e % N, col = e / N;
      Ma[r + col * NS] = Ms[rows[r] + col * d.sb];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T* __restrict__ orow = out + (size_t)(d.a0 + al) * d.nsk;
    for (int b0 = 0; b0 < d.nsk; b0 += GPW) {
      const int b = b0 + grp;
      const bool live = b < d.nsk;
      const T* colp = Ma + ((live && c < N) ? (int)kidx[b * N + c] * NS : 0);
      T a[N];
#pragma unroll
      for (int r = 0; r < N; ++r) a[r] = colp[r];
      const T det = det_group<T, N, G>(a, c, scratch);
      if (live && c == 0) orow[b] = sc<T>::mul(scale, det);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// n = 0: every "minor" is the empty determinant 1
template <typename T>
__global__ __launch_bounds__(256) void det_fill_kernel(const tmf_det_desc* __restrict__ desc) {
  const tmf_det_desc d = desc[blockIdx.x];
  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out) + (size_t)d.a0 * d.nsk;
  for (int e = threadIdx.x; e < (d.a1 - d.a0) * d.nsk; e += 256) out[e] = scale;
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

// Slow but general: any sometimes-matrix (read from global memory, not staged), minors of order n with n x n elements in
// LDS (n <= 97 complex, 138 real) - what the reference's numpy.linalg.det handles without a thought (slater.py:828-869).
// One determinant at a time per workgroup, one thread per column, partial pivoting.
// LDS: [ W : n*n T ]
template <typename T>
__global__ __launch_bounds__(256) void det_global_kernel(const tmf_det_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int s_piv;
  const tmf_det_desc d = desc[blockIdx.x];
  const int n = d.n, na = d.a1 - d.a0;
  T* W = reinterpret_cast<T*>(smem);
  const T* __restrict__ S = reinterpret_cast<const T*>(d.S);
  const uint8_t* __restrict__ gk = reinterpret_cast<const uint8_t*>(d.ket_idx);
  const uint8_t* __restrict__ gb = reinterpret_cast<const uint8_t*>(d.bra_idx) + (size_t)d.a0 * n;
  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const int c = threadIdx.x;
  for (int p = 0; p < na * d.nsk; ++p) {
    const int al = p / d.nsk, b = p % d.nsk;
    if (c < n) {
      const size_t col = gk[(size_t)b * n + c];
      for (int r = 0; r < n; ++r) W[r + c * n] = S[(size_t)gb[(size_t)al * n + r] + col * d.lds];
    }
    __syncthreads();
    T det = sc<T>::one();
    for (int j = 0; j < n; ++j) {
      if (c == j) {
        int best = j;
        double bv = sc<T>::abs2(W[j + j * n]);
        for (int r = j + 1; r < n; ++r) {
          const double v = sc<T>::abs2(W[r + j * n]);
          if (v > bv) bv = v, best = r;
        }
        s_piv = best;
      }
      __syncthreads();
      const int piv = s_piv;
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

template <typename T, int N>
static void launch_exact(dim3 g, int lds, hipStream_t s, const tmf_det_desc* d) {
  constexpr int G = N <= 8 ? 8 : (N <= 16 ? 16 : 32);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)det_kernel<T, N, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((det_kernel<T, N, G>), g, dim3(256), lds, s, d);
}

template <typename T>
static int launch_det(int n, const tmf_det_desc* d, int nt, int lds, hipStream_t s) {
  dim3 g(nt);
  switch (n) {
    case 0: hipLaunchKernelGGL((det_fill_kernel<T>), g, dim3(256), 0, s, d); break;
#define TMF_CASE(N) case N: launch_exact<T, N>(g, lds, s, d); break;
    TMF_CASE(1) TMF_CASE(2) TMF_CASE(3) TMF_CASE(4) TMF_CASE(5) TMF_CASE(6) TMF_CASE(7) TMF_CASE(8)
    TMF_CASE(9) TMF_CASE(10) TMF_CASE(11) TMF_CASE(12) TMF_CASE(13) TMF_CASE(14) TMF_CASE(15) TMF_CASE(16)
    TMF_CASE(17) TMF_CASE(18) TMF_CASE(19) TMF_CASE(20) TMF_CASE(21) TMF_CASE(22) TMF_CASE(23) TMF_CASE(24)
    TMF_CASE(25) TMF_CASE(26) TMF_CASE(27) TMF_CASE(28) TMF_CASE(29) TMF_CASE(30) TMF_CASE(31) TMF_CASE(32)
#undef TMF_CASE
    case 64: {
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)det_lds_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
      }
      hipLaunchKernelGGL((det_lds_kernel<T>), g, dim3(64), lds, s, d);
      break;
    }
    case 255: {
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)det_global_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);   // (+ 4 B static)
        attr = true;
      }
      hipLaunchKernelGGL((det_global_kernel<T>), g, dim3(256), lds, s, d);
      break;
    }
    default:
      set_error("tmf_det_gather_batched: order must be 0..32 (exact), 64 (generic) or 255 (global memory), got %d", n);
      return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_det_gather_batched launch");
}

}  // namespace tmf

extern "C" int tmf_det_gather_batched(int dtype, int order, const tmf_det_desc* d_desc, int ntiles, int lds_bytes,
                                      void* stream) {
  if (ntiles <= 0) return TMF_OK;
  if (lds_bytes < 0 || lds_bytes > 160 * 1024) {
    tmf::set_error("tmf_det_gather_batched: lds_bytes %d exceeds the 160 KiB LDS of a CU", lds_bytes);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128) return tmf::launch_det<tmf::cd>(order, d_desc, ntiles, lds_bytes, s);
  if (dtype == TMF_F64) return tmf::launch_det<double>(order, d_desc, ntiles, lds_bytes, s);
  tmf::set_error("tmf_det_gather_batched: bad dtype %d", dtype);
  return TMF_E_ARG;
}
