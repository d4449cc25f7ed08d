// Error plumbing + small utility kernels (transpose, RNG fill, signed gather, column normalise).
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>

#include "common.hpp"

namespace tmf {

static thread_local char g_err[512] = "";
static thread_local const int32_t* g_run_if = nullptr;
const int32_t* launch_condition() { return g_run_if; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return TMF_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return TMF_E_HIP;
}

// ---- transpose: row-major host layout -> column-major device layout -----------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int n) {
  __shared__ T tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int i = by + r, j = bx + tx;  // in[i][j] row-major
    if (i < n && j < n) tile[r][tx] = in[(size_t)i * n + j];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    // out column-major: out[i + j*n] = in[i*n + j] -> write runs along i
    const int j = bx + r, i = by + tx;
    if (i < n && j < n) out[(size_t)i + (size_t)j * n] = tile[tx][r];
  }
}

// ---- counter-based normal fill (splitmix64 + Box-Muller) -----------------------------
__device__ inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void fill_normal_kernel(double* __restrict__ out, int64_t pairs, int64_t count,
                                                          uint64_t seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < pairs; i += stride) {
    const uint64_t a = splitmix(seed ^ (uint64_t)(2 * i)), b = splitmix(seed ^ (uint64_t)(2 * i + 1));
    const double u1 = ((a >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0,1]
    const double u2 = (b >> 11) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    if (2 * i < count) out[2 * i] = r * c;
    if (2 * i + 1 < count) out[2 * i + 1] = r * s;
  }
}

// ---- signed gather: dst[r,c] = sr[r]*sc[c]*src[row_sel[r], col_sel[c]] -----------------
template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const tmf_gather_desc* __restrict__ desc, const int32_t* __restrict__ run_if) {
  if (run_if != nullptr && *run_if == 0) return;
  const tmf_gather_desc d = desc[blockIdx.x];
  const T* __restrict__ src = reinterpret_cast<const T*>(d.src);
  const T* __restrict__ phys = reinterpret_cast<const T*>(d.phys);
  T* __restrict__ dst = reinterpret_cast<T*>(d.dst);
  const int32_t* rs = reinterpret_cast<const int32_t*>(d.row_sel);
  const int32_t* cs = reinterpret_cast<const int32_t*>(d.col_sel);
  const int8_t* rg = reinterpret_cast<const int8_t*>(d.row_sign);
  const int8_t* cg = reinterpret_cast<const int8_t*>(d.col_sign);
  const int total = d.rows * d.cols;
  // (gridDim.y workgroups share one problem: a shard of 38 sites would otherwise copy 10 MB per site through 38 CUs)
  for (int e = threadIdx.x + blockIdx.y * blockDim.x; e < total; e += blockDim.x * gridDim.y) {
    const int r = e % d.rows, c = e / d.rows;
    const int sr = rs[r], sc_ = cs[c];
    T v = (sr < 0) ? phys[(size_t)(-sr - 1) + (size_t)sc_ * d.ldp] : src[(size_t)sr + (size_t)sc_ * d.lds_];
    const double sg = (double)(rg[r] * cg[c]);
    dst[(size_t)r + (size_t)c * d.ldd] = sc<T>::scale(v, sg);
  }
}

// ---- column normalise (+ optional reversal / conjugation / real part / odd sign flip, slater.py:410,
//      pfaffian.py:807-816) -------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colnorm_kernel(const tmf_colnorm_desc* __restrict__ desc) {
  const tmf_colnorm_desc d = desc[blockIdx.x];
  const T* __restrict__ src = reinterpret_cast<const T*>(d.src);
  T* __restrict__ dst = reinterpret_cast<T*>(d.dst);
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = 0; c < d.c; ++c) {
    double s = 0.0;
    for (int r = threadIdx.x; r < d.n; r += 256) {
      T v = src[(size_t)r + (size_t)c * d.lds_];
      if (d.reverse & 4) v = sc<T>::real_only(v);
      s += sc<T>::abs2(v);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
    const int cd_ = (d.reverse & 1) ? d.c - 1 - c : c;
    double f = nrm > 0.0 ? 1.0 / nrm : 0.0;
    if (d.flip_odd && (cd_ & 1)) f = -f;
    for (int r = threadIdx.x; r < d.n; r += 256) {
      T v = src[(size_t)r + (size_t)c * d.lds_];
      if (d.reverse & 4) v = sc<T>::real_only(v);
      if (d.reverse & 2) v = sc<T>::conj(v);
      dst[(size_t)r + (size_t)cd_ * d.ldd] = sc<T>::scale(v, f);
    }
  }
}

// ---- strided block copy with optional (conjugate) transposition ------------------------------
// dst[r, c] = src[r, c]  (flags 0)   or   dst[c, r] = [conj] src[r, c]  (flags & 1, conj with flags & 2);
// rows x cols is the shape of the SOURCE block.  32 x 32 tiles through LDS: both sides run along
// their leading dimension.
template <typename T>
__global__ __launch_bounds__(256) void copy_blocks_kernel(const tmf_copy_desc* __restrict__ desc) {
  __shared__ T tile[32][33];
  const tmf_copy_desc d = desc[blockIdx.x];
  const T* __restrict__ src = reinterpret_cast<const T*>(d.src);
  T* __restrict__ dst = reinterpret_cast<T*>(d.dst);
  const int tr = (d.rows + 31) / 32, tc = (d.cols + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int t = blockIdx.y; t < tr * tc; t += gridDim.y) {
    const int r0 = (t % tr) * 32, c0 = (t / tr) * 32;
    if (!(d.flags & 1)) {
      for (int c = ty; c < 32; c += 8) {
        const int r = r0 + tx, cc = c0 + c;
        if (r < d.rows && cc < d.cols) dst[(size_t)r + (size_t)cc * d.ldd] = src[(size_t)r + (size_t)cc * d.lds_];
      }
      continue;
    }
    __syncthreads();
    for (int c = ty; c < 32; c += 8) {
      const int r = r0 + tx, cc = c0 + c;
      if (r < d.rows && cc < d.cols) {
        T v = src[(size_t)r + (size_t)cc * d.lds_];
        tile[c][tx] = (d.flags & 2) ? sc<T>::conj(v) : v;
      }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int cc = c0 + tx, rr = r0 + r;   // dst[cc, rr] = src[rr, cc]: runs along cc
      if (rr < d.rows && cc < d.cols) dst[(size_t)cc + (size_t)rr * d.ldd] = tile[tx][r];
    }
  }
}

// ---- column norms -----------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void norms_kernel(const tmf_norms_desc* __restrict__ desc) {
  const tmf_norms_desc d = desc[blockIdx.x];
  const T* __restrict__ src = reinterpret_cast<const T*>(d.src);
  double* __restrict__ out = reinterpret_cast<double*>(d.out);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // one wavefront per column; gridDim.y workgroups share the columns of a problem (few problems: a rank's shard)
  for (int c = wave + 4 * blockIdx.y; c < d.c; c += 4 * gridDim.y) {
    double s = 0.0;
    for (int r = lane; r < d.n; r += 64) s += sc<T>::abs2(src[(size_t)r + (size_t)c * d.lds_]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[c] = sqrt(s);
  }
}

}  // namespace tmf

using namespace tmf;

extern "C" const char* tmf_last_error(void) { return tmf::g_err; }
extern "C" int tmf_version(void) { return 100; }

extern "C" void tmf_launch_condition(const int32_t* d_flag) { tmf::g_run_if = d_flag; }

extern "C" int tmf_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return TMF_E_HIP;
  }
  return n;
}

extern "C" int tmf_transpose(int dtype, const void* d_in, void* d_out, int n, void* stream) {
  if (n <= 0) return TMF_OK;
  dim3 g((n + 31) / 32, (n + 31) / 32), b(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(transpose_kernel<cd>, g, b, 0, s, (const cd*)d_in, (cd*)d_out, n);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(transpose_kernel<double>, g, b, 0, s, (const double*)d_in, (double*)d_out, n);
  else {
    set_error("tmf_transpose: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_transpose");
}

extern "C" int tmf_fill_normal(int dtype, void* d_out, int64_t count, uint64_t seed, void* stream) {
  if (count <= 0) return TMF_OK;
  const int64_t doubles = (dtype == TMF_C128) ? 2 * count : count;
  const int64_t pairs = (doubles + 1) / 2;
  int64_t blocks = (pairs + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (double*)d_out, pairs, doubles, seed);
  return check_hip(hipGetLastError(), "tmf_fill_normal");
}

extern "C" int tmf_gather_signed_batched(int dtype, const tmf_gather_desc* d_desc, int nprob, void* stream) {
  if (nprob <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int gy = 2048 / nprob;
  gy = gy < 1 ? 1 : (gy > 32 ? 32 : gy);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(gather_kernel<cd>, dim3(nprob, gy), dim3(256), 0, s, d_desc, tmf::launch_condition());
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(gather_kernel<double>, dim3(nprob, gy), dim3(256), 0, s, d_desc, tmf::launch_condition());
  else {
    set_error("tmf_gather_signed_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_gather_signed_batched");
}

// ---- common power-of-two rescaling of the blocks of one step (tmf_rescale_desc) --------------
template <typename T>
__global__ __launch_bounds__(1024) void rescale_pow2_kernel(const tmf_rescale_desc* __restrict__ desc, int nblk,
                                                            long long* __restrict__ acc, long long* __restrict__ out) {
  __shared__ double s_max[16];
  __shared__ int s_e;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double m = 0.0;
  for (int b = 0; b < nblk; ++b) {
    const tmf_rescale_desc d = desc[b];
    const T* __restrict__ A = reinterpret_cast<const T*>(d.A);
    const int total = d.rows * d.cols;
    for (int e = tid; e < total; e += 1024) {
      const T v = A[(size_t)(e % d.rows) + (size_t)(e / d.rows) * d.ld];
      m = fmax(m, fmax(fabs(sc<T>::real(v)), fabs(sc<T>::imag(v))));
    }
  }
  m = wave_max64(m);
  if (lane == 0) s_max[wave] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = 0.0;
    for (int w = 0; w < 16; ++w) mm = fmax(mm, s_max[w]);
    int e = 0;
    if (mm > 0.0 && mm < 1.7e308) e = ilogb(mm);
    s_e = e;
    const long long a = *acc + e;
    *acc = a;
    if (out) *out = a;
  }
  __syncthreads();
  const int e = s_e;
  if (e == 0) return;
  const double f = ldexp(1.0, -e);
  for (int b = 0; b < nblk; ++b) {
    const tmf_rescale_desc d = desc[b];
    T* __restrict__ A = reinterpret_cast<T*>(d.A);
    const int total = d.rows * d.cols;
    for (int x = tid; x < total; x += 1024) {
      const size_t o = (size_t)(x % d.rows) + (size_t)(x / d.rows) * d.ld;
      A[o] = sc<T>::scale(A[o], f);
    }
  }
}

extern "C" int tmf_rescale_pow2_batched(int dtype, const tmf_rescale_desc* d_desc, int nblk, int64_t* d_acc, int64_t* d_out,
                                        void* stream) {
  if (nblk <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(rescale_pow2_kernel<cd>, dim3(1), dim3(1024), 0, s, d_desc, nblk, (long long*)d_acc, (long long*)d_out);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(rescale_pow2_kernel<double>, dim3(1), dim3(1024), 0, s, d_desc, nblk, (long long*)d_acc, (long long*)d_out);
  else {
    set_error("tmf_rescale_pow2_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_rescale_pow2_batched");
}

// ---- canonical gauge of the entangled orbitals (tmf_gauge_desc) --------------------------
namespace {
constexpr int GAUGE_DMAX = 8;
template <typename T>
__device__ inline T gauge_weight(int t, int j);
template <>
__device__ inline double gauge_weight<double>(int t, int j) {
  const uint64_t a = splitmix(0x6A09E667F3BCC908ull ^ ((uint64_t)t << 40) ^ (uint64_t)j);
  return (double)(int64_t)(a >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
template <>
__device__ inline cd gauge_weight<cd>(int t, int j) {
  const uint64_t a = splitmix(0x6A09E667F3BCC908ull ^ ((uint64_t)t << 40) ^ (uint64_t)j);
  const uint64_t b = splitmix(a ^ 0xBB67AE8584CAA73Bull);
  return make_cd((double)(int64_t)(a >> 11) * (2.0 / 9007199254740992.0) - 1.0,
                 (double)(int64_t)(b >> 11) * (2.0 / 9007199254740992.0) - 1.0);
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void gauge_kernel(const tmf_gauge_desc* __restrict__ desc, int kcap, int w0_rows) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_gauge_desc d = desc[blockIdx.x];
  const int n = d.n, k = d.k;
  if (n <= 0 || k <= 0) return;
  T* __restrict__ V = reinterpret_cast<T*>(d.V);
  const int32_t* __restrict__ start = reinterpret_cast<const int32_t*>(d.start);
  // LDS (sized for the largest problem of the launch: kcap columns, w0_rows rows)
  T* M = reinterpret_cast<T*>(smem);                   // [GAUGE_DMAX][k]: <w_t | v_c>, then the rotation of the group of c
  int* size = reinterpret_cast<int*>(smem + (size_t)GAUGE_DMAX * kcap * sizeof(T));
  T* w0 = reinterpret_cast<T*>(smem + (size_t)GAUGE_DMAX * kcap * sizeof(T) + (((size_t)kcap * 4 + 15) & ~(size_t)15));
  const bool have_w0 = n <= w0_rows;                   // first weight vector by row, else recomputed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (have_w0)
    for (int r = tid; r < n; r += 256) w0[r] = gauge_weight<T>(0, d.from_top ? r : n - 1 - r);
  for (int c = tid; c < k; c += 256) {
    const int s0 = start[c];
    int e = c;
    while (e < k && start[e] == s0) ++e;
    size[c] = e - s0;
  }
  __syncthreads();
  // overlaps with the weight vectors
  for (int c = wave; c < k; c += 4) {
    const int dsz = size[c];
    if (dsz > GAUGE_DMAX) continue;
    T acc[GAUGE_DMAX];
#pragma unroll
    for (int t = 0; t < GAUGE_DMAX; ++t) acc[t] = sc<T>::zero();
    if (dsz == 1 && have_w0) {                          // (the usual case: a phase)
      for (int r = lane; r < n; r += 64) acc[0] = sc<T>::fmacc(acc[0], w0[r], V[r + (size_t)c * d.ld]);
    } else {
      for (int r = lane; r < n; r += 64) {
        const T v = V[r + (size_t)c * d.ld];
        const int j = d.from_top ? r : n - 1 - r;
#pragma unroll
        for (int t = 0; t < GAUGE_DMAX; ++t)
          if (t < dsz) acc[t] = sc<T>::fmacc(acc[t], gauge_weight<T>(t, j), v);
      }
    }
#pragma unroll
    for (int t = 0; t < GAUGE_DMAX; ++t) {
      if (t >= dsz) continue;                          // (uniform)
      const T sum = wave_sum64(acc[t]);
      if (lane == 0) M[(size_t)t * k + c] = sum;
    }
  }
  __syncthreads();
  // one thread per group: Q = orthonormal factor of B^H (B = the group's d x d block of M), so that B Q is lower
  // triangular with a positive diagonal; Q overwrites the block (Q[c][s] at M[c - s0][s0 + s] ... stored column-wise)
  for (int c = tid; c < k; c += 256) {
    const int s0 = start[c], dsz = size[c];
    if (c != s0 || dsz > GAUGE_DMAX) continue;
    T q[GAUGE_DMAX][GAUGE_DMAX];                       // q[s][i]: component i of vector s
    for (int s = 0; s < dsz; ++s) {
      for (int i = 0; i < dsz; ++i) q[s][i] = sc<T>::conj(M[(size_t)s * k + s0 + i]);      // b_s = conj(row s of B)
      for (int pass = 0; pass < 2; ++pass)
        for (int p = 0; p < s; ++p) {
          T dot = sc<T>::zero();
          for (int i = 0; i < dsz; ++i) dot = sc<T>::fmacc(dot, q[p][i], q[s][i]);          // <q_p | b_s>
          for (int i = 0; i < dsz; ++i) q[s][i] = sc<T>::fms(q[s][i], dot, q[p][i]);
        }
      double nn = 0.0;
      for (int i = 0; i < dsz; ++i) nn += sc<T>::abs2(q[s][i]);
      if (!(nn > 1e-280)) {                            // (no direction left: keep the basis of this group)
        for (int a = 0; a < dsz; ++a)
          for (int i = 0; i < dsz; ++i) q[a][i] = (a == i) ? sc<T>::one() : sc<T>::zero();
        break;
      }
      const double inv = 1.0 / sqrt(nn);
      for (int i = 0; i < dsz; ++i) q[s][i] = sc<T>::scale(q[s][i], inv);
    }
    for (int s = 0; s < dsz; ++s)
      for (int i = 0; i < dsz; ++i) M[(size_t)i * k + s0 + s] = q[s][i];                     // Q[i][s]
  }
  __syncthreads();
  // V_group <- V_group Q
  for (int r = tid; r < n; r += 256) {
    for (int c = 0; c < k;) {
      const int dsz = size[c];
      if (dsz <= GAUGE_DMAX) {
        T v[GAUGE_DMAX], o[GAUGE_DMAX];
        for (int i = 0; i < dsz; ++i) v[i] = V[r + (size_t)(c + i) * d.ld], o[i] = sc<T>::zero();
        for (int s = 0; s < dsz; ++s)
          for (int i = 0; i < dsz; ++i) o[s] = sc<T>::add(o[s], sc<T>::mul(v[i], M[(size_t)i * k + c + s]));
        for (int s = 0; s < dsz; ++s) V[r + (size_t)(c + s) * d.ld] = o[s];
      }
      c += dsz;
    }
  }
}

extern "C" int tmf_canonical_gauge_batched(int dtype, const tmf_gauge_desc* d_desc, int nprob, int max_n, int max_k, void* stream) {
  if (nprob <= 0) return TMF_OK;
  if (max_k <= 0 || max_k > 4096 || max_n <= 0) {
    set_error("tmf_canonical_gauge_batched: bad sizes (%d rows, %d columns)", max_n, max_k);
    return TMF_E_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const int kcap = (max_k + 3) & ~3;
  const size_t head = (size_t)GAUGE_DMAX * kcap * elem + (((size_t)kcap * 4 + 15) & ~(size_t)15);
  int w0_rows = max_n;
  if (head + (size_t)w0_rows * elem > 60 * 1024) w0_rows = head < 60 * 1024 ? (int)((60 * 1024 - head) / elem) : 0;
  const size_t lds = head + (size_t)w0_rows * elem;     // <= 60 KiB: two or more workgroups per CU
  if (lds > 150 * 1024) {
    set_error("tmf_canonical_gauge_batched: %d columns need %zu B of LDS", max_k, lds);
    return TMF_E_LIMIT;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gauge_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)gauge_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(gauge_kernel<cd>, dim3(nprob), dim3(256), lds, s, d_desc, kcap, w0_rows);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(gauge_kernel<double>, dim3(nprob), dim3(256), lds, s, d_desc, kcap, w0_rows);
  else {
    set_error("tmf_canonical_gauge_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_canonical_gauge_batched");
}

extern "C" int tmf_normalise_columns_batched(int dtype, const tmf_colnorm_desc* d_desc, int nprob, void* stream) {
  if (nprob <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(colnorm_kernel<cd>, dim3(nprob), dim3(256), 0, s, d_desc);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(colnorm_kernel<double>, dim3(nprob), dim3(256), 0, s, d_desc);
  else {
    set_error("tmf_normalise_columns_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_normalise_columns_batched");
}

extern "C" int tmf_column_norms_batched(int dtype, const tmf_norms_desc* d_desc, int nprob, void* stream) {
  if (nprob <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int gy = 2048 / nprob;
  gy = gy < 1 ? 1 : (gy > 32 ? 32 : gy);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(norms_kernel<cd>, dim3(nprob, gy), dim3(256), 0, s, d_desc);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(norms_kernel<double>, dim3(nprob, gy), dim3(256), 0, s, d_desc);
  else {
    set_error("tmf_column_norms_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_column_norms_batched");
}

extern "C" int tmf_copy_blocks_batched(int dtype, const tmf_copy_desc* d_desc, int nprob, int max_tiles, void* stream) {
  if (nprob <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int gy = max_tiles < 1 ? 1 : (max_tiles > 64 ? 64 : max_tiles);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(copy_blocks_kernel<cd>, dim3(nprob, gy), dim3(256), 0, s, d_desc);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(copy_blocks_kernel<double>, dim3(nprob, gy), dim3(256), 0, s, d_desc);
  else {
    set_error("tmf_copy_blocks_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_copy_blocks_batched");
}

// ---- host <-> device plumbing of the result path ---------------------------------------
// The tensors of a conversion leave the GPU through page-locked host memory: either memory
// the caller registered (a POSIX shared-memory segment that rank 0 maps as well, so that
// every GPU of a node writes its shard through its own PCIe link into one host-visible
// result) or ordinary pinned allocations.
// Small results into page-locked host memory by a kernel (the buffer is mapped into the device's address space): unlike a
// DMA copy it does not queue behind the tensor download of the previous conversion on the copy engine.
__global__ __launch_bounds__(256) void export_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, const int64_t words) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int tmf_export_words(void* host_mapped_dst, const void* d_src, int64_t bytes, void* stream) {
  if (bytes <= 0) return TMF_OK;
  if (bytes % 4 != 0) {
    tmf::set_error("tmf_export_words: %lld bytes is not a multiple of 4", (long long)bytes);
    return TMF_E_ARG;
  }
  const int64_t words = bytes / 4;
  const int grid = (int)std::min<int64_t>((words + 255) / 256, 256);
  hipLaunchKernelGGL(export_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), (uint32_t*)host_mapped_dst, (const uint32_t*)d_src, words);
  return tmf::check_hip(hipGetLastError(), "tmf_export_words");
}

extern "C" int tmf_host_register(void* ptr, int64_t bytes) {
  if (ptr == nullptr || bytes <= 0) {
    set_error("tmf_host_register: empty range");
    return TMF_E_ARG;
  }
  return check_hip(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable), "hipHostRegister");
}

extern "C" int tmf_host_unregister(void* ptr) { return check_hip(hipHostUnregister(ptr), "hipHostUnregister"); }

extern "C" int tmf_memcpy_async(void* dst, const void* src, int64_t bytes, int to_host, void* stream) {
  if (bytes <= 0) return TMF_OK;
  return check_hip(hipMemcpyAsync(dst, src, (size_t)bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice,
                                  static_cast<hipStream_t>(stream)),
                   "hipMemcpyAsync");
}
