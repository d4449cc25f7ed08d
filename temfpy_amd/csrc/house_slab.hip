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
#include "common.hpp"

namespace tmf {

template <typename T>
__device__ inline T wsum(T v) {      // DPP path: the butterfly through ds_bpermute cost 24 dependent LDS-crossbar hops per complex sum
  return wave_sum64(v);
}

constexpr int VB = 8;   // reflectors per LDS block

template <typename T, int RMAX>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, (RMAX * sizeof(T) <= 128) ? 8 : 4)))
void house_slab_kernel(const tmf_slab_desc* __restrict__ desc, int w) {
  extern __shared__ __align__(16) unsigned char smem[];
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
    const int kprev = p0 < K ? p0 : K;
    for (int kb = 0; kb < kprev; kb += VB) {      // reflectors of the earlier panels, VB at a time
      const int nb = (kprev - kb < VB) ? kprev - kb : VB;
      __syncthreads();
      load_block(kb, nb);
      __syncthreads();
      if (mine)
        for (int j = 0; j < nb; ++j) apply(kb + j, sc<T>::conj(taus[kb + j]), VBLK + (size_t)j * n);
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
    if (mine) {
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int r = lane + 64 * i;
        if (r < n) a[r] = col[i];
      }
    }
    __syncthreads();
  }
  // ---------------- R (c x c, zero rows beyond K), optionally as R^H ----------------
  if (R)
    for (int e = tid; e < c * c; e += NT) {
      const int r = e % c, cc = e / c;
      const T v = (r <= cc && r < K) ? A[r + (size_t)cc * lda] : sc<T>::zero();
      if (d.flags & 1) R[cc + (size_t)r * d.ldr] = sc<T>::conj(v);
      else R[r + (size_t)cc * d.ldr] = v;
    }
  if (d.flags & 4) return;      // only R wanted: no Q at all (A is left holding the reflectors)
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
      if (mine)
        for (int j = nb - 1; j >= 0; --j)
          if (kb + j <= jc) apply(kb + j, taus[kb + j], VBLK + (size_t)j * n);      // H_k e_j = e_j for k > j
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

}  // namespace tmf

extern "C" int tmf_house_slab_batched(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream) {
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
  const dim3 g(nprob), b(64 * w);
#define TMF_SLAB_LAUNCH(T, RM) hipLaunchKernelGGL((house_slab_kernel<T, RM>), g, b, lds, s, d_desc, w)
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
