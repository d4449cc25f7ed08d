// Batched gathered Pfaffians -- the hot kernel of the BCS / Pfaffian path.
//
// Reference: pfaffian.py:1429-1479 (`_tensor_block`) gathers, for every pair (bra set a, ket set b)
// of a block, the skew-symmetric sub-matrix N[idx, idx] with idx = [ket positions (n2), bra
// positions (n1)] and calls pfapack's Pfaffian on each matrix in a Python loop
// (pfaffian.py:1413-1426).
//
// Here the Pfaffian matrix N of a site (<= 64 KiB) and the index lists are staged in LDS; G = 8 / 16 /
// 32 lanes own one sub-matrix of even order m = n1 + n2, lane c holding column c in registers.
// Pfaffian by pivot-pair elimination: for the lowest unused index p pick q = argmax |A[p, q]|, then
//      Pf(A) = (-1)^(q' - 1) A[p, q] Pf(A'),   A'[r, c] = A[r, c] + (A[r,p] A[q,c] - A[r,q] A[p,c]) / A[p,q]
// (q' = position of q among the unused indices).  The two pivot columns go through an LDS scratch;
// the row entries A[q, c], A[p, c] a lane needs are -col_q[c] and its own register, so no register
// array is indexed at run time.  One straight-line kernel per even order m <= 32.
#include "det_common.hpp"

namespace tmf {

template <typename T, int M, int G>
__device__ inline T pf_group(T (&a)[M], const int c, T* __restrict__ scratch) {
  T pf = sc<T>::one();
  unsigned used = 0u;
  bool mine_used = c >= M;
  constexpr unsigned valid = (M >= 32) ? 0xffffffffu : ((1u << M) - 1u);
  T* colp = scratch;
  T* colq = scratch + M;
#pragma unroll
  for (int p = 0; p < M - 1; ++p) {
    if (!((used >> p) & 1u)) {  // uniform inside a group, may differ between the groups of a wave
      unsigned key = 0u;
      if (!mine_used && c != p) {
        const float v = (float)sc<T>::abs2(a[p]);
        key = ((__float_as_uint(v) >> 1) & ~(unsigned)(G - 1)) | (unsigned)c | 0x80000000u;
      }
      key = group_max<G>(key);
      const int q = (int)(key & (unsigned)(G - 1));
      if (c == p) {
#pragma unroll
        for (int r = 0; r < M; ++r) colp[r] = a[r];
      }
      if (c == q) {
#pragma unroll
        for (int r = 0; r < M; ++r) colq[r] = a[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const T piv = colq[p];  // A[p, q]
      pf = sc<T>::mul(pf, piv);
      const int qpos = __popc(~used & ((1u << q) - 1u) & valid);  // counts p as well
      if (!(qpos & 1)) pf = sc<T>::neg(pf);                       // (-1)^(q' - 1)
      used |= (1u << p) | (1u << q);
      mine_used = mine_used || c == p || c == q;
      const T pinv = sc<T>::abs2(piv) > 0.0 ? sc<T>::inv_fast(piv) : sc<T>::zero();
      // A[q, c] = -A[c, q] = -colq[c] ; A[p, c] = own register a[p]
      const int cc = c < M ? c : 0;
      const T aq = sc<T>::mul(sc<T>::neg(colq[cc]), pinv);
      const T ap = sc<T>::mul(a[p], pinv);
#pragma unroll
      for (int r = 0; r < M; ++r) {
        T x = sc<T>::fmac(a[r], colp[r], aq);  // + A[r,p] A[q,c] / piv
        a[r] = sc<T>::fms(x, colq[r], ap);     // - A[r,q] A[p,c] / piv
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  return pf;
}

// LDS (dynamic): [ N : nn*nn T ][ ket idx : nsk*n2 u8 ][ bra idx : na*n1 u8 ][ per group: 2*M T ]
template <typename T, int M, int G>
__global__ __launch_bounds__(256) void pf_kernel(const tmf_pf_desc* __restrict__ desc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_pf_desc d = desc[blockIdx.x];
  const int na = d.a1 - d.a0, n1 = d.n1, n2 = d.n2, nn = d.nn;
  T* Ns = reinterpret_cast<T*>(smem);
  size_t off = ((size_t)nn * nn * sizeof(T) + 15) & ~(size_t)15;
  uint8_t* kidx = smem + off;
  off += ((size_t)d.nsk * n2 + 15) & ~(size_t)15;
  uint8_t* bidx = smem + off;
  off += ((size_t)na * n1 + 15) & ~(size_t)15;
  T* scratch = reinterpret_cast<T*>(smem + off) + (size_t)(threadIdx.x / G) * (2 * M);

  const T* __restrict__ Ng = reinterpret_cast<const T*>(d.N);
  for (int e = threadIdx.x; e < nn * nn; e += 256) Ns[e] = Ng[(size_t)(e % nn) + (size_t)(e / nn) * d.ldn];
  const uint8_t* __restrict__ gk = reinterpret_cast<const uint8_t*>(d.ket_idx);
  const uint8_t* __restrict__ gb = reinterpret_cast<const uint8_t*>(d.bra_idx) + (size_t)d.a0 * n1;
  for (int e = threadIdx.x; e < d.nsk * n2; e += 256) kidx[e] = gk[e];
  for (int e = threadIdx.x; e < na * n1; e += 256) bidx[e] = gb[e];
  __syncthreads();

  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  constexpr int NG = 256 / G;
  const int grp = threadIdx.x / G, c = threadIdx.x % G;
  const int npairs = na * d.nsk;
  const int iters = (npairs + NG - 1) / NG;
  for (int it = 0; it < iters; ++it) {
    const int pidx = it * NG + grp;
    const bool live = pidx < npairs;
    const int al = live ? pidx / d.nsk : 0, b = live ? pidx % d.nsk : 0;
    // index of sub-matrix position r: ket positions first, then bra positions (pfaffian.py:1468-1473)
    auto pos = [&](int r) -> int { return r < n2 ? kidx[b * n2 + r] : bidx[al * n1 + (r - n2)]; };
    T a[M];
    const int mycol = c < M ? pos(c) : 0;
#pragma unroll
    for (int r = 0; r < M; ++r) a[r] = (c < M) ? Ns[pos(r) + mycol * nn] : sc<T>::zero();
    const T pf = pf_group<T, M, G>(a, c, scratch);
    if (live && c == 0) out[(size_t)(d.a0 + al) * d.nsk + b] = sc<T>::mul(scale, pf);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pf_fill_kernel(const tmf_pf_desc* __restrict__ desc) {
  const tmf_pf_desc d = desc[blockIdx.x];
  const T scale = *reinterpret_cast<const T*>(d.scale);
  T* __restrict__ out = reinterpret_cast<T*>(d.out) + (size_t)d.a0 * d.nsk;
  for (int e = threadIdx.x; e < (d.a1 - d.a0) * d.nsk; e += 256) out[e] = scale;
}

template <typename T, int M>
static void launch_pf_exact(dim3 g, int lds, hipStream_t s, const tmf_pf_desc* d) {
  constexpr int G = M <= 8 ? 8 : (M <= 16 ? 16 : 32);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)pf_kernel<T, M, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((pf_kernel<T, M, G>), g, dim3(256), lds, s, d);
}

template <typename T>
static int launch_pf(int m, const tmf_pf_desc* d, int nt, int lds, hipStream_t s) {
  dim3 g(nt);
  switch (m) {
    case 0: hipLaunchKernelGGL((pf_fill_kernel<T>), g, dim3(256), 0, s, d); break;
#define TMF_CASE(M) case M: launch_pf_exact<T, M>(g, lds, s, d); break;
    TMF_CASE(2) TMF_CASE(4) TMF_CASE(6) TMF_CASE(8) TMF_CASE(10) TMF_CASE(12) TMF_CASE(14) TMF_CASE(16)
    TMF_CASE(18) TMF_CASE(20) TMF_CASE(22) TMF_CASE(24) TMF_CASE(26) TMF_CASE(28) TMF_CASE(30) TMF_CASE(32)
#undef TMF_CASE
    default:
      set_error("tmf_pf_gather_batched: order must be even and <= 32, got %d", m);
      return (m % 2) ? TMF_E_ARG : TMF_E_LIMIT;
  }
  return check_hip(hipGetLastError(), "tmf_pf_gather_batched launch");
}

}  // namespace tmf

extern "C" int tmf_pf_gather_batched(int dtype, int order, const tmf_pf_desc* d_desc, int ntiles, int lds_bytes,
                                     void* stream) {
  if (ntiles <= 0) return TMF_OK;
  if (lds_bytes < 0 || lds_bytes > 160 * 1024) {
    tmf::set_error("tmf_pf_gather_batched: lds_bytes %d exceeds the 160 KiB LDS of a CU", lds_bytes);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128) return tmf::launch_pf<tmf::cd>(order, d_desc, ntiles, lds_bytes, s);
  if (dtype == TMF_F64) return tmf::launch_pf<double>(order, d_desc, ntiles, lds_bytes, s);
  tmf::set_error("tmf_pf_gather_batched: bad dtype %d", dtype);
  return TMF_E_ARG;
}
