// Products of NESTED blocks of one matrix with a shared random block, by running sums.
//
// Every entanglement cut x of a chain needs  Y_x = A_x Omega  (A_x = C[:x,:x] or C[x:,x:], the
// block numpy.linalg.eigh diagonalises at slater.py:347) and  Y_x = F_x Omega  (F_x = C[:x,x:] or
// C[x:,:x]).  Consecutive cuts differ by one row and one column, so with Omega indexed by the
// GLOBAL orbital index
//
//      Y_x[r, c] = sum_{j < x} C[r, j] Omega[j, c]      (or the sum over j >= x)
//
// is a prefix (suffix) sum over j: all cuts together cost O(D^2 c) flops instead of O(D^3 c) for
// one GEMM per cut, and the kernel is bound by its output stores (sum_x n_x c_x elements).
//
// Mapping: one wavefront = 64 consecutive rows r x CPT columns c (accumulators in registers); it
// walks j, loads C[r, j] (coalesced: C is column-major) and Omega[j, c..c+CPT) (wave-uniform), and
// whenever a cut sits at the current j it stores its CPT accumulators into that cut's slab.
#include "common.hpp"

namespace tmf {

template <typename T, int CPT>
__global__ __launch_bounds__(256) void nested_product_kernel(const tmf_nested_desc* __restrict__ desc) {
  const tmf_nested_desc d = desc[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rb = blockIdx.x * 64, r = rb + lane;
  const int c0 = (blockIdx.y * 4 + wave) * CPT;
  if (c0 >= d.maxc) return;                       // wave-uniform
  if (!d.rows_ge && rb >= d.x_hi) return;         // rows r < x only exist below the last cut
  if (d.rows_ge && rb + 63 < d.x_lo) return;      // rows r >= x only from the first cut on
  const T* __restrict__ C = reinterpret_cast<const T*>(d.C);
  const T* __restrict__ Om = reinterpret_cast<const T*>(d.Omega);
  const uint64_t* __restrict__ dest = reinterpret_cast<const uint64_t*>(d.dest);
  const int32_t* __restrict__ ncol = reinterpret_cast<const int32_t*>(d.ncol);
  const int32_t* __restrict__ ldd = reinterpret_cast<const int32_t*>(d.ld);
  const bool rv = r < d.D;
  T acc[CPT];
#pragma unroll
  for (int t = 0; t < CPT; ++t) acc[t] = sc<T>::zero();

  // The walk over j is a chain of short steps (one coalesced load of C[:, j], CPT wave-uniform loads of Omega[j, :], CPT
  // FMAs, and the stores of the cut that sits at j).  Loads and stores return in order on this architecture, so a load
  // issued after the stores of the previous step waits for them: the operands of the NEXT group of G steps are therefore
  // fetched before the current group computes and stores (software pipeline, two register buffers).
  constexpr int G = 4;
  struct Group {
    T cj[G];
    T om[G][CPT];
  };
  auto fetch = [&](Group& g, const int j0, const int dir, const int jlim) {   // steps j0, j0 + dir, ...; beyond jlim: zeros
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int j = j0 + dir * u;
      const bool in = dir > 0 ? j < jlim : j >= jlim;
      g.cj[u] = (in && rv) ? C[(size_t)r + (size_t)j * d.ldc] : sc<T>::zero();
#pragma unroll
      for (int t = 0; t < CPT; ++t) g.om[u][t] = (in && c0 + t < d.maxc) ? Om[(size_t)j + (size_t)(c0 + t) * d.ldo] : sc<T>::zero();
    }
  };
  auto emit = [&](const int x) {
    const uint64_t dp = dest[x];
    if (!dp) return;
    T* __restrict__ out = reinterpret_cast<T*>(dp);
    const int nc = ncol[x], ld = ldd[x];
    const bool ok = rv && (d.rows_ge ? r >= x : r < x);
    const int lr = d.rows_ge ? r - x : r;
#pragma unroll
    for (int t = 0; t < CPT; ++t)
      if (ok && c0 + t < nc) out[(size_t)lr + (size_t)(c0 + t) * ld] = acc[t];
  };

  Group ga, gb;      // (two named buffers, the loop handles both per trip: an indexed pair would live in scratch memory)
  if (!d.suffix) {  // sums over j < x: cut x is complete after step j = x - 1
    int jend = d.x_hi;
    if (d.rows_ge && rb + 63 < jend) jend = rb + 63;  // rows r >= x: nothing to store once x > r
    if (d.x_lo == 0) emit(0);
    auto run = [&](const Group& g, const int j0) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int j = j0 + u;
        if (j < jend) {
#pragma unroll
          for (int t = 0; t < CPT; ++t) acc[t] = sc<T>::fmac(acc[t], g.cj[u], g.om[u][t]);
          if (j + 1 >= d.x_lo) emit(j + 1);
        }
      }
    };
    fetch(ga, 0, 1, jend);
    for (int j0 = 0; j0 < jend; j0 += 2 * G) {
      fetch(gb, j0 + G, 1, jend);
      run(ga, j0);
      fetch(ga, j0 + 2 * G, 1, jend);
      run(gb, j0 + G);
    }
  } else {          // sums over j >= x: cut x is complete after step j = x
    int jbeg = d.x_lo;
    if (!d.rows_ge && rb + 1 > jbeg) jbeg = rb + 1;   // rows r < x: nothing to store once x <= r
    if (d.x_hi >= d.D) emit(d.D);
    auto run = [&](const Group& g, const int j0) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int j = j0 - u;
        if (j >= jbeg) {
#pragma unroll
          for (int t = 0; t < CPT; ++t) acc[t] = sc<T>::fmac(acc[t], g.cj[u], g.om[u][t]);
          if (j <= d.x_hi) emit(j);
        }
      }
    };
    fetch(ga, d.D - 1, -1, jbeg);
    for (int j0 = d.D - 1; j0 >= jbeg; j0 -= 2 * G) {
      fetch(gb, j0 - G, -1, jbeg);
      run(ga, j0);
      fetch(ga, j0 - 2 * G, -1, jbeg);
      run(gb, j0 - G);
    }
  }
}

}  // namespace tmf

extern "C" int tmf_nested_products_batched(int dtype, const tmf_nested_desc* d_desc, int ndesc, int D, int maxc,
                                           void* stream) {
  using namespace tmf;
  if (ndesc <= 0 || D <= 0 || maxc <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128) {
    constexpr int CPT = 4;
    dim3 g((D + 63) / 64, (maxc + 4 * CPT - 1) / (4 * CPT), ndesc);
    hipLaunchKernelGGL((nested_product_kernel<cd, CPT>), g, dim3(256), 0, s, d_desc);
  } else if (dtype == TMF_F64) {
    constexpr int CPT = 8;
    dim3 g((D + 63) / 64, (maxc + 4 * CPT - 1) / (4 * CPT), ndesc);
    hipLaunchKernelGGL((nested_product_kernel<double, CPT>), g, dim3(256), 0, s, d_desc);
  } else {
    set_error("tmf_nested_products_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_nested_products_batched");
}
