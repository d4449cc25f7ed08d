// Self-check of the Schmidt decomposition (testing.py:131-177): the largest deviation between a
// block of the correlation matrix and its reconstruction from the computed orbitals,
//
//   mode 0:  dev = max_{r,c} | T[r,c] - sum_j X[r,j] w[j] conj(Y[c,j']) |,  j' = j or q-1-j
//            ("vL does not diagonalise C_LL", "vL and vR do not SVD C_LR")
//   mode 1:  dev = max_{r,c} | T[r,c] - sum_i conj(X[i,r]) Y[i,c] |,  T = identity when its address is 0
//            ("vL is not unitary", evaluated as V^H V on the kept columns)
//
// Nothing is materialised: a 64 x 64 tile per workgroup, 4 x 4 outputs per thread (rows ty + 16 a, columns tx + 16 b), operands staged
// through LDS in slices of 8, the tile maximum folded into the problem's result with an atomic max
// on the (non-negative) bit pattern.
#include "common.hpp"

namespace tmf {

constexpr int RT = 64, RK = 8;

template <typename T>
__global__ __launch_bounds__(256) void recon_error_kernel(const tmf_recon_desc* __restrict__ desc,
                                                          const int32_t* __restrict__ tiles) {
  const int prob = tiles[3 * blockIdx.x], tr = tiles[3 * blockIdx.x + 1], tc = tiles[3 * blockIdx.x + 2];
  const tmf_recon_desc d = desc[prob];
  const T* __restrict__ X = reinterpret_cast<const T*>(d.X);
  const T* __restrict__ Y = reinterpret_cast<const T*>(d.Y);
  const T* __restrict__ Tg = reinterpret_cast<const T*>(d.T);
  const double* __restrict__ w = reinterpret_cast<const double*>(d.w);
  __shared__ T Xs[RK][RT + 1];
  __shared__ T Ys[RK][RT + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int r0 = tr * RT, c0 = tc * RT;
  T acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = sc<T>::zero();

  const int K = d.mode == 0 ? d.q : d.inner;  // contraction length
  for (int k0 = 0; k0 < K; k0 += RK) {
    // stage: Xs[kk][i] = factor of output row r0+i, Ys[kk][i] = factor of output column c0+i
    for (int e = tid; e < RK * RT; e += 256) {
      T xv = sc<T>::zero(), yv = sc<T>::zero();
      if (d.mode == 0) {
        const int i = e % RT, kk = e / RT, j = k0 + kk;
        if (j < K) {
          const int jy = d.y_reverse ? d.q - 1 - j : j;
          if (r0 + i < d.rows) xv = sc<T>::scale(X[(size_t)(r0 + i) + (size_t)j * d.ldx], w ? w[j] : 1.0);
          if (c0 + i < d.cols) yv = sc<T>::conj(Y[(size_t)(c0 + i) + (size_t)jy * d.ldy]);
        }
        Xs[kk][i] = xv;
        Ys[kk][i] = yv;
      } else {
        const int kk = e % RK, i = e / RK, j = k0 + kk;  // contraction index runs along the columns' rows
        if (j < K) {
          if (r0 + i < d.rows) xv = sc<T>::conj(X[(size_t)j + (size_t)(r0 + i) * d.ldx]);
          if (c0 + i < d.cols) yv = Y[(size_t)j + (size_t)(c0 + i) * d.ldy];
        }
        Xs[kk][i] = xv;
        Ys[kk][i] = yv;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < RK; ++kk) {
      T xa[4], yb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xa[a] = Xs[kk][ty + 16 * a];
#pragma unroll
      for (int b = 0; b < 4; ++b) yb[b] = Ys[kk][tx + 16 * b];      // (lanes on consecutive elements: tx * 4 + b cost 4.6 bank conflicts per LDS cycle)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = sc<T>::fmac(acc[a][b], xa[a], yb[b]);
    }
    __syncthreads();
  }
  double mx = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int r = r0 + ty + 16 * a, c = c0 + tx + 16 * b;
      if (r < d.rows && c < d.cols) {
        T t = Tg ? Tg[(size_t)r + (size_t)c * d.ldt] : ((r == c) ? sc<T>::one() : sc<T>::zero());
        const double v = sqrt(sc<T>::abs2(sc<T>::sub(t, acc[a][b])));
        mx = v > mx ? v : mx;  // (a NaN deviation is reported as such below)
        if (v != v) mx = v;
      }
    }
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_xor(mx, o);
    mx = (other > mx || other != other) ? other : mx;
  }
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    double m = red[0];
    for (int i = 1; i < 4; ++i) m = (red[i] > m || red[i] != red[i]) ? red[i] : m;
    if (m != m) m = __longlong_as_double(0x7ff0000000000000ll);  // NaN -> +inf: always reported
    atomicMax(reinterpret_cast<unsigned long long*>(d.out), (unsigned long long)__double_as_longlong(m));
  }
}

}  // namespace tmf

extern "C" int tmf_recon_error_batched(int dtype, const tmf_recon_desc* d_desc, const int32_t* d_tiles, int ntiles,
                                       void* stream) {
  using namespace tmf;
  if (ntiles <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(recon_error_kernel<cd>, dim3(ntiles), dim3(256), 0, s, d_desc, d_tiles);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(recon_error_kernel<double>, dim3(ntiles), dim3(256), 0, s, d_desc, d_tiles);
  else {
    set_error("tmf_recon_error_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_recon_error_batched");
}
