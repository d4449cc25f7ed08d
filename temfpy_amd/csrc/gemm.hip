// Batched variable-size GEMM on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
//   C[p] = alpha * op(A[p]) * B[p] + beta * C[p]       op = identity | conjugate-transpose
//
// One 256-thread workgroup (4 waves) owns one 64 x TN tile of one problem; a host-built
// tile table maps blockIdx -> (problem, tile_m, tile_n) so that thousands of problems of
// different shapes (one per entanglement cut / site) fill the 256 CUs in one launch.
//
// Replaces the dense products of the reference: slater.py:1071 (O = v_bra^H v_ket),
// :1080/:1087 and the GEMM-shaped share of the block diagonalisation (slater.py:347).
//
// MFMA operand maps (cdna_hip_programming.md section 3, "f64 MFMA does NOT use these maps"):
//   a-operand lane l : X[i = l & 15][k = l >> 4]        b-operand : Y[k = l >> 4][j = l & 15]
//   result reg r     : D[row = (l >> 4) + 4 r][col = l & 15]
// We feed a := B-values (i -> n) and b := A-values (j -> m), so that D[n][m] puts
// consecutive lanes on consecutive m: column-major C is then written in 128/256-byte runs.
#include "common.hpp"

namespace tmf {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64;
constexpr int KT = 16;
constexpr int LDT = KT + 1;  // padded k-stride of the LDS tiles (doubles)

template <typename T, int OPA, int TN, bool M4 = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const tmf_gemm_desc* __restrict__ desc,
                                                   const int32_t* __restrict__ tiles, double alpha, double beta,
                                                   const int32_t* __restrict__ run_if) {
  if (run_if != nullptr && *run_if == 0) return;   // conditional-launch scope (tmf_launch_condition)
  constexpr int CP = sc<T>::cplx;
  constexpr int NP = CP ? 2 : 1;               // planes (re, im)
  constexpr int WM = (TN == 64) ? 32 : 16;     // rows of C per wave
  constexpr int MI = WM / 16;
  constexpr int NI = (TN == 64) ? 2 : 1;

  __shared__ double As[NP][TM][LDT];
  __shared__ double Bs[NP][TN][LDT];

  const int4 tl = reinterpret_cast<const int4*>(tiles)[blockIdx.x];
  const tmf_gemm_desc d = desc[tl.x];
  const int m0 = tl.y * TM, n0 = tl.z * TN;
  const T* __restrict__ A = reinterpret_cast<const T*>(d.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(d.B);
  T* __restrict__ C = reinterpret_cast<T*>(d.C);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (TN == 64) ? (wave >> 1) * 32 : wave * 16;
  const int wn0 = (TN == 64) ? (wave & 1) * 32 : 0;
  const int l15 = lane & 15, l4 = lane >> 4;

  // Complex products by the 3M scheme: P1 = sum ar br, P2 = sum ai bi, P3 = sum (ar + ai)(br + bi);  Re = P1 - P2,
  // Im = P3 - P1 - P2.  Three real MFMAs per complex multiply-add instead of four: the fp64 MFMA pipe is the bound of this
  // kernel (PMC: busy 77 % of the time with four), its rate on this part is 32 flop / cycle / SIMD, and two extra
  // v_add_f64 per fragment are free next to a 64-cycle MFMA.  Error bound: normwise the same as the 4M form
  // (eps * sum |a||b| with a constant of 4 instead of 2 on the imaginary part); every product of the sweep is followed
  // by an orthogonalisation or a difference of O(1) quantities, where the normwise bound is the one that matters.
  // M4 (tmf_gemm_set_4m, the A/B switch of tests/test_gpu_fullsize.py): the four-product form, componentwise error bound.
  constexpr int NACC = CP ? (M4 ? 2 : 3) : 1;
  d4 acc[NACC][MI][NI];
#pragma unroll
  for (int p = 0; p < NACC; ++p)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[p][i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  // Software pipeline: the global loads of tile k+1 are issued into registers before the MFMAs of
  // tile k, so that one memory latency is hidden behind the other's arithmetic (and behind the other
  // workgroups of the CU) instead of being exposed once per 16 rows of K.
  constexpr int NA = TM * KT / 256, NB = (TN * KT + 255) / 256;
  T ra[NA], rb[NB];
  auto load_regs = [&](const int k0) {
    if (OPA == 0) {
      const int i = tid & 63;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int kk = (tid >> 6) + 4 * q;
        ra[q] = (m0 + i < d.M && k0 + kk < d.K) ? A[(size_t)(m0 + i) + (size_t)(k0 + kk) * d.lda] : sc<T>::zero();
      }
    } else {
      const int kk = tid & 15;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int i = (tid >> 4) + 16 * q;
        ra[q] = (m0 + i < d.M && k0 + kk < d.K) ? A[(size_t)(k0 + kk) + (size_t)(m0 + i) * d.lda] : sc<T>::zero();
      }
    }
    const int kk = tid & 15;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int j = (tid >> 4) + 16 * q;
      rb[q] = (j < TN && n0 + j < d.N && k0 + kk < d.K) ? B[(size_t)(k0 + kk) + (size_t)(n0 + j) * d.ldb] : sc<T>::zero();
    }
  };
  auto store_regs = [&]() {
    if (OPA == 0) {
      const int i = tid & 63;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int kk = (tid >> 6) + 4 * q;
        if constexpr (CP) {
          As[0][i][kk] = ra[q].x;
          As[1][i][kk] = ra[q].y;
        } else {
          As[0][i][kk] = ra[q];
        }
      }
    } else {
      const int kk = tid & 15;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int i = (tid >> 4) + 16 * q;
        if constexpr (CP) {
          As[0][i][kk] = ra[q].x;
          As[1][i][kk] = -ra[q].y;  // conjugate
        } else {
          As[0][i][kk] = ra[q];
        }
      }
    }
    const int kk = tid & 15;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int j = (tid >> 4) + 16 * q;
      if (j < TN) {
        if constexpr (CP) {
          Bs[0][j][kk] = rb[q].x;
          Bs[1][j][kk] = rb[q].y;
        } else {
          Bs[0][j][kk] = rb[q];
        }
      }
    }
  };

  if (d.K > 0) {
    load_regs(0);
    store_regs();
  }
  __syncthreads();
  for (int k0 = 0; k0 < d.K; k0 += KT) {
    const bool more = k0 + KT < d.K;
    if (more) load_regs(k0 + KT);
#pragma unroll
    for (int kc = 0; kc < KT; kc += 4) {
      double ar[MI], ai[MI], br[NI], bi[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        ar[i] = As[0][wm0 + i * 16 + l15][kc + l4];
        if constexpr (CP) ai[i] = As[1][wm0 + i * 16 + l15][kc + l4];
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        br[j] = Bs[0][wn0 + j * 16 + l15][kc + l4];
        if constexpr (CP) bi[j] = Bs[1][wn0 + j * 16 + l15][kc + l4];
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[j], ar[i], acc[0][i][j], 0, 0, 0);
          if constexpr (CP && M4) {
            acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bi[j], ai[i], acc[0][i][j], 0, 0, 0);
            acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[j], ar[i], acc[1][i][j], 0, 0, 0);
            acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[j], ai[i], acc[1][i][j], 0, 0, 0);
          } else if constexpr (CP) {
            acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[j], ai[i], acc[1][i][j], 0, 0, 0);
            acc[2][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[j] + bi[j], ar[i] + ai[i], acc[2][i][j], 0, 0, 0);
          }
        }
    }
    __syncthreads();
    if (more) {
      store_regs();
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds C[m = l15][n = l4 + 4 r] of each 16 x 16 sub-tile -----------
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm0 + i * 16 + l15;
        const int n = n0 + wn0 + j * 16 + l4 + 4 * r;
        if (m < d.M && n < d.N) {
          T* c = C + (size_t)m + (size_t)n * d.ldc;
          if constexpr (CP) {
            double vre, vim;
            if constexpr (M4) {
              vre = acc[0][i][j][r], vim = acc[1][i][j][r];
            } else {
              const double p1 = acc[0][i][j][r], p2 = acc[1][i][j][r], p3 = acc[NACC - 1][i][j][r];
              vre = p1 - p2, vim = p3 - p1 - p2;
            }
            cd v = make_cd(alpha * vre, alpha * vim);
            if (beta != 0.0) {
              cd o = *c;
              v.x = fma(beta, o.x, v.x);
              v.y = fma(beta, o.y, v.y);
            }
            *c = v;
          } else {
            double v = alpha * acc[0][i][j][r];
            if (beta != 0.0) v = fma(beta, *c, v);
            *c = v;
          }
        }
      }
}

// ---------------------------------------------------------------------------------------------
// Tall-skinny variant:  C (M x N, N <= 16) = alpha * A^H B + beta * C  with a long contraction
// (K = rows of an orbital slab) and few output columns - the coefficient products Q^H P of the blocked
// Gram-Schmidt.  The general kernel above advances 16 rows per iteration and exposes one memory
// latency per iteration (measured 2.9 us x K/16, and its 64-column A tile is mostly padding when
// M < 64); here a workgroup owns a 16 x 16 output tile and stages 64 rows per iteration (1 KiB
// contiguous per column), the four waves split the 64 rows between them and their partial tiles are
// summed in a fixed order at the end (deterministic).
// ---------------------------------------------------------------------------------------------
constexpr int TK = 64;
constexpr int LDK = TK + 1;

template <typename T>
__global__ __launch_bounds__(256) void gemm_tall_kernel(const tmf_gemm_desc* __restrict__ desc,
                                                        const int32_t* __restrict__ tiles, double alpha, double beta) {
  constexpr int CP = sc<T>::cplx;
  constexpr int NP = CP ? 2 : 1;
  __shared__ double As[NP][16][LDK];
  __shared__ double Bs[NP][16][LDK];
  __shared__ double red[4][NP][4][64];

  const int4 tl = reinterpret_cast<const int4*>(tiles)[blockIdx.x];
  const tmf_gemm_desc d = desc[tl.x];
  const int m0 = tl.y * 16;
  const T* __restrict__ A = reinterpret_cast<const T*>(d.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(d.B);
  T* __restrict__ C = reinterpret_cast<T*>(d.C);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;

  d4 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) acc[p] = (d4){0.0, 0.0, 0.0, 0.0};

  for (int k0 = 0; k0 < d.K; k0 += TK) {
    const int kk = lane;  // row inside the chunk; wave = column group
#pragma unroll
    for (int c = wave; c < 16; c += 4) {
      T va = sc<T>::zero(), vb = sc<T>::zero();
      if (k0 + kk < d.K) {
        if (m0 + c < d.M) va = A[(size_t)(k0 + kk) + (size_t)(m0 + c) * d.lda];
        if (c < d.N) vb = B[(size_t)(k0 + kk) + (size_t)c * d.ldb];
      }
      if constexpr (CP) {
        As[0][c][kk] = va.x, As[1][c][kk] = -va.y;  // conjugate
        Bs[0][c][kk] = vb.x, Bs[1][c][kk] = vb.y;
      } else {
        As[0][c][kk] = va;
        Bs[0][c][kk] = vb;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kc = wave * 16; kc < wave * 16 + 16; kc += 4) {
      const double ar = As[0][l15][kc + l4], br = Bs[0][l15][kc + l4];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(br, ar, acc[0], 0, 0, 0);
      if constexpr (CP) {
        const double ai = As[1][l15][kc + l4], bi = Bs[1][l15][kc + l4];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bi, ai, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi, ar, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(br, ai, acc[1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][p][r][lane] = acc[p][r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + l15, n = l4 + 4 * r;   // lane holds C[m = l15][n = l4 + 4 r]
      double vr = ((red[0][0][r][lane] + red[1][0][r][lane]) + red[2][0][r][lane]) + red[3][0][r][lane];
      double vi = 0.0;
      if constexpr (CP) vi = ((red[0][1][r][lane] + red[1][1][r][lane]) + red[2][1][r][lane]) + red[3][1][r][lane];
      if (m < d.M && n < d.N) {
        T* c = C + (size_t)m + (size_t)n * d.ldc;
        if constexpr (CP) {
          cd v = make_cd(alpha * vr, alpha * vi);
          if (beta != 0.0) {
            const cd o = *c;
            v.x = fma(beta, o.x, v.x);
            v.y = fma(beta, o.y, v.y);
          }
          *c = v;
        } else {
          double v = alpha * vr;
          if (beta != 0.0) v = fma(beta, *c, v);
          *c = v;
        }
      }
    }
  }
}

static bool g_four_products = false;

template <typename T>
static int launch(int opA, double alpha, double beta, const tmf_gemm_desc* d, const int32_t* t, int nt, int tile_n,
                  hipStream_t s) {
  dim3 g(nt), b(256);
  if (g_four_products && tile_n == 64) {
    if (opA) hipLaunchKernelGGL((gemm_kernel<T, 1, 64, true>), g, b, 0, s, d, t, alpha, beta, launch_condition());
    else hipLaunchKernelGGL((gemm_kernel<T, 0, 64, true>), g, b, 0, s, d, t, alpha, beta, launch_condition());
  } else if (g_four_products) {
    if (opA) hipLaunchKernelGGL((gemm_kernel<T, 1, 16, true>), g, b, 0, s, d, t, alpha, beta, launch_condition());
    else hipLaunchKernelGGL((gemm_kernel<T, 0, 16, true>), g, b, 0, s, d, t, alpha, beta, launch_condition());
  } else if (tile_n == 64) {
    if (opA) hipLaunchKernelGGL((gemm_kernel<T, 1, 64>), g, b, 0, s, d, t, alpha, beta, launch_condition());
    else hipLaunchKernelGGL((gemm_kernel<T, 0, 64>), g, b, 0, s, d, t, alpha, beta, launch_condition());
  } else {
    if (opA) hipLaunchKernelGGL((gemm_kernel<T, 1, 16>), g, b, 0, s, d, t, alpha, beta, launch_condition());
    else hipLaunchKernelGGL((gemm_kernel<T, 0, 16>), g, b, 0, s, d, t, alpha, beta, launch_condition());
  }
  return check_hip(hipGetLastError(), "tmf_gemm_batched launch");
}

}  // namespace tmf

// A/B switch: complex products of tmf_gemm_batched by four real MFMAs (componentwise error bound) instead of three.
extern "C" void tmf_gemm_set_4m(int on) { tmf::g_four_products = on != 0; }

extern "C" int tmf_gemm_batched(int dtype, int opA, double alpha, double beta, const tmf_gemm_desc* d_desc,
                                const int32_t* d_tiles, int ntiles, int tile_n, void* stream) {
  if (ntiles <= 0) return TMF_OK;
  if (tile_n != 64 && tile_n != 16) {
    tmf::set_error("tmf_gemm_batched: tile_n must be 64 or 16, got %d", tile_n);
    return TMF_E_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128) return tmf::launch<tmf::cd>(opA, alpha, beta, d_desc, d_tiles, ntiles, tile_n, s);
  if (dtype == TMF_F64) return tmf::launch<double>(opA, alpha, beta, d_desc, d_tiles, ntiles, tile_n, s);
  tmf::set_error("tmf_gemm_batched: bad dtype %d", dtype);
  return TMF_E_ARG;
}

extern "C" int tmf_gemm_tall_batched(int dtype, double alpha, double beta, const tmf_gemm_desc* d_desc,
                                     const int32_t* d_tiles, int ntiles, void* stream) {
  if (ntiles <= 0) return TMF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 g(ntiles), b(256);
  if (dtype == TMF_C128) hipLaunchKernelGGL((tmf::gemm_tall_kernel<tmf::cd>), g, b, 0, s, d_desc, d_tiles, alpha, beta);
  else if (dtype == TMF_F64) hipLaunchKernelGGL((tmf::gemm_tall_kernel<double>), g, b, 0, s, d_desc, d_tiles, alpha, beta);
  else {
    tmf::set_error("tmf_gemm_tall_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return tmf::check_hip(hipGetLastError(), "tmf_gemm_tall_batched launch");
}
