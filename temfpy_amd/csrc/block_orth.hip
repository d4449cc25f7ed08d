// Orthonormalisation of one 64-column block of every orbital slab in ONE launch.
//
// tmf_bcgs_batched (bcgs.hip) orthonormalises the filled-orbital bases (the orthonormal eigenvector blocks that
// numpy.linalg.eigh returns at slater.py:347) block by block: a 64-column block is first projected against everything
// before it by two batched MFMA GEMMs, and then orthonormalised in itself - four 16-column panels, each projected against
// the earlier panels of the block and put through Cholesky-QR twice.  As separate launches that inner part is a chain of
// ~26 short dependent kernels per block (descriptors, coefficients, update, Gram, Cholesky + apply, Gram, Cholesky +
// apply per panel), 8 - 30 us each however few slabs there are: it was the largest item of a multi-GPU shard's fixed cost
// (2.3 of 7.5 ms for 38 sites, profiles/r02/shard_8way_rank3_kernel_stats.csv).  Here one workgroup owns the block of one
// slab for the whole chain.
//
// Row wavefronts.  Wavefront w < ceil(n / 64) owns rows 64 w .. 64 w + 63.  Its 64 x 16 piece of the current panel lives
// in registers in the layout the fp64 MFMA (v_mfma_f64_16x16x4_f64) both consumes as an operand of a product that contracts
// over ROWS and produces as a result:
//
//     register m = 4 g + r of lane (l4 = lane >> 4, l15 = lane & 15)   <->   row 16 g + 4 r + l4, column l15
//
//   Gram matrix / coefficients  C = Q^H X  (contraction over the rows): a-operand conj(Q), b-operand X, both in that
//       layout, 16 k-steps of 4 rows; the 16 x 16 partial results of the wavefronts are summed through LDS in a fixed
//       order (deterministic: a shard of a multi-GPU conversion and the whole chain agree bit for bit).
//   update X -= Q C / apply X <- X R^-1  (contraction over the 16 columns): the a-operand needs the other layout
//       (lane <-> row 16 g + l15, column 4 s + l4).  Earlier panels are read from global memory in that layout directly;
//       the register-resident panel is turned through a 16 x 16 LDS tile per wavefront, one row group at a time.  The
//       result lands in the first layout again, so the panel never leaves the registers between its first load and its
//       final store.
//
// Helper wavefront.  The wavefront behind the last row wavefront holds no rows; it factors the 16 x 16 Gram matrices:
// Cholesky factor and the inverse of its conjugate transpose in ONE right-looking elimination of the augmented matrix
// [G | 1] (L^-1 [G | 1] = [R | L^-1], R^-1 = (L^-1)^H), the 16 x 32 matrix in registers (lane -> row a, columns cg + 4 c),
// the pivot row through LDS once per step.  In a row wavefront the same code spills: the panel takes 64 of the 128
// registers a lane has at 16 wavefronts per workgroup, and scratch reloads inside the 16 dependent steps cost 30 us per
// factorisation (measured).  Same rules as cholqr_apply_kernel: a pivot below 1e-13 of its diagonal entry, or below 1e-14
// of the column's norm before any projection, gives a ZERO column.
#include <stdlib.h>

#include "common.hpp"

namespace tmf {

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T>
struct pl;   // planes of an element
template <>
struct pl<double> {
  __device__ static inline double re(double a) { return a; }
  __device__ static inline double im(double) { return 0.0; }
};
template <>
struct pl<cd> {
  __device__ static inline double re(cd a) { return a.x; }
  __device__ static inline double im(cd a) { return a.y; }
};

constexpr int SCR_LD = 17;                 // elements per row of a wavefront's 16 x 16 turning tile

template <typename T>
struct BlockShared {
  static constexpr int NP = sc<T>::cplx ? 2 : 1;
  static constexpr int WV_BYTES = 16 * SCR_LD * (int)sizeof(T);
  // per row wavefront: its turning tile (16 x 17 elements); the partial sums of a reduction (NP x 4 x 64 doubles) share it
  unsigned char wv[15][WV_BYTES];
  double C[NP][16][17];                    // reduced 16 x 16 matrix (Gram matrix or coefficients), then R^-1
  T row[2][32];                            // pivot row of the elimination (double buffered)
  double diag[16], raw[16];
};

}  // namespace

template <typename T>
__global__ __launch_bounds__(1024) void block_orth_kernel(const tmf_bcgs_desc* __restrict__ desc, const int t,
                                                          unsigned long long* __restrict__ dbg) {
  constexpr int CP = sc<T>::cplx;
  constexpr int NP = CP ? 2 : 1;
  static_assert(NP * 4 * 64 * 8 <= BlockShared<T>::WV_BYTES, "partial sums must fit the wavefront's own tile");
  __shared__ __align__(16) BlockShared<T> sh;

  const tmf_bcgs_desc d = desc[blockIdx.x];
  const int n = d.rows, ld = d.ld;
  if (n <= 0 || d.c_end - d.c_begin <= t) return;
  const int nwv = (n + 63) >> 6;           // row wavefronts 0 .. nwv - 1, helper = wavefront nwv
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (wv > nwv) return;                    // (a finished wavefront is not waited for by the workgroup's barriers)
  const int nthr = 64 * (nwv + 1);
  const int jb = d.c_begin + t;
  const int wb = min(64, d.c_end - jb);
  const int npan = (wb + 15) >> 4;
  T* __restrict__ P0 = reinterpret_cast<T*>(d.base) + (size_t)jb * ld;
  const double* __restrict__ raw = d.norms ? reinterpret_cast<const double*>(d.norms) + t : nullptr;

  // diagnostics (TMF_BORTH_STAMPS=1): cycles per phase of the first wavefront and of the helper, summed over the launch
  unsigned long long tq = dbg ? __builtin_amdgcn_s_memtime() : 0ull, tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto lap = [&](int i) {
    if (dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[i] += now - tq, tq = now;
    }
  };

  // second half of a reduction: the row wavefronts' partial 16 x 16 matrices (lane holds entries [l4 + 4 r][l15]), summed
  // in a fixed order into sh.C; every wavefront of the workgroup (the helper too) takes a share
  auto sum_partials = [&]() {
    for (int e = tid; e < NP * 256; e += nthr) {
      const int p = e >> 8, r = (e >> 6) & 3, ln = e & 63;
      double v = 0.0;
      for (int w = 0; w < nwv; ++w) v += reinterpret_cast<const double*>(sh.wv[w])[(p * 4 + r) * 64 + ln];
      sh.C[p][(ln >> 4) + 4 * r][ln & 15] = v;
    }
  };

  if (wv == nwv) {
    // ================================================================= helper wavefront
    const int a = lane >> 2, cg = lane & 3;
    for (int p = 0; p < npan; ++p) {
      const int wp = min(16, wb - 16 * p);
      for (int q = 0; q < p; ++q) {
        __syncthreads();
        sum_partials();
        __syncthreads();
      }
      for (int rep = 0; rep < 2; ++rep) {
        __syncthreads();
        sum_partials();
        __syncthreads();
        lap(0);
        // Second pass: G = 1 + E with |E| ~ eps cond(panel)^2.  R = 1 + U, U = strict upper part of E + diag(E) / 2, and
        // R^-1 = 1 - U to O(E^2): below 1e-8 the elementwise formula is exact to rounding and the 16 dependent steps of the
        // elimination (the critical path of the workgroup: every row wavefront waits for them) are not needed.
        bool quick = false;
        if (rep == 1) {
          double dev = 0.0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int col = cg + 4 * c;
            const double er = sh.C[0][a][col] - (a == col ? 1.0 : 0.0), ei = CP ? sh.C[NP - 1][a][col] : 0.0;
            dev = fmax(dev, fmax(fabs(er), fabs(ei)));
          }
          dev = wave_max64(dev);
          quick = dev < 1e-8;       // (also false for NaN)
        }
        if (quick) {
          double ur[4], ui[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int col = cg + 4 * c;
            const double er = sh.C[0][a][col], ei = CP ? sh.C[NP - 1][a][col] : 0.0;
            ur[c] = col > a ? -er : (col == a ? 1.0 - 0.5 * (er - 1.0) : 0.0);
            ui[c] = col > a ? -ei : 0.0;
            if (col >= wp || a >= wp) ur[c] = ui[c] = 0.0;      // columns beyond the panel stay zero
          }
          wave_sync();
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            sh.C[0][a][cg + 4 * c] = ur[c];
            if constexpr (CP) sh.C[NP - 1][a][cg + 4 * c] = ui[c];
          }
        } else {
        T mreg[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int col = cg + 4 * c;
          mreg[c] = col < 16 ? sc<T>::from2(sh.C[0][a][col], CP ? sh.C[NP - 1][a][col] : 0.0)
                             : ((col - 16 == a) ? sc<T>::one() : sc<T>::zero());
        }
        if (lane < 16) {
          sh.diag[lane] = sh.C[0][lane][lane];
          sh.raw[lane] = (raw && lane < wp) ? raw[16 * p + lane] : 0.0;
        }
        if (a == 0) {
#pragma unroll
          for (int c = 0; c < 8; ++c) sh.row[0][cg + 4 * c] = mreg[c];
        }
        wave_sync();
#pragma unroll 1
        for (int j = 0; j < 16; ++j) {
          const T* __restrict__ rowj = sh.row[j & 1];     // pivot row, published by the previous step
          const double piv = sc<T>::real(rowj[j]), g0 = sh.diag[j], rw = sh.raw[j];
          const bool drop = j >= wp || !(piv > 1e-13 * g0) || !(g0 > 0.0) || !(piv > 1e-28 * rw * rw);
          if (!drop) {
            if (a > j) {
              // 1 / piv: v_rcp_f64 + two Newton steps (the pivots are O(1e-13 .. 1) times the Gram diagonal)
              double r = __builtin_amdgcn_rcp(piv);
              r = fma(fma(-piv, r, 1.0), r, r);
              r = fma(fma(-piv, r, 1.0), r, r);
              const T f = sc<T>::scale(sc<T>::conj(rowj[a]), r);
#pragma unroll
              for (int c = 0; c < 8; ++c) mreg[c] = sc<T>::fms(mreg[c], f, rowj[cg + 4 * c]);
            } else if (a == j) {
              double q_ = __builtin_amdgcn_rsq(piv);       // 1 / sqrt(piv), two Newton steps
              q_ = q_ * fma(-0.5 * piv * q_, q_, 1.5);
              q_ = q_ * fma(-0.5 * piv * q_, q_, 1.5);
#pragma unroll
              for (int c = 0; c < 8; ++c) mreg[c] = sc<T>::scale(mreg[c], q_);
            }
          } else if (a == j) {   // a column that depends on the earlier ones to rounding: ZERO column of the result
#pragma unroll
            for (int c = 0; c < 8; ++c) mreg[c] = sc<T>::zero();
          }
          if (a == j + 1) {      // the next pivot row is final now: publish it (other buffer: this one is still being read)
#pragma unroll
            for (int c = 0; c < 8; ++c) sh.row[(j + 1) & 1][cg + 4 * c] = mreg[c];
          }
          wave_sync();
        }
        // R^-1 = (L^-1)^H into sh.C: entry [k][a] = conj(W[a][k]), W = columns 16 .. 31
#pragma unroll
        for (int c = 4; c < 8; ++c) {
          const int k = cg + 4 * c - 16;
          sh.C[0][k][a] = pl<T>::re(mreg[c]);
          if constexpr (CP) sh.C[NP - 1][k][a] = -pl<T>::im(mreg[c]);
        }
        }
        lap(1);
        __syncthreads();
      }
      __syncthreads();
    }
    if (dbg && lane == 0) {
      atomicAdd(&dbg[11], tacc[0]), atomicAdd(&dbg[12], tacc[1]);
    }
    return;
  }

  // ===================================================================== row wavefronts
  const int l15 = lane & 15, l4 = lane >> 4;
  const int r0 = 64 * wv;
  T* scr = reinterpret_cast<T*>(sh.wv[wv]);
  double* red = reinterpret_cast<double*>(sh.wv[wv]);
  T x[16];

  auto reduce = [&](const d4 (&acc)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(p * 4 + r) * 64 + lane] = acc[p][r];
    __syncthreads();
    sum_partials();
    __syncthreads();
  };

  // C = Y^H x over this wavefront's rows; Y given by a loader of register m.  Complex products by the 3M scheme as in
  // gemm.hip (the fp64 MFMA pipe, 64 cycles per instruction and SIMD, is what these phases wait for): with conj(y) = a + i b,
  // x = c + i d:  P1 = sum a c, P2 = sum b d, P3 = sum (a + b)(c + d);  Re = P1 - P2, Im = P3 - P1 - P2.
  auto contract = [&](auto&& load_y, d4 (&acc)[NP]) {
    d4 p1 = (d4){0.0, 0.0, 0.0, 0.0}, p2 = p1, p3 = p1;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const T y = load_y(m);
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pl<T>::re(y), pl<T>::re(x[m]), p1, 0, 0, 0);
      if constexpr (CP) {
        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pl<T>::im(y), pl<T>::im(x[m]), p2, 0, 0, 0);
        p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(pl<T>::re(y) - pl<T>::im(y), pl<T>::re(x[m]) + pl<T>::im(x[m]), p3, 0, 0, 0);
      }
      // (registers: the panel takes 64 of the 128 a lane has at 16 wavefronts per workgroup; without the fence the
      // scheduler hoists all 16 loads of Y to the top and spills the panel)
      if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (CP) {
      acc[0] = p1 - p2;
      acc[1] = p3 - p1 - p2;
    } else {
      acc[0] = p1;
    }
  };

  for (int p = 0; p < npan; ++p) {
    const int wp = min(16, wb - 16 * p);
    T* __restrict__ Pp = P0 + (size_t)(16 * p) * ld;
    // ---- the panel into registers
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int row = r0 + 4 * m + l4;
      x[m] = (row < n && l15 < wp) ? Pp[(size_t)row + (size_t)l15 * ld] : sc<T>::zero();
    }
    lap(0);
    // ---- against the earlier panels of the block (one pass, as tmf_bcgs_batched does inside a block)
    for (int q = 0; q < p; ++q) {
      const T* __restrict__ Q = P0 + (size_t)(16 * q) * ld;
      d4 acc[NP];
      contract([&](int m) {
        const int row = r0 + 4 * m + l4;
        return row < n ? Q[(size_t)row + (size_t)l15 * ld] : sc<T>::zero();
      }, acc);
      lap(1);
      reduce(acc);
      lap(2);
      // x -= Q C
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        d4 a[NP];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a[0][r] = pl<T>::re(x[4 * g + r]);
          if constexpr (CP) a[1][r] = pl<T>::im(x[4 * g + r]);
        }
        const int row = r0 + 16 * g + l15;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const T z = row < n ? Q[(size_t)row + (size_t)(4 * s + l4) * ld] : sc<T>::zero();
          const double cr = sh.C[0][4 * s + l4][l15];
          a[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pl<T>::re(z), cr, a[0], 0, 0, 0);
          if constexpr (CP) {
            const double ci = sh.C[1][4 * s + l4][l15];
            a[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pl<T>::im(z), ci, a[0], 0, 0, 0);
            a[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pl<T>::re(z), ci, a[1], 0, 0, 0);
            a[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pl<T>::im(z), cr, a[1], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) x[4 * g + r] = sc<T>::from2(a[0][r], CP ? a[NP - 1][r] : 0.0);
        __builtin_amdgcn_sched_barrier(0);
      }
      lap(3);
      // (sh.C is rewritten only behind the next reduction's first barrier, which every wavefront reaches after this loop)
    }
    // ---- Cholesky-QR, twice
    for (int rep = 0; rep < 2; ++rep) {
      d4 acc[NP];
      contract([&](int m) { return x[m]; }, acc);
      lap(4);
      reduce(acc);
      lap(2);
      __syncthreads();           // the helper wavefront has put R^-1 into sh.C
      lap(5);
      // x <- x R^-1: per row group, turn the 16 x 16 piece through the wavefront's tile into the other layout
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[(4 * r + l4) * SCR_LD + l15] = x[4 * g + r];
        wave_sync();
        d4 a[NP];
#pragma unroll
        for (int q_ = 0; q_ < NP; ++q_) a[q_] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const T z = scr[l15 * SCR_LD + 4 * s + l4];
          const double br = sh.C[0][4 * s + l4][l15];
          a[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pl<T>::re(z), br, a[0], 0, 0, 0);
          if constexpr (CP) {
            const double bi = sh.C[1][4 * s + l4][l15];
            a[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pl<T>::im(z), bi, a[0], 0, 0, 0);
            a[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pl<T>::re(z), bi, a[1], 0, 0, 0);
            a[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pl<T>::im(z), br, a[1], 0, 0, 0);
          }
        }
        wave_sync();     // the tile is rewritten by the next row group
#pragma unroll
        for (int r = 0; r < 4; ++r) x[4 * g + r] = sc<T>::from2(a[0][r], CP ? a[NP - 1][r] : 0.0);
      }
      lap(6);
    }
    // ---- store (the later panels of the block read it back, each wavefront its own rows)
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int row = r0 + 4 * m + l4;
      if (row < n && l15 < wp) Pp[(size_t)row + (size_t)l15 * ld] = x[m];
    }
    __syncthreads();
    lap(7);
  }
  if (dbg && tid == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&dbg[i], tacc[i]);
    atomicAdd(&dbg[8], 1ull), atomicAdd(&dbg[9], (unsigned long long)npan), atomicAdd(&dbg[10], (unsigned long long)n);
  }
}

}  // namespace tmf

static unsigned long long* borth_stamps() {
  static unsigned long long* p = nullptr;
  static bool on = getenv("TMF_BORTH_STAMPS") != nullptr;
  if (on && !p && hipMalloc((void**)&p, 16 * 8) == hipSuccess) (void)hipMemset(p, 0, 16 * 8);
  return on ? p : nullptr;
}

extern "C" int tmf_block_orth_stamps(uint64_t* out16) {
  unsigned long long* p = borth_stamps();
  if (!p) return TMF_E_ARG;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out16, p, 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) return TMF_E_HIP;
  (void)hipMemset(p, 0, 16 * 8);
  return TMF_OK;
}

// Orthonormalises the columns [c_begin + t, min(c_end, c_begin + t + 64)) of every slab in themselves (they have been
// projected against all earlier columns already).  rows <= max_rows <= 960 (15 row wavefronts + the helper).
extern "C" int tmf_block_orth_batched(int dtype, const tmf_bcgs_desc* d_desc, int nprob, int t, int max_rows, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  if (max_rows < 1 || max_rows > 960) {
    set_error("tmf_block_orth_batched: %d rows (1 .. 960)", max_rows);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned long long* dbg = borth_stamps();
  const dim3 g(nprob), b(64 * ((max_rows + 63) / 64 + 1));
  if (dtype == TMF_C128) hipLaunchKernelGGL(block_orth_kernel<cd>, g, b, 0, s, d_desc, t, dbg);
  else if (dtype == TMF_F64) hipLaunchKernelGGL(block_orth_kernel<double>, g, b, 0, s, d_desc, t, dbg);
  else {
    set_error("tmf_block_orth_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_block_orth_batched launch");
}
