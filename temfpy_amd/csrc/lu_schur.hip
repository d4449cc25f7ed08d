// Blocked LU restricted to the leading k x k "always" block, producing its determinant
// and the Schur complement onto the "sometimes" orbitals in one pass:
//
//   W = [ A  B ]      det_always = det(A),      S = D - C A^-1 B   (left in W[k:, k:])
//       [ C  D ]
//
// Reference: slater.py:1077-1090 (numpy det + inv + two matrix products per site).
// One 256-thread workgroup per site.  A panel of NB columns (all rows below the diagonal)
// is factored in LDS with partial pivoting among the always rows only; the trailing matrix
// is then updated once per panel (each thread keeps its row of the L panel in registers and
// streams the columns), so the matrix crosses L2/HBM k/NB times instead of k times.
#include "common.hpp"

namespace tmf {

constexpr int NBMAX = 16;
constexpr int UCH = 32;  // columns of U12 staged per chunk

template <typename T>
__global__ __launch_bounds__(256) void lu_schur_kernel(const tmf_schur_desc* __restrict__ desc, const int NB) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_schur_desc d = desc[blockIdx.x];
  const int mb = d.mb, mk = d.mk, k = d.k, ldw = d.ldw;
  T* __restrict__ W = reinterpret_cast<T*>(d.W);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // LDS: panel P[c * rows + r] (rows <= mb), U chunk [UCH][NBMAX], reduction scratch
  T* P = reinterpret_cast<T*>(smem);
  T* Uc = P + (size_t)mb * NB;
  double* rv = reinterpret_cast<double*>(Uc + UCH * NBMAX);
  int* ri = reinterpret_cast<int*>(rv + 4);
  int* pivs = ri + 4;  // NBMAX pivots of the current panel

  T det = sc<T>::one();

  for (int j0 = 0; j0 < k; j0 += NB) {
    const int jb = min(NB, k - j0);
    const int rows = mb - j0;       // panel rows j0 .. mb-1
    const int prow = k - j0;        // rows eligible as pivots: local [j, prow)
    for (int e = tid; e < rows * jb; e += 256) {
      const int r = e % rows, c = e / rows;
      P[(size_t)c * rows + r] = W[(size_t)(j0 + r) + (size_t)(j0 + c) * ldw];
    }
    __syncthreads();
    // ---- factor the panel -------------------------------------------------------------
    for (int j = 0; j < jb; ++j) {
      T* pj = P + (size_t)j * rows;
      double bv = -1.0;
      int bi = j;
      for (int r = j + tid; r < prow; r += 256) {
        const double v = sc<T>::abs2(pj[r]);
        if (v > bv) bv = v, bi = r;
      }
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
      }
      if (lane == 0) rv[wave] = bv, ri[wave] = bi;
      __syncthreads();
      if (tid == 0) {
        double b = rv[0];
        int ix = ri[0];
        for (int q = 1; q < 4; ++q)
          if (rv[q] > b || (rv[q] == b && ri[q] < ix)) b = rv[q], ix = ri[q];
        pivs[j] = ix;
      }
      __syncthreads();
      const int piv = pivs[j];
      if (piv != j && tid < jb) {  // swap rows j <-> piv inside the panel
        const T t = P[(size_t)tid * rows + j];
        P[(size_t)tid * rows + j] = P[(size_t)tid * rows + piv];
        P[(size_t)tid * rows + piv] = t;
      }
      __syncthreads();
      const T pv = pj[j];
      det = sc<T>::mul(det, pv);
      if (piv != j) det = sc<T>::neg(det);
      const T pinv = sc<T>::abs2(pv) > 0.0 ? sc<T>::inv(pv) : sc<T>::zero();
      for (int r = j + 1 + tid; r < rows; r += 256) {
        const T l = sc<T>::mul(pj[r], pinv);
        pj[r] = l;
        for (int c = j + 1; c < jb; ++c) {
          T* pc = P + (size_t)c * rows;
          pc[r] = sc<T>::fms(pc[r], l, pc[j]);
        }
      }
      __syncthreads();
    }
    // ---- apply the row swaps to the trailing columns ---------------------------------------
    const int c0 = j0 + jb;
    const int ncols = mk - c0;
    for (int c = tid; c < ncols; c += 256) {
      T* col = W + (size_t)(c0 + c) * ldw + j0;
      for (int j = 0; j < jb; ++j) {
        const int piv = pivs[j];
        if (piv != j) {
          const T t = col[j];
          col[j] = col[piv];
          col[piv] = t;
        }
      }
    }
    __syncthreads();
    // ---- trailing update in column chunks: U12 = L11^-1 A12 ; A22 -= L21 U12 ----------------
    const int r2 = rows - jb;  // rows below the panel's diagonal block
    for (int cc = 0; cc < ncols; cc += UCH) {
      const int nc = min(UCH, ncols - cc);
      if (tid < nc) {  // one thread per column: forward substitution with unit-lower L11
        T* col = W + (size_t)(c0 + cc + tid) * ldw + j0;
        T u[NBMAX];
#pragma unroll
        for (int q = 0; q < NBMAX; ++q) u[q] = (q < jb) ? col[q] : sc<T>::zero();
#pragma unroll
        for (int q = 0; q < NBMAX; ++q) {
          if (q < jb) {
#pragma unroll
            for (int r = q + 1; r < NBMAX; ++r)
              if (r < jb) u[r] = sc<T>::fms(u[r], P[(size_t)q * rows + r], u[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NBMAX; ++q) {
          if (q < jb) {
            col[q] = u[q];
            Uc[tid * NBMAX + q] = u[q];
          }
        }
      }
      __syncthreads();
      for (int rb = 0; rb < r2; rb += 256) {
        const int r = rb + tid;
        if (r < r2) {
          T l[NBMAX];
#pragma unroll
          for (int q = 0; q < NBMAX; ++q) l[q] = (q < jb) ? P[(size_t)q * rows + jb + r] : sc<T>::zero();
          T* rowp = W + (size_t)(j0 + jb + r) + (size_t)(c0 + cc) * ldw;
          // eight columns at a time: all eight loads are issued before the first store (a load after a
          // store through the same pointer cannot be hoisted by the compiler, and with one workgroup per
          // site nothing else hides the memory latency)
          for (int cb = 0; cb < nc; cb += 8) {
            T x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (cb + u < nc) ? rowp[(size_t)(cb + u) * ldw] : sc<T>::zero();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
              for (int q = 0; q < NBMAX; ++q)
                if (q < jb) x[u] = sc<T>::fms(x[u], l[q], Uc[(cb + u) * NBMAX + q]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (cb + u < nc) rowp[(size_t)(cb + u) * ldw] = x[u];
          }
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) *reinterpret_cast<T*>(d.det) = det;
  if (d.S) {  // optional compact copy of the Schur complement
    T* __restrict__ S = reinterpret_cast<T*>(d.S);
    const int sr = mb - k, scn = mk - k;
    __syncthreads();
    for (int e = tid; e < sr * scn; e += 256) {
      const int r = e % sr, c = e / sr;
      S[(size_t)r + (size_t)c * d.lds] = W[(size_t)(k + r) + (size_t)(k + c) * ldw];
    }
  }
}


// -------------------------------------------------------------------------------------------
// Blocked form over several launches (tmf_lu_block_batched / tmf_lu_trsm_batched + the MFMA GEMM):
// the one-workgroup-per-site kernel above streams the whole trailing matrix through L2 once per
// 16-column panel with VALU arithmetic (13 % of the fp64 peak, 4.7x the algorithmic traffic,
// measured).  Here an outer block of WB = 64 columns is factored by one workgroup per site
// (lu_block_kernel: the same panel code, trailing updates confined to the block), the block's row
// interchanges and the triangular solve U12 = L11^-1 A12 run one thread per trailing column
// (lu_trsm_kernel), and the rank-64 update A22 -= L21 U12 of ALL sites is one batched MFMA GEMM
// launch: the trailing matrix is read and written k / 64 times, by every CU.
// -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void lu_block_kernel(const int32_t* __restrict__ run_if, const tmf_lublock_desc* __restrict__ desc, const int j0, const int WB,
                                                       const int NB) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (run_if != nullptr && *run_if == 0) return;   // conditional-launch scope (tmf_launch_condition)
  const tmf_lublock_desc d = desc[blockIdx.x];
  const int mb = d.mb, k = d.k, ldw = d.ldw;
  T* __restrict__ W = reinterpret_cast<T*>(d.W);
  int32_t* __restrict__ gpiv = reinterpret_cast<int32_t*>(d.piv);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (j0 >= k) {
    if (j0 == 0 && tid == 0) *reinterpret_cast<T*>(d.det) = sc<T>::one();  // no always-block: det of the empty matrix
    return;
  }
  const int cend = min(k, j0 + WB);  // columns of this outer block: [j0, cend)

  T* P = reinterpret_cast<T*>(smem);
  T* Uc = P + (size_t)(mb - j0) * NB;
  double* rv = reinterpret_cast<double*>(Uc + UCH * NBMAX);
  int* ri = reinterpret_cast<int*>(rv + 4);
  int* pivs = ri + 4;

  T det = (j0 == 0) ? sc<T>::one() : *reinterpret_cast<const T*>(d.det);

  for (int jj = j0; jj < cend; jj += NB) {
    const int jb = min(NB, cend - jj);
    const int rows = mb - jj;   // panel rows jj .. mb-1
    const int prow = k - jj;    // rows eligible as pivots: local [j, prow)
    for (int e = tid; e < rows * jb; e += 256) {
      const int r = e % rows, c = e / rows;
      P[(size_t)c * rows + r] = W[(size_t)(jj + r) + (size_t)(jj + c) * ldw];
    }
    __syncthreads();
    for (int j = 0; j < jb; ++j) {
      T* pj = P + (size_t)j * rows;
      double bv = -1.0;
      int bi = j;
      for (int r = j + tid; r < prow; r += 256) {
        const double v = sc<T>::abs2(pj[r]);
        if (v > bv) bv = v, bi = r;
      }
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
      }
      if (lane == 0) rv[wave] = bv, ri[wave] = bi;
      __syncthreads();
      if (tid == 0) {
        double b = rv[0];
        int ix = ri[0];
        for (int q = 1; q < 4; ++q)
          if (rv[q] > b || (rv[q] == b && ri[q] < ix)) b = rv[q], ix = ri[q];
        pivs[j] = ix;
      }
      __syncthreads();
      const int piv = pivs[j];
      if (piv != j && tid < jb) {
        const T t = P[(size_t)tid * rows + j];
        P[(size_t)tid * rows + j] = P[(size_t)tid * rows + piv];
        P[(size_t)tid * rows + piv] = t;
      }
      __syncthreads();
      const T pv = pj[j];
      det = sc<T>::mul(det, pv);
      if (piv != j) det = sc<T>::neg(det);
      const T pinv = sc<T>::abs2(pv) > 0.0 ? sc<T>::inv(pv) : sc<T>::zero();
      for (int r = j + 1 + tid; r < rows; r += 256) {
        const T l = sc<T>::mul(pj[r], pinv);
        pj[r] = l;
        for (int c = j + 1; c < jb; ++c) {
          T* pc = P + (size_t)c * rows;
          pc[r] = sc<T>::fms(pc[r], l, pc[j]);
        }
      }
      __syncthreads();
    }
    // the factored panel goes back: L21 is the A operand of the GEMM, L11 that of the triangular solve
    for (int e = tid; e < rows * jb; e += 256) {
      const int r = e % rows, c = e / rows;
      W[(size_t)(jj + r) + (size_t)(jj + c) * ldw] = P[(size_t)c * rows + r];
    }
    if (tid < jb) gpiv[jj + tid] = jj + pivs[tid];
    // ---- row interchanges on the other columns of this block: [j0, jj) (earlier panels' L) and [jj+jb, cend) ----
    // (composing the jb interchanges into one permutation per panel, with independent loads per column, measured
    // SLOWER: 411 vs 362 us per launch - the arrays it needs go to scratch)
    const int nleft = jj - j0, nright = cend - (jj + jb);
    for (int c = tid; c < nleft + nright; c += 256) {
      const int cc = c < nleft ? j0 + c : jj + jb + (c - nleft);
      T* col = W + (size_t)cc * ldw + jj;
      for (int j = 0; j < jb; ++j) {
        const int piv = pivs[j];
        if (piv != j) {
          const T t = col[j];
          col[j] = col[piv];
          col[piv] = t;
        }
      }
    }
    __syncthreads();
    // ---- trailing update inside the block: U12 = L11^-1 A12 ; A22 -= L21 U12 for columns [jj+jb, cend) ----
    const int c0 = jj + jb;
    const int ncols = nright;
    const int r2 = rows - jb;
    for (int cc = 0; cc < ncols; cc += UCH) {
      const int nc = min(UCH, ncols - cc);
      if (tid < nc) {
        T* col = W + (size_t)(c0 + cc + tid) * ldw + jj;
        T u[NBMAX];
#pragma unroll
        for (int q = 0; q < NBMAX; ++q) u[q] = (q < jb) ? col[q] : sc<T>::zero();
#pragma unroll
        for (int q = 0; q < NBMAX; ++q) {
          if (q < jb) {
#pragma unroll
            for (int r = q + 1; r < NBMAX; ++r)
              if (r < jb) u[r] = sc<T>::fms(u[r], P[(size_t)q * rows + r], u[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NBMAX; ++q) {
          if (q < jb) {
            col[q] = u[q];
            Uc[tid * NBMAX + q] = u[q];
          }
        }
      }
      __syncthreads();
      for (int rb = 0; rb < r2; rb += 256) {
        const int r = rb + tid;
        if (r < r2) {
          T l[NBMAX];
#pragma unroll
          for (int q = 0; q < NBMAX; ++q) l[q] = (q < jb) ? P[(size_t)q * rows + jb + r] : sc<T>::zero();
          T* rowp = W + (size_t)(jj + jb + r) + (size_t)(c0 + cc) * ldw;
          for (int cb = 0; cb < nc; cb += 8) {
            T x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (cb + u < nc) ? rowp[(size_t)(cb + u) * ldw] : sc<T>::zero();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
              for (int q = 0; q < NBMAX; ++q)
                if (q < jb) x[u] = sc<T>::fms(x[u], l[q], Uc[(cb + u) * NBMAX + q]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (cb + u < nc) rowp[(size_t)(cb + u) * ldw] = x[u];
          }
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) *reinterpret_cast<T*>(d.det) = det;
}

// Row interchanges of the outer block [j0, cend) applied to every trailing column, then U12 = L11^-1 A12 into the
// scratch T (nbk x ncols, leading dimension WB): one thread per column, L11 (unit lower) in LDS, 16 rows at a time
// in registers.  (Composing the interchanges into one permutation with gathered loads measured slower: 207 vs 183 us.)
template <typename T>
__global__ __launch_bounds__(256) void lu_trsm_kernel(const int32_t* __restrict__ run_if, const tmf_lublock_desc* __restrict__ desc, const int j0, const int WB) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (run_if != nullptr && *run_if == 0) return;
  const tmf_lublock_desc d = desc[blockIdx.x];
  const int k = d.k, ldw = d.ldw;
  if (j0 >= k) return;
  const int cend = min(k, j0 + WB), nbk = cend - j0, ncols = d.mk - cend;
  if ((int)blockIdx.y * 256 >= ncols) return;
  T* __restrict__ W = reinterpret_cast<T*>(d.W);
  T* Ls = reinterpret_cast<T*>(smem);                        // Ls[c * nbk + r] = L11[r][c]
  int* pv = reinterpret_cast<int*>(Ls + (size_t)WB * WB);
  const int tid = threadIdx.x;
  for (int e = tid; e < nbk * nbk; e += 256) {
    const int r = e % nbk, c = e / nbk;
    if (r > c) Ls[e] = W[(size_t)(j0 + r) + (size_t)(j0 + c) * ldw];
  }
  if (tid < nbk) pv[tid] = reinterpret_cast<const int32_t*>(d.piv)[j0 + tid];
  __syncthreads();
  const int col = blockIdx.y * 256 + tid;
  if (col >= ncols) return;
  T* __restrict__ cp = W + (size_t)(cend + col) * ldw;
  for (int j = 0; j < nbk; ++j) {
    const int pr = pv[j];
    if (pr != j0 + j) {
      const T t = cp[j0 + j];
      cp[j0 + j] = cp[pr];
      cp[pr] = t;
    }
  }
  T* __restrict__ tp = reinterpret_cast<T*>(d.T) + (size_t)col * WB;
  for (int p0 = 0; p0 < nbk; p0 += NBMAX) {
    const int pb = min(NBMAX, nbk - p0);
    T a[NBMAX];
#pragma unroll
    for (int i = 0; i < NBMAX; ++i) a[i] = (i < pb) ? cp[j0 + p0 + i] : sc<T>::zero();
    for (int q = 0; q < p0; ++q) {
      const T tq = tp[q];
      const T* lq = Ls + (size_t)q * nbk + p0;
#pragma unroll
      for (int i = 0; i < NBMAX; ++i)
        if (i < pb) a[i] = sc<T>::fms(a[i], lq[i], tq);
    }
#pragma unroll
    for (int i = 0; i < NBMAX; ++i) {
      if (i < pb) {
        const T* li = Ls + (size_t)(p0 + i) * nbk + p0;
#pragma unroll
        for (int r = i + 1; r < NBMAX; ++r)
          if (r < pb) a[r] = sc<T>::fms(a[r], li[r], a[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NBMAX; ++i)
      if (i < pb) tp[p0 + i] = a[i];
  }
}


// -------------------------------------------------------------------------------------------
// Block-local pivoting (tmf_diag_inverse_batched): the filled-orbital bases of neighbouring cuts are built from the SAME
// random block (column f of Omega_f -> column f of the basis, Gram-Schmidt in column order), so the overlap of the
// always-occupied orbitals of a site is close to a (block-)diagonal matrix once tmf_site_prepare has put the paired
// filled orbitals on the diagonal.  Block Gaussian elimination is stable for such matrices as long as ||D^-1 A12|| stays
// bounded (Demmel, Higham, Schreiber 1995), so the row search is confined to the 64 x 64 diagonal block D of each outer
// step and everything else is MFMA GEMM work:
//     D^-1 (this kernel)        X = D^-1 A12        A22 -= A21 X        det(A) = prod det(D)
// The diagonal blocks are counted from the END of the always-block (block 0 has (k - 1) % 64 + 1 columns, the others 64):
// the orbitals without a partner come last (tmf_site_prepare), and the loss of rank of the paired part that they repair
// only shows in its last pivots - both must sit in the same diagonal block for the local search to find them.
// Reported per matrix: the smallest pivot magnitude and the largest |entry| of D^-1 over the blocks that have always-rows
// below them (where a search over all rows could have done better).  The caller repeats the factorisation fully pivoted
// (tmf_lu_block_batched) when that entry is large, so robustness does not rest on the structure assumed here.
//
// The kernel: in-place Gauss-Jordan inversion with (implicit) row pivoting, entirely in registers.  One 256-thread
// workgroup per matrix; thread (lane r, wave w) holds row r of the 16 columns c = w (mod 4).  Per elimination step the
// wave that owns column j finds the pivot row (DPP max over the unused rows), publishes the multipliers through LDS
// (one barrier per step), every wave reads the pivot row of its own columns with v_readlane (the pivot row index is
// wave-uniform) and updates its 16 columns: 64 readlanes + 64 FMAs per step and wave, no LDS traffic for the matrix.
// Rows are never moved: row p_j (the pivot row of column j) of the result is row j of (P D)^-1, un-permuted on output.
// -------------------------------------------------------------------------------------------
__device__ inline double readlane_t(double v, int l) { return readlane_d(v, l); }
__device__ inline cd readlane_t(cd v, int l) { return make_cd(readlane_d(v.x, l), readlane_d(v.y, l)); }

template <typename T>
__global__ __launch_bounds__(256) void diag_inverse_kernel(const tmf_diaginv_desc* __restrict__ desc, const int step, double* __restrict__ stats) {
  constexpr int WB = 64;
  __shared__ T colbuf[2][WB];
  __shared__ T pvbuf[2];
  __shared__ int pivbuf[2];
  __shared__ int perm[WB], pos[WB];
  __shared__ T wdet[4];
  __shared__ double wmin[4], wmax[4];
  const tmf_diaginv_desc d = desc[blockIdx.x];
  const int k = d.k, ldw = d.ldw;
  const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
  // blocks counted from the END of the always-block: a ragged first block, full ones after it
  const int nb0 = k > 0 ? (k - 1) % WB + 1 : 0;
  const int j0 = step == 0 ? 0 : nb0 + (step - 1) * WB;
  if (j0 >= k) {
    if (j0 == 0 && tid == 0) *reinterpret_cast<T*>(d.det) = sc<T>::one(), stats[2 * blockIdx.x] = 1e300, stats[2 * blockIdx.x + 1] = 0.0;
    return;
  }
  const int cend = step == 0 ? nb0 : j0 + WB, nb = cend - j0;
  const T* __restrict__ W = reinterpret_cast<const T*>(d.W);
  // the block, padded with the identity to 64 x 64 (the padding pivots on itself with pivot 1)
  T x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = cg + 4 * i;
    x[i] = (r < nb && c < nb) ? W[(size_t)(j0 + r) + (size_t)(j0 + c) * ldw] : (r == c ? sc<T>::one() : sc<T>::zero());
  }
  bool used = false;
  T det = sc<T>::one();        // product of the pivots of the columns this wave owns
  double minp = 1e300;
#pragma unroll 1
  for (int jj = 0; jj < 16; ++jj) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = 4 * jj + b, buf = b & 1;
      if (cg == b) {            // this wave owns column j: it sits in x[0] (the registers rotate once per jj)
        const T v = x[0];
        const double a2 = used ? -1.0 : sc<T>::abs2(v);
        const double mx = wave_max64(a2);
        const int piv = __builtin_ctzll(__ballot(a2 == mx));
        const T pv = readlane_t(v, piv);
        const T pvinv = mx > 0.0 ? sc<T>::inv(pv) : sc<T>::zero();     // singular block: det = 0, finite garbage
        colbuf[buf][r] = (r == piv) ? sc<T>::sub(sc<T>::one(), pvinv) : sc<T>::mul(v, pvinv);
        if (r == 0) pivbuf[buf] = piv, pvbuf[buf] = pvinv, perm[j] = piv;
        det = sc<T>::mul(det, pv);
        if (j < nb) minp = fmin(minp, mx);
      }
      __syncthreads();
      // unified update  x[r][c] -= g[r] x[p][c]:  g[r] = x[r][j] / pv for the other rows, g[p] = 1 - 1 / pv turns row p
      // into x[p][c] / pv; column j itself becomes the column of the inverse: -g[r], and 1 / pv in row p
      const T g = colbuf[buf][r];
      const int piv = __builtin_amdgcn_readfirstlane(pivbuf[buf]);
      if (r == piv) used = true;
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = sc<T>::fms(x[i], g, readlane_t(x[i], piv));
      if (cg == b) x[0] = (r == piv) ? pvbuf[buf] : sc<T>::neg(g);
    }
    const T t = x[0];
#pragma unroll
    for (int i = 0; i < 15; ++i) x[i] = x[i + 1];
    x[15] = t;
  }
  // ---- bookkeeping: det(D) = prod pivots * sign(perm); statistics; un-permuted output -------------------------
  double big = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const double a = sc<T>::abs2(x[i]);
    big = (a > big || a != a) ? a : big;
  }
  {
    // NaN-aware maximum over the wave (wave_max64 would drop a NaN)
    for (int o = 32; o > 0; o >>= 1) {
      const double a = __shfl_xor(big, o);
      big = (a > big || a != a) ? a : big;
    }
  }
  const double wm = minp;      // wave-uniform already
  // every wave owns pivots: combine the four partial products / minima
  if (r == 0) wdet[cg] = det, wmin[cg] = wm, wmax[cg] = big;
  __syncthreads();
  if (tid < WB) pos[tid] = WB;          // (a NaN block can leave perm short of a permutation: such rows are not written)
  __syncthreads();
  if (tid < WB) pos[perm[tid]] = tid;
  if (tid == 0) {
    T dd = sc<T>::mul(sc<T>::mul(wdet[0], wdet[1]), sc<T>::mul(wdet[2], wdet[3]));
    unsigned long long seen = 0;
    int transpositions = 0;
    for (int i = 0; i < WB; ++i) {          // parity from the cycle lengths
      if (seen >> i & 1) continue;
      int len = 0;
      for (int q = i; !(seen >> q & 1); q = perm[q]) seen |= 1ull << q, ++len;
      transpositions += len - 1;
    }
    if (transpositions & 1) dd = sc<T>::neg(dd);
    const T prev = (step == 0) ? sc<T>::one() : *reinterpret_cast<const T*>(d.det);
    *reinterpret_cast<T*>(d.det) = sc<T>::mul(prev, dd);
    const double m = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
    double g = wmax[0];
    for (int w = 1; w < 4; ++w) g = (wmax[w] > g || wmax[w] != wmax[w]) ? wmax[w] : g;
    const double m0 = (step == 0) ? 1e300 : stats[2 * blockIdx.x];
    const double g0 = (step == 0) ? 0.0 : stats[2 * blockIdx.x + 1];
    stats[2 * blockIdx.x] = fmin(m, m0);
    stats[2 * blockIdx.x + 1] = (cend < k) ? ((g > g0 || g != g) ? g : g0) : g0;   // only blocks with always-rows below them
  }
  __syncthreads();
  // (P D)^-1 = D^-1 P^T: storage row p_i is row i of it, its column j belongs to column p_j of D^-1
  T* __restrict__ inv = reinterpret_cast<T*>(d.inv);
  const int row = pos[r];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int col = perm[cg + 4 * i];
    if ((unsigned)row < (unsigned)nb && (unsigned)col < (unsigned)nb) inv[(size_t)col * WB + row] = x[i];
  }
}


// Verdict on the block-local elimination of a whole launch series: *flag = 1 when any block inverse has an entry above
// cap (or a NaN), or when forced; summary (page-locked host memory mapped into the device, or device memory) receives
// {smallest |pivot|, largest |D^-1 entry|, flag}.  The fully pivoted kernels that follow run under
// tmf_launch_condition(flag): the decision stays on the device, the host never waits for it.
__global__ __launch_bounds__(256) void diag_verdict_kernel(const double* __restrict__ stats, const int n, const double cap2, const int force,
                                                           int32_t* __restrict__ flag, double* __restrict__ summary) {
  __shared__ double smin[256], smax[256];
  double mn = 1e300, mx = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    mn = fmin(mn, stats[2 * i]);
    const double a = stats[2 * i + 1];
    mx = (a > mx || a != a) ? a : mx;
  }
  smin[threadIdx.x] = mn, smax[threadIdx.x] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 256; ++i) {
      mn = fmin(mn, smin[i]);
      mx = (smax[i] > mx || smax[i] != smax[i]) ? smax[i] : mx;
    }
    const int f = (force || !(mx <= cap2)) ? 1 : 0;
    *flag = f;
    summary[0] = sqrt(mn), summary[1] = sqrt(mx), summary[2] = (double)f;
  }
}

}  // namespace tmf

extern "C" int tmf_lu_schur_batched(int dtype, const tmf_schur_desc* d_desc, int nprob, int max_mb, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  int nb = NBMAX;  // panel width: shrink until the mb x nb panel fits the 160 KiB LDS
  auto need = [&](int w) { return ((size_t)max_mb * w + UCH * NBMAX) * elem + 256; };
  while (need(nb) > 160 * 1024 && nb > 1) nb >>= 1;
  const size_t lds = need(nb);
  if (lds > 160 * 1024) {
    set_error("tmf_lu_schur_batched: mb = %d needs %zu B of LDS (> 160 KiB)", max_mb, lds);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)lu_schur_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lu_schur_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_done = true;
  }
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(lu_schur_kernel<cd>, dim3(nprob), dim3(256), lds, s, d_desc, nb);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(lu_schur_kernel<double>, dim3(nprob), dim3(256), lds, s, d_desc, nb);
  else {
    set_error("tmf_lu_schur_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_lu_schur_batched launch");
}

extern "C" int tmf_lu_block_batched(int dtype, const tmf_lublock_desc* d_desc, int nprob, int j0, int wb, int max_mb, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  if (wb != 64 || j0 < 0 || (j0 % wb) != 0) {
    set_error("tmf_lu_block_batched: block width %d (supported: 64), first column %d", wb, j0);
    return TMF_E_ARG;
  }
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const int rows = max_mb - j0 > 1 ? max_mb - j0 : 1;
  int nb = NBMAX;
  auto need = [&](int w) { return ((size_t)rows * w + UCH * NBMAX) * elem + 512; };
  // narrower panels for tall blocks: three workgroups per CU instead of one (one workgroup per site, ~1000 sites, 256 CUs)
  // (measured: 52 KiB budget 362 us per launch, 80 / 160 KiB 390 us, 34 KiB 407 us - the kernel is bound by its
  // dependent steps, not by occupancy)
  while (need(nb) > 52 * 1024 && nb > 4) nb >>= 1;
  while (need(nb) > 160 * 1024 && nb > 1) nb >>= 1;
  const size_t lds = need(nb);
  if (lds > 160 * 1024) {
    set_error("tmf_lu_block_batched: %d rows need %zu B of LDS (> 160 KiB)", rows, lds);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)lu_block_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lu_block_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(lu_block_kernel<cd>, dim3(nprob), dim3(256), lds, s, launch_condition(), d_desc, j0, wb, nb);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(lu_block_kernel<double>, dim3(nprob), dim3(256), lds, s, launch_condition(), d_desc, j0, wb, nb);
  else {
    set_error("tmf_lu_block_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_lu_block_batched launch");
}

extern "C" int tmf_lu_trsm_batched(int dtype, const tmf_lublock_desc* d_desc, int nprob, int j0, int wb, int max_cols, void* stream) {
  using namespace tmf;
  if (nprob <= 0 || max_cols <= 0) return TMF_OK;
  if (wb != 64) {
    set_error("tmf_lu_trsm_batched: block width %d (supported: 64)", wb);
    return TMF_E_ARG;
  }
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const size_t lds = (size_t)wb * wb * elem + (size_t)wb * 4 * 4 + 64;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)lu_trsm_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lu_trsm_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const dim3 grid(nprob, (max_cols + 255) / 256);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(lu_trsm_kernel<cd>, grid, dim3(256), lds, s, launch_condition(), d_desc, j0, wb);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(lu_trsm_kernel<double>, grid, dim3(256), lds, s, launch_condition(), d_desc, j0, wb);
  else {
    set_error("tmf_lu_trsm_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_lu_trsm_batched launch");
}

extern "C" int tmf_diag_inverse_batched(int dtype, const tmf_diaginv_desc* d_desc, int nprob, int step, void* d_stats, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  if (step < 0) {
    set_error("tmf_diag_inverse_batched: outer step %d", step);
    return TMF_E_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(diag_inverse_kernel<cd>, dim3(nprob), dim3(256), 0, s, d_desc, step, (double*)d_stats);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(diag_inverse_kernel<double>, dim3(nprob), dim3(256), 0, s, d_desc, step, (double*)d_stats);
  else {
    set_error("tmf_diag_inverse_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_diag_inverse_batched launch");
}

extern "C" int tmf_diag_inverse_verdict(const void* d_stats, int nprob, double cap, int force, int32_t* d_flag, double* summary,
                                        void* stream) {
  using namespace tmf;
  hipLaunchKernelGGL(diag_verdict_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), (const double*)d_stats, nprob, cap * cap,
                     force, d_flag, summary);
  return check_hip(hipGetLastError(), "tmf_diag_inverse_verdict launch");
}
