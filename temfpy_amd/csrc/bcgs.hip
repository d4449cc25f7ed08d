// Blocked classical Gram-Schmidt (BCGS with re-orthogonalisation) of many slabs, driven from C++.
//
// Orthonormalises the columns [c_begin, c_end) of every slab against all columns before them, panel
// by panel: per panel and pass two MFMA GEMMs (coefficients Q^H P, update P -= Q c) and then the LDS
// panel kernel (orth_panel.hip).  It is the QR of the range-finder slabs and of the filled-orbital
// bases that replace numpy.linalg.eigh (slater.py:347).
//
// The descriptors and tile tables of every step are BUILT ON THE DEVICE from one small per-slab
// record (a 256-thread kernel: pointer arithmetic + a block scan for the tile offsets), so a whole QR
// needs a single descriptor upload and ~8 launches per panel from this loop - the Python driver
// spent 0.15 ms per GEMM on building and uploading them (154 GEMMs per conversion, measured).
#include "common.hpp"

namespace tmf {

// Block = columns [c_begin + t, c_begin + t + w); it is projected against the columns
// [from, c_begin + t) with from = 0 (t0 < 0: everything before the block) or from = c_begin + t0
// (the earlier panels of the same 16-column block).
__global__ __launch_bounds__(256) void bcgs_prepare_kernel(const tmf_bcgs_desc* __restrict__ desc, int nprob, int t,
                                                           int w, int t0, int coef_rows, size_t elem,
                                                           tmf_gemm_desc* __restrict__ g_coef,
                                                           tmf_gemm_desc* __restrict__ g_upd,
                                                           tmf_panel_desc* __restrict__ pd, int32_t* __restrict__ tiles_coef,
                                                           int32_t* __restrict__ tiles_upd, tmf_gemm_desc* __restrict__ g_gram,
                                                           int32_t* __restrict__ tiles_gram) {
  __shared__ int scan_c[257], scan_u[257];
  const int tid = threadIdx.x;
  const int per = (nprob + 255) / 256;
  const int i0 = tid * per, i1 = min(nprob, i0 + per);
  int nc = 0, nu = 0;
  for (int i = i0; i < i1; ++i) {
    const tmf_bcgs_desc d = desc[i];
    const int span = d.c_end - d.c_begin;
    const bool act = span > t && d.rows > 0;
    const int jb = d.c_begin + t;                      // first column of the block
    const int from = t0 < 0 ? 0 : d.c_begin + t0;
    const int j = jb - from;                           // columns to project against
    const int wj = act ? min(w, d.c_end - jb) : 0;
    const uint64_t colp = d.base + (uint64_t)jb * d.ld * elem;
    const uint64_t qp = d.base + (uint64_t)from * d.ld * elem;
    tmf_gemm_desc c, u;
    c.A = qp, c.B = colp, c.C = d.scratch;              // coefficients (j x wj) = Q^H P
    c.M = act ? j : 0, c.N = wj, c.K = d.rows, c.lda = d.ld, c.ldb = d.ld, c.ldc = j > 1 ? j : 1;
    u.A = qp, u.B = d.scratch, u.C = colp;              // P -= Q c
    u.M = act ? d.rows : 0, u.N = (act && j > 0) ? wj : 0, u.K = j, u.lda = d.ld, u.ldb = j > 1 ? j : 1, u.ldc = d.ld;
    if (j == 0) c.N = 0;                                // nothing to project against
    g_coef[i] = c, g_upd[i] = u;
    tmf_panel_desc p;
    p.A = colp, p.norms = d.norms ? d.norms + 8ull * t : 0ull, p.n = act ? d.rows : 0, p.w = wj, p.lda = d.ld, p.pad = 0;
    pd[i] = p;
    if (g_gram) {  // Gram matrix of the panel (Cholesky-QR mode): wj x wj into the slab's scratch, one tile each
      tmf_gemm_desc g;
      g.A = colp, g.B = colp, g.C = d.scratch, g.M = wj, g.N = wj, g.K = act ? d.rows : 0, g.lda = d.ld, g.ldb = d.ld;
      g.ldc = wj > 1 ? wj : 1;
      g_gram[i] = g;
      tiles_gram[4 * i] = i, tiles_gram[4 * i + 1] = 0, tiles_gram[4 * i + 2] = 0, tiles_gram[4 * i + 3] = 0;
    }
    nc += (c.M > 0 && c.N > 0) ? (c.M + coef_rows - 1) / coef_rows : 0;   // tall kernel: 16 x 16 output tiles; wide blocks: 64 x 64
    nu += (u.M > 0 && u.N > 0) ? (u.M + 63) / 64 : 0;
  }
  scan_c[tid + 1] = nc, scan_u[tid + 1] = nu;
  if (tid == 0) scan_c[0] = scan_u[0] = 0;
  __syncthreads();
  if (tid == 0)
    for (int k = 1; k <= 256; ++k) scan_c[k] += scan_c[k - 1], scan_u[k] += scan_u[k - 1];
  __syncthreads();
  int oc = scan_c[tid], ou = scan_u[tid];
  for (int i = i0; i < i1; ++i) {
    const tmf_gemm_desc c = g_coef[i], u = g_upd[i];
    if (c.M > 0 && c.N > 0)
      for (int m = 0; m < (c.M + coef_rows - 1) / coef_rows; ++m, ++oc) {
        tiles_coef[4 * oc] = i, tiles_coef[4 * oc + 1] = m, tiles_coef[4 * oc + 2] = 0, tiles_coef[4 * oc + 3] = 0;
      }
    if (u.M > 0 && u.N > 0)
      for (int m = 0; m < (u.M + 63) / 64; ++m, ++ou) {
        tiles_upd[4 * ou] = i, tiles_upd[4 * ou + 1] = m, tiles_upd[4 * ou + 2] = 0, tiles_upd[4 * ou + 3] = 0;
      }
  }
}

// Cholesky-QR step of one panel per workgroup: G = P^H P (w x w, in the slab's scratch) -> upper R with
// G = R^H R -> X = R^{-1} -> P <- P X.  A pivot below 1e-13 of its diagonal entry (column dependent on the
// earlier ones to rounding) gives a ZERO column, like the LDS panel kernel.  Run twice per panel
// ("CholQR2"): the first pass leaves an orthogonality error ~ eps * cond(P)^2, the second removes it.
// Only for well-conditioned panels (the filled-orbital bases); the range-finder slabs, whose columns
// can be pure rounding noise, keep the Gram-Schmidt panel kernel.
template <typename T>
__global__ __launch_bounds__(256) void cholqr_apply_kernel(const tmf_panel_desc* __restrict__ pd,
                                                           const tmf_gemm_desc* __restrict__ gram) {
  const tmf_panel_desc d = pd[blockIdx.x];
  const int n = d.n, w = d.w;
  if (n <= 0 || w <= 0) return;
  __shared__ T R[16][17], X[16][17];
  __shared__ int dropped[16];
  const T* __restrict__ G = reinterpret_cast<const T*>(gram[blockIdx.x].C);
  const int ldg = gram[blockIdx.x].ldc;
  const int tid = threadIdx.x;
  if (tid < 256) {
    const int r = tid & 15, c = tid >> 4;
    R[r][c] = (r < w && c < w) ? G[(size_t)r + (size_t)c * ldg] : sc<T>::zero();
    X[r][c] = sc<T>::zero();
  }
  __syncthreads();
  // Cholesky (row by row of the upper factor), thread i owns column i
  for (int j = 0; j < w; ++j) {
    if (tid < 16) {
      const int i = tid;
      if (i >= j && i < w) {
        T v = R[j][i];   // G_ji
        for (int k = 0; k < j; ++k) v = sc<T>::fms(v, sc<T>::conj(R[k][j]), R[k][i]);
        R[j][i] = v;     // unscaled: divided by the pivot below
      }
    }
    __syncthreads();
    if (tid < 16) {
      const int i = tid;
      const double piv = sc<T>::real(R[j][j]);
      const double g0 = sc<T>::real(reinterpret_cast<const T*>(G)[(size_t)j + (size_t)j * ldg]);
      bool drop = !(piv > 1e-13 * g0) || !(g0 > 0.0);
      if (d.norms) {  // residual below 1e-14 of the column's norm BEFORE any projection: rounding noise
        const double raw = reinterpret_cast<const double*>(d.norms)[j];
        drop = drop || !(piv > 1e-28 * raw * raw);
      }
      if (i == 0) dropped[j] = drop;
      if (i >= j && i < w) {
        if (drop) R[j][i] = (i == j) ? sc<T>::one() : sc<T>::zero();
        else R[j][i] = sc<T>::scale(R[j][i], 1.0 / sqrt(piv));
      }
    }
    __syncthreads();
  }
  // X = R^{-1} (upper triangular), thread c owns column c
  if (tid < 16 && tid < w) {
    const int c = tid;
    if (!dropped[c]) {
      X[c][c] = sc<T>::scale(sc<T>::one(), 1.0 / sc<T>::real(R[c][c]));
      for (int i = c - 1; i >= 0; --i) {
        T acc = sc<T>::zero();
        for (int k = i + 1; k <= c; ++k) acc = sc<T>::fmac(acc, R[i][k], X[k][c]);
        X[i][c] = sc<T>::scale(acc, -1.0 / sc<T>::real(R[i][i]));
      }
    }
  }
  __syncthreads();
  T* __restrict__ P = reinterpret_cast<T*>(d.A);
  for (int r = tid; r < n; r += 256) {
    T x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = c < w ? P[(size_t)r + (size_t)c * d.lda] : sc<T>::zero();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < w) {
        T y = sc<T>::zero();
#pragma unroll
        for (int k = 0; k <= c; ++k) y = sc<T>::fmac(y, x[k], X[k][c]);
        P[(size_t)r + (size_t)c * d.lda] = y;
      }
    }
  }
}

}  // namespace tmf

extern "C" int64_t tmf_bcgs_work_bytes(const tmf_bcgs_desc* h_desc, int nprob) {
  int64_t tc = 0, tu = 0;
  for (int i = 0; i < nprob; ++i) {
    tc += (h_desc[i].c_end + 15) / 16;
    tu += (h_desc[i].rows + 63) / 64;
  }
  return (int64_t)nprob * (3 * sizeof(tmf_gemm_desc) + sizeof(tmf_panel_desc) + 16) + 16 * (tc + tu) + 1024;
}

extern "C" int tmf_bcgs_batched(int dtype, const tmf_bcgs_desc* d_desc, const tmf_bcgs_desc* h_desc, int nprob,
                                int passes, int flags, void* d_work, int64_t work_bytes, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  if (dtype != TMF_C128 && dtype != TMF_F64) {
    set_error("tmf_bcgs_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  if (work_bytes < tmf_bcgs_work_bytes(h_desc, nprob)) {
    set_error("tmf_bcgs_batched: workspace of %lld bytes is too small (tmf_bcgs_work_bytes)", (long long)work_bytes);
    return TMF_E_ARG;
  }
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  int max_rows = 0, max_span = 0;
  int64_t cap_c = 0;
  for (int i = 0; i < nprob; ++i) {
    max_rows = h_desc[i].rows > max_rows ? h_desc[i].rows : max_rows;
    const int span = h_desc[i].c_end - h_desc[i].c_begin;
    max_span = span > max_span ? span : max_span;
    cap_c += (h_desc[i].c_end + 15) / 16;
  }
  const bool cholqr = (flags & 1) != 0;   // Cholesky-QR panels (no LDS panel: always 16 wide)
  // flags & 2: outer blocks of 64 columns.  The projection of a block against everything before it streams those
  // columns twice (coefficients, update) and is bound by that traffic (PMC: 22 GB per conversion for the filled
  // bases with 16-column blocks, 4 TB/s); 64-column blocks read them a quarter as often.  The coefficient and update
  // products of such a block are ordinary 64-wide MFMA GEMMs; inside it the 16-column panels are projected against
  // the earlier panels of the block only.  Needs c_end x 64 elements of scratch per slab.
  const bool wide = (flags & 2) != 0;
  // flags & 4: the inner part of a 64-column block (its four panels) in one launch, tmf_block_orth_batched
  const bool fused = (flags & 4) != 0 && cholqr && wide && max_rows <= 960;
  int w = 16;  // widest panel that fits the LDS of orth_panel_kernel
  while (!cholqr && (size_t)max_rows * w * elem + 1024 > 150 * 1024 && w > 1) w >>= 1;
  char* wk = static_cast<char*>(d_work);
  tmf_gemm_desc* g_coef = reinterpret_cast<tmf_gemm_desc*>(wk);
  tmf_gemm_desc* g_upd = g_coef + nprob;
  tmf_panel_desc* pd = reinterpret_cast<tmf_panel_desc*>(g_upd + nprob);
  int32_t* tiles_coef = reinterpret_cast<int32_t*>(pd + nprob);
  int32_t* tiles_upd = tiles_coef + 4 * cap_c;
  int64_t cap_u = 0;
  for (int i = 0; i < nprob; ++i) cap_u += (h_desc[i].rows + 63) / 64;
  tmf_gemm_desc* g_gram = reinterpret_cast<tmf_gemm_desc*>(tiles_upd + 4 * cap_u);
  int32_t* tiles_gram = reinterpret_cast<int32_t*>(g_gram + nprob);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto orth_panel = [&]() -> int {
    if (!cholqr) return tmf_orth_panel_batched(dtype, pd, nprob, max_rows, w, stream);
    for (int rep = 0; rep < 2; ++rep) {
      int st = tmf_gemm_tall_batched(dtype, 1.0, 0.0, g_gram, tiles_gram, nprob, stream);
      if (st) return st;
      if (dtype == TMF_C128) hipLaunchKernelGGL(cholqr_apply_kernel<cd>, dim3(nprob), dim3(256), 0, s, pd, g_gram);
      else hipLaunchKernelGGL(cholqr_apply_kernel<double>, dim3(nprob), dim3(256), 0, s, pd, g_gram);
      st = check_hip(hipGetLastError(), "tmf_bcgs_batched cholqr");
      if (st) return st;
    }
    return TMF_OK;
  };
  // Outer blocks of 16 columns are projected against everything before them in one go (the GEMMs are
  // bound by streaming those columns, so fewer, wider projections = less traffic); tall slabs whose
  // LDS panel is narrower (w < 16) then orthonormalise the block panel by panel, projecting each
  // later panel only against the earlier panels of its own block.
  const int wo = wide ? 64 : 16;
  auto project = [&](int t, int wb, int t0) -> int {
    const int crow = wb > 16 ? 64 : 16;   // rows of a coefficient tile: MFMA GEMM kernel / tall-skinny kernel
    int64_t nc = 0, nu = 0;
    for (int i = 0; i < nprob; ++i) {
      const tmf_bcgs_desc& d = h_desc[i];
      if (d.c_end - d.c_begin <= t || d.rows <= 0) continue;
      const int j = d.c_begin + t - (t0 < 0 ? 0 : d.c_begin + t0);
      if (j > 0) {
        nc += (j + crow - 1) / crow;
        nu += (d.rows + 63) / 64;
      }
    }
    hipLaunchKernelGGL(bcgs_prepare_kernel, dim3(1), dim3(256), 0, s, d_desc, nprob, t, wb, t0, crow, elem, g_coef, g_upd, pd,
                       tiles_coef, tiles_upd, cholqr ? g_gram : (tmf_gemm_desc*)nullptr, tiles_gram);
    int st = check_hip(hipGetLastError(), "tmf_bcgs_batched prepare");
    for (int p = 0; p < passes && nc > 0 && !st; ++p) {
      if (wb > 16) {
        st = tmf_gemm_batched(dtype, 1, 1.0, 0.0, g_coef, tiles_coef, (int)nc, 64, stream);
        if (!st) st = tmf_gemm_batched(dtype, 0, -1.0, 1.0, g_upd, tiles_upd, (int)nu, 64, stream);
      } else {
        st = tmf_gemm_tall_batched(dtype, 1.0, 0.0, g_coef, tiles_coef, (int)nc, stream);
        if (!st) st = tmf_gemm_batched(dtype, 0, -1.0, 1.0, g_upd, tiles_upd, (int)nu, 16, stream);
      }
    }
    return st;
  };
  for (int t = 0; t < max_span; t += wo) {
    int st = project(t, wo, -1);
    if (st) return st;
    if (fused) {
      st = tmf_block_orth_batched(dtype, d_desc, nprob, t, max_rows, stream);
      if (st) return st;
      continue;
    }
    if (w == wo) {  // the block is one LDS panel: the descriptors of `project` are the panel's
      st = orth_panel();
      if (st) return st;
      continue;
    }
    for (int ti = 0; ti < wo && t + ti < max_span; ti += w) {
      st = project(t + ti, w, t);
      if (st) return st;
      st = orth_panel();
      if (st) return st;
    }
  }
  return TMF_OK;
}
