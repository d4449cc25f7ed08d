// One-sided (Hestenes) Jacobi for small dense matrices, one workgroup per problem,
// X (p x p) and the accumulated rotations V resident in LDS:   X V = U diag(s).
//
// Used twice per entanglement cut instead of numpy.linalg.eigh (slater.py:347):
//  * on the column-graded triangular factor of the projected off-diagonal block, where
//    its high relative accuracy resolves singular values around sqrt(cutoff) = 1e-6 that a
//    Gram-matrix eigensolver would bury in rounding noise, and
//  * on the Rayleigh-Ritz matrix of the entangled subspace (Hermitian PSD: s = eigenvalues,
//    V = eigenvectors).
// Rotations of one round (p/2 disjoint pairs, round-robin tournament) run in parallel, each
// pair on TPP lanes of one wavefront; dot products are reduced with width-TPP shuffles.
#include "common.hpp"

namespace tmf {

// 1 / sqrt(x): v_rsq_f64 (about 1e-8 relative... refined by two Newton steps to ~1 ulp); x > 0, normal range
__device__ inline double rsqrt_fast(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}

template <typename T, bool WITH_V>
__global__ __launch_bounds__(512) void jacobi_kernel(const tmf_jacobi_desc* __restrict__ desc,
                                                     int32_t* __restrict__ sweeps_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_jacobi_desc d = desc[blockIdx.x];
  const int p = d.p;
  if (p <= 0) return;
  T* X = reinterpret_cast<T*>(smem);        // X[c * p + r]
  T* V = X + (size_t)p * p;                 // only with WITH_V
  double* nrm = reinterpret_cast<double*>(X + (WITH_V ? 2 : 1) * (size_t)p * p);  // p norms
  int* flag = reinterpret_cast<int*>(nrm + p);                 // rotation counter

  const int tid = threadIdx.x;
  const T* __restrict__ Xg = reinterpret_cast<const T*>(d.X);
  for (int e = tid; e < p * p; e += 512) {
    const int r = e % p, c = e / p;
    X[e] = Xg[(size_t)r + (size_t)c * d.ldx];
    if (WITH_V) V[e] = (r == c) ? sc<T>::one() : sc<T>::zero();
  }
  if (tid == 0) *flag = 0;
  __syncthreads();

  // Only columns that are not identically zero take part in the tournament (the Rayleigh-Ritz matrix
  // has a zero row and column for every direction below the range-finder threshold: ~half of p).
  __shared__ int act[128];
  __shared__ int s_nact;
  if (tid < 128) act[tid] = 0;
  __syncthreads();
  for (int c = tid; c < p; c += 512) {
    bool nz = false;
    for (int r = 0; r < p && !nz; ++r) nz = sc<T>::abs2(X[(size_t)c * p + r]) > 0.0;
    act[c] = nz;
  }
  __syncthreads();
  if (tid == 0) {
    int k = 0;
    for (int c = 0; c < p; ++c)
      if (act[c]) act[k++] = c;   // in place: k <= c
    s_nact = k;
  }
  __syncthreads();
  const int nact = s_nact;
  const int pe = (nact + 1) & ~1;   // players (even)
  const int m = pe - 1;             // players on the circle; player m sits still
  const int npairs = pe / 2;
  int tpp = 64;                     // lanes per pair: largest power of two with npairs*tpp <= 512
  while (tpp * npairs > 512) tpp >>= 1;
  const int pair = tid >> (31 - __builtin_clz(tpp)), pl = tid & (tpp - 1);      // tpp is a power of two
  // rotate while |<x_i, x_j>| > sqrt(max(p, 16)) eps |x_i| |x_j| (the rounding level of the p-term inner product
  // itself; a tighter bound makes pairs at that level rotate for ever: 18 of 1023 cuts hit the cap).  The floor of 4 eps
  // is the accuracy of a rotation (c, s from rsq / rcp + Newton steps, a few ulp): below it a rotated pair is as
  // orthogonal as it gets, and without the floor every 3 x 3 problem with three significant columns kept "rotating" by
  // rounding-level angles until the sweep cap (results correct, LinAlgError raised: tests/soak/soak_small.py seed 1919).
  const double tol2 = 1.1e-16 * 1.1e-16 * (double)(p > 16 ? p : 16);

  int sweep = 0;
  for (; sweep < 60; ++sweep) {
    for (int rho = 0; rho < m; ++rho) {
      if (pair < npairs) {  // a pair never straddles wavefronts (tpp divides 64)
        int i, j;
        if (pair == 0) {
          i = m;
          j = rho;
        } else {
          i = rho + pair;          // (rho < m, pair <= m: one conditional wrap instead of two integer divisions per round -
          j = rho - pair;          //  the kernel is bound by its VALU instruction count, a quarter of it integer arithmetic)
          if (i >= m) i -= m;
          if (j < 0) j += m;
        }
        if (i > j) {
          const int t = i;
          i = j;
          j = t;
        }
        if (j < nact) {
          i = act[i], j = act[j];
          T* xi = X + (size_t)i * p;
          T* xj = X + (size_t)j * p;
          double al = 0.0, be = 0.0;
          T ga = sc<T>::zero();
          for (int r = pl; r < p; r += tpp) {
            const T a = xi[r], b = xj[r];
            al += sc<T>::abs2(a);
            be += sc<T>::abs2(b);
            ga = sc<T>::fmacc(ga, a, b);
          }
          group_sum3<T>(al, be, ga, tpp);      // DPP butterfly (the ds_bpermute one cost ~60 cycles per word and step)
          const double g2 = sc<T>::abs2(ga);
          if (g2 > tol2 * al * be && g2 > 1e-280) {   // (below: rsq / rcp without denormal handling)
            // Rotation parameters with v_rsq_f64 / v_rcp_f64 + Newton steps instead of IEEE sqrt / div
            // sequences (every lane of the pair computes them redundantly, and that scalar math was about
            // half of the kernel's instruction count).  Only c^2 + s^2 = 1 has to hold to eps (it does:
            // s = c t with c refined to eps); a last-bit error in the angle costs nothing.
            const double ginv = rsqrt_fast(g2);                       // 1 / |gamma|
            const double zeta = 0.5 * (be - al) * ginv;
            const double az = fabs(zeta);
            const double w1 = 1.0 + zeta * zeta;
            const double den = az + w1 * rsqrt_fast(w1);              // |zeta| + sqrt(1 + zeta^2)
            const double t = copysign(sc<double>::inv_fast(den), zeta);
            const double c = rsqrt_fast(1.0 + t * t), s = c * t;
            // phase e^{-i theta} of gamma, folded into column j
            const T ph = sc<T>::scale(sc<T>::conj(ga), ginv);
            const T sph = sc<T>::scale(ph, s), cph = sc<T>::scale(ph, c);
            for (int r = pl; r < p; r += tpp) {
              const T a = xi[r], b = xj[r];
              xi[r] = sc<T>::sub(sc<T>::scale(a, c), sc<T>::mul(sph, b));
              xj[r] = sc<T>::add(sc<T>::scale(a, s), sc<T>::mul(cph, b));
            }
            if (WITH_V) {
              T* vi = V + (size_t)i * p;
              T* vj = V + (size_t)j * p;
              for (int r = pl; r < p; r += tpp) {
                const T a = vi[r], b = vj[r];
                vi[r] = sc<T>::sub(sc<T>::scale(a, c), sc<T>::mul(sph, b));
                vj[r] = sc<T>::add(sc<T>::scale(a, s), sc<T>::mul(cph, b));
              }
            }
            if (pl == 0) atomicAdd(flag, 1);
          }
        }
      }
      __syncthreads();
    }
    const int rot = *flag;
    __syncthreads();
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (rot == 0) break;
  }
  if (tid == 0 && sweeps_out) sweeps_out[blockIdx.x] = sweep;

  // column norms, rank by descending norm (ties: lower index first)
  for (int c = tid; c < p; c += 512) {
    double s = 0.0;
    for (int r = 0; r < p; ++r) s += sc<T>::abs2(X[(size_t)c * p + r]);
    nrm[c] = sqrt(s);
  }
  __syncthreads();
  T* __restrict__ Vg = reinterpret_cast<T*>(d.V);
  T* __restrict__ Ug = reinterpret_cast<T*>(d.U);
  double* __restrict__ sg = reinterpret_cast<double*>(d.s);
  for (int e = tid; e < p * p; e += 512) {
    const int r = e % p, c = e / p;
    const double sc_ = nrm[c];
    int rank = 0;
    for (int c2 = 0; c2 < p; ++c2) rank += (nrm[c2] > sc_) || (nrm[c2] == sc_ && c2 < c);
    const bool keep = !(d.thresh2 > 0.0) || sc_ * sc_ >= d.thresh2;
    if (WITH_V) Vg[(size_t)r + (size_t)rank * d.ldv] = keep ? V[e] : sc<T>::zero();
    if (Ug) Ug[(size_t)r + (size_t)rank * d.ldu] = sc<T>::scale(X[e], (keep && sc_ > 0.0) ? 1.0 / sc_ : 0.0);
    if (r == 0) sg[rank] = sc_;
  }
  if (tid == 0 && d.count) {
    int cnt = 0;
    for (int c = 0; c < p; ++c) cnt += !(d.thresh2 > 0.0) || nrm[c] * nrm[c] >= d.thresh2;
    *reinterpret_cast<int32_t*>(d.count) = cnt;
  }
}

}  // namespace tmf

template <bool WITH_V>
static int launch_jacobi(int dtype, const tmf_jacobi_desc* d_desc, int nprob, int max_p, int32_t* d_sweeps, void* stream,
                         const char* who) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const size_t lds = (WITH_V ? 2 : 1) * (size_t)max_p * max_p * elem + (size_t)max_p * 8 + 64;
  if (max_p <= 0 || max_p > 128 || lds > 159 * 1024) {
    set_error("%s: p = %d needs %zu B of LDS (limit 159 KiB)", who, max_p, lds);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    // (the kernel also has ~0.5 KiB of static LDS: ask for 159 KiB of dynamic LDS, not the full 160)
    (void)hipFuncSetAttribute((const void*)jacobi_kernel<cd, WITH_V>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              159 * 1024);
    (void)hipFuncSetAttribute((const void*)jacobi_kernel<double, WITH_V>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              159 * 1024);
    attr_done = true;
  }
  if (dtype == TMF_C128)
    hipLaunchKernelGGL((jacobi_kernel<cd, WITH_V>), dim3(nprob), dim3(512), lds, s, d_desc, d_sweeps);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL((jacobi_kernel<double, WITH_V>), dim3(nprob), dim3(512), lds, s, d_desc, d_sweeps);
  else {
    set_error("%s: bad dtype %d", who, dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), who);
}

extern "C" int tmf_jacobi_batched(int dtype, const tmf_jacobi_desc* d_desc, int nprob, int max_p, int32_t* d_sweeps,
                                  void* stream) {
  return launch_jacobi<true>(dtype, d_desc, nprob, max_p, d_sweeps, stream, "tmf_jacobi_batched");
}

// Left singular vectors and singular values only (desc.V is ignored, desc.U must be set): no
// rotation accumulator, half the LDS, two workgroups per CU.
extern "C" int tmf_svd_left_batched(int dtype, const tmf_jacobi_desc* d_desc, int nprob, int max_p, int32_t* d_sweeps,
                                    void* stream) {
  return launch_jacobi<false>(dtype, d_desc, nprob, max_p, d_sweeps, stream, "tmf_svd_left_batched");
}

// ---------------------------------------------------------------------------------------------
// Compacting variant for rank-deficient factors (canonicalisation sweeps of projected MPS, p up to 4096):
// columns whose squared norm is below thresh2 * 1e-4 / p cannot lift a singular value over the threshold
// (together they perturb the spectrum by < 1e-2 sqrt(thresh2) in absolute terms; the rounding noise of the
// QR that produced the factor sits at 1e-16, four decades below a 1e-12 cutoff) and take no part; only the nact active
// columns of X and of the rotation accumulator V are held, p x nact each, in LDS when they fit (otherwise V,
// or both, stay in global memory: d.X in place, d.V as p x nact workspace).  Output: d.U = right singular
// vectors sorted by descending singular value (zero columns for everything below the threshold), d.s, d.count.
// With d.V == 0 no rotations are accumulated and d.U receives the LEFT singular vectors (normalised rotated
// columns) instead: fed with the conjugate transpose of a (twice) QR-preconditioned factor, whose columns are
// graded, these are the wanted right vectors at half the work and half the LDS.
// After a Gutzwiller projection about half of the columns are inactive: 4x fewer rotations per sweep and
// p = 130 (real) fits the LDS that the plain kernel exhausts at p = 100.
// ---------------------------------------------------------------------------------------------
namespace tmf {

template <typename T>
__global__ __launch_bounds__(512) void jacobi_compact_kernel(const tmf_jacobi_desc* __restrict__ desc, int lds_elems, int head_cols,
                                                             int32_t* __restrict__ sweeps_out) {
  extern __shared__ __align__(16) unsigned char smem_all[];
  __shared__ int s_nact, flag;
  const tmf_jacobi_desc d = desc[blockIdx.x];
  const int p = d.p;
  if (p <= 0) return;
  // head of the dynamic LDS: column norms, active list and a second list for its reordering (16 B per column of the
  // LARGEST problem of the launch: `head_cols`), then the columns themselves
  double* nrm = reinterpret_cast<double*>(smem_all);
  int* act = reinterpret_cast<int*>(nrm + head_cols);
  int* act2 = act + head_cols;
  unsigned char* smem = smem_all + (size_t)head_cols * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* Xg = reinterpret_cast<T*>(d.X);
  T* Vg = reinterpret_cast<T*>(d.V);
  T* Ug = reinterpret_cast<T*>(d.U);
  double* sg = reinterpret_cast<double*>(d.s);

  for (int c = wave; c < p; c += 8) {
    double s = 0.0;
    for (int r = lane; r < p; r += 64) s += sc<T>::abs2(Xg[(size_t)r + (size_t)c * d.ldx]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) nrm[c] = s;
  }
  __syncthreads();
  if (tid == 0) {
    const double lim = d.thresh2 * 1e-4 / (double)p;
    int k = 0;
    for (int c = 0; c < p; ++c)
      if (nrm[c] > 0.0 && nrm[c] >= lim) act[k++] = c;
    s_nact = k;
    flag = 0;
  }
  __syncthreads();
  const int nact = s_nact;
  // de Rijk ordering: active columns by descending norm (fewer sweeps on graded factors)
  for (int a = tid; a < nact; a += 512) {
    const int my_col = act[a];
    const double v = nrm[my_col];
    int my_rank = 0;
    for (int b = 0; b < nact; ++b) {
      const int cb = act[b];
      my_rank += (nrm[cb] > v) || (nrm[cb] == v && cb < my_col);
    }
    act2[my_rank] = my_col;
  }
  __syncthreads();
  for (int a = tid; a < nact; a += 512) act[a] = act2[a];
  __syncthreads();
  const bool with_v = Vg != nullptr;    // desc.V == 0: left vectors only (no accumulator), see the entry point
  const bool x_lds = (size_t)p * nact <= (size_t)lds_elems;
  const bool v_lds = with_v && 2 * (size_t)p * nact <= (size_t)lds_elems;
  T* Xs = reinterpret_cast<T*>(smem);
  T* Vs = Xs + (size_t)p * nact;
  auto xcol = [&](int a) -> T* { return x_lds ? Xs + (size_t)a * p : Xg + (size_t)act[a] * d.ldx; };
  auto vcol = [&](int a) -> T* { return v_lds ? Vs + (size_t)a * p : Vg + (size_t)a * d.ldv; };
  for (int a = wave; a < nact; a += 8) {
    T* x = xcol(a);
    T* v = vcol(a);
    const int c = act[a];
    for (int r = lane; r < p; r += 64) {
      if (x_lds) x[r] = Xg[(size_t)r + (size_t)c * d.ldx];
      if (with_v) v[r] = (r == c) ? sc<T>::one() : sc<T>::zero();
    }
  }
  __syncthreads();

  const int pe = (nact + 1) & ~1, m = pe - 1, npairs = pe / 2;
  int tpp = 64;
  while (tpp * npairs > 512 && tpp > 1) tpp >>= 1;
  const int slots = 512 / tpp;                    // pairs processed at once (npairs may exceed it for tpp = 1)
  const double tol2 = 1.1e-16 * 1.1e-16 * (double)(p > 16 ? p : 16);      // (floor: see jacobi_kernel)
  int sweep = 0;
  for (; sweep < 60 && nact > 1; ++sweep) {
    for (int rho = 0; rho < m; ++rho) {
      for (int pair = tid >> (31 - __builtin_clz(tpp)); pair < npairs; pair += slots) {
        const int pl = tid & (tpp - 1);      // tpp is a power of two
        int i, j;
        if (pair == 0) {
          i = m;
          j = rho;
        } else {
          i = rho + pair;          // (rho < m, pair <= m: one conditional wrap instead of two integer divisions per round -
          j = rho - pair;          //  the kernel is bound by its VALU instruction count, a quarter of it integer arithmetic)
          if (i >= m) i -= m;
          if (j < 0) j += m;
        }
        if (i > j) {
          const int t = i;
          i = j;
          j = t;
        }
        if (j < nact) {
          T* xi = xcol(i);
          T* xj = xcol(j);
          double al = 0.0, be = 0.0;
          T ga = sc<T>::zero();
          for (int r = pl; r < p; r += tpp) {
            const T a = xi[r], b = xj[r];
            al += sc<T>::abs2(a);
            be += sc<T>::abs2(b);
            ga = sc<T>::fmacc(ga, a, b);
          }
          group_sum3<T>(al, be, ga, tpp);      // DPP butterfly (the ds_bpermute one cost ~60 cycles per word and step)
          const double g2 = sc<T>::abs2(ga);
          if (g2 > tol2 * al * be && g2 > 1e-280) {
            const double ginv = rsqrt_fast(g2);
            const double zeta = 0.5 * (be - al) * ginv;
            const double w1 = 1.0 + zeta * zeta;
            const double den = fabs(zeta) + w1 * rsqrt_fast(w1);
            const double t = copysign(sc<double>::inv_fast(den), zeta);
            const double c = rsqrt_fast(1.0 + t * t), s = c * t;
            const T ph = sc<T>::scale(sc<T>::conj(ga), ginv);
            const T sph = sc<T>::scale(ph, s), cph = sc<T>::scale(ph, c);
            for (int r = pl; r < p; r += tpp) {
              const T a = xi[r], b = xj[r];
              xi[r] = sc<T>::sub(sc<T>::scale(a, c), sc<T>::mul(sph, b));
              xj[r] = sc<T>::add(sc<T>::scale(a, s), sc<T>::mul(cph, b));
            }
            if (with_v) {
              T* vi = vcol(i);
              T* vj = vcol(j);
              for (int r = pl; r < p; r += tpp) {
                const T a = vi[r], b = vj[r];
                vi[r] = sc<T>::sub(sc<T>::scale(a, c), sc<T>::mul(sph, b));
                vj[r] = sc<T>::add(sc<T>::scale(a, s), sc<T>::mul(cph, b));
              }
            }
            if (pl == 0) atomicAdd(&flag, 1);
          }
        }
      }
      __syncthreads();
    }
    const int rot = flag;
    __syncthreads();
    if (tid == 0) flag = 0;
    __syncthreads();
    if (rot == 0) break;
  }
  if (tid == 0 && sweeps_out) sweeps_out[blockIdx.x] = sweep;

  // singular values = norms of the rotated active columns; rank among them
  for (int a = wave; a < nact; a += 8) {
    const T* x = xcol(a);
    double s = 0.0;
    for (int r = lane; r < p; r += 64) s += sc<T>::abs2(x[r]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) nrm[a] = sqrt(s);
  }
  __syncthreads();
  for (int a = wave; a < p; a += 8) {          // output column `rank`; inactive columns fill the tail with zeros
    if (a < nact) {
      const double sa = nrm[a];
      int rank = 0;
      for (int b = 0; b < nact; ++b) rank += (nrm[b] > sa) || (nrm[b] == sa && b < a);
      const bool keep = sa * sa >= d.thresh2;
      if (with_v) {
        const T* v = vcol(a);
        for (int r = lane; r < p; r += 64) Ug[(size_t)r + (size_t)rank * d.ldu] = keep ? v[r] : sc<T>::zero();
      } else {      // left singular vectors: the rotated columns, normalised
        const T* x = xcol(a);
        const double f = (keep && sa > 0.0) ? 1.0 / sa : 0.0;
        for (int r = lane; r < p; r += 64) Ug[(size_t)r + (size_t)rank * d.ldu] = sc<T>::scale(x[r], f);
      }
      if (lane == 0) sg[rank] = sa;
    } else {
      for (int r = lane; r < p; r += 64) Ug[(size_t)r + (size_t)a * d.ldu] = sc<T>::zero();
      if (lane == 0) sg[a] = 0.0;
    }
  }
  if (tid == 0 && d.count) {
    int cnt = 0;
    for (int a = 0; a < nact; ++a) cnt += nrm[a] * nrm[a] >= d.thresh2;
    *reinterpret_cast<int32_t*>(d.count) = cnt;
  }
}

}  // namespace tmf

extern "C" int tmf_jacobi_compact_batched(int dtype, const tmf_jacobi_desc* d_desc, int nprob, int max_p,
                                          int32_t* d_sweeps, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  if (max_p <= 0 || max_p > 4096) {      // (p > ~100: the columns live in global memory, slow but correct)
    set_error("tmf_jacobi_compact_batched: p = %d not in 1..4096", max_p);
    return TMF_E_LIMIT;
  }
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  const int head_cols = (max_p + 15) & ~15;
  const size_t head = (size_t)head_cols * 16;
  size_t lds = 2 * (size_t)max_p * max_p * elem;          // enough for everything active
  if (lds > 158 * 1024 - head) lds = 158 * 1024 - head;
  if (lds < 1024) lds = 1024;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)jacobi_compact_kernel<cd>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    (void)hipFuncSetAttribute((const void*)jacobi_compact_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    attr_done = true;
  }
  const int lds_elems = (int)(lds / elem);
  if (dtype == TMF_C128)
    hipLaunchKernelGGL(jacobi_compact_kernel<cd>, dim3(nprob), dim3(512), lds + head, s, d_desc, lds_elems, head_cols, d_sweeps);
  else if (dtype == TMF_F64)
    hipLaunchKernelGGL(jacobi_compact_kernel<double>, dim3(nprob), dim3(512), lds + head, s, d_desc, lds_elems, head_cols, d_sweeps);
  else {
    set_error("tmf_jacobi_compact_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_jacobi_compact_batched");
}

// ---------------------------------------------------------------------------------------------
// Block variant for p > 64: X (and V) stay in global memory (L2), two column blocks of width bw are
// staged in LDS at a time and swept against each other (block one-sided Jacobi).  One workgroup per
// problem, no inter-workgroup synchronisation.  Used only when a cut's entanglement rank exceeds
// what the 64-column range finder resolves (the engine then repeats the stage with 128 or 256
// columns), so it is built for generality, not speed.
// X is destroyed; WITH_V: V is the workspace of the accumulated rotations and the sorted right
// vectors are written to U; otherwise U receives the sorted, normalised left vectors.
// ---------------------------------------------------------------------------------------------
namespace tmf {

// widest block (a power of two <= 32) whose pair, x2 with the rotation accumulator, fits 150 KiB of LDS for a p-row problem
__host__ __device__ inline int jacobi_block_width(int p, bool with_v, size_t elem) {
  int bw = 32;
  while (bw > 1 && (size_t)(with_v ? 2 : 1) * 2 * bw * p * elem + 64 > 150 * 1024) bw >>= 1;
  return bw;
}

template <typename T, bool WITH_V>
__global__ __launch_bounds__(512) void jacobi_block_kernel(const tmf_jacobi_desc* __restrict__ desc, int,
                                                           int32_t* __restrict__ sweeps_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const tmf_jacobi_desc d = desc[blockIdx.x];
  const int p = d.p;
  if (p <= 0) return;
  // The block width follows from the problem's OWN size, not from the largest problem of the launch: the order of the
  // rotations - and with it every rounding - must not depend on what else is in the batch.  (A shard of a multi-GPU
  // conversion and its neighbour both compute the cut between them; with the width taken from the launch they disagreed
  // by 1e-16, which exactly degenerate spectra turn into different bases: tests/soak/soak_shards.py seed 1952.)
  const int bw = jacobi_block_width(p, WITH_V, sizeof(T));
  const int nc = 2 * bw;                               // columns resident in LDS
  T* Xs = reinterpret_cast<T*>(smem);                  // Xs[c * p + r], c < nc
  T* Vs = Xs + (size_t)nc * p;                         // only WITH_V
  int* flag = reinterpret_cast<int*>(Xs + (WITH_V ? 2 : 1) * (size_t)nc * p);
  T* __restrict__ Xg = reinterpret_cast<T*>(d.X);
  T* __restrict__ Vg = reinterpret_cast<T*>(d.V);
  const int tid = threadIdx.x;
  if (WITH_V)
    for (int e = tid; e < p * p; e += 512) {
      const int r = e % p, c = e / p;
      Vg[(size_t)r + (size_t)c * d.ldv] = (r == c) ? sc<T>::one() : sc<T>::zero();
    }
  if (tid == 0) *flag = 0;
  __syncthreads();

  const int nb = (p + bw - 1) / bw;
  const int m = nc - 1, npairs = nc / 2;               // round-robin over nc players
  int tpp = 64;
  while (tpp * npairs > 512) tpp >>= 1;
  const int pair = tid >> (31 - __builtin_clz(tpp)), pl = tid & (tpp - 1);      // tpp is a power of two
  const double tol2 = 1.1e-16 * 1.1e-16 * (double)(p > 16 ? p : 16);      // (floor: see jacobi_kernel)

  int sweep = 0;
  for (; sweep < 60; ++sweep) {
    for (int I = 0; I < nb; ++I) {
      for (int J = (nb == 1 ? 0 : I + 1); J < (nb == 1 ? 1 : nb); ++J) {
        // global column of LDS column c: block I for c < bw, block J after (J == I: second half unused)
        auto gcol = [&](int c) { return c < bw ? I * bw + c : (J == I ? p : J * bw + (c - bw)); };
        for (int e = tid; e < nc * p; e += 512) {
          const int r = e % p, c = e / p, g = gcol(c);
          Xs[e] = g < p ? Xg[(size_t)r + (size_t)g * d.ldx] : sc<T>::zero();
          if (WITH_V) Vs[e] = g < p ? Vg[(size_t)r + (size_t)g * d.ldv] : sc<T>::zero();
        }
        __syncthreads();
        for (int rho = 0; rho < m; ++rho) {
          if (pair < npairs) {
            int i, j;
            if (pair == 0) {
              i = m;
              j = rho;
            } else {
              i = rho + pair;
              j = rho - pair;
              if (i >= m) i -= m;
              if (j < 0) j += m;
            }
            if (i > j) {
              const int t = i;
              i = j;
              j = t;
            }
            if (gcol(i) < p && gcol(j) < p) {
              T* xi = Xs + (size_t)i * p;
              T* xj = Xs + (size_t)j * p;
              double al = 0.0, be = 0.0;
              T ga = sc<T>::zero();
              for (int r = pl; r < p; r += tpp) {
                const T a = xi[r], b = xj[r];
                al += sc<T>::abs2(a);
                be += sc<T>::abs2(b);
                ga = sc<T>::fmacc(ga, a, b);
              }
              group_sum3<T>(al, be, ga, tpp);
              const double g2 = sc<T>::abs2(ga);
              if (g2 > tol2 * al * be && g2 > 0.0) {
                const double g = sqrt(g2);
                const double zeta = (be - al) / (2.0 * g);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                const T ph = sc<T>::scale(sc<T>::conj(ga), 1.0 / g);
                const T sph = sc<T>::scale(ph, s), cph = sc<T>::scale(ph, c);
                for (int r = pl; r < p; r += tpp) {
                  const T a = xi[r], b = xj[r];
                  xi[r] = sc<T>::sub(sc<T>::scale(a, c), sc<T>::mul(sph, b));
                  xj[r] = sc<T>::add(sc<T>::scale(a, s), sc<T>::mul(cph, b));
                }
                if (WITH_V) {
                  T* vi = Vs + (size_t)i * p;
                  T* vj = Vs + (size_t)j * p;
                  for (int r = pl; r < p; r += tpp) {
                    const T a = vi[r], b = vj[r];
                    vi[r] = sc<T>::sub(sc<T>::scale(a, c), sc<T>::mul(sph, b));
                    vj[r] = sc<T>::add(sc<T>::scale(a, s), sc<T>::mul(cph, b));
                  }
                }
                if (pl == 0) atomicAdd(flag, 1);
              }
            }
          }
          __syncthreads();
        }
        for (int e = tid; e < nc * p; e += 512) {
          const int r = e % p, c = e / p, g = gcol(c);
          if (g < p) {
            Xg[(size_t)r + (size_t)g * d.ldx] = Xs[e];
            if (WITH_V) Vg[(size_t)r + (size_t)g * d.ldv] = Vs[e];
          }
        }
        __syncthreads();
      }
    }
    const int rot = *flag;
    __syncthreads();
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (rot == 0) break;
  }
  if (tid == 0 && sweeps_out) sweeps_out[blockIdx.x] = sweep;

  // column norms (reusing the LDS image as scratch), rank by descending norm, sorted output
  double* nrm = reinterpret_cast<double*>(smem);
  __syncthreads();
  for (int c = tid; c < p; c += 512) {
    double s = 0.0;
    for (int r = 0; r < p; ++r) s += sc<T>::abs2(Xg[(size_t)r + (size_t)c * d.ldx]);
    nrm[c] = sqrt(s);
  }
  __syncthreads();
  T* __restrict__ Ug = reinterpret_cast<T*>(d.U);
  double* __restrict__ sg = reinterpret_cast<double*>(d.s);
  for (int e = tid; e < p * p; e += 512) {
    const int r = e % p, c = e / p;
    const double sc_ = nrm[c];
    int rank = 0;
    for (int c2 = 0; c2 < p; ++c2) rank += (nrm[c2] > sc_) || (nrm[c2] == sc_ && c2 < c);
    const bool keep = !(d.thresh2 > 0.0) || sc_ * sc_ >= d.thresh2;
    T v;
    if (WITH_V) v = keep ? Vg[(size_t)r + (size_t)c * d.ldv] : sc<T>::zero();
    else v = sc<T>::scale(Xg[(size_t)r + (size_t)c * d.ldx], (keep && sc_ > 0.0) ? 1.0 / sc_ : 0.0);
    Ug[(size_t)r + (size_t)rank * d.ldu] = v;
    if (r == 0) sg[rank] = sc_;
  }
  if (tid == 0 && d.count) {
    int cnt = 0;
    for (int c = 0; c < p; ++c) cnt += !(d.thresh2 > 0.0) || nrm[c] * nrm[c] >= d.thresh2;
    *reinterpret_cast<int32_t*>(d.count) = cnt;
  }
}

}  // namespace tmf

extern "C" int tmf_jacobi_block_batched(int dtype, int with_v, const tmf_jacobi_desc* d_desc, int nprob, int max_p,
                                        int32_t* d_sweeps, void* stream) {
  using namespace tmf;
  if (nprob <= 0) return TMF_OK;
  const size_t elem = (dtype == TMF_C128) ? 16 : 8;
  if (max_p <= 0 || max_p > 512) {
    set_error("tmf_jacobi_block_batched: p = %d not in 1..512", max_p);
    return TMF_E_LIMIT;
  }
  size_t lds = 0;          // every problem picks its own block width (jacobi_block_width): room for the largest need
  for (int q = 1; q <= max_p; ++q) {
    const size_t need = (size_t)(with_v ? 2 : 1) * 2 * jacobi_block_width(q, with_v != 0, elem) * q * elem + 64;
    lds = need > lds ? need : lds;
  }
  const int bw = 0;        // (kernel argument kept for the launch signature; unused)
  if (lds > 160 * 1024 || lds < (size_t)max_p * 8) {
    set_error("tmf_jacobi_block_batched: p = %d does not fit the LDS staging", max_p);
    return TMF_E_LIMIT;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)jacobi_block_kernel<cd, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)jacobi_block_kernel<cd, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)jacobi_block_kernel<double, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)jacobi_block_kernel<double, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const dim3 g(nprob), b(512);
  if (dtype == TMF_C128) {
    if (with_v) hipLaunchKernelGGL((jacobi_block_kernel<cd, true>), g, b, lds, s, d_desc, bw, d_sweeps);
    else hipLaunchKernelGGL((jacobi_block_kernel<cd, false>), g, b, lds, s, d_desc, bw, d_sweeps);
  } else if (dtype == TMF_F64) {
    if (with_v) hipLaunchKernelGGL((jacobi_block_kernel<double, true>), g, b, lds, s, d_desc, bw, d_sweeps);
    else hipLaunchKernelGGL((jacobi_block_kernel<double, false>), g, b, lds, s, d_desc, bw, d_sweeps);
  } else {
    set_error("tmf_jacobi_block_batched: bad dtype %d", dtype);
    return TMF_E_ARG;
  }
  return check_hip(hipGetLastError(), "tmf_jacobi_block_batched");
}
