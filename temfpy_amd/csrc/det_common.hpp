// Device helpers shared by det_gather.hip and det_reduced.hip.
#pragma once
#include "common.hpp"

namespace tmf {

__device__ inline double sel(bool m, double a, double b) { return m ? a : b; }
__device__ inline cd sel(bool m, cd a, cd b) { return make_cd(m ? a.x : b.x, m ? a.y : b.y); }

// max over the G lanes of a group, result in every lane.  Rows of 16 lanes use DPP
// (quad_perm xor 1, xor 2, then row rotations), no LDS traffic.
template <int G>
__device__ inline unsigned group_max(unsigned k) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
  k = o > k ? o : k;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
  k = o > k ? o : k;
  if (G == 8) {
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x141, 0xF, 0xF, false);  // row_half_mirror
    k = o > k ? o : k;
  } else {
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x124, 0xF, 0xF, false);  // row_ror:4
    k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x128, 0xF, 0xF, false);  // row_ror:8
    k = o > k ? o : k;
    if (G == 32) {
      o = (unsigned)__shfl_xor((int)k, 16, 32);
      k = o > k ? o : k;
    }
  }
  return k;
}

// One determinant per G-lane group: lane c holds column c of the N x N minor in registers
// (N is a template parameter: straight-line code, no predicates on the row index).
// Gaussian elimination by COLUMN operations with column pivoting: at step j the pivot is the
// largest |entry| of row j among the unused columns, so "swapping" is a change of lane roles
// and costs nothing (a row swap would need a select chain over a register array).  The pivot
// column is handed to the other lanes through a per-group LDS scratch (one ds_read_b128 per
// element instead of four ds_bpermute).  det = prod(pivots) * sign(permutation).
template <typename T, int N, int G>
__device__ inline T det_group(T (&a)[N], const int c, T* __restrict__ scratch) {
  T det = sc<T>::one();
  unsigned used = 0u;         // columns already used as pivots (uniform inside the group)
  bool mine_used = c >= N;    // padding lanes never take part
  constexpr unsigned valid = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    // pivot = arg max |a[j]| over unused lanes; key = (float bits >> 1, lane), max-reduced
    unsigned key = 0u;
    if (!mine_used) {
      const float m = (float)sc<T>::abs2(a[j]);
      key = ((__float_as_uint(m) >> 1) & ~(unsigned)(G - 1)) | (unsigned)c | 0x80000000u;
    }
    key = group_max<G>(key);
    const int piv = (int)(key & (unsigned)(G - 1));
    const bool is_piv = (c == piv);
    if (is_piv) {
#pragma unroll
      for (int r = j; r < N; ++r) scratch[r] = a[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const T p = scratch[j];
    det = sc<T>::mul(det, p);
    if (__popc(~used & ((1u << piv) - 1u) & valid) & 1) det = sc<T>::neg(det);
    used |= 1u << piv;
    mine_used = mine_used || is_piv;
    const T pinv = sc<T>::abs2(p) > 0.0 ? sc<T>::inv_fast(p) : sc<T>::zero();
    const T m = sc<T>::mul(a[j], pinv);
#pragma unroll
    for (int r = j + 1; r < N; ++r) a[r] = sc<T>::fms(a[r], m, scratch[r]);
    __builtin_amdgcn_wave_barrier();  // next step's pivot column overwrites the scratch
  }
  return det;
}

}  // namespace tmf
