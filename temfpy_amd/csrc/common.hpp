// Shared device/host helpers for the gfx950 kernels.  Wavefront = 64 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/temfpy_hip.h"

namespace tmf {

struct __attribute__((aligned(16))) cd {  // complex double, interleaved like numpy complex128
  double x, y;                            // 16-byte aligned: one dwordx4 / ds_*_b128 per element
};

__host__ __device__ inline cd make_cd(double x, double y) {
  cd r;
  r.x = x;
  r.y = y;
  return r;
}

// ---- scalar traits: T = double | cd ------------------------------------------------
template <typename T>
struct sc;

template <>
struct sc<double> {
  static constexpr int cplx = 0;
  __device__ static inline double zero() { return 0.0; }
  __device__ static inline double one() { return 1.0; }
  __device__ static inline double conj(double a) { return a; }
  __device__ static inline double real_only(double a) { return a; }
  __device__ static inline double mul(double a, double b) { return a * b; }
  __device__ static inline double cmul(double a, double b) { return a * b; }  // conj(a)*b
  __device__ static inline double add(double a, double b) { return a + b; }
  __device__ static inline double sub(double a, double b) { return a - b; }
  __device__ static inline double fms(double c, double a, double b) { return fma(-a, b, c); }  // c - a*b
  __device__ static inline double fmac(double c, double a, double b) { return fma(a, b, c); }  // c + a*b
  __device__ static inline double fmacc(double c, double a, double b) { return fma(a, b, c); } // c + conj(a)*b
  __device__ static inline double abs2(double a) { return a * a; }
  __device__ static inline double scale(double a, double s) { return a * s; }
  __device__ static inline double inv(double a) { return 1.0 / a; }
  __device__ static inline double inv_fast(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
  }
  __device__ static inline double real(double a) { return a; }
  __device__ static inline double from_real(double a) { return a; }
  __device__ static inline double from2(double re, double) { return re; }
  __device__ static inline double imag(double) { return 0.0; }
  __device__ static inline double neg(double a) { return -a; }
};

template <>
struct sc<cd> {
  static constexpr int cplx = 1;
  __device__ static inline cd zero() { return make_cd(0.0, 0.0); }
  __device__ static inline cd one() { return make_cd(1.0, 0.0); }
  __device__ static inline cd conj(cd a) { return make_cd(a.x, -a.y); }
  __device__ static inline cd real_only(cd a) { return make_cd(a.x, 0.0); }
  __device__ static inline cd mul(cd a, cd b) { return make_cd(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x)); }
  __device__ static inline cd cmul(cd a, cd b) { return make_cd(fma(a.x, b.x, a.y * b.y), fma(a.x, b.y, -a.y * b.x)); }
  __device__ static inline cd add(cd a, cd b) { return make_cd(a.x + b.x, a.y + b.y); }
  __device__ static inline cd sub(cd a, cd b) { return make_cd(a.x - b.x, a.y - b.y); }
  __device__ static inline cd fms(cd c, cd a, cd b) {
    return make_cd(fma(a.y, b.y, fma(-a.x, b.x, c.x)), fma(-a.y, b.x, fma(-a.x, b.y, c.y)));
  }
  __device__ static inline cd fmac(cd c, cd a, cd b) {
    return make_cd(fma(-a.y, b.y, fma(a.x, b.x, c.x)), fma(a.y, b.x, fma(a.x, b.y, c.y)));
  }
  __device__ static inline cd fmacc(cd c, cd a, cd b) {  // c + conj(a) * b
    return make_cd(fma(a.y, b.y, fma(a.x, b.x, c.x)), fma(-a.y, b.x, fma(a.x, b.y, c.y)));
  }
  __device__ static inline double abs2(cd a) { return fma(a.x, a.x, a.y * a.y); }
  __device__ static inline cd scale(cd a, double s) { return make_cd(a.x * s, a.y * s); }
  __device__ static inline cd inv(cd a) {
    double d = 1.0 / fma(a.x, a.x, a.y * a.y);
    return make_cd(a.x * d, -a.y * d);
  }
  // 1/a with v_rcp_f64 + two Newton steps (no denormal / inf fix-ups: pivots are O(1e-300..1e300))
  __device__ static inline cd inv_fast(cd a) {
    const double q = fma(a.x, a.x, a.y * a.y);
    double r = __builtin_amdgcn_rcp(q);
    r = fma(fma(-q, r, 1.0), r, r);
    r = fma(fma(-q, r, 1.0), r, r);
    return make_cd(a.x * r, -a.y * r);
  }
  __device__ static inline double real(cd a) { return a.x; }
  __device__ static inline cd from_real(double a) { return make_cd(a, 0.0); }
  __device__ static inline cd from2(double re, double im) { return make_cd(re, im); }
  __device__ static inline double imag(cd a) { return a.y; }
  __device__ static inline cd neg(cd a) { return make_cd(-a.x, -a.y); }
};

// ---- wave-level helpers (64 lanes) -------------------------------------------------
__device__ inline double shfl_d(double v, int src, int width = 64) { return __shfl(v, src, width); }
__device__ inline double shfl_xor_d(double v, int mask, int width = 64) { return __shfl_xor(v, mask, width); }

template <typename T>
__device__ inline T shfl_t(T v, int src, int width);
template <>
__device__ inline double shfl_t<double>(double v, int src, int width) {
  return __shfl(v, src, width);
}
template <>
__device__ inline cd shfl_t<cd>(cd v, int src, int width) {
  return make_cd(__shfl(v.x, src, width), __shfl(v.y, src, width));
}
template <typename T>
__device__ inline T shfl_xor_t(T v, int mask, int width);
template <>
__device__ inline double shfl_xor_t<double>(double v, int mask, int width) {
  return __shfl_xor(v, mask, width);
}
template <>
__device__ inline cd shfl_xor_t<cd>(cd v, int mask, int width) {
  return make_cd(__shfl_xor(v.x, mask, width), __shfl_xor(v.y, mask, width));
}

// ---- full-wave reductions on the DPP path (VALU cross-lane moves; __shfl_xor goes through the LDS crossbar:
// ds_bpermute_b32, ~60 cycles of latency per 32-bit word and step, 6 dependent steps per reduction) -----------------
template <int CTRL>
__device__ inline double dpp_mov_keep(double v) {       // lanes without a source keep their own value
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ inline double dpp_mov_zero(double v) {       // lanes without a source read 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double readlane_d(double v, int l) {   // l wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// sum over the 64 lanes, the same value in every lane: inclusive scan inside every row of 16 lanes (row_shr 1, 2, 4, 8),
// lane 15 of a row into the next row (row_bcast:15), lane 31 into rows 2 and 3 (row_bcast:31): lane 63 holds the total
__device__ inline double wave_sum64(double v) {
  v += dpp_mov_zero<0x111>(v);
  v += dpp_mov_zero<0x112>(v);
  v += dpp_mov_zero<0x114>(v);
  v += dpp_mov_zero<0x118>(v);
  v += dpp_mov_zero<0x142>(v);
  v += dpp_mov_zero<0x143>(v);
  return readlane_d(v, 63);
}
__device__ inline cd wave_sum64(cd v) { return make_cd(wave_sum64(v.x), wave_sum64(v.y)); }
// Four sums over the 64 lanes for the price of one and a half: v_permlane32_swap / v_permlane16_swap (gfx950) fold the
// four values into one register whose rows of 16 lanes hold the partial sums of a, c, b, d; one row scan serves all four.
// 29 instructions against 80 for four wave_sum64 (the reductions are what the register QR kernels issue most).
__device__ inline double swap_sum32(double a, double b) {      // lanes 0-31: a(l) + a(l + 32), lanes 32-63: b(l - 32) + b(l)
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ inline double swap_sum16(double a, double b) {      // rows 0, 2: a(row) + a(row + 1); rows 1, 3: b(row - 1) + b(row)
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ inline void wave_sum64x4(double& a, double& b, double& c, double& d) {
  double z = swap_sum16(swap_sum32(a, b), swap_sum32(c, d));      // rows: a, c, b, d
  z += dpp_mov_zero<0x111>(z);
  z += dpp_mov_zero<0x112>(z);
  z += dpp_mov_zero<0x114>(z);
  z += dpp_mov_zero<0x118>(z);
  a = readlane_d(z, 15), c = readlane_d(z, 31), b = readlane_d(z, 47), d = readlane_d(z, 63);
}
// maximum over the 64 lanes (non-NaN input), the same value in every lane
__device__ inline double wave_max64(double v) {
  v = fmax(v, dpp_mov_keep<0x111>(v));
  v = fmax(v, dpp_mov_keep<0x112>(v));
  v = fmax(v, dpp_mov_keep<0x114>(v));
  v = fmax(v, dpp_mov_keep<0x118>(v));
  v = fmax(v, dpp_mov_keep<0x142>(v));
  v = fmax(v, dpp_mov_keep<0x143>(v));
  return readlane_d(v, 63);
}

// sum over aligned groups of W lanes (W = 2 .. 64, a power of two), the group's total in each of its lanes.  Butterfly on
// DPP: xor 1 and xor 2 as quad permutations, xor 4 as row_half_mirror, xor 8 as row_mirror (after the earlier steps all
// lanes of a quad / half row hold the same partial sum, so a mirror pairs the right partners); 32 and 64 through
// row_bcast + readlane.
template <int W>
__device__ inline double group_sum(double v) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group width");
  if constexpr (W >= 2) v += dpp_mov_keep<0xB1>(v);     // quad_perm [1,0,3,2]
  if constexpr (W >= 4) v += dpp_mov_keep<0x4E>(v);     // quad_perm [2,3,0,1]
  if constexpr (W >= 8) v += dpp_mov_keep<0x141>(v);    // row_half_mirror
  if constexpr (W >= 16) v += dpp_mov_keep<0x140>(v);   // row_mirror
  if constexpr (W == 32) {
    const double w = v + dpp_mov_zero<0x142>(v);        // rows 1 and 3: own row + previous row
    const double lo = readlane_d(w, 31), hi = readlane_d(w, 63);
    v = (__lane_id() < 32) ? lo : hi;
  }
  if constexpr (W == 64) {
    double w = v + dpp_mov_zero<0x142>(v);
    w += dpp_mov_zero<0x143>(w);
    v = readlane_d(w, 63);
  }
  return v;
}
template <int W>
__device__ inline cd group_sum(cd v) { return make_cd(group_sum<W>(v.x), group_sum<W>(v.y)); }
// the three sums of a Jacobi rotation (two squared norms and the inner product) over groups of `w` lanes, w run-time
template <typename T>
__device__ inline void group_sum3(double& al, double& be, T& ga, int w) {
  switch (w) {
    case 64: al = group_sum<64>(al), be = group_sum<64>(be), ga = group_sum<64>(ga); break;
    case 32: al = group_sum<32>(al), be = group_sum<32>(be), ga = group_sum<32>(ga); break;
    case 16: al = group_sum<16>(al), be = group_sum<16>(be), ga = group_sum<16>(ga); break;
    case 8: al = group_sum<8>(al), be = group_sum<8>(be), ga = group_sum<8>(ga); break;
    case 4: al = group_sum<4>(al), be = group_sum<4>(be), ga = group_sum<4>(ga); break;
    case 2: al = group_sum<2>(al), be = group_sum<2>(be), ga = group_sum<2>(ga); break;
    default: break;
  }
}

// error plumbing (host)
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
// device flag of the calling thread's conditional-launch scope (tmf_launch_condition), or nullptr
const int32_t* launch_condition();

}  // namespace tmf
