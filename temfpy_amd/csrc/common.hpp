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

// error plumbing (host)
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
// device flag of the calling thread's conditional-launch scope (tmf_launch_condition), or nullptr
const int32_t* launch_condition();

}  // namespace tmf
