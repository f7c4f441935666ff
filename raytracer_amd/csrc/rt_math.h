// Device FP64 vector helpers for the gfx950 render path.
//
// Parity with the reference's CPU arithmetic depends on operation ORDER:
// x86-64 g++ -O3 evaluates the reference's Vec3d expressions
// (src/vecmath/vec.h:241-260,737-785,818-820) with separate IEEE multiplies,
// adds, divides and square roots and never fuses a*b+c.  gfx950 has IEEE
// binary64 add/mul/div/sqrt, so spelling each expression in the same order
// and compiling with -ffp-contract=off reproduces the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace rtd {

struct V3 {
  double x, y, z;
  __device__ __forceinline__ V3() : x(0.0), y(0.0), z(0.0) {}
  __device__ __forceinline__ V3(double a, double b, double c) : x(a), y(b), z(c) {}
  __device__ __forceinline__ explicit V3(const double* p) : x(p[0]), y(p[1]), z(p[2]) {}
  __device__ __forceinline__ double get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

__device__ __forceinline__ V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(const V3& a, double d) { return V3(a.x * d, a.y * d, a.z * d); }
__device__ __forceinline__ V3 operator*(double d, const V3& a) { return V3(a.x * d, a.y * d, a.z * d); }
// vec.h:759-761: (a0*b0 + a1*b1) + a2*b2
__device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vec.h:818-820
__device__ __forceinline__ V3 prod(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ double length2(const V3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ double length(const V3& a) { return sqrt(length2(a)); }
// vec.h:257-260: three divisions by the length
__device__ __forceinline__ V3 normalized(const V3& a) {
  double len = length(a);
  return V3(a.x / len, a.y / len, a.z / len);
}
__device__ __forceinline__ bool iszero(const V3& a) { return a.x == 0 && a.y == 0 && a.z == 0; }
__device__ __forceinline__ bool anynan(const V3& a) { return a.x != a.x || a.y != a.y || a.z != a.z; }

// Affine transform by rows 0..2 of a row-major 4x4 (vec.h:746-751)
__device__ __forceinline__ V3 xformPoint(const double* m, const V3& v) {
  return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3],
            m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7],
            m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]);
}
// Row-major 3x3 times vector (vec.h:764-768)
__device__ __forceinline__ V3 mul33(const double* m, const V3& v) {
  return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z,
            m[3] * v.x + m[4] * v.y + m[5] * v.z,
            m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

// pow(x, y) for an exponent y = n + delta that is a small non-negative integer n
// up to a rounding residue delta (|delta| <= 1e-12; Material::shininess is a
// weighted sum 0.299 s + 0.587 s + 0.114 s of the scene file's number, which
// comes out as 31.999999999999996 for 32), without the general power function,
// whose double-precision library code is some two hundred instructions and
// doubles the registers of the kernel that holds it.  x^n by binary powering in
// double-double arithmetic (products split exactly with fused multiply-adds:
// relative error around 2^-100), times exp(delta ln x) = 1 + delta ln x, whose
// second-order term is below 1e-18; ln x only has to be good to six digits
// (three for the residues scenes really have, a few 1e-15).
// The result is the correctly rounded power in practice (0 misses in 20 000
// random cases against exact rational arithmetic, where the C library's pow
// misses 19 by one unit in the last place).  Special cases as pow: y == 0 gives
// 1 whatever x is; x == 0 gives 0; a negative x with delta != 0 gives NaN; a
// power beyond the binary64 range gives infinity, one below it 0.
__device__ __forceinline__ void ddMul(double& ah, double& al, double bh, double bl) {
  const double p = ah * bh;
  if (!(fabs(p) <= 1.7976931348623157e308)) {   // overflow (or NaN): no error term to form, fma(inf, ., -inf) is NaN
    ah = p;
    al = 0.0;
    return;
  }
  double e = __builtin_fma(ah, bh, -p);
  e = e + (ah * bl + al * bh);
  const double hi = p + e;
  al = e - (hi - p);
  ah = hi;
}
__device__ __forceinline__ double powNearInt(double x, int n, double delta) {
  if (n == 0 && delta == 0.0) return 1.0;
  if (x == 0.0) return (n & 1) && delta == 0.0 ? x : 0.0;
  if (!(fabs(x) <= 1.0e100)) {   // infinities and NaN: plain products do what pow does
    double r = 1.0;
    for (int k = 0; k < n && k < 4096; k++) r = r * x;
    return r;
  }
  double rh = 1.0, rl = 0.0, bh = x, bl = 0.0;
  bool have = false;
  for (int m = n;;) {
    if (m & 1) {
      if (have) ddMul(rh, rl, bh, bl);
      else { rh = bh; rl = bl; have = true; }
    }
    m >>= 1;
    if (!m) break;
    ddMul(bh, bl, bh, bl);
  }
  if (delta != 0.0) {
    // x < 0 with a fractional exponent: the logarithm is NaN, as pow's result.
    // A positive x below binary32's range takes the logarithm of the smallest
    // normal float (its power has underflowed to 0 long before, and 0 times a
    // finite correction stays 0 where 0 * log(0) would be NaN); an overflowed
    // power stays infinite.
    float xf = (float)x;
    if (x > 0.0 && xf < 1.17549435e-38f) xf = 1.17549435e-38f;
    const double c = delta * (double)__logf(xf);
    if (fabs(rh) <= 1.7976931348623157e308) rl = rl + rh * c;
  }
  return rh + rl;
}

// SoA access: component c of element i in an array of `stride` elements per component
__device__ __forceinline__ V3 loadV3(const double* base, size_t stride, size_t i) {
  return V3(base[i], base[stride + i], base[2 * stride + i]);
}
__device__ __forceinline__ void storeV3(double* base, size_t stride, size_t i, const V3& v) {
  base[i] = v.x;
  base[stride + i] = v.y;
  base[2 * stride + i] = v.z;
}

}  // namespace rtd
