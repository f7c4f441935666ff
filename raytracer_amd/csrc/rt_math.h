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
