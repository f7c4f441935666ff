// Host-side FP64 vector/matrix helpers.
//
// Every function spells out the operation order of the reference's header-only
// templates (src/vecmath/vec.h:198-303,718-848, src/vecmath/mat.h:80-119,263-302,
// 449-482) so that transforms, bounding boxes and camera frames computed here
// carry the same bits as the reference computes for the same scene file.
// Build with -ffp-contract=off: x86-64 g++ -O3 never fuses a*b+c for the
// reference, so no fused multiply-add may appear here either.
#pragma once
#include <cmath>
#include <cstring>
#include <utility>

namespace rta {

struct Vec3 {
  double n[3];
  Vec3() : n{0.0, 0.0, 0.0} {}
  Vec3(double x, double y, double z) : n{x, y, z} {}
  double& operator[](int i) { return n[i]; }
  double operator[](int i) const { return n[i]; }
};

struct Vec4 {
  double n[4];
  Vec4() : n{0.0, 0.0, 0.0, 0.0} {}
  Vec4(double x, double y, double z, double w) : n{x, y, z, w} {}
  double& operator[](int i) { return n[i]; }
  double operator[](int i) const { return n[i]; }
};

inline Vec3 add(const Vec3& a, const Vec3& b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3 sub(const Vec3& a, const Vec3& b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3 scale(const Vec3& a, double d) { return Vec3(a[0] * d, a[1] * d, a[2] * d); }
inline Vec3 neg(const Vec3& a) { return Vec3(-a[0], -a[1], -a[2]); }
// vec.h:759-761: left-to-right sum of the three products
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// vec.h:781-785
inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline double length2(const Vec3& a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }
inline double length(const Vec3& a) { return std::sqrt(length2(a)); }
// vec.h:257-260: three divisions by the length, not a reciprocal multiply
inline Vec3 normalized(const Vec3& a) {
  double len = length(a);
  return Vec3(a[0] / len, a[1] / len, a[2] / len);
}
inline bool iszero(const Vec3& a) { return a[0] == 0 && a[1] == 0 && a[2] == 0; }
inline Vec3 vmin(const Vec3& a, const Vec3& b) {
  return Vec3(std::min(a[0], b[0]), std::min(a[1], b[1]), std::min(a[2], b[2]));
}
inline Vec3 vmax(const Vec3& a, const Vec3& b) {
  return Vec3(std::max(a[0], b[0]), std::max(a[1], b[1]), std::max(a[2], b[2]));
}

struct Mat3 {
  double n[9];
  Mat3() { std::memset(n, 0, sizeof n); n[0] = n[4] = n[8] = 1.0; }
  double* operator[](int r) { return &n[r * 3]; }
  const double* operator[](int r) const { return &n[r * 3]; }
  Mat3 transpose() const {
    Mat3 t;
    const int idx[9] = {0, 3, 6, 1, 4, 7, 2, 5, 8};
    for (int i = 0; i < 9; i++) t.n[i] = n[idx[i]];
    return t;
  }
};

struct Mat4 {
  double n[16];
  Mat4() { std::memset(n, 0, sizeof n); n[0] = n[5] = n[10] = n[15] = 1.0; }
  double* operator[](int r) { return &n[r * 4]; }
  const double* operator[](int r) const { return &n[r * 4]; }
  Mat3 upper33() const {
    Mat3 m;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) m[r][c] = n[r * 4 + c];
    return m;
  }
};

// Gauss-Jordan elimination with partial pivoting on an N x N row-major matrix,
// same pivot search (strict > on fabs), swap, scale and eliminate sequence as
// mat.h:80-119 (N=3) and mat.h:263-302 (N=4).
template <int N>
inline void gaussJordanInverse(const double* src, double* dst) {
  double a[N * N], b[N * N];
  std::memcpy(a, src, sizeof a);
  for (int i = 0; i < N * N; i++) b[i] = 0.0;
  for (int i = 0; i < N; i++) b[i * N + i] = 1.0;
  for (int j = 0; j < N; j++) {
    int piv = j;
    for (int i = j + 1; i < N; i++)
      if (std::fabs(a[i * N + j]) > std::fabs(a[piv * N + j])) piv = i;
    for (int i = 0; i < N; i++) {
      std::swap(a[piv * N + i], a[j * N + i]);
      std::swap(b[piv * N + i], b[j * N + i]);
    }
    double s = a[j * N + j];
    for (int i = 0; i < N; i++) {
      b[j * N + i] /= s;
      a[j * N + i] /= s;
    }
    for (int i = 0; i < N; i++) {
      if (i == j) continue;
      s = a[i * N + j];
      for (int k = 0; k < N; k++) {
        b[i * N + k] -= s * b[j * N + k];
        a[i * N + k] -= s * a[j * N + k];
      }
    }
  }
  std::memcpy(dst, b, sizeof b);
}

inline Mat4 inverse(const Mat4& m) { Mat4 r; gaussJordanInverse<4>(m.n, r.n); return r; }
inline Mat3 inverse(const Mat3& m) { Mat3 r; gaussJordanInverse<3>(m.n, r.n); return r; }

// mat.h:516-533: each entry is the left-to-right sum of four products
inline Mat4 mul(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j];
  return r;
}
// vec.h:746-751: affine point transform, rows 0..2 only
inline Vec3 mulPoint(const Mat4& a, const Vec3& v) {
  return Vec3(a.n[0] * v[0] + a.n[1] * v[1] + a.n[2] * v[2] + a.n[3],
              a.n[4] * v[0] + a.n[5] * v[1] + a.n[6] * v[2] + a.n[7],
              a.n[8] * v[0] + a.n[9] * v[1] + a.n[10] * v[2] + a.n[11]);
}
// vec.h:843-848
inline Vec4 mul(const Mat4& a, const Vec4& v) {
  return Vec4(a.n[0] * v[0] + a.n[1] * v[1] + a.n[2] * v[2] + a.n[3] * v[3],
              a.n[4] * v[0] + a.n[5] * v[1] + a.n[6] * v[2] + a.n[7] * v[3],
              a.n[8] * v[0] + a.n[9] * v[1] + a.n[10] * v[2] + a.n[11] * v[3],
              a.n[12] * v[0] + a.n[13] * v[1] + a.n[14] * v[2] + a.n[15] * v[3]);
}
// vec.h:764-768
inline Vec3 mul(const Mat3& a, const Vec3& v) {
  return Vec3(a.n[0] * v[0] + a.n[1] * v[1] + a.n[2] * v[2],
              a.n[3] * v[0] + a.n[4] * v[1] + a.n[5] * v[2],
              a.n[6] * v[0] + a.n[7] * v[1] + a.n[8] * v[2]);
}

inline Mat4 translation(double x, double y, double z) {
  Mat4 m; m[0][3] = x; m[1][3] = y; m[2][3] = z; return m;
}
inline Mat4 scaling(double x, double y, double z) {
  Mat4 m; m[0][0] = x; m[1][1] = y; m[2][2] = z; return m;
}
// mat.h:449-461: the axis arrives as float (three narrowing conversions) and is
// normalised in double before the Rodrigues entries are formed.
inline Mat4 rotation(double angle, float fx, float fy, float fz) {
  double c = std::cos(angle), s = std::sin(angle), t = 1.0 - c;
  Vec3 a = normalized(Vec3(fx, fy, fz));
  Mat4 m;
  m[0][0] = t * a[0] * a[0] + c;        m[0][1] = t * a[0] * a[1] - s * a[2]; m[0][2] = t * a[0] * a[2] + s * a[1];
  m[1][0] = t * a[0] * a[1] + s * a[2]; m[1][1] = t * a[1] * a[1] + c;        m[1][2] = t * a[1] * a[2] - s * a[0];
  m[2][0] = t * a[0] * a[2] - s * a[1]; m[2][1] = t * a[1] * a[2] + s * a[0]; m[2][2] = t * a[2] * a[2] + c;
  return m;
}

}  // namespace rta
