// Device-side surface reconstruction, lights, Phong terms, reflection /
// refraction ray generation and cube-map lookup.  Every function keeps the
// operation order of the reference function it cites.
#pragma once
#include "rt_trace.h"

namespace rtd {

struct MatD {
  V3 ke, ka, ks, kd, kr, kt, shin, index;
  bool shinMapped;
};

__device__ __forceinline__ MatD loadMat(const RtMaterial& m) {
  MatD r;
  r.ke = V3(m.ke); r.ka = V3(m.ka); r.ks = V3(m.ks); r.kd = V3(m.kd);
  r.kr = V3(m.kr); r.kt = V3(m.kt); r.shin = V3(m.shininess); r.index = V3(m.index);
  r.shinMapped = m.shininess_mapped != 0;
  return r;
}

// MaterialParameter::intensityValue (material.cpp:158-167)
__device__ __forceinline__ double intensityValue(const V3& v) {
  return (0.299 * v.x) + (0.587 * v.y) + (0.114 * v.z);
}
// Material::shininess (material.h:209-215)
__device__ __forceinline__ double shininessOf(const MatD& m) {
  return m.shinMapped ? 128.0 * intensityValue(m.shin) : intensityValue(m.shin);
}

struct Surface {
  V3 N, bary;
  int objectID;
};

// Interpolated per-vertex material: Material() + b0*A + b1*B + b2*C where the
// default-constructed Material has index (1,1,1) and everything else zero
// (material.h:162-172,183-195,277-289; trimesh.cpp:189-199).
__device__ __forceinline__ MatD interpMaterial(const DevScene& S, int faceIdx, const V3& bary) {
  MatD pm;
  pm.index = V3(1.0, 1.0, 1.0);
  pm.shinMapped = false;
  const double w[3] = {bary.x, bary.y, bary.z};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const MatD m = loadMat(S.materials[S.vmaterials[S.face_ids[3 * faceIdx + k]]]);
    pm.ke = pm.ke + m.ke * w[k];
    pm.ka = pm.ka + m.ka * w[k];
    pm.ks = pm.ks + m.ks * w[k];
    pm.kd = pm.kd + m.kd * w[k];
    pm.kr = pm.kr + m.kr * w[k];
    pm.kt = pm.kt + m.kt * w[k];
    pm.index = pm.index + m.index * w[k];
    pm.shin = pm.shin + m.shin * w[k];
  }
  return pm;
}

// isect of a winning triangle: barycentrics, interpolated or face normal in
// world space, objectID (trimesh.cpp:173-204 after Geometry::intersect,
// scene.cpp:13-36).  wantN false skips the normal.
__device__ __forceinline__ void surfaceAtFace(const DevScene& S, int fi, const V3& p, const V3& d, Surface& sf,
                                              bool wantN) {
  const DevTri& T = S.tris[fi];
  const RtObject& o = S.objects[T.owner];
  const RtMesh& m = S.meshes[o.mesh];
  const LocalRay lr = toLocal(o.inverse, p, d);
  double rt;
  V3 bary;
  triTest<false>(T, lr.p, lr.d, false, rt, &bary);
  sf.bary = bary;
  sf.objectID = T.object_id;
  if (wantN) {
    V3 Nl;
    if (S.smooth && m.has_vertex_normals) {
      const double* fn = S.face_normals + 9 * (size_t)fi;
      const V3 nA(fn), nB(fn + 3), nC(fn + 6);
      Nl = (bary.x * nA) + (bary.y * nB) + (bary.z * nC);
    } else {
      Nl = V3(T.n);
    }
    Nl = normalized(Nl);
    sf.N = normalized(mul33(o.normi, Nl));
  }
}

// surfaceAtFace's normal for a face whose RT_FI_* word is at hand.  A face of an
// identity-transform mesh goes through the same arithmetic with the identity
// written out (Geometry::intersect multiplies by the matrices whatever they
// hold, scene.cpp:17-20, scene.h:168-172) instead of loading its owner.
__device__ __forceinline__ void surfaceNormalOfFace(const DevScene& S, int fi, int info, const V3& p, const V3& d,
                                                    Surface& sf) {
  const DevTri& T = S.tris[fi];
  const bool smoothN = S.smooth && (info & RT_FI_VNORMALS);
  sf.objectID = T.object_id;
  V3 Nl = V3(T.n);
  if (T.axes & 16) {
    const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (smoothN) {
      const LocalRay lr = toLocal(I4, p, d);
      double rt;
      V3 bary;
      triTest<false>(T, lr.p, lr.d, false, rt, &bary);
      const double* fn = S.face_normals + 9 * (size_t)fi;
      Nl = (bary.x * V3(fn)) + (bary.y * V3(fn + 3)) + (bary.z * V3(fn + 6));
    }
    sf.N = normalized(mul33(I3, normalized(Nl)));
  } else {
    const RtObject& o = S.objects[T.owner];
    if (smoothN) {
      const LocalRay lr = toLocal(o.inverse, p, d);
      double rt;
      V3 bary;
      triTest<false>(T, lr.p, lr.d, false, rt, &bary);
      const double* fn = S.face_normals + 9 * (size_t)fi;
      Nl = (bary.x * V3(fn)) + (bary.y * V3(fn + 3)) + (bary.z * V3(fn + 6));
    }
    sf.N = normalized(mul33(o.normi, normalized(Nl)));
  }
}

// surfaceNormalOfFace for a face of an identity-transform mesh whose record T
// and corner normals fn (nine doubles; read only where the face has vertex
// normals) the caller has already loaded: the same arithmetic, no memory
// access in between.
__device__ __forceinline__ void surfaceNormalLoaded(const DevScene& S, const DevTri& T, const double* fn, int info,
                                                    const V3& p, const V3& d, Surface& sf) {
  const bool smoothN = S.smooth && (info & RT_FI_VNORMALS);
  sf.objectID = T.object_id;
  V3 Nl = V3(T.n);
  const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (smoothN) {
    const LocalRay lr = toLocal(I4, p, d);
    double rt;
    V3 bary;
    triTest<false>(T, lr.p, lr.d, false, rt, &bary);
    Nl = (bary.x * V3(fn)) + (bary.y * V3(fn + 3)) + (bary.z * V3(fn + 6));
  }
  sf.N = normalized(mul33(I3, normalized(Nl)));
}

// Full isect of the winning primitive: what Geometry::intersect +
// <Primitive>::intersectLocal left in `i` for the closest candidate
// (scene.cpp:13-36; trimesh.cpp:173-204).  A null `mat` skips the material;
// ktOnly skips the normal (shadow attenuation only needs the material).
__device__ __forceinline__ void surfaceAt(const DevScene& S, int prim, const V3& p, const V3& d,
                                          int rayType, Surface& sf, MatD* mat, bool ktOnly) {
  if (prim & RT_PRIM_FACE) {
    const int fi = prim & ~RT_PRIM_FACE;
    surfaceAtFace(S, fi, p, d, sf, !ktOnly);
    if (mat) {
      const RtObject& o = S.objects[S.tris[fi].owner];
      if (S.smooth && S.meshes[o.mesh].has_vertex_materials) *mat = interpMaterial(S, fi, sf.bary);
      else *mat = loadMat(S.materials[o.material]);
    }
  } else {
    const RtObject& o = S.objects[prim];
    double t;
    V3 N;
    objectIntersect<true>(o, p, d, t, N);
    sf.N = N;
    sf.bary = V3();
    sf.objectID = o.object_id;
    if (mat) *mat = loadMat(S.materials[o.material]);
  }
  (void)rayType;
}

// TextureMap::getPixelAt (material.cpp:130-148)
__device__ __forceinline__ V3 cubePixel(const DevScene& S, int face, int x, int y) {
  if (x >= S.cube_w) x = S.cube_w - 1;
  if (y >= S.cube_h) y = S.cube_h - 1;
  const unsigned char* data = S.cube[face] + ((size_t)y * S.cube_w + x) * 3;
  return V3(double(data[0]) / 255.0, double(data[1]) / 255.0, double(data[2]) / 255.0);
}

// ---- CubeMap::getColor (cubeMap.cpp:6-208) -----------------------------------
// What the reference spells as nested branches is a table: for each of the six
// faces a direction can look at (axis, sign), the map slot of that face, and for
// a filter tap that leaves it across its bottom / top / left / right edge the
// slot of the neighbouring map and where the tap lands in it.  A landing
// coordinate is always  base + sign * variable  with base one of
// 0, -1, W, W-1, H, H-1 and the variable one of the tap's in-face indices
// (xi, yi) or its loop offsets (ii, jj) -- the reference uses the offsets, not
// the indices, along the crossed edge (cubeMap.cpp:125-205).
//
// Landing code: bits 0-2 base (bit 0: minus one; bits 1-2: 0 / W / H),
// bits 3-5 variable (0 none, 1 xi, 2 yi, 3 ii, 4 jj), bit 6 negative.
#define RT_CL(base, var, neg) ((base) | ((var) << 3) | ((neg) << 6))
#define RT_CL_0 0
#define RT_CL_M1 1
#define RT_CL_W 2
#define RT_CL_W1 3
#define RT_CL_H 4
#define RT_CL_H1 5
#define RT_CL_XI 1
#define RT_CL_YI 2
#define RT_CL_II 3
#define RT_CL_JJ 4
struct CubeEdge { unsigned char slot, x, y; };
struct CubeFace { unsigned char front; CubeEdge edge[4]; };   // edge: 0 bottom (yi < 0), 1 top, 2 left (xi < 0), 3 right
#define RT_CL_SAME_X RT_CL(RT_CL_0, RT_CL_XI, 0)
#define RT_CL_SAME_Y RT_CL(RT_CL_0, RT_CL_YI, 0)
__constant__ CubeFace kCubeFaces[6] = {
    // looking down +x: slot 0; below slot 3, above slot 2, left slot 4, right slot 5
    {0, {{3, RT_CL(RT_CL_W, RT_CL_JJ, 0), RT_CL(RT_CL_H1, RT_CL_XI, 1)}, {2, RT_CL(RT_CL_W, RT_CL_JJ, 1), RT_CL_SAME_X},
         {4, RT_CL(RT_CL_W, RT_CL_II, 0), RT_CL_SAME_Y}, {5, RT_CL(RT_CL_M1, RT_CL_II, 0), RT_CL_SAME_Y}}},
    // -x: slot 1
    {1, {{3, RT_CL(RT_CL_M1, RT_CL_JJ, 1), RT_CL_SAME_X}, {2, RT_CL(RT_CL_M1, RT_CL_JJ, 0), RT_CL(RT_CL_H1, RT_CL_XI, 1)},
         {5, RT_CL(RT_CL_W, RT_CL_II, 0), RT_CL_SAME_Y}, {4, RT_CL(RT_CL_M1, RT_CL_II, 0), RT_CL_SAME_Y}}},
    // +y: slot 2; below slot 4, above slot 5, left slot 1, right slot 0
    {2, {{4, RT_CL_SAME_X, RT_CL(RT_CL_H, RT_CL_JJ, 0)}, {5, RT_CL(RT_CL_W1, RT_CL_XI, 1), RT_CL(RT_CL_H, RT_CL_JJ, 1)},
         {1, RT_CL(RT_CL_W1, RT_CL_YI, 1), RT_CL(RT_CL_H, RT_CL_II, 0)}, {0, RT_CL_SAME_Y, RT_CL(RT_CL_H, RT_CL_II, 1)}}},
    // -y: slot 3; below slot 5, above slot 4
    {3, {{5, RT_CL(RT_CL_W1, RT_CL_XI, 1), RT_CL(RT_CL_M1, RT_CL_JJ, 1)}, {4, RT_CL_SAME_X, RT_CL(RT_CL_M1, RT_CL_JJ, 0)},
         {1, RT_CL_SAME_Y, RT_CL(RT_CL_M1, RT_CL_II, 1)}, {0, RT_CL(RT_CL_W1, RT_CL_YI, 1), RT_CL(RT_CL_M1, RT_CL_II, 0)}}},
    // +z reads slot 5 (the reference's "Zneg" slot, cubeMap.cpp:32-41,63-72); left slot 0, right slot 1
    {5, {{3, RT_CL(RT_CL_W1, RT_CL_XI, 1), RT_CL(RT_CL_M1, RT_CL_JJ, 1)}, {2, RT_CL(RT_CL_W1, RT_CL_XI, 1), RT_CL(RT_CL_H, RT_CL_JJ, 1)},
         {0, RT_CL(RT_CL_W, RT_CL_II, 0), RT_CL_SAME_Y}, {1, RT_CL(RT_CL_M1, RT_CL_II, 0), RT_CL_SAME_Y}}},
    // -z reads slot 4; left slot 1, right slot 0
    {4, {{3, RT_CL_SAME_X, RT_CL(RT_CL_H, RT_CL_JJ, 0)}, {2, RT_CL_SAME_X, RT_CL(RT_CL_M1, RT_CL_JJ, 0)},
         {1, RT_CL(RT_CL_W, RT_CL_II, 0), RT_CL_SAME_Y}, {0, RT_CL(RT_CL_M1, RT_CL_II, 0), RT_CL_SAME_Y}}},
};

__device__ __forceinline__ int cubeLanding(int code, int W, int H, int xi, int yi, int ii, int jj) {
  const int span = (code >> 1) & 3;
  const int base = (span == 1 ? W : (span == 2 ? H : 0)) - (code & 1);
  const int which = (code >> 3) & 7;
  const int var = which == 1 ? xi : (which == 2 ? yi : (which == 3 ? ii : (which == 4 ? jj : 0)));
  return (code & 64) ? base - var : base + var;
}

// A NaN direction (refraction past the critical angle, RayTracer.cpp:193-199)
// makes the reference skip every filter tap and return 0/0, reproduced here as
// a NaN colour.
__device__ V3 cubeColor(const DevScene& S, const V3& dir, int filterwidth) {
  if (anynan(dir)) {
    const double q = __longlong_as_double(0x7ff8000000000000LL);
    return V3(q, q, q);
  }
  // the face: the axis with the strictly largest magnitude, z on ties (cubeMap.cpp:12-73)
  const double ax = fabs(dir.x), ay = fabs(dir.y), az = fabs(dir.z);
  const int axis = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  const double major = axis == 0 ? dir.x : (axis == 1 ? dir.y : dir.z);
  const bool pos = major > 0.0;
  const CubeFace& F = kCubeFaces[2 * axis + (pos ? 0 : 1)];
  // face coordinates (cubeMap.cpp:75-92): one quotient as it is, the other negated on the negative side
  double u, v;
  if (axis == 0) {
    u = dir.z / dir.x;
    v = pos ? dir.y / dir.x : -dir.y / dir.x;
  } else if (axis == 1) {
    u = pos ? dir.x / dir.y : -dir.x / dir.y;
    v = dir.z / dir.y;
  } else {
    u = -dir.x / dir.z;
    v = pos ? dir.y / dir.z : -dir.y / dir.z;
  }
  u = (u + 1.0) / 2.0;
  v = (v + 1.0) / 2.0;
  const int W = S.cube_w, H = S.cube_h;
  const int before = (filterwidth + 1) / 2 - 1, after = filterwidth - before;   // taps -before .. after-1
  const double texelU = 1.0 / W, texelV = 1.0 / H;
  const int x = (int)floor(0.5 + u * (W - 1));
  const int y = (int)floor(0.5 + v * (H - 1));
  const double firstU = texelU * (x - before), firstV = texelV * (y - before);
  const double slopeU = 1.0 / (((double)filterwidth + 1) / 2.0 * texelU);
  const double slopeV = 1.0 / (((double)filterwidth + 1) / 2.0 * texelV);
  double total = 0.0;
  V3 sum(0.0, 0.0, 0.0);
  double tapV = firstV;
  for (int jj = -before; jj < after; jj++) {
    double tapU = firstU;
    for (int ii = -before; ii < after; ii++) {
      const int xi = x + ii, yi = y + jj;
      const bool below = yi < 0, above = yi >= H, leftOf = xi < 0, rightOf = xi >= W;
      if ((below || above) && (leftOf || rightOf)) continue;   // a corner belongs to no face (cubeMap.cpp:117-120)
      int slot = F.front, tx = xi, ty = yi;
      if (below || above || leftOf || rightOf) {
        const CubeEdge e = F.edge[below ? 0 : (above ? 1 : (leftOf ? 2 : 3))];
        slot = e.slot;
        tx = cubeLanding(e.x, W, H, xi, yi, ii, jj);
        ty = cubeLanding(e.y, W, H, xi, yi, ii, jj);
      }
      // the reference indexes the texel array unguarded below zero; clamp so a
      // landing coordinate can never leave the face
      if (tx < 0) tx = 0;
      if (ty < 0) ty = 0;
      const double du = u - tapU, dv = v - tapV;
      const double weight = (1.0 - fabs(du * slopeU)) * (1.0 - fabs(dv * slopeV));   // tent
      tapU += texelU;
      total += weight;
      sum = sum + cubePixel(S, slot, tx, ty) * weight;
    }
    tapV += texelV;
  }
  return V3(sum.x / total, sum.y / total, sum.z / total);
}

// Light::getDirection followed by the extra normalize in Material::shade
// (light.cpp:33-37,59-64,105-110; material.cpp:56-57)
__device__ __forceinline__ V3 lightDirection(const RtLight& l, const V3& P) {
  if (l.type == RT_LIGHT_DIRECTIONAL) return -V3(l.orientation);
  return normalized(V3(l.position) - P);
}

// PointLight::distanceAttenuation (light.cpp:39-52); 1 for the other kinds
__device__ __forceinline__ double distanceAttenuation(const RtLight& l, const V3& P) {
  if (l.type != RT_LIGHT_POINT) return 1.0;
  V3 direction = V3(l.position) - P;
  double distanceSq = length2(direction);
  double distance = sqrt(distanceSq);
  return fmin(1.0, 1 / (l.constant_term + l.linear_term * distance + l.quadratic_term * distanceSq));
}

}  // namespace rtd
