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

// CubeMap::getColor (cubeMap.cpp:6-208).  A NaN direction (refraction past the
// critical angle, RayTracer.cpp:193-199) makes the reference skip every filter
// tap and return 0/0, reproduced here as a NaN colour.
__device__ V3 cubeColor(const DevScene& S, const V3& dir, int filterwidth) {
  if (anynan(dir)) {
    const double q = __longlong_as_double(0x7ff8000000000000LL);
    return V3(q, q, q);
  }
  int axis, front, left = 0, right = 0, top = 0, bottom = 0;
  double u, v;
  if (fabs(dir.x) > fabs(dir.y)) {
    if (fabs(dir.x) > fabs(dir.z)) {
      axis = 0; bottom = 3; top = 2;
      if (dir.x > 0.0) { front = 0; left = 4; right = 5; } else { front = 1; left = 5; right = 4; }
    } else {
      axis = 2; bottom = 3; top = 2;
      if (dir.z > 0.0) { front = 5; left = 0; right = 1; } else { front = 4; left = 1; right = 0; }
    }
  } else {
    if (fabs(dir.y) > fabs(dir.z)) {
      axis = 1; left = 1; right = 0;
      if (dir.y > 0.0) { front = 2; bottom = 4; top = 5; } else { front = 3; bottom = 5; top = 4; }
    } else {
      axis = 2; bottom = 3; top = 2;
      if (dir.z > 0.0) { front = 5; left = 0; right = 1; } else { front = 4; left = 1; right = 0; }
    }
  }
  const bool pos = (axis == 0) ? (dir.x > 0) : ((axis == 1) ? (dir.y > 0) : (dir.z > 0));
  if (axis == 0) {
    u = dir.z / dir.x;
    v = (dir.x > 0.0) ? dir.y / dir.x : -dir.y / dir.x;
  } else if (axis == 1) {
    u = (dir.y > 0.0) ? dir.x / dir.y : -dir.x / dir.y;
    v = dir.z / dir.y;
  } else {
    u = -dir.x / dir.z;
    v = (dir.z > 0.0) ? dir.y / dir.z : -dir.y / dir.z;
  }
  u = (u + 1.0) / 2.0;
  v = (v + 1.0) / 2.0;
  const int fw = (filterwidth + 1) / 2 - 1;
  const int rm = filterwidth - fw;
  const int width = S.cube_w, height = S.cube_h;
  const double delu = 1.0 / width, delv = 1.0 / height;
  const int x = (int)floor(0.5 + u * (width - 1));
  const int y = (int)floor(0.5 + v * (height - 1));
  const double startu = delu * (x - fw), startv = delv * (y - fw);
  double nextu = startu, nextv = startv;
  const double xslope = 1.0 / (((double)filterwidth + 1) / 2.0 * delu);
  const double yslope = 1.0 / (((double)filterwidth + 1) / 2.0 * delv);
  double correct = 0.0;
  V3 thePixel(0.0, 0.0, 0.0);
  for (int jj = -fw; jj < rm; jj++) {
    for (int ii = -fw; ii < rm; ii++) {
      int xindex = x + ii, yindex = y + jj;
      if (xindex < 0 && yindex < 0) continue;
      if (xindex >= width && yindex >= height) continue;
      if (xindex >= width && yindex < 0) continue;
      if (xindex < 0 && yindex >= height) continue;
      int theMap = front;
      if (yindex < 0) {
        theMap = bottom;
        if (axis == 0) {
          if (pos) { yindex = height - 1 - xindex; xindex = width + jj; }
          else { yindex = xindex; xindex = -1 - jj; }
        } else if (axis == 1) {
          if (pos) yindex = height + jj;
          else { yindex = -1 - jj; xindex = width - 1 - xindex; }
        } else {
          if (pos) { yindex = -1 - jj; xindex = width - 1 - xindex; }
          else yindex = height + jj;
        }
      } else if (yindex >= height) {
        theMap = top;
        if (axis == 0) {
          if (pos) { yindex = xindex; xindex = width - jj; }
          else { yindex = height - 1 - xindex; xindex = jj - 1; }
        } else if (axis == 1) {
          if (pos) { yindex = height - jj; xindex = width - 1 - xindex; }
          else yindex = jj - 1;
        } else {
          if (pos) { yindex = height - jj; xindex = width - 1 - xindex; }
          else yindex = jj - 1;
        }
      } else if (xindex < 0) {
        theMap = left;
        if (axis == 0 || axis == 2) xindex = width + ii;
        else {
          if (pos) { xindex = width - 1 - yindex; yindex = height + ii; }
          else { xindex = yindex; yindex = -1 - ii; }
        }
      } else if (xindex >= width) {
        theMap = right;
        if (axis == 0 || axis == 2) xindex = ii - 1;
        else {
          if (pos) { xindex = yindex; yindex = height - ii; }
          else { xindex = width - 1 - yindex; yindex = ii - 1; }
        }
      }
      // the reference indexes the texel array unguarded below zero; clamp so a
      // remapped index can never leave the face
      if (xindex < 0) xindex = 0;
      if (yindex < 0) yindex = 0;
      const double du = u - nextu, dv = v - nextv;
      const double weight = (1.0 - fabs(du * xslope)) * (1.0 - fabs(dv * yslope));
      nextu += delu;
      correct += weight;
      thePixel = thePixel + cubePixel(S, theMap, xindex, yindex) * weight;
    }
    nextu = startu;
    nextv += delv;
  }
  return V3(thePixel.x / correct, thePixel.y / correct, thePixel.z / correct);
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
