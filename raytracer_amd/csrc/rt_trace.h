// Device-side scene layout, primitive tests and the two-level KD traversal.
//
// Closest-hit query = Scene::intersect (src/scene/scene.cpp:133-159).  Two
// flavours share the primitive arithmetic:
//   REFERENCE  the caller's KD-trees walked with Scene::intersectKdTree's exact
//              push rules, far child first, no early exit, per-candidate
//              bounding-box cull through Geometry::intersect
//              (scene.cpp:59-129,13-36).
//   FAST       the library's own bounding-volume hierarchy (rt_bvh.h), near
//              child first, subtrees culled once they start beyond the best
//              hit, no per-face box cull, barycentric acceptance without
//              divisions, optional any-hit termination for shadow rays.
// Both compute t of an accepted hit with the same operations, so when they pick
// the same primitive the hit is bit-identical.
#pragma once
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_math.h"

namespace rtd {

#define RT_EPS 0.00000001  // RAY_EPSILON, src/scene/ray.h:114

// 16-byte KD node.  Internal: a = (left << 2) | dim, b = right.
// Leaf: a = (first_item << 2) | 3, b = item count.
struct DevNode {
  double split;
  uint32_t a, b;
};

// 128-byte triangle record = one TrimeshFace (trimesh.h:82-149) with the
// ray-independent parts of TrimeshFace::intersectLocal hoisted:
// dconst = -(v0 . n) (trimesh.cpp:122), abc = projected area of the triangle
// (trimesh.cpp:162), ax/ay = the two kept axes (trimesh.cpp:138-153).
struct DevTri {
  double v0[3], v1[3], v2[3];
  double n[3];
  double dconst;
  double abc;
  int axes;       // ax | (ay << 2) | (16 if the owner's transform is the identity)
  int object_id;  // Geometry::objectID of the face
  int owner;      // index of the owning Trimesh in objects[]
  int face;       // global face index (position in the caller's faces[] array)
};

// What the hit decision of TrimeshFace::intersectLocal reads of a triangle,
// with the two projected coordinates of each vertex already picked out
// (trimesh.cpp:133-165): 96 bytes instead of 128 and no per-test selection.
struct DevTriFast {
  double n[3];
  double dconst;
  double a0, a1, b0, b1, c0, c1;
  double abc2;    // twice the projected area: the three sub-areas are summed without their halvings (exact scalings)
  int axes;       // bit 0: the first kept axis is y (else x); bit 1: the second kept axis is z (else y)
  int face;
};

// 64-byte 4-wide BVH node with 8-bit quantised child boxes, see rt_bvh.h.
// Byte c of qlo[a] / qhi[a] is the lower / upper plane of child c on axis a.
struct DevBvhNode {
  float origin[3];
  float scaleX;       // plane q of axis a lies at origin[a] + q * scale (a power of two)
  int child[4];
  uint32_t qlo[3];
  uint32_t qhi[3];
  float scaleY, scaleZ;
};
#define RT_BVH_OBJECT_BIT 0x40000000
#define RT_BVH_EMPTY 0x7fffffff

// The same node with its child boxes as plain binary32 world-space planes
// (lower planes rounded down, upper planes up), 128 bytes, axis by axis:
// lo[a][c] / hi[a][c] = lower / upper plane of child c on axis a.  This form is
// read through the scalar cache when every lane of a wavefront is at the same
// node (bvhNodeSlabsUniform); DevBvhNode serves the lanes that are not.  The
// array holds eight of them per node, number o for rays that run down the axes
// of o's set bits, with the children sorted near to far for such rays.
struct DevBvhNodeF {
  float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];
  int child[4];
  int pad[4];
  // Normal cone of each child: every triangle below child c has a face normal n
  // with cone[c][0..2] . n / |n| >= cos(theta_c), and cone[c][3] bounds
  // sin(theta_c) from above (2: no bound, or a child that holds something other
  // than triangles).  A ray with unit direction d and cone . d > cone[c][3]
  // meets every such triangle from behind: where back faces are culled
  // (TrimeshFace::intersectLocal, trimesh.cpp:93-104: normal . d > 0 -> miss)
  // nothing below the child can be hit and the child is not entered.
  float cone[4][4];
};
#define RT_BVHF_BYTES 192

// What the shading kernels need of one (non-interpolated) material, derived
// once at scene creation with the reference's own expressions.
struct DevMat {
  double kd[3], ks[3], kr[3], kt[3];
  double base[3];   // ke + ka (*) ambient            (material.cpp:46)
  double shin;      // Material::shininess            (material.h:209-215)
  double ior;       // Material::index as a scalar    (material.h:217)
  int flags;        // RT_MATF_*
  int shinN;        // shininess = shinN + shinD with |shinD| <= 1e-12 when it is that close to a small
  double shinD;     // non-negative integer (powNearInt), else shinN = -1
};
#define RT_MATF_KD 1   // kd != 0
#define RT_MATF_KS 2
#define RT_MATF_KR 4
#define RT_MATF_KT 8

// Per-face word: the owner's material index and lit flag, and whether the mesh
// carries vertex normals / per-vertex materials, so that a hit on the face does
// not have to walk face -> object -> mesh -> material.
#define RT_FI_MATERIAL 0x0fffffff
#define RT_FI_LIT 0x10000000      // the material's kd or ks is non-zero
#define RT_FI_VNORMALS 0x20000000
#define RT_FI_VMATERIALS 0x40000000
#define RT_FI_REFLECTS 0x80000000u  // the material's kr is non-zero (RayTracer.cpp:161)

struct DevScene {
  const DevMat* devMats;     // one per material
  int interp_materials;      // 1: some mesh interpolates per-vertex materials
  const DevNode* nodes;
  const int* items;
  const DevTri* tris;
  const int* face_ids;       // 3 global vertex indices per face
  const RtObject* objects;
  const RtMesh* meshes;
  const RtMaterial* materials;
  const RtLight* lights;
  // per light: cos(atten_angle + 1e-9 deg), cos(atten_angle - 1e-9 deg).  A spot
  // light's cone test acos(x) * 180 / pi > atten_angle (light.cpp:129-133,141-145)
  // is decided by x against these two wherever x is not within that hair of the
  // cone's edge; only there is the acos itself needed.
  const double* spotCone;      // per light: cosines bracketing the cone's edge, then fall_rate as n, delta (powNearInt)
  const double* vnormals;    // 3 per global vertex
  const double* face_normals;  // 9 per face: the vertex normals of its corners (NULL: none / flat shading)
  const int* faceInfo;         // RT_FI_* word per face: what a hit needs to know about its mesh and owner
  const int* vmaterials;     // material index per global vertex
  const unsigned char* cube[6];
  int cube_w, cube_h;
  double eye[3], look[3], u[3], v[3], ambient[3];
  double scene_bb_min[3], scene_bb_max[3];
  int n_objects, n_lights, scene_root, use_kd, cull, smooth;
  // library-built BVH (RT_TRAVERSE_FAST)
  const DevBvhNode* bvh;
  const DevBvhNodeF* bvhF;   // the same nodes, binary32 planes (wavefront-uniform visits)
  const DevTri* bvhTris;     // triangles in leaf order
  const struct DevTriFast* bvhTrisFast;   // the same triangles, what the hit decision needs only
  double bvh_bb_min[3], bvh_bb_max[3];
  float bvh_bb_f[6];         // the same box in binary32: lower planes rounded down, then upper planes rounded up
  // per axis, an upper bound of 255 * scale + |origin| over every node of the BVH
  // (what a slab distance's rounding error grows with, see bvhRaySetup)
  float bvhErrG[3];
  int bvhLaneF;              // 1: bvhF is small enough for 32-bit byte offsets (bvhNodeSlabsLane)
  int bvhCones;              // 1: the normal cones of bvhF are worth testing (some child of some node can be culled by
                             // one) for every ray but camera rays; 2: for camera rays too; 0: not at all
  int bvh_valid;             // 0: empty scene
  int opaque;                // 1: no material transmits (kt == 0 everywhere)
};

#define RT_PRIM_NONE (-1)
#define RT_PRIM_FACE 0x40000000  // prim = RT_PRIM_FACE | global face index, else object index

// nodes visited, triangles tested, non-mesh objects tested (counted in STATS builds)
struct TraceCounters { uint32_t nodes, prims, objs; };

// BoundingBox::intersect (src/scene/bbox.h:62-89).  On an early `false` the
// outputs keep the partial values, which the reference's callers go on to use.
__device__ __forceinline__ bool boxIntersect(const double* bmin, const double* bmax, const V3& p,
                                             const V3& d, double& tMin, double& tMax) {
  tMin = -1.0e308;
  tMax = 1.0e308;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double vd = d.get(a);
    if (vd == 0.0) continue;
    double v1 = bmin[a] - p.get(a);
    double v2 = bmax[a] - p.get(a);
    double t1 = v1 / vd;
    double t2 = v2 / vd;
    if (t1 > t2) { double tt = t1; t1 = t2; t2 = tt; }
    if (t1 > tMin) tMin = t1;
    if (t2 < tMax) tMax = t2;
    if (tMin > tMax) return false;
    if (tMax < RT_EPS) return false;
  }
  return true;
}

// ------------------------------------------------------------------ primitives
// Each returns the local-space t and (optionally) the local-space normal of
// <Primitive>::intersectLocal.

// Sphere.cpp:8-41
__device__ __forceinline__ bool sphereLocal(const V3& p, const V3& d, double& t, V3& N) {
  V3 v = -p;
  double b = dot(v, d);
  double disc = b * b - dot(v, v) + 1;
  if (disc < 0.0) return false;
  disc = sqrt(disc);
  double t2 = b + disc;
  if (t2 <= RT_EPS) return false;
  double t1 = b - disc;
  t = (t1 > RT_EPS) ? t1 : t2;
  N = normalized(p + (t * d));
  return true;
}

// Box.cpp:11-79: the unit cube as three pairs of planes.  One plane: the ray
// parameter where it meets the plane `at` of axis A, kept when it lies in front
// of the ray, is not beyond the best so far and lands inside the face.  The
// reference visits the planes in the order -x -y -z +x +y +z and replaces the
// best only by a strictly smaller parameter; boxLocal keeps that order.
template <int A>
__device__ __forceinline__ double axisOf(const V3& v) { return A == 0 ? v.x : (A == 1 ? v.y : v.z); }

template <int A>
__device__ __forceinline__ void boxPlane(const V3& p, const V3& d, double at, int face, double& nearest, int& nearestFace) {
  const double along = axisOf<A>(d);
  if (along == 0) return;
  const double tt = (at - axisOf<A>(p)) / along;
  if (tt < RT_EPS || tt > nearest) return;
  const double e1 = axisOf<(A + 1) % 3>(p) + tt * axisOf<(A + 1) % 3>(d);
  const double e2 = axisOf<(A + 2) % 3>(p) + tt * axisOf<(A + 2) % 3>(d);
  if (e1 <= 0.5 && e1 >= -0.5 && e2 <= 0.5 && e2 >= -0.5 && nearest > tt) {
    nearest = tt;
    nearestFace = face;
  }
}

__device__ __forceinline__ bool boxLocal(const V3& p, const V3& d, double& t, V3& N) {
  double nearest = 1e100;
  int face = -1;
  boxPlane<0>(p, d, -0.5, 0, nearest, face);
  boxPlane<1>(p, d, -0.5, 1, nearest, face);
  boxPlane<2>(p, d, -0.5, 2, nearest, face);
  boxPlane<0>(p, d, 0.5, 3, nearest, face);
  boxPlane<1>(p, d, 0.5, 4, nearest, face);
  boxPlane<2>(p, d, 0.5, 5, nearest, face);
  if (face < 0) return false;
  t = nearest;
  // outward normal of the face: -axis for the three lower planes, +axis for the upper ones
  const double s = face < 3 ? -1.0 : 1.0;
  const int axis = face < 3 ? face : face - 3;
  N = V3(s * double(axis == 0), s * double(axis == 1), s * double(axis == 2));
  return true;
}

// Square.cpp:9-45
__device__ __forceinline__ bool squareLocal(const V3& p, const V3& d, double& t, V3& N) {
  if (d.z == 0.0) return false;
  double tt = -p.z / d.z;
  if (tt <= RT_EPS) return false;
  V3 P = p + (tt * d);
  if (P.x < -0.5 || P.x > 0.5) return false;
  if (P.y < -0.5 || P.y > 0.5) return false;
  t = tt;
  N = (d.z > 0.0) ? V3(0.0, 0.0, -1.0) : V3(0.0, 0.0, 1.0);
  return true;
}

// Cylinder.cpp:28-93 (body), 95-150 (caps), 8-26 (combination); always capped (Cylinder.h:11)
__device__ __forceinline__ bool cylinderBody(const V3& p, const V3& d, double& t, V3& N) {
  double x0 = p.x, y0 = p.y, x1 = d.x, y1 = d.y;
  double a = x1 * x1 + y1 * y1;
  double b = 2.0 * (x0 * x1 + y0 * y1);
  double c = x0 * x0 + y0 * y0 - 1.0;
  if (0.0 == a) return false;
  double disc = b * b - 4.0 * a * c;
  if (disc < 0.0) return false;
  disc = sqrt(disc);
  double t2 = (-b + disc) / (2.0 * a);
  if (t2 <= RT_EPS) return false;
  double t1 = (-b - disc) / (2.0 * a);
  if (t1 > RT_EPS) {
    V3 P = p + (t1 * d);
    if (P.z >= 0.0 && P.z <= 1.0) {
      t = t1;
      N = normalized(V3(P.x, P.y, 0.0));
      return true;
    }
  }
  V3 P = p + (t2 * d);
  if (P.z >= 0.0 && P.z <= 1.0) {
    t = t2;
    N = normalized(V3(P.x, P.y, 0.0));
    return true;
  }
  return false;
}
__device__ __forceinline__ bool cylinderCaps(const V3& p, const V3& d, double& t, V3& N) {
  double pz = p.z, dz = d.z;
  if (0.0 == dz) return false;
  double t1, t2;
  if (dz > 0.0) { t1 = (-pz) / dz; t2 = (1.0 - pz) / dz; }
  else { t1 = (1.0 - pz) / dz; t2 = (-pz) / dz; }
  if (t2 < RT_EPS) return false;
  if (t1 >= RT_EPS) {
    V3 q = p + (t1 * d);
    if ((q.x * q.x + q.y * q.y) <= 1.0) {
      t = t1;
      N = (dz > 0.0) ? V3(0.0, 0.0, -1.0) : V3(0.0, 0.0, 1.0);
      return true;
    }
  }
  V3 q = p + (t2 * d);
  if ((q.x * q.x + q.y * q.y) <= 1.0) {
    t = t2;
    N = (dz > 0.0) ? V3(0.0, 0.0, 1.0) : V3(0.0, 0.0, -1.0);
    return true;
  }
  return false;
}
__device__ __forceinline__ bool cylinderLocal(const V3& p, const V3& d, double& t, V3& N) {
  if (cylinderCaps(p, d, t, N)) {
    double tb;
    V3 Nb;
    if (cylinderBody(p, d, tb, Nb)) {
      if (tb < t) { t = tb; N = Nb; }
    }
    return true;
  }
  return cylinderBody(p, d, t, N);
}

// Cone.cpp:7-114
__device__ __forceinline__ bool coneLocal(const RtObject& o, const V3& p, const V3& d, double& t, V3& N) {
  const double beta_squared = o.cone_beta_squared, gamma = o.cone_gamma, height = o.cone_height;
  const bool capped = o.capped != 0;
  V3 normal;
  double pz = p.z, dz = d.z;
  double a = d.x * d.x + d.y * d.y - beta_squared * d.z * d.z;
  if (a == 0.0) return false;
  double b = 2 * (p.x * d.x + p.y * d.y - beta_squared * ((p.z + gamma) * d.z));
  double c = -beta_squared * (gamma + p.z) * (gamma + p.z) + p.x * p.x + p.y * p.y;
  double disc = b * b - 4 * a * c;
  double theRoot = RT_EPS;
  if (disc <= 0) return false;
  disc = sqrt(disc);
  double nearRoot = (-b + disc) / (2 * a);
  double farRoot = (-b - disc) / (2 * a);
  V3 qn = p + (nearRoot * d);
  bool nearGood = !(qn.z < 0 || qn.z > height);
  if (nearGood && (nearRoot > theRoot)) {
    theRoot = nearRoot;
    normal = V3(qn.x, qn.y, -2.0 * beta_squared * (qn.z + gamma));
  }
  V3 qf = p + (farRoot * d);
  bool farGood = !(qf.z < 0 || qf.z > height);
  if (farGood && ((nearGood && farRoot < theRoot) || farRoot > RT_EPS)) {
    theRoot = farRoot;
    normal = V3(qf.x, qf.y, -2.0 * beta_squared * (qf.z + gamma));
  }
  if (!capped && dot(normal, d) > 0) normal = -normal;
  double t1 = (-pz) / dz;
  double t2 = (height - pz) / dz;
  if (capped) {
    V3 q1 = p + (t1 * d);
    if (q1.x * q1.x + q1.y * q1.y <= o.cone_b_radius * o.cone_b_radius) {
      if (t1 < theRoot && t1 > RT_EPS) {
        theRoot = t1;
        normal = (dz > 0.0) ? V3(0.0, 0.0, -1.0) : V3(0.0, 0.0, 1.0);
      }
    }
    V3 q2 = p + (t2 * d);
    if (q2.x * q2.x + q2.y * q2.y <= o.cone_t_radius * o.cone_t_radius) {
      if (t2 < theRoot && t2 > RT_EPS) {
        theRoot = t2;
        normal = (dz > 0.0) ? V3(0.0, 0.0, 1.0) : V3(0.0, 0.0, -1.0);
      }
    }
  }
  if (theRoot <= RT_EPS) return false;
  t = theRoot;
  N = normalized(normal);
  return true;
}

// World ray -> object space, Geometry::intersect (scene.cpp:17-20)
struct LocalRay { V3 p, d; double len; };
__device__ __forceinline__ LocalRay toLocal(const double* inverse, const V3& p, const V3& d) {
  LocalRay r;
  r.p = xformPoint(inverse, p);
  V3 dir = xformPoint(inverse, p + d) - r.p;
  r.len = length(dir);
  r.d = V3(dir.x / r.len, dir.y / r.len, dir.z / r.len);
  return r;
}

// Geometry::intersect for a non-mesh object: world t and (optionally) world N.
template <bool WANT_N>
__device__ __forceinline__ bool objectIntersect(const RtObject& o, const V3& p, const V3& d,
                                                double& t, V3& N) {
  double tmin, tmax;
  if (!boxIntersect(o.bbox_min, o.bbox_max, p, d, tmin, tmax)) return false;
  LocalRay lr = toLocal(o.inverse, p, d);
  double tl = 0.0;
  V3 Nl;
  bool hit = false;
  switch (o.kind) {
    case RT_OBJ_SPHERE: hit = sphereLocal(lr.p, lr.d, tl, Nl); break;
    case RT_OBJ_BOX: hit = boxLocal(lr.p, lr.d, tl, Nl); break;
    case RT_OBJ_SQUARE: hit = squareLocal(lr.p, lr.d, tl, Nl); break;
    case RT_OBJ_CYLINDER: hit = cylinderLocal(lr.p, lr.d, tl, Nl); break;
    case RT_OBJ_CONE: hit = coneLocal(o, lr.p, lr.d, tl, Nl); break;
    default: break;
  }
  if (!hit) return false;
  t = tl / lr.len;
  if (WANT_N) N = normalized(mul33(o.normi, Nl));  // scene.h:168-172
  return true;
}

// TrimeshFace::intersectLocal (trimesh.cpp:91-208), hit decision and local t.
// sqrt(0 + 0 + z*z) of the reference's 3-D cross product of two vectors with
// zero third component equals |z| exactly in binary64 (no over/underflow at
// scene scales), so the four square roots are written as fabs.
template <bool FAST>
__device__ __forceinline__ bool triTest(const DevTri& T, const V3& lp, const V3& ld, bool cullBack,
                                        double& rayT, V3* bary) {
  const V3 n(T.n);
  const double nd = dot(n, ld);
  if (cullBack && nd > 0) return false;
  if (nd == 0) return false;
  rayT = -(dot(n, lp) + T.dconst) / nd;
  if (rayT < RT_EPS) return false;
  const V3 P = lp + rayT * ld;
  const int ax = T.axes & 3, ay = (T.axes >> 2) & 3;
  const double a0 = T.v0[ax], a1 = T.v0[ay], b0 = T.v1[ax], b1 = T.v1[ay];
  const double c0 = T.v2[ax], c1 = T.v2[ay], p0 = P.get(ax), p1 = P.get(ay);
  // z components of (b-p)x(c-p), (p-a)x(c-a), (b-a)x(p-a)  (trimesh.cpp:163-165)
  const double pbc = fabs((b0 - p0) * (c1 - p1) - (b1 - p1) * (c0 - p0)) / 2;
  const double apc = fabs((p0 - a0) * (c1 - a1) - (p1 - a1) * (c0 - a0)) / 2;
  const double abp = fabs((b0 - a0) * (p1 - a1) - (b1 - a1) * (p0 - a0)) / 2;
  if (FAST && bary == nullptr) {
    const double s = pbc + apc + abp;
    return s <= T.abc * (1 + RT_EPS) && s >= T.abc * (1 - RT_EPS);
  }
  const double w0 = pbc / T.abc, w1 = apc / T.abc, w2 = abp / T.abc;
  const double total = w0 + w1 + w2;
  if (bary) *bary = V3(w0, w1, w2);
  return total <= (1 + RT_EPS) && total >= (1 - RT_EPS);
}

// triTest<true> without barycentrics on the compact record: same operations on
// the same values, except that the three halvings of the sub-areas and of the
// triangle's area are left out on both sides of the comparison (powers of two:
// exact) and the two kept coordinates of the hit point are picked by two
// selects (the kept axes are (y,z), (x,z) or (x,y)).  rCull: a plane distance
// beyond it cannot matter to the caller (it lies behind the best hit / the light).
__device__ __forceinline__ bool triTestFast(const DevTriFast& T, const V3& lp, const V3& ld, bool cullBack,
                                            double rCull, double& rayT) {
  // The whole 96-byte record is asked for at once, and pinned in registers
  // before the first early return: left alone, the compiler sinks each load
  // behind the returns that come before its first use, and a test then waits
  // for memory three times in a row (normal, plane constant, corners).  1 % of
  // a whole frame, 2 % of a thin one.  RT_TRI_LAZY restores that for A/B runs.
  const V3 n(T.n);
#ifndef RT_TRI_LAZY
  double dconst = T.dconst, a0 = T.a0, a1 = T.a1, b0 = T.b0, b1 = T.b1, c0 = T.c0, c1 = T.c1, abc2 = T.abc2;
  int axes = T.axes;
  asm volatile("" : "+v"(dconst), "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(abc2), "+v"(axes));
#else
  const double dconst = T.dconst;
#endif
  const double nd = dot(n, ld);
  if (cullBack && nd > 0) return false;
  if (nd == 0) return false;
  rayT = -(dot(n, lp) + dconst) / nd;
  if (rayT < RT_EPS) return false;
#ifndef RT_NO_RCULL
  if (rayT > rCull) return false;
#endif
  const V3 P = lp + rayT * ld;
#ifdef RT_TRI_LAZY
  const double a0 = T.a0, a1 = T.a1, b0 = T.b0, b1 = T.b1, c0 = T.c0, c1 = T.c1, abc2 = T.abc2;
  const int axes = T.axes;
#endif
  const double p0 = (axes & 1) ? P.y : P.x, p1 = (axes & 2) ? P.z : P.y;
  const double pbc = fabs((b0 - p0) * (c1 - p1) - (b1 - p1) * (c0 - p0));
  const double apc = fabs((p0 - a0) * (c1 - a1) - (p1 - a1) * (c0 - a0));
  const double abp = fabs((b0 - a0) * (p1 - a1) - (b1 - a1) * (p0 - a0));
  const double s = pbc + apc + abp;
  return s <= abc2 * (1 + RT_EPS) && s >= abc2 * (1 - RT_EPS);
}

// Local bounding box of a face = componentwise extrema of its vertices
// (TrimeshFace::ComputeLocalBoundingBox, trimesh.h:136-145)
__device__ __forceinline__ bool faceBoxCull(const DevTri& T, const V3& p, const V3& d) {
  double lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    hi[a] = fmax(T.v2[a], fmax(T.v0[a], T.v1[a]));
    lo[a] = fmin(T.v2[a], fmin(T.v0[a], T.v1[a]));
  }
  double t0, t1;
  return boxIntersect(lo, hi, p, d, t0, t1);
}

// ------------------------------------------------------------------- traversal
#define RT_STACK_DEPTH 96
#define RT_ENTRY_RESUME 0x80000000u

struct KdEntry { uint32_t node; double tmin, tmax; };

// The first RT_KD_LDS_STACK entries of a lane's stack live in the workgroup's
// LDS as three planes -- node words, then lower, then upper interval ends, each
// [entry][thread] -- so that a wavefront's accesses fall on 64 different banks;
// deeper entries go to a per-lane array in scratch memory.  The intervals stay
// binary64: they are compared with tStar exactly as the reference compares them.
#ifndef RT_KD_LDS_STACK
#define RT_KD_LDS_STACK 8
#endif
#define RT_KD_LDS_STACK_BYTES (RT_KD_LDS_STACK * 20)   // per thread
typedef __attribute__((address_space(3))) uint32_t rt_kd_lds_u32;
typedef __attribute__((address_space(3))) double rt_kd_lds_f64;

// Scene::intersect with the reference's own traversal.  Returns the winning
// primitive (RT_PRIM_NONE on a miss) and its world-space t.
template <bool STATS>
__device__ int traceKdReference(const DevScene& S, const V3& p, const V3& d, int rayType,
                                double& tOut, TraceCounters& cnt, rt_kd_lds_u32* ldsStack) {
  bool haveOne = false;
  double bestT = 0.0;
  int bestPrim = RT_PRIM_NONE;
  const bool cullBack = S.cull && rayType != RT_RAY_REFRACTION;

  if (!(S.use_kd && S.scene_root >= 0)) {
    // Linear scan of Scene::objects (scene.cpp:144-153); meshes through
    // Trimesh::intersectLocal (trimesh.cpp:62-82): no per-face box cull.
    for (int j = 0; j < S.n_objects; j++) {
      const RtObject& o = S.objects[j];
      if (o.kind == RT_OBJ_TRIMESH) {
        double tmin, tmax;
        if (!boxIntersect(o.bbox_min, o.bbox_max, p, d, tmin, tmax)) continue;
        const LocalRay lr = toLocal(o.inverse, p, d);
        const RtMesh& m = S.meshes[o.mesh];
        bool meshHave = false;
        double meshT = 0.0;
        int meshFace = -1;
        for (int k = 0; k < m.n_faces; k++) {
          const DevTri& T = S.tris[m.first_face + k];
          double rt;
          V3 bary;
          if (STATS) cnt.prims++;
          if (triTest<false>(T, lr.p, lr.d, cullBack, rt, &bary)) {
            if (!meshHave || rt < meshT) { meshHave = true; meshT = rt; meshFace = m.first_face + k; }
          }
        }
        if (meshHave) {
          double tw = meshT / lr.len;
          if (!haveOne || tw < bestT) { haveOne = true; bestT = tw; bestPrim = RT_PRIM_FACE | meshFace; }
        }
      } else {
        double t;
        V3 N;
        if (STATS) cnt.objs++;
        if (objectIntersect<false>(o, p, d, t, N)) {
          if (!haveOne || t < bestT) { haveOne = true; bestT = t; bestPrim = j; }
        }
      }
    }
    tOut = bestT;
    return bestPrim;
  }

  KdEntry deep[RT_STACK_DEPTH - RT_KD_LDS_STACK];   // entries past the LDS part
  const unsigned int ldsStride = blockDim.x;
  rt_kd_lds_u32* const colNode = ldsStack + threadIdx.x;
  rt_kd_lds_f64* const colMin = (rt_kd_lds_f64*)(ldsStack + RT_KD_LDS_STACK * ldsStride) + threadIdx.x;
  rt_kd_lds_f64* const colMax = colMin + RT_KD_LDS_STACK * ldsStride;
#define RT_KD_PUSH(NODE, TMIN, TMAX)                                               \
  {                                                                                \
    if (sp < RT_KD_LDS_STACK) {                                                    \
      colNode[(unsigned int)sp * ldsStride] = (uint32_t)(NODE);                    \
      colMin[(unsigned int)sp * ldsStride] = (TMIN);                               \
      colMax[(unsigned int)sp * ldsStride] = (TMAX);                               \
    } else {                                                                       \
      KdEntry& e_ = deep[sp - RT_KD_LDS_STACK];                                    \
      e_.node = (uint32_t)(NODE); e_.tmin = (TMIN); e_.tmax = (TMAX);              \
    }                                                                              \
    sp++;                                                                          \
  }
  int sp = 0;
  // function-level interval of the reference's current intersectKdTree call
  double fMin = 0.0, fMax = 0.0;
  boxIntersect(S.scene_bb_min, S.scene_bb_max, p, d, fMin, fMax);  // result ignored (scene.cpp:139)
  RT_KD_PUSH(S.scene_root, fMin, fMax)
  const double sceneFMin = fMin, sceneFMax = fMax;
  bool inMesh = false;
  LocalRay lr;
  lr.len = 1.0;

  while (sp > 0) {
    --sp;
    uint32_t nidx;
    double eMin, eMax;
    if (sp < RT_KD_LDS_STACK) {
      nidx = colNode[(unsigned int)sp * ldsStride];
      eMin = colMin[(unsigned int)sp * ldsStride];
      eMax = colMax[(unsigned int)sp * ldsStride];
    } else {
      const KdEntry& e_ = deep[sp - RT_KD_LDS_STACK];
      nidx = e_.node; eMin = e_.tmin; eMax = e_.tmax;
    }
    int k0 = 0;
    if (nidx & RT_ENTRY_RESUME) {
      // back in the scene-level leaf loop after a nested mesh traversal
      nidx &= ~RT_ENTRY_RESUME;
      k0 = (int)eMin;
      inMesh = false;
      fMin = sceneFMin;
      fMax = sceneFMax;
    }
    const DevNode nd = S.nodes[nidx];
    if (STATS) cnt.nodes++;
    if ((nd.a & 3u) == 3u) {
      const int first = (int)(nd.a >> 2), count = (int)nd.b;
      if (!inMesh) {
        for (int k = k0; k < count; k++) {
          const int oi = S.items[first + k];
          const RtObject& o = S.objects[oi];
          if (o.kind == RT_OBJ_TRIMESH) {
            const RtMesh& m = S.meshes[o.mesh];
            double a0 = 0.0, a1 = 0.0;
            boxIntersect(m.root_bb_min, m.root_bb_max, p, d, a0, a1);  // result ignored (scene.cpp:79)
            if (m.kd_root < 0) continue;
            if (sp + 2 > RT_STACK_DEPTH) continue;
            RT_KD_PUSH(nidx | RT_ENTRY_RESUME, (double)(k + 1), 0.0)
            RT_KD_PUSH(m.kd_root, a0, a1)
            inMesh = true;
            fMin = a0;
            fMax = a1;
            lr = toLocal(o.inverse, p, d);
            break;
          }
          double t;
          V3 N;
          if (STATS) cnt.objs++;
          if (objectIntersect<false>(o, p, d, t, N)) {
            if (!haveOne || t < bestT) { haveOne = true; bestT = t; bestPrim = oi; }
          }
        }
      } else {
        for (int k = 0; k < count; k++) {
          const int fi = S.items[first + k];
          const DevTri& T = S.tris[fi];
          if (STATS) cnt.prims++;
          if (!faceBoxCull(T, p, d)) continue;
          double rt;
          if (triTest<false>(T, lr.p, lr.d, cullBack, rt, nullptr)) {
            double tw = rt / lr.len;
            if (!haveOne || tw < bestT) { haveOne = true; bestT = tw; bestPrim = RT_PRIM_FACE | fi; }
          }
        }
      }
    } else {
      const int dim = (int)(nd.a & 3u);
      const uint32_t left = nd.a >> 2, right = nd.b;
      const double po = p.get(dim), di = d.get(dim);
      const double tStar = (nd.split - po) / di;
      // scene.cpp:96-124: pushed intervals come from the function-level
      // tMin/tMax, not from the popped element
      uint32_t nearN, farN;
      if (po < nd.split) { nearN = left; farN = right; } else { nearN = right; farN = left; }
      if (sp + 2 > RT_STACK_DEPTH) continue;
      if (tStar > eMax || tStar < 0) {
        RT_KD_PUSH(nearN, fMin, fMax)
      } else if (tStar < eMin) {
        RT_KD_PUSH(farN, fMin, fMax)
      } else if (eMin <= tStar && tStar <= eMax) {
        RT_KD_PUSH(nearN, fMin, tStar)
        RT_KD_PUSH(farN, tStar, fMax)
      }
    }
  }
#undef RT_KD_PUSH
  tOut = bestT;
  return bestPrim;
}

// -------------------------------------------------------------- BVH traversal
#define RT_BVH_STACK 96

// Where a lane keeps the subtrees it still has to visit.  The first
// RT_LDS_STACK entries of every lane live in the workgroup's LDS, laid out so
// that entry k of the workgroup's thread t sits at word (k * blockDim.x + t)
// (times two for 8-byte entries): whatever depth each lane is at, the lanes of
// a wavefront touch 64 different banks.  Only deeper entries go to a per-lane
// array in scratch memory, which most rays never reach.  (With the whole stack
// in scratch, two thirds of the bytes the closest-hit kernel wrote to HBM were
// stack entries leaving the caches.)
//
// A node visit writes its (up to three) new entries WITHOUT branches: each
// candidate is stored at the lane's top of stack and the top moves up only if
// the candidate counts, so a candidate that does not count is overwritten by the
// next store.  That needs three free rows above the top; as long as every lane
// of the wavefront has them (checked with one comparison per visit) the visit
// is straight-line code.
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 11
#endif
typedef __attribute__((address_space(3))) uint32_t rt_lds_u32;
typedef __attribute__((address_space(3))) unsigned long long rt_lds_u64;

// Bytes of dynamic LDS a traversal kernel needs per thread: closest-hit entries
// are 8 bytes (subtree, entry distance), occlusion entries 4 (subtree).
#define RT_LDS_STACK_BYTES_CLOSEST (RT_LDS_STACK * 8)
#define RT_LDS_STACK_BYTES_ANYHIT (RT_LDS_STACK * 4)

template <bool ANYHIT> struct BvhSlot;
template <> struct BvhSlot<false> {   // closest hit: subtree and its entry distance
  typedef rt_lds_u64 lds_t;
  typedef unsigned long long word_t;
  static __device__ __forceinline__ word_t pack(int ref, float tn) {
    return ((unsigned long long)__float_as_uint(tn) << 32) | (uint32_t)ref;
  }
  static __device__ __forceinline__ int refOf(word_t w) { return (int)(uint32_t)w; }
  static __device__ __forceinline__ float tnOf(word_t w) { return __uint_as_float((uint32_t)(w >> 32)); }
};
template <> struct BvhSlot<true> {    // occlusion: every overlapped subtree is searched, no distance needed
  typedef rt_lds_u32 lds_t;
  typedef uint32_t word_t;
  static __device__ __forceinline__ word_t pack(int ref, float) { return (uint32_t)ref; }
  static __device__ __forceinline__ int refOf(word_t w) { return (int)w; }
  static __device__ __forceinline__ float tnOf(word_t) { return 0.0f; }
};

// Three-operand minimum / maximum as the instructions they are.  (Written as
// fminf / fmaxf the compiler first passes every operand that comes out of the
// packed multiply-adds through a quieting v_max_f32 x, x: eight extra
// instructions per node.  No operand here can be a NaN: planes, reciprocals and
// offsets are finite, see bvhRaySetup.)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min2f(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// One bit per lane of the wavefront, in a scalar register pair: the outcome of
// a per-lane test as the comparison instruction leaves it.  Combining such
// masks is scalar work; a lane picks between two values by its bit with one
// v_cndmask.  (Left to itself the compiler turns tests that are used in more
// than one basic block into 0/1 integers in vector registers and combines
// those: a dozen vector instructions per node visit.)
typedef unsigned long long rt_mask;
__device__ __forceinline__ unsigned int pickByMask(unsigned int ifClear, unsigned int ifSet, rt_mask m) {
  unsigned int r;
  asm("" : "+s"(m));   // a mask the compiler knows the value of must still arrive in a register pair
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(ifClear), "v"(ifSet), "s"(m));
  return r;
}
typedef __attribute__((address_space(3))) unsigned char rt_lds_u8;

__device__ __forceinline__ float roundDownF(double x) {
  float f = (float)x;
  return ((double)f > x) ? nextafterf(f, -INFINITY) : f;
}
// a float that is not below x (x > 0)
__device__ __forceinline__ float roundUpF(double x) {
  const float f = (float)x;
  return fmaf(fabsf(f), 0x1p-23f, f) + 1.0e-37f;
}

typedef float rt_f2 __attribute__((ext_vector_type(2)));

// Slab distances of two children at once: q * B + A on both lanes of a packed
// fused multiply-add.
__device__ __forceinline__ rt_f2 slab2(uint32_t q, int c, float B, float A) {
  rt_f2 qf;
  qf.x = (float)((q >> (8 * c)) & 255u);
  qf.y = (float)((q >> (8 * c + 8)) & 255u);
  const rt_f2 b = {B, B}, a = {A, A};
  return __builtin_elementwise_fma(qf, b, a);
}

// ---- pieces of the BVH query, shared by the per-ray and the streaming form ----

// What a query keeps per ray besides p and d.
struct BvhRay {
  V3 ldI;                 // object-space direction of every identity-transform mesh
  double lenI;            // (Geometry::intersect, scene.cpp:17-20: ((p+d)-p) renormalised)
  float ivx, ivy, ivz;    // 1 / d in binary32
  // -(p * inv) in binary32, moved down (a0*) and up (a1*) by a bound of the
  // rounding error of a plane distance on that axis: what entry and exit plane
  // distances are offset by
  float a0x, a1x, a0y, a1y, a0z, a1z;
  bool negx, negy, negz;  // direction sign per axis (false on a degenerate axis)
  bool zx, zy, zz;        // |d| < 1e-30 on that axis: inside / outside test instead of a slab
  bool cullBack;
  // the direction in binary32 where back faces are culled for this ray and the
  // tree carries normal cones (DevBvhNodeF::cone), else (0, 0, 0): no cone test
  // can then come out above its positive threshold
  float dfx, dfy, dfz;
};

__device__ __forceinline__ void bvhRaySetup(const DevScene& S, const V3& p, const V3& d, int rayType, BvhRay& r) {
  r.cullBack = S.cull && rayType != RT_RAY_REFRACTION;
  r.zx = !(fabs(d.x) >= 1.0e-30);
  r.zy = !(fabs(d.y) >= 1.0e-30);
  r.zz = !(fabs(d.z) >= 1.0e-30);
  // Everything the box tests use is binary32 from the start (three binary64
  // divisions per ray cost as much as three node visits): inv = 1 / d and
  // p * inv carry a relative error of 3 * 2^-24 and 5 * 2^-24 of their own.
  // (the hardware reciprocal is good to one unit in the last place: with the
  // conversion of d that is 1.5 * 2^-23, inside the bound below)
  r.ivx = __builtin_amdgcn_rcpf((float)d.x); r.ivy = __builtin_amdgcn_rcpf((float)d.y); r.ivz = __builtin_amdgcn_rcpf((float)d.z);
  // A plane at coordinate c lies at distance c * inv - p * inv.  In binary32,
  // with c either a float (bvhNodeSlabsUniform) or origin + q * scale evaluated
  // as q * (scale * inv) + origin * inv (bvhNodeSlabs), the errors of inv and of
  // p * inv plus the roundings of the products, of their sum and of the fused
  // multiply-add stay below 2^-21 (G |inv| + |p inv|), where G = bvhErrG bounds
  // 255 * scale + |origin|, and so every |c|, over all nodes.  Entry distances
  // are moved down and exit distances up by twice that: they stay lower and
  // upper bounds.  (Far from the world origin the widening grows relative to
  // the boxes: culling gets weaker, never wrong.)
  {
    const float pix = (float)p.x * r.ivx, piy = (float)p.y * r.ivy, piz = (float)p.z * r.ivz;
    const float ex = 0x1p-20f * fmaf(S.bvhErrG[0], fabsf(r.ivx), fabsf(pix)) + 1.0e-37f;
    const float ey = 0x1p-20f * fmaf(S.bvhErrG[1], fabsf(r.ivy), fabsf(piy)) + 1.0e-37f;
    const float ez = 0x1p-20f * fmaf(S.bvhErrG[2], fabsf(r.ivz), fabsf(piz)) + 1.0e-37f;
    r.a0x = -pix - ex; r.a1x = -pix + ex;
    r.a0y = -piy - ey; r.a1y = -piy + ey;
    r.a0z = -piz - ez; r.a1z = -piz + ez;
  }
  r.negx = (d.x < 0.0) & !r.zx;
  r.negy = (d.y < 0.0) & !r.zy;
  r.negz = (d.z < 0.0) & !r.zz;
  // (every ray of the render path has a unit direction; the cones' thresholds leave room for its rounding)
  // Camera rays look at most of what they meet from its front and get little
  // for the cone tests (a quarter of a visit's arithmetic): bvhCones == 1 leaves
  // them out, 2 tests them too.  (Measured and dropped: a wavefront that stops
  // asking after a few visits without a culled child -- the culls come deep in
  // the tree, after visits that cull nothing.)
  const bool cones = r.cullBack && (S.bvhCones > 1 || (S.bvhCones == 1 && rayType != RT_RAY_VISIBILITY));
  r.dfx = cones ? (float)d.x : 0.0f;
  r.dfy = cones ? (float)d.y : 0.0f;
  r.dfz = cones ? (float)d.z : 0.0f;
}

// The rest of the setup, for rays that get past the root box: what the
// triangle tests need.
__device__ __forceinline__ void bvhRaySetupLocal(const V3& p, const V3& d, BvhRay& r) {
  const V3 dirI = (p + d) - p;
  r.lenI = length(dirI);
  r.ldI = V3(dirI.x / r.lenI, dirI.y / r.lenI, dirI.z / r.lenI);
}

// A NaN ray.  A refraction past the critical angle leaves a NaN direction
// (RayTracer.cpp:193-199), and a hit of such a ray a NaN shade point.  No
// comparison on a NaN orders anything, so no box can be culled; the reference's
// scan over Scene::objects (scene.cpp:144-153) then accepts the first object
// whose tests all fall through, and no triangle (trimesh.cpp:91-208 ends in
// range checks).
template <bool STATS>
__device__ __forceinline__ int bvhScanNaN(const DevScene& S, const V3& p, const V3& d, double& tOut, TraceCounters& cnt) {
  bool have = false;
  double bt = 0.0;
  int bp = RT_PRIM_NONE;
  for (int j = 0; j < S.n_objects; j++) {
    const RtObject& o = S.objects[j];
    if (o.kind == RT_OBJ_TRIMESH) continue;
    double t;
    V3 N;
    if (STATS) cnt.objs++;
    if (objectIntersect<false>(o, p, d, t, N) && (!have || t < bt)) { have = true; bt = t; bp = j; }
  }
  tOut = bt;
  return bp;
}

// Slab tests of the four children of node `ref`: tn[c] = entry distance (a
// lower bound) of child c, +inf when the ray misses it or it lies beyond
// bestTf; kid[c] = its reference.
//
// The box tests only decide which subtrees are skipped, never a result, so they
// run in binary32 with every plane distance widened by a bound of its rounding
// error (entry distances are lower bounds, exit distances upper bounds).
__device__ __forceinline__ void bvhNodeSlabs(const DevScene& S, const V3& p, const BvhRay& r, int ref, float bestTf,
                                             float tn[4], int kid[4]) {
  const float kInf = __int_as_float(0x7f800000);
  const uint4* np = reinterpret_cast<const uint4*>(S.bvh + ref);
  const uint4 w0 = np[0], w1 = np[1], w2 = np[2];
  const uint4 w3 = np[3];
  const float ox = __uint_as_float(w0.x), oy = __uint_as_float(w0.y), oz = __uint_as_float(w0.z);
  const float sx = __uint_as_float(w0.w), sy = __uint_as_float(w3.z), sz = __uint_as_float(w3.w);
  kid[0] = (int)w1.x; kid[1] = (int)w1.y; kid[2] = (int)w1.z; kid[3] = (int)w1.w;
  // B = scale / d: plane q lies at distance q * B + origin * inv - p * inv; the
  // last term comes widened from the ray's setup.
  float bx = r.ivx * sx, by = r.ivy * sy, bz = r.ivz * sz;
  float a0x = fmaf(ox, r.ivx, r.a0x), a1x = fmaf(ox, r.ivx, r.a1x);
  float a0y = fmaf(oy, r.ivy, r.a0y), a1y = fmaf(oy, r.ivy, r.a1y);
  float a0z = fmaf(oz, r.ivz, r.a0z), a1z = fmaf(oz, r.ivz, r.a1z);
  if (r.zx | r.zy | r.zz) {
    // The ray runs parallel to the slabs of this axis: with u = the origin in
    // grid units, K (q_lo - u) and K (q_hi - u) for a huge K put the entry at
    // -huge and the exit at +huge when the origin lies between the planes and
    // the entry at +huge or the exit at -huge when it does not.
    const float K = 0x1p40f;
    if (r.zx) {
      const float u = fminf(fmaxf((float)((p.x - (double)ox) / (double)sx), -1.0e9f), 1.0e9f);
      const float m = fmaf(fabsf(u), 0x1p-21f, 0x1p-20f);
      bx = K; a0x = -K * (u + m); a1x = -K * (u - m);
    }
    if (r.zy) {
      const float u = fminf(fmaxf((float)((p.y - (double)oy) / (double)sy), -1.0e9f), 1.0e9f);
      const float m = fmaf(fabsf(u), 0x1p-21f, 0x1p-20f);
      by = K; a0y = -K * (u + m); a1y = -K * (u - m);
    }
    if (r.zz) {
      const float u = fminf(fmaxf((float)((p.z - (double)oz) / (double)sz), -1.0e9f), 1.0e9f);
      const float m = fmaf(fabsf(u), 0x1p-21f, 0x1p-20f);
      bz = K; a0z = -K * (u + m); a1z = -K * (u - m);
    }
  }
  // entry / exit plane words by the sign of the direction
  const uint32_t q0x = r.negx ? w2.w : w2.x, q1x = r.negx ? w2.x : w2.w;
  const uint32_t q0y = r.negy ? w3.x : w2.y, q1y = r.negy ? w2.y : w3.x;
  const uint32_t q0z = r.negz ? w3.y : w2.z, q1z = r.negz ? w2.z : w3.y;
#pragma unroll
  for (int c = 0; c < 4; c += 2) {
    const rt_f2 nx = slab2(q0x, c, bx, a0x), fx = slab2(q1x, c, bx, a1x);
    const rt_f2 ny = slab2(q0y, c, by, a0y), fy = slab2(q1y, c, by, a1y);
    const rt_f2 nz = slab2(q0z, c, bz, a0z), fz = slab2(q1z, c, bz, a1z);
    const float tnear0 = max3f(nx.x, ny.x, nz.x), tnear1 = max3f(nx.y, ny.y, nz.y);
    const float tfar0 = min2f(min3f(fx.x, fy.x, fz.x), bestTf);
    const float tfar1 = min2f(min3f(fx.y, fy.y, fz.y), bestTf);
    // (an unused slot has an inverted box; should rounding ever let one pass,
    // it leads to a leaf of one null triangle)
    const bool h0 = (tnear0 <= tfar0) & (tfar0 >= 0.0f);
    const bool h1 = (tnear1 <= tfar1) & (tfar1 >= 0.0f);
    tn[c] = h0 ? tnear0 : kInf;
    tn[c + 1] = h1 ? tnear1 : kInf;
  }
}

typedef float rt_f4 __attribute__((ext_vector_type(4)));
typedef int rt_i4 __attribute__((ext_vector_type(4)));

// bvhNodeSlabs for a node that every active lane of the wavefront is visiting
// (`ref` is wavefront-uniform) by rays that share their direction signs
// (`negMask`: bit a set = direction negative on axis a, wavefront-uniform) and
// have no axis-parallel component.  The node comes through the scalar cache
// into scalar registers -- no vector-memory instruction, no byte-to-float
// conversion -- and each plane distance is one multiply-add plane * inv - p *
// inv, widened by the ray's error bound: |plane| <= bvhErrG, so the bound of
// bvhRaySetup covers the roundings of this form as well.
// RAW: tn[c] is the entry distance whether the child is hit or not and hit[c]
// says which it is (what the ordered descent of a closest-hit query wants);
// otherwise tn[c] is +inf for a child that is not hit and hit[] is not written.
typedef float rt_f8 __attribute__((ext_vector_type(8)));

// CONES (with RAW): children whose normal cone says that the ray meets every
// triangle below them from behind are struck from the hit masks (see
// DevBvhNodeF::cone); wave-uniform switch `cones`.
template <bool RAW>
__device__ __forceinline__ void bvhNodeSlabsUniform(const DevScene& S, const BvhRay& r, int ref, unsigned int negMask,
                                                    unsigned int copy, float bestTf, float tn[4], int kid[4],
                                                    rt_mask hit[4], bool cones = false) {
  const float kInf = __int_as_float(0x7f800000);
  // eight copies of every node, one per combination of direction signs, each
  // with its children listed near to far for directions of those signs
  const char* np = reinterpret_cast<const char*>(S.bvhF + ((size_t)ref * 8 + copy));
  // entry planes of an axis: the lower ones unless the rays run down that axis
  const unsigned int oNX = (negMask & 1u) ? 16u : 0u, oNY = (negMask & 2u) ? 48u : 32u, oNZ = (negMask & 4u) ? 80u : 64u;
  const unsigned int oFX = oNX ^ 16u, oFY = oNY ^ 16u, oFZ = oNZ ^ 16u;
  rt_f4 nX, fX, nY, fY, nZ, fZ;
  rt_i4 kids;
  rt_f8 cone01, cone23;
  if (RAW && cones) {
    // (one statement, so that nothing that reads a result can come before the wait)
    asm volatile(
        "s_load_dwordx4 %0, %9, %10\n\t"
        "s_load_dwordx4 %1, %9, %11\n\t"
        "s_load_dwordx4 %2, %9, %12\n\t"
        "s_load_dwordx4 %3, %9, %13\n\t"
        "s_load_dwordx4 %4, %9, %14\n\t"
        "s_load_dwordx4 %5, %9, %15\n\t"
        "s_load_dwordx4 %6, %9, 0x60\n\t"
        "s_load_dwordx8 %7, %9, 0x80\n\t"
        "s_load_dwordx8 %8, %9, 0xa0\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(nX), "=&s"(fX), "=&s"(nY), "=&s"(fY), "=&s"(nZ), "=&s"(fZ), "=&s"(kids), "=&s"(cone01), "=&s"(cone23)
        : "s"(np), "s"(oNX), "s"(oFX), "s"(oNY), "s"(oFY), "s"(oNZ), "s"(oFZ)
        : "memory");
  } else {
    asm volatile(
        "s_load_dwordx4 %0, %7, %8\n\t"
        "s_load_dwordx4 %1, %7, %9\n\t"
        "s_load_dwordx4 %2, %7, %10\n\t"
        "s_load_dwordx4 %3, %7, %11\n\t"
        "s_load_dwordx4 %4, %7, %12\n\t"
        "s_load_dwordx4 %5, %7, %13\n\t"
        "s_load_dwordx4 %6, %7, 0x60\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(nX), "=&s"(fX), "=&s"(nY), "=&s"(fY), "=&s"(nZ), "=&s"(fZ), "=&s"(kids)
        : "s"(np), "s"(oNX), "s"(oFX), "s"(oNY), "s"(oFY), "s"(oNZ), "s"(oFZ)
        : "memory");
  }
  kid[0] = kids.x; kid[1] = kids.y; kid[2] = kids.z; kid[3] = kids.w;
  // Two children per packed multiply-add: planes from a scalar register pair,
  // the reciprocal and the offset picked out of register pairs of the ray's
  // state by the instruction's operand selects (written out by hand: left to
  // itself the compiler copies every selected value into a pair of its own,
  // eighteen registers more per lane).
  const rt_f2 ivXY = {r.ivx, r.ivy}, ivZZ = {r.ivz, r.ivz};
  const rt_f2 aX = {r.a0x, r.a1x}, aY = {r.a0y, r.a1y}, aZ = {r.a0z, r.a1z};
#define RT_PKFMA(dst, planes, ivp, ap, sel)                                                     \
  asm("v_pk_fma_f32 %0, %1, %2, %3 " sel : "=v"(dst) : "s"(planes), "v"(ivp), "v"(ap))
#pragma unroll
  for (int c = 0; c < 4; c += 2) {
    const rt_f2 pnx = {nX[c], nX[c + 1]}, pfx = {fX[c], fX[c + 1]};
    const rt_f2 pny = {nY[c], nY[c + 1]}, pfy = {fY[c], fY[c + 1]};
    const rt_f2 pnz = {nZ[c], nZ[c + 1]}, pfz = {fZ[c], fZ[c + 1]};
    rt_f2 nx, fx, ny, fy, nz, fz;
    RT_PKFMA(nx, pnx, ivXY, aX, "op_sel_hi:[1,0,0]");                   // planes * ivx + a0x
    RT_PKFMA(fx, pfx, ivXY, aX, "op_sel:[0,0,1] op_sel_hi:[1,0,1]");    // planes * ivx + a1x
    RT_PKFMA(ny, pny, ivXY, aY, "op_sel:[0,1,0] op_sel_hi:[1,1,0]");    // planes * ivy + a0y
    RT_PKFMA(fy, pfy, ivXY, aY, "op_sel:[0,1,1] op_sel_hi:[1,1,1]");    // planes * ivy + a1y
    RT_PKFMA(nz, pnz, ivZZ, aZ, "op_sel_hi:[1,0,0]");                   // planes * ivz + a0z
    RT_PKFMA(fz, pfz, ivZZ, aZ, "op_sel:[0,0,1] op_sel_hi:[1,0,1]");    // planes * ivz + a1z
    const float tnear0 = max3f(nx.x, ny.x, nz.x), tnear1 = max3f(nx.y, ny.y, nz.y);
    const float tfar0 = min2f(min3f(fx.x, fy.x, fz.x), bestTf);
    const float tfar1 = min2f(min3f(fx.y, fy.y, fz.y), bestTf);
    const bool h0 = (tnear0 <= tfar0) & (tfar0 >= 0.0f);
    const bool h1 = (tnear1 <= tfar1) & (tfar1 >= 0.0f);
    if (RAW) {
      tn[c] = tnear0; tn[c + 1] = tnear1;
      hit[c] = __ballot(tnear0 <= tfar0) & __ballot(tfar0 >= 0.0f);
      hit[c + 1] = __ballot(tnear1 <= tfar1) & __ballot(tfar1 >= 0.0f);
    } else {
      tn[c] = h0 ? tnear0 : kInf;
      tn[c + 1] = h1 ? tnear1 : kInf;
    }
  }
  if (RAW && cones) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (hit[c] == 0) continue;   // nobody enters this child anyway (wave-uniform)
      const float ax = c < 2 ? cone01[4 * c] : cone23[4 * c - 8], ay = c < 2 ? cone01[4 * c + 1] : cone23[4 * c - 7];
      const float az = c < 2 ? cone01[4 * c + 2] : cone23[4 * c - 6], thr = c < 2 ? cone01[4 * c + 3] : cone23[4 * c - 5];
      const float along = fmaf(az, r.dfz, fmaf(ay, r.dfy, ax * r.dfx));
      hit[c] &= ~__ballot(along > thr);
    }
  }
}

// The same slab test for a lane that is at a node of its own: the planes of the
// copy that matches the lane's direction signs (`copy`; 0 for queries that
// order the children themselves) through vector loads -- 112 bytes per node
// instead of the quantised node's 64, but no byte-to-float conversions, no
// per-node origin and scale arithmetic, tighter boxes, and children that
// arrive near to far.  No axis-parallel component (the caller checked).
#ifndef RT_LANE_CONES_EARLY
#define RT_LANE_CONES_EARLY 0
#endif
template <bool RAW>
__device__ __forceinline__ void bvhNodeSlabsLane(const DevScene& S, const BvhRay& r, int ref, unsigned int copy,
                                                 float bestTf, float tn[4], int kid[4], rt_mask hit[4],
                                                 bool cones = false) {
  const float kInf = __int_as_float(0x7f800000);
  const char* base = reinterpret_cast<const char*>(S.bvhF);
  const unsigned int node = ((unsigned int)ref * 8u + copy) * (unsigned int)RT_BVHF_BYTES;   // (node count checked at scene creation)
  const unsigned int oNX = r.negx ? 16u : 0u, oNY = r.negy ? 48u : 32u, oNZ = r.negz ? 80u : 64u;
  const rt_f4 nX = *reinterpret_cast<const rt_f4*>(base + (node + oNX));
  const rt_f4 fX = *reinterpret_cast<const rt_f4*>(base + (node + (oNX ^ 16u)));
  const rt_f4 nY = *reinterpret_cast<const rt_f4*>(base + (node + oNY));
  const rt_f4 fY = *reinterpret_cast<const rt_f4*>(base + (node + (oNY ^ 16u)));
  const rt_f4 nZ = *reinterpret_cast<const rt_f4*>(base + (node + oNZ));
  const rt_f4 fZ = *reinterpret_cast<const rt_f4*>(base + (node + (oNZ ^ 16u)));
  const rt_i4 kids = *reinterpret_cast<const rt_i4*>(base + (node + 96u));
#if RT_LANE_CONES_EARLY
  rt_f4 cnEarly[4];
  if (RAW && cones) {
#pragma unroll
    for (int c = 0; c < 4; c++) cnEarly[c] = *reinterpret_cast<const rt_f4*>(base + (node + 128u + 16u * c));
  }
#endif
  kid[0] = kids.x; kid[1] = kids.y; kid[2] = kids.z; kid[3] = kids.w;
  const rt_f2 ivXY = {r.ivx, r.ivy}, ivZZ = {r.ivz, r.ivz};
  const rt_f2 aX = {r.a0x, r.a1x}, aY = {r.a0y, r.a1y}, aZ = {r.a0z, r.a1z};
#define RT_PKFMA_V(dst, planes, ivp, ap, sel)                                                   \
  asm("v_pk_fma_f32 %0, %1, %2, %3 " sel : "=v"(dst) : "v"(planes), "v"(ivp), "v"(ap))
#pragma unroll
  for (int c = 0; c < 4; c += 2) {
    const rt_f2 pnx = {nX[c], nX[c + 1]}, pfx = {fX[c], fX[c + 1]};
    const rt_f2 pny = {nY[c], nY[c + 1]}, pfy = {fY[c], fY[c + 1]};
    const rt_f2 pnz = {nZ[c], nZ[c + 1]}, pfz = {fZ[c], fZ[c + 1]};
    rt_f2 nx, fx, ny, fy, nz, fz;
    RT_PKFMA_V(nx, pnx, ivXY, aX, "op_sel_hi:[1,0,0]");                   // planes * ivx + a0x
    RT_PKFMA_V(fx, pfx, ivXY, aX, "op_sel:[0,0,1] op_sel_hi:[1,0,1]");    // planes * ivx + a1x
    RT_PKFMA_V(ny, pny, ivXY, aY, "op_sel:[0,1,0] op_sel_hi:[1,1,0]");    // planes * ivy + a0y
    RT_PKFMA_V(fy, pfy, ivXY, aY, "op_sel:[0,1,1] op_sel_hi:[1,1,1]");    // planes * ivy + a1y
    RT_PKFMA_V(nz, pnz, ivZZ, aZ, "op_sel_hi:[1,0,0]");                   // planes * ivz + a0z
    RT_PKFMA_V(fz, pfz, ivZZ, aZ, "op_sel:[0,0,1] op_sel_hi:[1,0,1]");    // planes * ivz + a1z
    const float tnear0 = max3f(nx.x, ny.x, nz.x), tnear1 = max3f(nx.y, ny.y, nz.y);
    const float tfar0 = min2f(min3f(fx.x, fy.x, fz.x), bestTf);
    const float tfar1 = min2f(min3f(fx.y, fy.y, fz.y), bestTf);
    const bool h0 = (tnear0 <= tfar0) & (tfar0 >= 0.0f);
    const bool h1 = (tnear1 <= tfar1) & (tfar1 >= 0.0f);
    if (RAW) {
      tn[c] = tnear0; tn[c + 1] = tnear1;
      hit[c] = __ballot(tnear0 <= tfar0) & __ballot(tfar0 >= 0.0f);
      hit[c + 1] = __ballot(tnear1 <= tfar1) & __ballot(tfar1 >= 0.0f);
    } else {
      tn[c] = h0 ? tnear0 : kInf;
      tn[c + 1] = h1 ? tnear1 : kInf;
    }
  }
#undef RT_PKFMA_V
  if (RAW && cones) {   // see bvhNodeSlabsUniform
#if RT_LANE_CONES_EARLY
    const rt_f4* cn = cnEarly;
#else
    // (all four at once: asked for one by one, only where a child is entered, each load waits for itself)
    rt_f4 cn[4];
#pragma unroll
    for (int c = 0; c < 4; c++) cn[c] = *reinterpret_cast<const rt_f4*>(base + (node + 128u + 16u * c));
#endif
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float along = fmaf(cn[c].z, r.dfz, fmaf(cn[c].y, r.dfy, cn[c].x * r.dfx));
      hit[c] &= ~__ballot(along > cn[c].w);
    }
  }
}

// (tn, kid) ascending by tn: network (0,1)(2,3)(0,2)(1,3)(1,2)
__device__ __forceinline__ void bvhSort4(float tn[4], int kid[4]) {
#define RT_CSWAP(i, j)                                                 \
  {                                                                    \
    const bool sw = tn[j] < tn[i];                                     \
    const float tlo = sw ? tn[j] : tn[i], thi = sw ? tn[i] : tn[j];    \
    const int klo = sw ? kid[j] : kid[i], khi = sw ? kid[i] : kid[j];  \
    tn[i] = tlo; tn[j] = thi; kid[i] = klo; kid[j] = khi;              \
  }
  RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
}

// Which of two primitives at exactly the same distance the reference's scan over
// Scene::objects keeps: the one it meets first -- the earlier object, and within
// a mesh the earlier face (strict < at scene.cpp:148 and in
// Trimesh::intersectLocal, trimesh.cpp:62-82).  The BVH meets primitives in its
// own order, so ties are settled by this instead of by arrival.
__device__ __forceinline__ bool sceneOrderBefore(const DevScene& S, int prim, int other) {
  if (other == RT_PRIM_NONE) return true;
  const int fa = prim & ~RT_PRIM_FACE, fb = other & ~RT_PRIM_FACE;
  const int oa = (prim & RT_PRIM_FACE) ? S.tris[fa].owner : prim;
  const int ob = (other & RT_PRIM_FACE) ? S.tris[fb].owner : other;
  if (oa != ob) return oa < ob;
  return fa < fb;   // two faces of one mesh
}

// The primitives of leaf reference `ref` (< 0).  Closest hit: updates the best
// hit.  Occlusion: returns true at the first hit that satisfies the reference's
// test |(p + t d) - p|^2 < occDist2 (occDist2 < 0: any hit), with tHit set.
// rCull: object-space plane distances beyond it are of no interest (closest
// hit: they lie behind the best hit so far, and rCull follows it; occlusion:
// behind the light).
//
// TRI: the scene holds nothing but triangles of identity-transform meshes (the
// caller checked), so the code for the other primitive kinds and for object
// space is left out -- it costs the traversal kernels a third of their
// registers even when it never runs.
template <bool ANYHIT, bool STATS, bool TRI>
__device__ __forceinline__ bool bvhLeaf(const DevScene& S, const V3& p, const V3& d, const BvhRay& r, int ref,
                                        double occDist2, bool& haveOne, double& bestT, float& bestTf, double& rCull,
                                        int& bestPrim, double& tHit, TraceCounters& cnt) {
  const int x = ~ref;
  if (!TRI && (x & RT_BVH_OBJECT_BIT)) {
    const int oi = x & 0x3fffffff;
    double t;
    V3 N;
    if (STATS) cnt.objs++;
    if (objectIntersect<false>(S.objects[oi], p, d, t, N)) {
      if (ANYHIT) {
        const V3 Qh = p + (t * d);
        if (occDist2 < 0.0 || length2(Qh - p) < occDist2) { tHit = t; return true; }
      } else if (!haveOne || t < bestT || (t == bestT && sceneOrderBefore(S, oi, bestPrim))) {
        haveOne = true; bestT = t; bestPrim = oi;
        bestTf = roundUpF(bestT);
      }
    }
  } else {
    const int first = x >> 3, count = (x & 7) + 1;
    for (int k = 0; k < count; k++) {
#ifdef RT_DIVERGENCE_STATS
      if (STATS && RT_DIVERGENCE_STATS != 3 && (threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cnt.prims++;
#else
      if (STATS) cnt.prims++;
#endif
      double rt, tw;
      bool hit;
      int face;
      if (TRI) {
        const DevTriFast& T = S.bvhTrisFast[first + k];
        hit = triTestFast(T, p, r.ldI, r.cullBack, rCull, rt);
        tw = rt / r.lenI;
        face = T.face;
      } else {
        const DevTri& T = S.bvhTris[first + k];
        face = T.face;
        if (T.axes & 16) {
          hit = triTest<true>(T, p, r.ldI, r.cullBack, rt, nullptr);
          tw = rt / r.lenI;
        } else {
          const LocalRay lr = toLocal(S.objects[T.owner].inverse, p, d);
          hit = triTest<true>(T, lr.p, lr.d, r.cullBack, rt, nullptr);
          tw = rt / lr.len;
        }
      }
#if defined(RT_DIVERGENCE_STATS) && RT_DIVERGENCE_STATS == 2   // triangle iterations in which no lane hits
      if (STATS && __ballot(hit) == 0 && (threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cnt.nodes++;
#endif
      if (!hit) continue;
      if (ANYHIT) {
        const V3 Qh = p + (tw * d);
        if (occDist2 < 0.0 || length2(Qh - p) < occDist2) { tHit = tw; return true; }
      } else if (!haveOne || tw < bestT || (tw == bestT && sceneOrderBefore(S, RT_PRIM_FACE | face, bestPrim))) {
        haveOne = true; bestT = tw; bestPrim = RT_PRIM_FACE | face;
        bestTf = roundUpF(bestT);
        // rt / lenI <= bestT implies rt <= bestT * lenI * (1 + 2^-52); a hair more
        if (TRI) rCull = (bestT * r.lenI) * (1.0 + 1.0e-12);
      }
    }
  }
  return false;
}

// (Measured and dropped: pushing the closest-hit children unsorted, +4 %
// time; parking one leaf per lane so that lanes stay in the node loop together,
// +14 %.)
//
// Closest hit (ANYHIT false) or occlusion query (ANYHIT true) on the library's
// BVH.  Closest hit: returns the primitive and t like traceKdReference; hits
// beyond tLimit are of no interest.  Occlusion: returns 0 as soon as any hit
// satisfies the reference's occlusion test |(p + t d) - p|^2 < occDist2
// (light.cpp:77-80,123-126; occDist2 < 0: any hit counts, light.cpp:21), else
// RT_PRIM_NONE.
template <bool ANYHIT, bool STATS, bool TRI = false>
__device__ int traceBvh(const DevScene& S, const V3& p, const V3& d, int rayType, double tLimit,
                        double occDist2, double& tOut, TraceCounters& cnt, rt_lds_u32* ldsStack) {
  tOut = 0.0;
  if (!S.bvh_valid) return RT_PRIM_NONE;
  if (anynan(d) || anynan(p)) {
    if (TRI) return RT_PRIM_NONE;   // no triangle accepts a NaN ray, see bvhScanNaN
    double bt;
    const int bp = bvhScanNaN<STATS>(S, p, d, bt, cnt);
    if (!ANYHIT) { tOut = bt; return bp; }
    if (bp != RT_PRIM_NONE) {
      const V3 Qh = p + (bt * d);
      if (occDist2 < 0.0 || length2(Qh - p) < occDist2) { tOut = bt; return 0; }
    }
    return RT_PRIM_NONE;
  }
  if (TRI && S.bvhCones && S.cull && rayType != RT_RAY_REFRACTION && (S.bvhCones > 1 || rayType != RT_RAY_VISIBILITY)) {
    // The normal cones of the root's four children before anything else: a ray
    // that meets all of the scene from behind (a shadow ray leaving a terrain
    // for a light above it) ends here, before its set-up.  (Measured and
    // dropped: putting off the object-space direction -- a square root and
    // three divisions -- until a ray's first leaf; the code that moved into
    // the loop cost every other ray 3 %.)
    const DevBvhNodeF& root = S.bvhF[0];
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    bool behindAll = true;
#pragma unroll
    for (int c = 0; c < 4; c++)
      behindAll &= fmaf(root.cone[c][2], dz, fmaf(root.cone[c][1], dy, root.cone[c][0] * dx)) > root.cone[c][3];
    if (behindAll) return RT_PRIM_NONE;
  }
  BvhRay r;
  bvhRaySetup(S, p, d, rayType, r);
  if (rayType == RT_RAY_VISIBILITY) {   // every other ray starts on a surface, inside the box
    // Early out for rays that pass the whole structure by (sky): the root box
    // through the same widened binary32 slab distances as any node (its planes
    // are within bvhErrG too), so rounding can only let a ray through, never
    // turn one away.  An axis the ray runs parallel to constrains nothing here.
    const float kInf = __int_as_float(0x7f800000);
    const float* bb = S.bvh_bb_f;
    const float nx = r.zx ? -kInf : fmaf(r.negx ? bb[3] : bb[0], r.ivx, r.a0x);
    const float fx = r.zx ? kInf : fmaf(r.negx ? bb[0] : bb[3], r.ivx, r.a1x);
    const float ny = r.zy ? -kInf : fmaf(r.negy ? bb[4] : bb[1], r.ivy, r.a0y);
    const float fy = r.zy ? kInf : fmaf(r.negy ? bb[1] : bb[4], r.ivy, r.a1y);
    const float nz = r.zz ? -kInf : fmaf(r.negz ? bb[5] : bb[2], r.ivz, r.a0z);
    const float fz = r.zz ? kInf : fmaf(r.negz ? bb[2] : bb[5], r.ivz, r.a1z);
    const float t0 = fmaxf(fmaxf(nx, ny), nz), t1 = fminf(fminf(fx, fy), fz);
    if (t0 > t1 || t1 < 0.0f) return RT_PRIM_NONE;
  }
  bvhRaySetupLocal(p, d, r);
  bool haveOne = false;
  double bestT = tLimit;   // doubles as the far cull bound until a hit exists
  float bestTf = roundUpF(bestT);
  // plane distances (object space of the identity-transform meshes) that cannot matter, see bvhLeaf
  double rCull = (tLimit * r.lenI) * (1.0 + 1.0e-12);
  int bestPrim = RT_PRIM_NONE;

  typedef BvhSlot<ANYHIT> Slot;
  typedef typename Slot::lds_t lds_t;
  typedef typename Slot::word_t word_t;
  const unsigned int stride = blockDim.x;
  lds_t* const base = (lds_t*)ldsStack + threadIdx.x;
  lds_t* const ldsEnd = base + RT_LDS_STACK * stride;
  lds_t* const fastLimit = base + (RT_LDS_STACK - 3) * stride;   // a visit may write three rows at the top
  lds_t* top = base;                                             // next free row of this lane
  word_t deep[RT_BVH_STACK - RT_LDS_STACK];                      // entries past the LDS part (depth checked at scene creation)
  int nDeep = 0;
  const float kInf = __int_as_float(0x7f800000);

  // One visit's new entries.  Children 0..3 are ordered near to far, the masks
  // say which of them each lane's ray enters: the nearest of those is visited
  // next, the others go onto the stack farthest first.
  const unsigned int rowBytes = stride * (unsigned int)sizeof(word_t);
#ifdef RT_STACK_TOP
  // The newest entry stays in registers (topRef / topTn); a new entry moves it
  // to the LDS rows.  A pop then hands the register entry out at once and the
  // load that refills it overlaps the work on the popped subtree.
  int topRef = RT_BVH_EMPTY;
  float topTn = 0.0f;
#define RT_DESCEND_STEP(C, P)                                                                      \
      *(lds_t*)t_ = Slot::pack(topRef, topTn); t_ += pickByMask(0u, rowBytes, (P) & valid_);       \
      topRef = (int)pickByMask((unsigned int)topRef, (unsigned int)kid[C], (P));                   \
      topTn = __uint_as_float(pickByMask(__float_as_uint(topTn), __float_as_uint(tn[C]), (P)));    \
      valid_ |= (P);
#define RT_DESCEND(M0, M1, M2, M3)                                                                 \
  {                                                                                                \
    const rt_mask m0_ = (M0), m1_ = (M1), m2_ = (M2), m3_ = (M3);                                  \
    const rt_mask p3_ = m3_ & (m0_ | m1_ | m2_), p2_ = m2_ & (m0_ | m1_), p1_ = m1_ & m0_;         \
    if (__ballot(top > fastLimit) == 0) {                                                          \
      rt_lds_u8* t_ = (rt_lds_u8*)top;                                                             \
      rt_mask valid_ = __ballot(topRef != RT_BVH_EMPTY);                                           \
      RT_DESCEND_STEP(3, p3_) RT_DESCEND_STEP(2, p2_) RT_DESCEND_STEP(1, p1_)                      \
      top = (lds_t*)t_;                                                                            \
    } else {                                                                                       \
      if (pickByMask(0u, 1u, p3_)) RT_PUSH_TOP_SLOW(kid[3], tn[3])                                 \
      if (pickByMask(0u, 1u, p2_)) RT_PUSH_TOP_SLOW(kid[2], tn[2])                                 \
      if (pickByMask(0u, 1u, p1_)) RT_PUSH_TOP_SLOW(kid[1], tn[1])                                 \
    }                                                                                              \
    unsigned int next_ = pickByMask((unsigned int)RT_BVH_EMPTY, (unsigned int)kid[3], m3_);        \
    next_ = pickByMask(next_, (unsigned int)kid[2], m2_);                                          \
    next_ = pickByMask(next_, (unsigned int)kid[1], m1_);                                          \
    ref = (int)pickByMask(next_, (unsigned int)kid[0], m0_);                                       \
  }
#define RT_PUSH_TOP_SLOW(R, T)                                                                     \
  {                                                                                                \
    if (topRef != RT_BVH_EMPTY) RT_PUSH_SLOW(topRef, topTn)                                        \
    topRef = (R); topTn = (T);                                                                     \
  }
#else
#ifndef RT_PUSH_ALWAYS
#define RT_PUSH_ALWAYS 0
#endif
  // (a slot that no lane of the wavefront uses is skipped as a whole -- rays of a
  // wavefront mostly enter the same children; inside a slot no lane branches)
#define RT_DESCEND(M0, M1, M2, M3)                                                                 \
  {                                                                                                \
    const rt_mask m0_ = (M0), m1_ = (M1), m2_ = (M2), m3_ = (M3);                                  \
    const rt_mask p3_ = m3_ & (m0_ | m1_ | m2_), p2_ = m2_ & (m0_ | m1_), p1_ = m1_ & m0_;         \
    if (__ballot(top > fastLimit) == 0) {                                                          \
      rt_lds_u8* t_ = (rt_lds_u8*)top;                                                             \
      if (RT_PUSH_ALWAYS || p3_) { *(lds_t*)t_ = Slot::pack(kid[3], tn[3]); t_ += pickByMask(0u, rowBytes, p3_); }   \
      if (RT_PUSH_ALWAYS || p2_) { *(lds_t*)t_ = Slot::pack(kid[2], tn[2]); t_ += pickByMask(0u, rowBytes, p2_); }   \
      if (RT_PUSH_ALWAYS || p1_) { *(lds_t*)t_ = Slot::pack(kid[1], tn[1]); t_ += pickByMask(0u, rowBytes, p1_); }   \
      top = (lds_t*)t_;                                                                            \
    } else {                                                                                       \
      if (pickByMask(0u, 1u, p3_)) RT_PUSH_SLOW(kid[3], tn[3])                                     \
      if (pickByMask(0u, 1u, p2_)) RT_PUSH_SLOW(kid[2], tn[2])                                     \
      if (pickByMask(0u, 1u, p1_)) RT_PUSH_SLOW(kid[1], tn[1])                                     \
    }                                                                                              \
    unsigned int next_ = pickByMask((unsigned int)RT_BVH_EMPTY, (unsigned int)kid[3], m3_);        \
    next_ = pickByMask(next_, (unsigned int)kid[2], m2_);                                          \
    next_ = pickByMask(next_, (unsigned int)kid[1], m1_);                                          \
    ref = (int)pickByMask(next_, (unsigned int)kid[0], m0_);                                       \
  }
#endif
#define RT_PUSH_SLOW(R, T)                                                                         \
  {                                                                                                \
    if (top != ldsEnd) { *top = Slot::pack(R, T); top += stride; }                                 \
    else deep[nDeep++] = Slot::pack(R, T);                                                         \
  }

#ifndef RT_NO_UNIFORM_PATH
  // wavefront-wide facts about the rays that are still here: do they share their
  // direction signs, and is none of them parallel to an axis
  const unsigned int myNeg = (r.negx ? 1u : 0u) | (r.negy ? 2u : 0u) | (r.negz ? 4u : 0u);
  const unsigned int negMask = (unsigned int)__builtin_amdgcn_readfirstlane((int)myNeg);
  const bool uniformOk = __ballot((myNeg != negMask) | r.zx | r.zy | r.zz) == 0;
  // this lane can read the binary32 planes by itself: no axis-parallel component, offsets fit 32 bits
  const bool laneOk = S.bvhLaneF && !(r.zx | r.zy | r.zz);
  // some ray of the wavefront culls back faces and the tree has normal cones for that
  const bool coneWave = TRI && __ballot((r.dfx != 0.0f) | (r.dfy != 0.0f) | (r.dfz != 0.0f)) != 0;
#endif
  int ref = 0;
  for (;;) {
    if (ref >= 0) {
#ifdef RT_DIVERGENCE_STATS   // diagnostics build: count wavefront iterations instead of lane visits
      if (STATS && RT_DIVERGENCE_STATS != 2 && (threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cnt.nodes++;
#if RT_DIVERGENCE_STATS == 3   // ... and, as "primitives", those in which every lane at a node is at the same node
      {
        const int uref = __builtin_amdgcn_readfirstlane(ref);
        if (STATS && __ballot(ref != uref) == 0 && (threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cnt.prims++;
      }
#endif
#else
      if (STATS) cnt.nodes++;
#endif
      float tn[4];
      int kid[4];
#ifndef RT_NO_UNIFORM_PATH
      // Three node visits in four find every lane of the wavefront at the same
      // node (neighbouring samples, neighbouring shade points): those take the
      // scalar form of the slab test, on the copy of the node that lists its
      // children near to far for the wavefront's direction signs (by box centre
      // along the signs' diagonal, fixed when the scene was created): no
      // per-lane sort.  The order is only nearly the order of entry distances;
      // a closer hit found late still culls what was pushed before it (pop), so
      // the result does not depend on it.  (Mesh scenes only: in the kernels
      // that also carry the code of the other primitive kinds the extra
      // registers push them over an occupancy step.)
      const int uref = __builtin_amdgcn_readfirstlane(ref);
      const bool sameNode = TRI && uniformOk && __ballot(ref != uref) == 0;
      if (TRI && sameNode) {
        rt_mask hk[4];
        bvhNodeSlabsUniform<true>(S, r, uref, negMask, negMask, bestTf, tn, kid, hk, coneWave);
        RT_DESCEND(hk[0], hk[1], hk[2], hk[3])
      } else if (TRI && (ANYHIT || coneWave) && laneOk) {
        // an occlusion query's lane at a node of its own: the same binary32
        // planes through vector loads, the copy for its own direction signs
        // (for closest-hit queries the quantised node and a sort were measured
        // faster: 64 instead of 112 bytes per visit)
        rt_mask hk[4];
        bvhNodeSlabsLane<true>(S, r, ref, myNeg, bestTf, tn, kid, hk, coneWave);
        RT_DESCEND(hk[0], hk[1], hk[2], hk[3])
      } else
#endif
      {
        bvhNodeSlabs(S, p, r, ref, bestTf, tn, kid);
        bvhSort4(tn, kid);
        RT_DESCEND(__ballot(tn[0] < kInf), __ballot(tn[1] < kInf), __ballot(tn[2] < kInf), __ballot(tn[3] < kInf))
      }
      // (The shape of this loop matters: with this `continue` the compiler keeps
      // the lanes that are still descending in a loop of node visits of their
      // own and parks the lanes that have reached a leaf or need a pop until
      // the last of them has, so that the triangle tests below run with as many
      // lanes as the wavefront can muster.  Without it every iteration runs a
      // node visit, a leaf and a pop for whichever lanes want one: a third more
      // vector-memory instructions and 13 % more time.)
      if (ref != RT_BVH_EMPTY) continue;
    } else {
      double tHit;
      if (bvhLeaf<ANYHIT, STATS, TRI>(S, p, d, r, ref, occDist2, haveOne, bestT, bestTf, rCull, bestPrim, tHit, cnt)) {
        tOut = tHit;
        return 0;
      }
    }
    {
      ref = RT_BVH_EMPTY;
      // pop the next subtree that can still contain a closer hit
#ifdef RT_STACK_TOP
      while (topRef != RT_BVH_EMPTY) {
        const int candRef = topRef;
        const float candTn = topTn;
        if (nDeep > 0) {
          const word_t e = deep[--nDeep];
          topRef = Slot::refOf(e); topTn = Slot::tnOf(e);
        } else if (top != base) {
          top -= stride;
          const word_t e = *top;
          topRef = Slot::refOf(e); topTn = Slot::tnOf(e);
        } else {
          topRef = RT_BVH_EMPTY;
        }
        if (ANYHIT || candTn <= bestTf) { ref = candRef; break; }
      }
#else
      if (__ballot(nDeep > 0) == 0) {
        while (top != base) {
          top -= stride;
          const word_t e = *top;
          if (ANYHIT || Slot::tnOf(e) <= bestTf) { ref = Slot::refOf(e); break; }
        }
      } else {
        for (;;) {
          word_t e;
          if (nDeep > 0) e = deep[--nDeep];
          else if (top != base) { top -= stride; e = *top; }
          else break;
          if (ANYHIT || Slot::tnOf(e) <= bestTf) { ref = Slot::refOf(e); break; }
        }
      }
#endif
      if (ref == RT_BVH_EMPTY) break;
    }
  }
#undef RT_DESCEND
#undef RT_PUSH_SLOW
#undef RT_DESCEND_STEP
#undef RT_PUSH_TOP_SLOW
  tOut = haveOne ? bestT : 0.0;
  return bestPrim;
}

}  // namespace rtd
