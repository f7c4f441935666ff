// rt_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A plain, recursive, single-source CPU restatement of the reference ray
// tracer's per-pixel render path (kaivanwadia/RayTracer, all FP64):
//   RayTracer::tracePixel / tracePixelAntiAlias / trace / traceRay
//   Scene::intersect / intersectKdTree, Geometry::intersect, BoundingBox::intersect
//   Sphere / Box / Square / Cylinder / Cone / Trimesh / TrimeshFace::intersectLocal
//   Material::shade, {Directional,Point,Spot}Light, CubeMap::getColor
//   GraphicalUI::applyFilter + doAntiAliasing (adaptive anti-aliasing driver)
// Each function cites the reference file:line it follows and keeps the
// reference's operation order (compile with -ffp-contract=off).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; the HIP product path never does.
//
// PARITY UNPINNED: the reference cannot be built in this image without
// stand-ins (every translation unit includes <FL/gl.h> through
// src/vecmath/vec.h:20, FLTK is absent, and the virtual glDraw* members it
// declares are defined only in the FLTK/OpenGL UI sources), and the reference
// ships no tests, fixtures, scenes or golden images.  The only reference
// outputs available are the few values recorded while the reference was
// surveyed (anti-aliasing jitter values for seed 1234567 and the NaN -> byte 0
// rule); tests/test_oracle_pins.py checks those.  Everything else in this file
// is pinned by reading the source only.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt_amd.h"

namespace {

const double RAY_EPSILON = 0.00000001;  // src/scene/ray.h:114

struct V3 {
  double n[3];
  V3() : n{0.0, 0.0, 0.0} {}
  V3(double x, double y, double z) : n{x, y, z} {}
  explicit V3(const double* p) : n{p[0], p[1], p[2]} {}
  double& operator[](int i) { return n[i]; }
  double operator[](int i) const { return n[i]; }
};
// src/vecmath/vec.h:241-242,737-744,759-761,781-785,818-820
inline V3 operator+(const V3& a, const V3& b) { return V3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3 operator-(const V3& a) { return V3(-a[0], -a[1], -a[2]); }
inline V3 operator*(const V3& a, double d) { return V3(a[0] * d, a[1] * d, a[2] * d); }
inline V3 operator*(double d, const V3& a) { return a * d; }
inline double dot(const V3& a, const V3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline V3 cross(const V3& a, const V3& b) {
  return V3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline V3 prod(const V3& a, const V3& b) { return V3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline double length2(const V3& a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }
inline double length(const V3& a) { return std::sqrt(length2(a)); }
inline void normalize(V3& a) {  // vec.h:257-260
  double len = length(a);
  a[0] /= len; a[1] /= len; a[2] /= len;
}
inline bool iszero(const V3& a) { return a[0] == 0 && a[1] == 0 && a[2] == 0; }

struct Ray { V3 p, d; int type; };

struct Mat { V3 ke, ka, ks, kd, kr, kt, shin, index; bool shinMapped; };

struct Isect {
  double t = 0.0;
  V3 N, bary;
  int objectID = 0;
  Mat mat;
};

struct Counters { unsigned long long ray[5] = {0, 0, 0, 0, 0}; unsigned long long shadePoints = 0; };

Mat toMat(const RtMaterial& m) {
  Mat r;
  r.ke = V3(m.ke); r.ka = V3(m.ka); r.ks = V3(m.ks); r.kd = V3(m.kd);
  r.kr = V3(m.kr); r.kt = V3(m.kt); r.shin = V3(m.shininess); r.index = V3(m.index);
  r.shinMapped = m.shininess_mapped != 0;
  return r;
}
// MaterialParameter::intensityValue, src/scene/material.cpp:158-167
inline double intensityValue(const V3& v) { return (0.299 * v[0]) + (0.587 * v[1]) + (0.114 * v[2]); }
inline double shininessOf(const Mat& m) {  // material.h:209-215
  return m.shinMapped ? 128.0 * intensityValue(m.shin) : intensityValue(m.shin);
}

// BoundingBox::intersect, src/scene/bbox.h:62-89.  tMin/tMax keep whatever
// they held when the function returns early.
bool boxIntersect(const double* bmin, const double* bmax, const Ray& r, double& tMin, double& tMax) {
  tMin = -1.0e308;
  tMax = 1.0e308;
  for (int a = 0; a < 3; a++) {
    double vd = r.d[a];
    if (vd == 0.0) continue;
    double v1 = bmin[a] - r.p[a];
    double v2 = bmax[a] - r.p[a];
    double t1 = v1 / vd;
    double t2 = v2 / vd;
    if (t1 > t2) { double tt = t1; t1 = t2; t2 = tt; }
    if (t1 > tMin) tMin = t1;
    if (t2 < tMax) tMax = t2;
    if (tMin > tMax) return false;
    if (tMax < RAY_EPSILON) return false;
  }
  return true;
}

class Tracer {
 public:
  Tracer(const RtSceneDesc* d, const RtRenderParams* p) : S(*d), P(*p) {}
  const RtSceneDesc& S;
  const RtRenderParams& P;
  Counters cnt;

  inline V3 at(const Ray& r, double t) const { return r.p + (t * r.d); }  // ray.h:42-43

  // ---------------------------------------------------------------- primitives
  // Sphere::intersectLocal, src/SceneObjects/Sphere.cpp:8-41
  bool sphereLocal(const Ray& r, Isect& i) const {
    V3 v = -r.p;
    double b = dot(v, r.d);
    double disc = b * b - dot(v, v) + 1;
    if (disc < 0.0) return false;
    disc = std::sqrt(disc);
    double t2 = b + disc;
    if (t2 <= RAY_EPSILON) return false;
    double t1 = b - disc;
    if (t1 > RAY_EPSILON) {
      i.t = t1;
      i.N = at(r, t1);
      normalize(i.N);
    } else {
      i.t = t2;
      i.N = at(r, t2);
      normalize(i.N);
    }
    return true;
  }

  // Box::intersectLocal, src/SceneObjects/Box.cpp:11-79
  bool boxLocal(const Ray& r, Isect& i) const {
    const V3& p = r.p;
    const V3& d = r.d;
    double bestT = 1e100;
    int bestIndex = -1;
    for (int it = 0; it < 6; it++) {
      int mod0 = it % 3;
      if (d[mod0] == 0) continue;
      double t = ((it / 3) - 0.5 - p[mod0]) / d[mod0];
      if (t < RAY_EPSILON || t > bestT) continue;
      int mod1 = (it + 1) % 3, mod2 = (it + 2) % 3;
      double x = p[mod1] + t * d[mod1];
      double y = p[mod2] + t * d[mod2];
      if (x <= 0.5 && x >= -0.5 && y <= 0.5 && y >= -0.5) {
        if (bestT > t) { bestT = t; bestIndex = it; }
      }
    }
    if (bestIndex < 0) return false;
    i.t = bestT;
    if (bestIndex < 3)
      i.N = V3(-double(bestIndex == 0), -double(bestIndex == 1), -double(bestIndex == 2));
    else
      i.N = V3(double(bestIndex == 3), double(bestIndex == 4), double(bestIndex == 5));
    return true;
  }

  // Square::intersectLocal, src/SceneObjects/Square.cpp:9-45
  bool squareLocal(const Ray& r, Isect& i) const {
    if (r.d[2] == 0.0) return false;
    double t = -r.p[2] / r.d[2];
    if (t <= RAY_EPSILON) return false;
    V3 Pp = at(r, t);
    if (Pp[0] < -0.5 || Pp[0] > 0.5) return false;
    if (Pp[1] < -0.5 || Pp[1] > 0.5) return false;
    i.t = t;
    i.N = (r.d[2] > 0.0) ? V3(0.0, 0.0, -1.0) : V3(0.0, 0.0, 1.0);
    return true;
  }

  // Cylinder::intersectBody / intersectCaps / intersectLocal, Cylinder.cpp:8-150
  bool cylBody(const Ray& r, Isect& i, bool capped) const {
    double x0 = r.p[0], y0 = r.p[1], x1 = r.d[0], y1 = r.d[1];
    double a = x1 * x1 + y1 * y1;
    double b = 2.0 * (x0 * x1 + y0 * y1);
    double c = x0 * x0 + y0 * y0 - 1.0;
    if (0.0 == a) return false;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return false;
    disc = std::sqrt(disc);
    double t2 = (-b + disc) / (2.0 * a);
    if (t2 <= RAY_EPSILON) return false;
    double t1 = (-b - disc) / (2.0 * a);
    if (t1 > RAY_EPSILON) {
      V3 Pp = at(r, t1);
      double z = Pp[2];
      if (z >= 0.0 && z <= 1.0) {
        i.t = t1;
        i.N = V3(Pp[0], Pp[1], 0.0);
        normalize(i.N);
        return true;
      }
    }
    V3 Pp = at(r, t2);
    double z = Pp[2];
    if (z >= 0.0 && z <= 1.0) {
      i.t = t2;
      V3 normal(Pp[0], Pp[1], 0.0);
      if (!capped && dot(normal, r.d) > 0) normal = -normal;
      i.N = normal;
      normalize(i.N);
      return true;
    }
    return false;
  }
  bool cylCaps(const Ray& r, Isect& i, bool capped) const {
    if (!capped) return false;
    double pz = r.p[2], dz = r.d[2];
    if (0.0 == dz) return false;
    double t1, t2;
    if (dz > 0.0) { t1 = (-pz) / dz; t2 = (1.0 - pz) / dz; }
    else { t1 = (1.0 - pz) / dz; t2 = (-pz) / dz; }
    if (t2 < RAY_EPSILON) return false;
    if (t1 >= RAY_EPSILON) {
      V3 p = at(r, t1);
      if ((p[0] * p[0] + p[1] * p[1]) <= 1.0) {
        i.t = t1;
        i.N = (dz > 0.0) ? V3(0.0, 0.0, -1.0) : V3(0.0, 0.0, 1.0);
        return true;
      }
    }
    V3 p = at(r, t2);
    if ((p[0] * p[0] + p[1] * p[1]) <= 1.0) {
      i.t = t2;
      i.N = (dz > 0.0) ? V3(0.0, 0.0, 1.0) : V3(0.0, 0.0, -1.0);
      return true;
    }
    return false;
  }
  bool cylinderLocal(const Ray& r, Isect& i, bool capped) const {
    if (cylCaps(r, i, capped)) {
      Isect ii;
      if (cylBody(r, ii, capped)) {
        if (ii.t < i.t) { i.t = ii.t; i.N = ii.N; }
      }
      return true;
    }
    return cylBody(r, i, capped);
  }

  // Cone::intersectLocal, src/SceneObjects/Cone.cpp:7-114
  bool coneLocal(const RtObject& o, const Ray& r, Isect& i) const {
    const double beta_squared = o.cone_beta_squared, gamma = o.cone_gamma, height = o.cone_height;
    const double b_radius = o.cone_b_radius, t_radius = o.cone_t_radius;
    const bool capped = o.capped != 0;
    V3 normal;
    const V3& R0 = r.p;
    const V3& Rd = r.d;
    double pz = R0[2], dz = Rd[2];
    double a = Rd[0] * Rd[0] + Rd[1] * Rd[1] - beta_squared * Rd[2] * Rd[2];
    if (a == 0.0) return false;
    double b = 2 * (R0[0] * Rd[0] + R0[1] * Rd[1] - beta_squared * ((R0[2] + gamma) * Rd[2]));
    double c = -beta_squared * (gamma + R0[2]) * (gamma + R0[2]) + R0[0] * R0[0] + R0[1] * R0[1];
    double disc = b * b - 4 * a * c;
    double theRoot = RAY_EPSILON;
    if (disc <= 0) return false;
    disc = std::sqrt(disc);
    double nearRoot = (-b + disc) / (2 * a);
    double farRoot = (-b - disc) / (2 * a);
    auto good = [&](const V3& q) { return !(q[2] < 0 || q[2] > height); };
    bool nearGood = good(at(r, nearRoot));
    if (nearGood && (nearRoot > theRoot)) {
      theRoot = nearRoot;
      V3 q = at(r, theRoot);
      normal = V3(q[0], q[1], -2.0 * beta_squared * (q[2] + gamma));
    }
    bool farGood = good(at(r, farRoot));
    if (farGood && ((nearGood && farRoot < theRoot) || farRoot > RAY_EPSILON)) {
      theRoot = farRoot;
      V3 q = at(r, theRoot);
      normal = V3(q[0], q[1], -2.0 * beta_squared * (q[2] + gamma));
    }
    if (!capped && dot(normal, r.d) > 0) normal = -normal;
    double t1 = (-pz) / dz;
    double t2 = (height - pz) / dz;
    V3 p = at(r, t1);
    if (capped) {
      if (p[0] * p[0] + p[1] * p[1] <= b_radius * b_radius) {
        if (t1 < theRoot && t1 > RAY_EPSILON) {
          theRoot = t1;
          normal = (dz > 0.0) ? V3(0.0, 0.0, -1.0) : V3(0.0, 0.0, 1.0);
        }
      }
      V3 q = at(r, t2);
      if (q[0] * q[0] + q[1] * q[1] <= t_radius * t_radius) {
        if (t2 < theRoot && t2 > RAY_EPSILON) {
          theRoot = t2;
          normal = (dz > 0.0) ? V3(0.0, 0.0, 1.0) : V3(0.0, 0.0, -1.0);
        }
      }
    }
    if (theRoot <= RAY_EPSILON) return false;
    i.t = theRoot;
    normalize(normal);
    i.N = normal;
    return true;
  }

  // TrimeshFace::intersectLocal, src/SceneObjects/trimesh.cpp:91-208
  bool faceLocal(int faceIdx, const RtMesh& mesh, const RtObject& owner, const Ray& r, Isect& i) const {
    const RtFace& f = S.faces[faceIdx];
    const V3 normal(f.normal);
    if (S.backface_cull) {
      if (r.type != RT_RAY_REFRACTION) {
        double cosineAngle = dot(normal, r.d);
        if (cosineAngle > 0) return false;
      }
    }
    const double* vb = S.vertices + 3 * (size_t)mesh.first_vertex;
    const V3 a(vb + 3 * f.ids[0]), b(vb + 3 * f.ids[1]), c(vb + 3 * f.ids[2]);
    double dConstant = -(dot(a, normal));
    if (dot(normal, r.d) == 0) return false;
    double rayT = -(dot(normal, r.p) + dConstant) / (dot(normal, r.d));
    if (rayT < RAY_EPSILON) return false;
    V3 p = r.p + rayT * r.d;
    int x, y;
    V3 na(std::fabs(normal[0]), std::fabs(normal[1]), std::fabs(normal[2]));
    if (na[0] >= na[1] && na[0] >= na[2]) { x = 1; y = 2; }
    else if (na[1] >= na[0] && na[1] >= na[2]) { x = 0; y = 2; }
    else { x = 0; y = 1; }
    V3 aUV(a[x], a[y], 0.0), bUV(b[x], b[y], 0.0), cUV(c[x], c[y], 0.0), pUV(p[x], p[y], 0.0);
    double ABCarea = length(cross(bUV - aUV, cUV - aUV)) / 2;
    double PBCarea = length(cross(bUV - pUV, cUV - pUV)) / 2;
    double APCarea = length(cross(pUV - aUV, cUV - aUV)) / 2;
    double ABParea = length(cross(bUV - aUV, pUV - aUV)) / 2;
    V3 bary(PBCarea / ABCarea, APCarea / ABCarea, ABParea / ABCarea);
    double total = bary[0] + bary[1] + bary[2];
    if (total <= (1 + RAY_EPSILON) && total >= (1 - RAY_EPSILON)) {
      i.t = rayT;
      if (S.smooth_shading && mesh.has_vertex_normals) {
        const double* nb = S.vnormals + 3 * (size_t)mesh.first_vertex;
        V3 nA(nb + 3 * f.ids[0]), nB(nb + 3 * f.ids[1]), nC(nb + 3 * f.ids[2]);
        i.N = (bary[0] * nA) + (bary[1] * nB) + (bary[2] * nC);
      } else {
        i.N = normal;
      }
      normalize(i.N);
      i.bary = bary;
      if (S.smooth_shading && mesh.has_vertex_materials) {
        // Material() + b0*A + b1*B + b2*C with Material()'s index = 1
        // (material.h:162-172,183-195,277-289)
        const int* vm = S.vmaterials + mesh.first_vertex;
        Mat m[3] = {toMat(S.materials[vm[f.ids[0]]]), toMat(S.materials[vm[f.ids[1]]]),
                    toMat(S.materials[vm[f.ids[2]]])};
        Mat pm;
        pm.index = V3(1.0, 1.0, 1.0);
        pm.shinMapped = false;
        for (int k = 0; k < 3; k++) {
          pm.ke = pm.ke + m[k].ke * bary[k];
          pm.ka = pm.ka + m[k].ka * bary[k];
          pm.ks = pm.ks + m[k].ks * bary[k];
          pm.kd = pm.kd + m[k].kd * bary[k];
          pm.kr = pm.kr + m[k].kr * bary[k];
          pm.kt = pm.kt + m[k].kt * bary[k];
          pm.index = pm.index + m[k].index * bary[k];
          pm.shin = pm.shin + m[k].shin * bary[k];
        }
        i.mat = pm;
      } else {
        i.mat = toMat(S.materials[owner.material]);
      }
      i.objectID = f.object_id;
      return true;
    }
    return false;
  }

  // Trimesh::intersectLocal, src/SceneObjects/trimesh.cpp:62-82 (brute-force path)
  bool trimeshLocal(const RtObject& o, const Ray& r, Isect& i) const {
    const RtMesh& mesh = S.meshes[o.mesh];
    bool have_one = false;
    for (int k = 0; k < mesh.n_faces; k++) {
      Isect cur;
      if (faceLocal(mesh.first_face + k, mesh, o, r, cur)) {
        if (!have_one || (cur.t < i.t)) { i = cur; have_one = true; }
      }
    }
    if (!have_one) i.t = 1000.0;
    return have_one;
  }

  // Geometry::intersect, src/scene/scene.cpp:13-36.  `faceIdx` >= 0 runs it
  // for a TrimeshFace (bounds = the face's local box, transform = the mesh's).
  bool geometryIntersect(const RtObject& o, int faceIdx, const Ray& r, Isect& i) const {
    double tmin, tmax;
    if (faceIdx >= 0) {
      const RtMesh& mesh = S.meshes[o.mesh];
      const RtFace& f = S.faces[faceIdx];
      const double* vb = S.vertices + 3 * (size_t)mesh.first_vertex;
      double lo[3], hi[3];
      for (int a = 0; a < 3; a++) {  // TrimeshFace::ComputeLocalBoundingBox, trimesh.h:136-145
        double v0 = vb[3 * f.ids[0] + a], v1 = vb[3 * f.ids[1] + a], v2 = vb[3 * f.ids[2] + a];
        hi[a] = std::max(v2, std::max(v0, v1));
        lo[a] = std::min(v2, std::min(v0, v1));
      }
      if (!boxIntersect(lo, hi, r, tmin, tmax)) return false;
    } else {
      if (!boxIntersect(o.bbox_min, o.bbox_max, r, tmin, tmax)) return false;
    }
    const double* m = o.inverse;
    auto xf = [m](const V3& v) {  // vec.h:746-751
      return V3(m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3],
                m[4] * v[0] + m[5] * v[1] + m[6] * v[2] + m[7],
                m[8] * v[0] + m[9] * v[1] + m[10] * v[2] + m[11]);
    };
    Ray lr;
    lr.type = r.type;
    lr.p = xf(r.p);
    lr.d = xf(r.p + r.d) - lr.p;
    double len = length(lr.d);
    lr.d = V3(lr.d[0] / len, lr.d[1] / len, lr.d[2] / len);
    bool hit = false;
    if (faceIdx >= 0) {
      hit = faceLocal(faceIdx, S.meshes[o.mesh], o, lr, i);
    } else {
      switch (o.kind) {
        case RT_OBJ_SPHERE: hit = sphereLocal(lr, i); break;
        case RT_OBJ_BOX: hit = boxLocal(lr, i); break;
        case RT_OBJ_SQUARE: hit = squareLocal(lr, i); break;
        case RT_OBJ_CYLINDER: hit = cylinderLocal(lr, i, true); break;
        case RT_OBJ_CONE: hit = coneLocal(o, lr, i); break;
        case RT_OBJ_TRIMESH: hit = trimeshLocal(o, lr, i); break;
      }
      if (hit && o.kind != RT_OBJ_TRIMESH) {
        i.mat = toMat(S.materials[o.material]);
        i.objectID = o.object_id;
      }
    }
    if (hit) {
      const double* nm = o.normi;  // TransformNode::localToGlobalCoordsNormal, scene.h:168-172
      V3 g(nm[0] * i.N[0] + nm[1] * i.N[1] + nm[2] * i.N[2],
           nm[3] * i.N[0] + nm[4] * i.N[1] + nm[5] * i.N[2],
           nm[6] * i.N[0] + nm[7] * i.N[1] + nm[8] * i.N[2]);
      normalize(g);
      i.N = g;
      i.t /= len;
    }
    return hit;
  }

  // ------------------------------------------------------------------ traversal
  struct StackElement { int node; double tMin, tMax; };

  // Scene::intersectKdTree, src/scene/scene.cpp:59-129.  meshObj < 0: scene
  // tree (leaf items are objects); else the tree of that Trimesh object.
  void intersectKdTree(const Ray& r, Isect& i, int root, int meshObj, bool& have_one,
                       double tMin, double tMax) const {
    std::vector<StackElement> st;
    st.push_back({root, tMin, tMax});
    while (!st.empty()) {
      StackElement e = st.back();
      st.pop_back();
      const int n = e.node;
      if (S.node_left[n] < 0 && S.node_right[n] < 0) {
        const int b = S.node_item_start[n], c = S.node_item_count[n];
        for (int k = 0; k < c; k++) {
          const int item = S.leaf_items[b + k];
          Isect cur;
          if (meshObj < 0 && S.objects[item].kind == RT_OBJ_TRIMESH) {
            const RtMesh& mesh = S.meshes[S.objects[item].mesh];
            double a0 = 0.0, a1 = 0.0;
            boxIntersect(mesh.root_bb_min, mesh.root_bb_max, r, a0, a1);
            intersectKdTree(r, i, mesh.kd_root, item, have_one, a0, a1);
          } else {
            bool hit = (meshObj < 0) ? geometryIntersect(S.objects[item], -1, r, cur)
                                     : geometryIntersect(S.objects[meshObj], item, r, cur);
            if (hit) {
              if (!have_one || cur.t < i.t) { i = cur; have_one = true; }
            }
          }
        }
      } else {
        const int dim = S.node_dim[n];
        const double split = S.node_split[n];
        int nearNode, farNode;
        if (r.p[dim] < split) { nearNode = S.node_left[n]; farNode = S.node_right[n]; }
        else { nearNode = S.node_right[n]; farNode = S.node_left[n]; }
        double tStar = (split - r.p[dim]) / r.d[dim];
        if (tStar > e.tMax || tStar < 0) {
          st.push_back({nearNode, tMin, tMax});
        } else if (tStar < e.tMin) {
          st.push_back({farNode, tMin, tMax});
        } else if (e.tMin <= tStar && tStar <= e.tMax) {
          st.push_back({nearNode, tMin, tStar});
          st.push_back({farNode, tStar, tMax});
        }
      }
    }
  }

  // Scene::intersect, src/scene/scene.cpp:133-159
  bool intersect(const Ray& r, Isect& i) {
    cnt.ray[r.type]++;
    bool have_one = false;
    if (S.use_kdtree && S.scene_kd_root >= 0) {
      double tMin = 0.0, tMax = 0.0;
      boxIntersect(S.scene_bb_min, S.scene_bb_max, r, tMin, tMax);
      intersectKdTree(r, i, S.scene_kd_root, -1, have_one, tMin, tMax);
    } else {
      for (int j = 0; j < S.n_objects; j++) {
        Isect cur;
        if (geometryIntersect(S.objects[j], -1, r, cur)) {
          if (!have_one || (cur.t < i.t)) { i = cur; have_one = true; }
        }
      }
    }
    if (!have_one) { i.t = 1000.0; i.objectID = 0; }
    return have_one;
  }

  // -------------------------------------------------------------------- lights
  V3 lightDirection(const RtLight& l, const V3& Pp) const {  // light.cpp:33-37,59-64,105-110
    if (l.type == RT_LIGHT_DIRECTIONAL) return -V3(l.orientation);
    V3 ret = V3(l.position) - Pp;
    normalize(ret);
    return ret;
  }
  double distanceAttenuation(const RtLight& l, const V3& Pp) const {  // light.cpp:7-11,39-52,89-98
    if (l.type != RT_LIGHT_POINT) return 1.0;
    V3 direction = V3(l.position) - Pp;
    double distanceSq = length2(direction);
    double distance = std::sqrt(distanceSq);
    return std::min(1.0, 1 / (l.constant_term + l.linear_term * distance + l.quadratic_term * distanceSq));
  }
  V3 shadowAttenuation(const RtLight& l, const V3& p) {  // light.cpp:14-26,67-87,113-149
    const V3 color(l.color);
    if (l.type == RT_LIGHT_DIRECTIONAL) {
      V3 sd = lightDirection(l, p);
      normalize(sd);
      Isect i;
      Ray sr{p, sd, RT_RAY_SHADOW};
      if (intersect(sr, i)) return V3(0, 0, 0);
      return color;
    }
    const V3 position(l.position);
    V3 sd = position - p;
    normalize(sd);
    Isect i;
    Ray sr{p, sd, RT_RAY_SHADOW};
    bool occluded = false;
    if (intersect(sr, i)) {
      double lightDistance = length2(position - p);
      V3 Q = at(sr, i.t);
      double distanceSq = length2(Q - p);
      if (distanceSq < lightDistance) occluded = true;
    }
    if (l.type == RT_LIGHT_POINT) {
      if (occluded) return prod(i.mat.kt, color);
      return color;
    }
    const V3 orientation(l.orientation);
    sd = sd * -1;
    double angle = std::acos(dot(orientation, sd)) * 180 / M_PI;
    if (angle > l.atten_angle) return V3(0.0, 0.0, 0.0);
    double fallFactor = std::pow(dot(orientation, sd), l.fall_rate);
    if (occluded) return prod((color * fallFactor), i.mat.kt);
    return (color * fallFactor);
  }

  // Material::shade, src/scene/material.cpp:16-76
  V3 shade(const Ray& r, const Isect& i) {
    const Mat& m = i.mat;
    V3 Q = at(r, i.t);
    V3 intensity = m.ke + prod(m.ka, V3(S.ambient));
    cnt.shadePoints++;
    for (int li = 0; li < S.n_lights; li++) {
      if (iszero(m.kd) && iszero(m.ks)) continue;
      const RtLight& l = S.lights[li];
      V3 L = lightDirection(l, Q);
      normalize(L);
      V3 lightIntensity = distanceAttenuation(l, Q) * shadowAttenuation(l, Q);
      if (!iszero(m.kd))
        intensity = intensity + prod(m.kd, lightIntensity) * std::max(dot(i.N, L), 0.0);
      if (!iszero(m.ks)) {
        V3 view = V3(S.eye) - Q;
        normalize(view);
        V3 cosVector = i.N * dot(L, i.N);
        V3 sinVector = cosVector - L;
        V3 refl = cosVector + sinVector;
        normalize(refl);
        intensity = intensity + prod(m.ks, lightIntensity) * std::pow(std::max(dot(refl, view), 0.0), shininessOf(m));
      }
    }
    return intensity;
  }

  // TextureMap::getPixelAt, src/scene/material.cpp:130-148
  V3 cubePixel(int face, int x, int y) const {
    const unsigned char* data = S.cube_rgb[face];
    if (x >= S.cube_w) x = S.cube_w - 1;
    if (y >= S.cube_h) y = S.cube_h - 1;
    int pos = (y * S.cube_w + x) * 3;
    return V3(double(data[pos]) / 255.0, double(data[pos + 1]) / 255.0, double(data[pos + 2]) / 255.0);
  }

  // CubeMap::getColor, src/scene/cubeMap.cpp:6-208
  V3 cubeColor(const Ray& r) const {
    const V3& dir = r.d;
    if (std::isnan(dir[0]) || std::isnan(dir[1]) || std::isnan(dir[2])) {
      // x86 converts the NaN texel coordinate to INT_MIN, every tap is skipped
      // and 0/0 is returned (cubeMap.cpp:103-104,117-120,206)
      double q = std::nan("");
      return V3(q, q, q);
    }
    int axis, front, left = 0, right = 0, top = 0, bottom = 0;
    double u = 0, v = 0;
    if (std::fabs(dir[0]) > std::fabs(dir[1])) {
      if (std::fabs(dir[0]) > std::fabs(dir[2])) {
        axis = 0; bottom = 3; top = 2;
        if (dir[0] > 0.0) { front = 0; left = 4; right = 5; } else { front = 1; left = 5; right = 4; }
      } else {
        axis = 2; bottom = 3; top = 2;
        if (dir[2] > 0.0) { front = 5; left = 0; right = 1; } else { front = 4; left = 1; right = 0; }
      }
    } else {
      if (std::fabs(dir[1]) > std::fabs(dir[2])) {
        axis = 1; left = 1; right = 0;
        if (dir[1] > 0.0) { front = 2; bottom = 4; top = 5; } else { front = 3; bottom = 5; top = 4; }
      } else {
        axis = 2; bottom = 3; top = 2;
        if (dir[2] > 0.0) { front = 5; left = 0; right = 1; } else { front = 4; left = 1; right = 0; }
      }
    }
    if (axis == 0) {
      u = dir[2] / dir[0];
      if (dir[0] > 0.0) v = dir[1] / dir[0]; else v = -dir[1] / dir[0];
    } else if (axis == 1) {
      if (dir[1] > 0.0) u = dir[0] / dir[1]; else u = -dir[0] / dir[1];
      v = dir[2] / dir[1];
    } else {
      u = -dir[0] / dir[2];
      if (dir[2] > 0.0) v = dir[1] / dir[2]; else v = -dir[1] / dir[2];
    }
    u = (u + 1.0) / 2.0;
    v = (v + 1.0) / 2.0;
    int filterwidth = P.filter_width;
    int fw = (filterwidth + 1) / 2 - 1;
    int rm = filterwidth - fw;
    int width = S.cube_w, height = S.cube_h;
    double delu = 1.0 / width, delv = 1.0 / height;
    int x = (int)std::floor(0.5 + u * (width - 1));
    int y = (int)std::floor(0.5 + v * (height - 1));
    double startu = delu * (x - fw), startv = delv * (y - fw);
    double nextu = startu, nextv = startv;
    double xslope = 1.0 / (((double)filterwidth + 1) / 2.0 * delu);
    double yslope = 1.0 / (((double)filterwidth + 1) / 2.0 * delv);
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
            if (dir[0] > 0) { yindex = height - 1 - xindex; xindex = width + jj; }
            else { yindex = xindex; xindex = -1 - jj; }
          } else if (axis == 1) {
            if (dir[1] > 0) yindex = height + jj;
            else { yindex = -1 - jj; xindex = width - 1 - xindex; }
          } else {
            if (dir[2] > 0) { yindex = -1 - jj; xindex = width - 1 - xindex; }
            else yindex = height + jj;
          }
        } else if (yindex >= height) {
          theMap = top;
          if (axis == 0) {
            if (dir[0] > 0) { yindex = xindex; xindex = width - jj; }
            else { yindex = height - 1 - xindex; xindex = jj - 1; }
          } else if (axis == 1) {
            if (dir[1] > 0) { yindex = height - jj; xindex = width - 1 - xindex; }
            else yindex = jj - 1;
          } else {
            if (dir[2] > 0) { yindex = height - jj; xindex = width - 1 - xindex; }
            else yindex = jj - 1;
          }
        } else if (xindex < 0) {
          theMap = left;
          if (axis == 0 || axis == 2) xindex = width + ii;
          else {
            if (dir[1] > 0) { xindex = width - 1 - yindex; yindex = height + ii; }
            else { xindex = yindex; yindex = -1 - ii; }
          }
        } else if (xindex >= width) {
          theMap = right;
          if (axis == 0 || axis == 2) xindex = ii - 1;
          else {
            if (dir[1] > 0) { xindex = yindex; yindex = height - ii; }
            else { xindex = width - 1 - yindex; yindex = ii - 1; }
          }
        }
        double du = u - nextu, dv = v - nextv;
        double weight = (1.0 - std::fabs(du * xslope)) * (1.0 - std::fabs(dv * yslope));
        nextu += delu;
        correct += weight;
        thePixel = thePixel + cubePixel(theMap, xindex, yindex) * weight;
      }
      nextu = startu;
      nextv += delv;
    }
    thePixel = V3(thePixel[0] / correct, thePixel[1] / correct, thePixel[2] / correct);
    return thePixel;
  }

  bool haveCubeMap() const { return P.use_cubemap && S.cube_w > 0 && S.cube_rgb[0] != nullptr; }

  // RayTracer::traceRay, src/RayTracer.cpp:132-226
  V3 traceRay(const Ray& r, int depth, int* primaryHit) {
    Isect i;
    if (intersect(r, i)) {
      if (primaryHit) *primaryHit = i.objectID;
      const Mat& material = i.mat;
      V3 intensity = shade(r, i);
      if (depth == 0) return intensity;
      V3 Qpoint = at(r, i.t);
      V3 minusD = -1 * r.d;
      V3 cosVector = i.N * dot(minusD, i.N);
      V3 sinVector = cosVector + r.d;
      if (!iszero(material.kr)) {
        V3 reflectedDirection = cosVector + sinVector;
        normalize(reflectedDirection);
        Ray reflected{Qpoint, reflectedDirection, RT_RAY_REFLECTION};
        intensity = intensity + prod(material.kr, traceRay(reflected, depth - 1, nullptr));
      }
      if (!iszero(material.kt)) {
        double cosineAngle = std::acos(dot(i.N, r.d)) * 180 / M_PI;
        double n_i, n_r;
        int iDirection;
        if (cosineAngle > 90) { n_i = 1; n_r = intensityValue(material.index); iDirection = 1; }
        else { n_i = intensityValue(material.index); n_r = 1; iDirection = -1; }
        V3 sinT = (n_i / n_r) * sinVector;
        V3 cosT = (-1 * i.N) * std::sqrt(1 - dot(sinT, sinT));
        // criticalAngle is the constant 360 (RayTracer.cpp:173,195): always true
        // for real angles, false for a NaN angle
        if (cosineAngle < 360) {
          V3 refractedDirection = cosT + iDirection * sinT;
          normalize(refractedDirection);
          Ray refracted{Qpoint, iDirection * refractedDirection, RT_RAY_REFRACTION};
          intensity = intensity + prod(material.kt, traceRay(refracted, depth - 1, nullptr));
        }
      }
      return intensity;
    }
    if (primaryHit) *primaryHit = 0;
    if (haveCubeMap()) return cubeColor(r);
    return V3(0.0, 0.0, 0.0);
  }

  // Camera::rayThrough + RayTracer::trace, camera.cpp:20-31, RayTracer.cpp:34-43
  V3 trace(double x, double y, int* primaryHit) {
    x -= 0.5;
    y -= 0.5;
    V3 dir = V3(S.look) + x * V3(S.u) + y * V3(S.v);
    normalize(dir);
    Ray r{V3(S.eye), dir, RT_RAY_VISIBILITY};
    V3 ret = traceRay(r, P.depth, primaryHit);
    for (int k = 0; k < 3; k++) {  // Vec3::clamp, vec.h:262-268
      if (ret[k] < 0) ret[k] = 0.0;
      if (ret[k] > 1) ret[k] = 1.0;
    }
    return ret;
  }

  static unsigned char toByte(double c) {
    // (int)(255.0*c) stored into an unsigned char (RayTracer.cpp:58-60); a NaN
    // converts to INT_MIN on x86-64 and therefore to byte 0
    double v = 255.0 * c;
    if (std::isnan(v)) return 0;
    return (unsigned char)(int)v;
  }

  // RayTracer::tracePixel, src/RayTracer.cpp:45-62
  void tracePixel(int i, int j, unsigned char* px, int* hitId) {
    double x = double(i) / double(P.width);
    double y = double(j) / double(P.height);
    V3 col = trace(x, y, hitId);
    px[0] = toByte(col[0]); px[1] = toByte(col[1]); px[2] = toByte(col[2]);
  }

  // RayTracer::tracePixelAntiAlias, src/RayTracer.cpp:64-103 with the jitter
  // values supplied by the caller in draw order (x then y; column outer, row inner)
  void tracePixelAntiAlias(int i, int j, unsigned char* px, int* hitId) {
    V3 col(0, 0, 0);
    double x = double(i) / double(P.width);
    double y = double(j) / double(P.height);
    const int n = P.aa_samples;
    double deltaX = (1.0 / double(P.width)) / n;
    double deltaY = (1.0 / double(P.height)) / n;
    int k = 0;
    for (int sc = 0; sc < n; sc++)
      for (int sr = 0; sr < n; sr++) {
        double jx = P.jitter ? P.jitter[2 * k] : 0.0, jy = P.jitter ? P.jitter[2 * k + 1] : 0.0;
        double xTemp = x + sc * deltaX + jx * deltaX;
        double yTemp = y + sr * deltaY + jy * deltaY;
        V3 tCol = trace(xTemp, yTemp, (k == 0) ? hitId : nullptr);
        col = V3(col[0] + tCol[0], col[1] + tCol[1], col[2] + tCol[2]);
        k++;
      }
    double nn = double(n * n);
    col = V3(col[0] / nn, col[1] / nn, col[2] / nn);
    px[0] = toByte(col[0]); px[1] = toByte(col[1]); px[2] = toByte(col[2]);
  }
};

// GraphicalUI::applyFilter, src/ui/GraphicalUI.cpp:437-486.  The reference's
// bounds test lets column == width through (that reads the next row's first
// pixel) and row == height (past the buffer); the latter is read as 0 here.
void edgeMask(const unsigned char* src, int w, int h, int cutOff, unsigned char* dst) {
  std::vector<unsigned char> gray((size_t)w * h);
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) {
      const unsigned char* p = src + 3 * ((size_t)r * w + c);
      double g = 0.299 * p[0] + 0.587 * p[1] + 0.114 * p[2];
      gray[(size_t)r * w + c] = (unsigned char)g;
    }
  const int kernel[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) {
      double sum = 0;
      for (int fr = 0; fr < 3; fr++)
        for (int fc = 0; fc < 3; fc++) {
          int tr = r - 1 + fr, tc = c - 1 + fc;
          if (tr < 0 || tc < 0 || tr > h || tc > w) continue;
          size_t idx = (size_t)tr * w + tc;
          double g = (idx < (size_t)w * h) ? gray[idx] : 0;
          sum = sum + kernel[3 * fr + fc] * g;
        }
      dst[(size_t)r * w + c] = (sum > cutOff) ? 255 : 0;
    }
}

void addCounters(Counters& a, const Counters& b) {
  for (int k = 0; k < 5; k++) a.ray[k] += b.ray[k];
  a.shadePoints += b.shadePoints;
}

void fillStats(RtStats* st, const Counters& c) {
  if (!st) return;
  std::memset(st, 0, sizeof *st);
  st->rays_primary = c.ray[RT_RAY_VISIBILITY];
  st->rays_reflect = c.ray[RT_RAY_REFLECTION];
  st->rays_refract = c.ray[RT_RAY_REFRACTION];
  st->rays_shadow = c.ray[RT_RAY_SHADOW];
  st->shade_points = c.shadePoints;
}

// Runs fn(thread, Tracer&) on nThreads threads and sums their counters.
template <class F>
Counters runThreads(const RtSceneDesc* d, const RtRenderParams* p, int nThreads, F fn) {
  if (nThreads < 1) nThreads = 1;
  std::vector<Counters> cs(nThreads);
  std::vector<std::thread> th;
  auto body = [&](int t) {
    Tracer tr(d, p);
    fn(t, tr);
    cs[t] = tr.cnt;
  };
  for (int t = 1; t < nThreads; t++) th.emplace_back(body, t);
  body(0);
  for (auto& x : th) x.join();
  Counters total;
  for (auto& c : cs) addCounters(total, c);
  return total;
}

}  // namespace

extern "C" {

// Full frame.  Threads take vertical column strips of ceil(w/T) columns like
// CommandLineUI::run / renderThread (src/ui/CommandLineUI.cpp:54-66,83-100),
// clipped at the right border.  aa_mode RT_AA_ADAPTIVE follows
// GraphicalUI::cb_render + doAntiAliasing (src/ui/GraphicalUI.cpp:285-435).
int orc_render(const RtSceneDesc* d, const RtRenderParams* p, unsigned char* rgb, int* hitId,
               RtStats* stats, int nThreads) {
  if (!d || !p || !rgb || p->width <= 0 || p->height <= 0) return RT_ERR_INVALID;
  const int w = p->width, h = p->height;
  if (nThreads < 1) nThreads = 1;
  const int cols = (w + nThreads - 1) / nThreads;
  std::memset(rgb, 0, (size_t)w * h * 3);
  Counters total;
  std::vector<unsigned char> mask;
  auto pass = [&](bool aa, const unsigned char* onlyMask) {
    Counters c = runThreads(d, p, nThreads, [&](int t, Tracer& tr) {
      int x0 = t * cols, x1 = std::min(w, x0 + cols);
      for (int y = 0; y < h; y++)
        for (int x = x0; x < x1; x++) {
          size_t idx = (size_t)x + (size_t)y * w;
          if (onlyMask && onlyMask[idx] != 255) continue;
          int* hp = (hitId && !onlyMask) ? hitId + idx : nullptr;
          if (aa) tr.tracePixelAntiAlias(x, y, rgb + 3 * idx, hp);
          else tr.tracePixel(x, y, rgb + 3 * idx, hp);
        }
    });
    addCounters(total, c);
  };
  if (p->aa_mode == RT_AA_FULL) {
    pass(true, nullptr);
  } else {
    pass(false, nullptr);
    if (p->aa_mode == RT_AA_ADAPTIVE) {
      mask.resize((size_t)w * h);
      edgeMask(rgb, w, h, p->aa_threshold, mask.data());
      pass(true, mask.data());
    }
  }
  fillStats(stats, total);
  return RT_OK;
}

// A list of pixels (x,y pairs) of the frame params describe; rgb gets 3 bytes
// per listed pixel.  Used to time a bounded sample of a large frame.
int orc_render_pixels(const RtSceneDesc* d, const RtRenderParams* p, int n, const int* xy,
                      unsigned char* rgb, RtStats* stats, int nThreads) {
  if (!d || !p || n < 0) return RT_ERR_INVALID;
  if (nThreads < 1) nThreads = 1;
  const bool aa = p->aa_mode == RT_AA_FULL;
  Counters total = runThreads(d, p, nThreads, [&](int t, Tracer& tr) {
    for (int k = t; k < n; k += nThreads) {
      if (aa) tr.tracePixelAntiAlias(xy[2 * k], xy[2 * k + 1], rgb + 3 * (size_t)k, nullptr);
      else tr.tracePixel(xy[2 * k], xy[2 * k + 1], rgb + 3 * (size_t)k, nullptr);
    }
  });
  fillStats(stats, total);
  return RT_OK;
}

// Geometry::objectID of what the FIRST sample of each listed pixel sees (0: nothing):
// the value tracePixel / tracePixelAntiAlias's first trace() call finds in
// Scene::intersect (src/RayTracer.cpp:45-62,84-93; src/scene/scene.cpp:133-159).
// One closest-hit query per pixel, no shading: cheap enough to hold a frame's
// primary-hit map against the oracle on 10^5 pixels.
int orc_primary_ids(const RtSceneDesc* d, const RtRenderParams* p, int n, const int* xy, int* ids,
                    int nThreads) {
  if (!d || !p || n < 0 || (n > 0 && (!xy || !ids))) return RT_ERR_INVALID;
  if (nThreads < 1) nThreads = 1;
  const bool aa = p->aa_mode == RT_AA_FULL;
  runThreads(d, p, nThreads, [&](int t, Tracer& tr) {
    for (int k = t; k < n; k += nThreads) {
      double x = double(xy[2 * k]) / double(p->width);
      double y = double(xy[2 * k + 1]) / double(p->height);
      if (aa) {
        // sample (column 0, row 0) of tracePixelAntiAlias
        const int ns = p->aa_samples;
        double deltaX = (1.0 / double(p->width)) / ns;
        double deltaY = (1.0 / double(p->height)) / ns;
        double jx = p->jitter ? p->jitter[0] : 0.0, jy = p->jitter ? p->jitter[1] : 0.0;
        x = x + 0 * deltaX + jx * deltaX;
        y = y + 0 * deltaY + jy * deltaY;
      }
      x -= 0.5;
      y -= 0.5;
      V3 dir = V3(tr.S.look) + x * V3(tr.S.u) + y * V3(tr.S.v);
      normalize(dir);
      Ray r{V3(tr.S.eye), dir, RT_RAY_VISIBILITY};
      Isect i;
      ids[k] = tr.intersect(r, i) ? i.objectID : 0;
    }
  });
  return RT_OK;
}

int orc_intersect_batch(const RtSceneDesc* d, int n, const double* raysPd, const int* types,
                        RtHit* hits) {
  if (!d || n < 0) return RT_ERR_INVALID;
  RtRenderParams p;
  std::memset(&p, 0, sizeof p);
  Tracer tr(d, &p);
  for (int k = 0; k < n; k++) {
    Ray r{V3(raysPd + 6 * k), V3(raysPd + 6 * k + 3), types ? types[k] : RT_RAY_VISIBILITY};
    Isect i;
    bool hit = tr.intersect(r, i);
    RtHit& o = hits[k];
    std::memset(&o, 0, sizeof o);
    o.t = i.t;
    if (hit) {
      for (int a = 0; a < 3; a++) { o.N[a] = i.N[a]; o.bary[a] = i.bary[a]; }
      o.object_id = i.objectID;
    }
  }
  return RT_OK;
}

int orc_edge_mask(const unsigned char* rgb, int w, int h, int cutoff, unsigned char* mask) {
  if (!rgb || !mask || w <= 0 || h <= 0) return RT_ERR_INVALID;
  edgeMask(rgb, w, h, cutoff, mask);
  return RT_OK;
}

// srand(seed) then 2*n*n draws of (double)rand()/(RAND_MAX+1) as
// RayTracer.cpp:70,91-92 evaluates it with g++ on x86-64: RAND_MAX+1 wraps to
// INT_MIN, so the divisor is -2147483648.0 and every value lies in (-1, 0].
int orc_jitter_table(unsigned seed, int n, double* out) {
  if (n <= 0 || !out) return RT_ERR_INVALID;
  srand(seed);
  for (int k = 0; k < 2 * n * n; k++) out[k] = (double)rand() / -2147483648.0;
  return RT_OK;
}

}  // extern "C"
