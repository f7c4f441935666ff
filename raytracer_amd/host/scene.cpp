// Host scene model: bounding boxes, transforms, camera, KD-tree builder and the
// flattener that produces the C-ABI scene description.
#include "scene.h"

#include <algorithm>
#include <cstdlib>
#include <limits>
#include <stdexcept>

namespace rta {

// ---------------------------------------------------------------- BoundingBox
// src/scene/bbox.h:118-126
void BoundingBox::merge(const BoundingBox& o) {
  if (o.empty) return;
  for (int a = 0; a < 3; a++) {
    if (empty || o.bmin[a] < bmin[a]) bmin[a] = o.bmin[a];
    if (empty || o.bmax[a] > bmax[a]) bmax[a] = o.bmax[a];
  }
  empty = false;
}

// src/scene/bbox.h:100-107
double BoundingBox::area() const {
  if (empty) return 0.0;
  double dx = bmax[0] - bmin[0], dy = bmax[1] - bmin[1], dz = bmax[2] - bmin[2];
  return 2.0 * (dx * dy + dy * dz + dz * dx);
}

// -------------------------------------------------------------- TransformNode
// src/scene/scene.h:180-186
TransformNode::TransformNode() {
  inverse = rta::inverse(xform);
  normi = rta::inverse(xform.upper33()).transpose();
}

TransformNode::TransformNode(const TransformNode& parent, const Mat4& local) {
  xform = mul(parent.xform, local);
  inverse = rta::inverse(xform);
  normi = rta::inverse(xform.upper33()).transpose();
}

TransformNode* TransformNode::createChild(const Mat4& local) {
  children.emplace_back(new TransformNode(*this, local));
  return children.back().get();
}

// --------------------------------------------------------------------- Camera
Camera::Camera() {  // src/scene/camera.cpp:9-18
  eye = Vec3(0, 0, 0);
  u = Vec3(1, 0, 0);
  v = Vec3(0, 1, 0);
  look = Vec3(0, 0, -1);
}

// src/scene/camera.cpp:39-60
void Camera::setLook(double r, double i, double j, double k) {
  m[0][0] = 1.0 - 2.0 * (i * i + j * j);
  m[0][1] = 2.0 * (r * i - j * k);
  m[0][2] = 2.0 * (j * r + i * k);
  m[1][0] = 2.0 * (r * i + j * k);
  m[1][1] = 1.0 - 2.0 * (j * j + r * r);
  m[1][2] = 2.0 * (i * j - r * k);
  m[2][0] = 2.0 * (j * r - i * k);
  m[2][1] = 2.0 * (i * j + r * k);
  m[2][2] = 1.0 - 2.0 * (i * i + r * r);
  update();
}

// src/scene/camera.cpp:62-72: rows x,y,z then transposed, so they become columns
void Camera::setLook(const Vec3& viewDir, const Vec3& upDir) {
  Vec3 z = neg(viewDir);
  Vec3 x = cross(upDir, z);
  Mat3 rows;
  for (int c = 0; c < 3; c++) {
    rows[0][c] = x[c];
    rows[1][c] = upDir[c];
    rows[2][c] = z[c];
  }
  m = rows.transpose();
  update();
}

// src/scene/camera.cpp:74-81 with its truncated PI (camera.cpp:4)
void Camera::setFOV(double fov) {
  const double PI = 3.14159265359;
  fov /= (180.0 / PI);
  normalizedHeight = 2 * std::tan(fov / 2);
  update();
}

void Camera::setAspectRatio(double ar) {
  aspectRatio = ar;
  update();
}

// src/scene/camera.cpp:91-97: (m*e1)*nh*ar evaluated left to right
void Camera::update() {
  u = scale(scale(mul(m, Vec3(1, 0, 0)), normalizedHeight), aspectRatio);
  v = scale(mul(m, Vec3(0, 1, 0)), normalizedHeight);
  look = mul(m, Vec3(0, 0, -1));
}

// ------------------------------------------------------------------- Geometry
BoundingBox Geometry::localBounds() const {
  switch (kind) {
    case RT_OBJ_SPHERE:    // Sphere.h:18-24
      return BoundingBox(Vec3(-1.0f, -1.0f, -1.0f), Vec3(1.0f, 1.0f, 1.0f));
    case RT_OBJ_BOX:       // Box.h:16-22
      return BoundingBox(Vec3(-0.5, -0.5, -0.5), Vec3(0.5, 0.5, 0.5));
    case RT_OBJ_SQUARE:    // Square.h:18-24
      return BoundingBox(Vec3(-0.5f, -0.5f, -RAY_EPSILON), Vec3(0.5f, 0.5f, RAY_EPSILON));
    case RT_OBJ_CYLINDER:  // Cylinder.h:18-24
      return BoundingBox(Vec3(-1.0f, -1.0f, 0.0f), Vec3(1.0f, 1.0f, 1.0f));
    case RT_OBJ_CONE: {    // Cone.h:42-50
      double big = (bRadius > tRadius) ? bRadius : tRadius;
      return BoundingBox(Vec3(-big, -big, (height < 0.0f) ? height : 0.0f),
                         Vec3(big, big, (height < 0.0f) ? 0.0f : height));
    }
    case RT_OBJ_TRIMESH: { // trimesh.h:60-74
      BoundingBox b;
      if (mesh->vertices.empty()) return b;
      Vec3 lo = mesh->vertices[0], hi = mesh->vertices[0];
      for (const Vec3& p : mesh->vertices) {
        hi = vmax(hi, p);
        lo = vmin(lo, p);
      }
      return BoundingBox(lo, hi);
    }
  }
  return BoundingBox();
}

// src/scene/scene.h:214-252: transform the 8 corners, take componentwise extrema
void Geometry::computeBoundingBox() {
  BoundingBox lb = localBounds();
  if (lb.empty) {  // bogus local box: the reference transforms default (zero) corners
    lb = BoundingBox(Vec3(), Vec3());
  }
  const Vec3 lo = lb.bmin, hi = lb.bmax;
  Vec3 nmin, nmax;
  for (int c = 0; c < 8; c++) {
    Vec4 p((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2], 1);
    Vec4 w = mul(transform->xform, p);
    Vec3 w3(w[0], w[1], w[2]);
    if (c == 0) {
      nmin = nmax = w3;
    } else {
      nmax = vmax(nmax, w3);
      nmin = vmin(nmin, w3);
    }
  }
  bounds = BoundingBox(nmin, nmax);
}

// src/SceneObjects/trimesh.cpp:210-238
void Trimesh::generateNormals() {
  size_t cnt = vertices.size();
  normals.resize(cnt);  // explicit normals, if any, stay and are added to
  std::vector<int> numFaces(cnt, 0);
  for (const TrimeshFace& f : faces)
    for (int i = 0; i < 3; i++) {
      normals[f.ids[i]] = add(normals[f.ids[i]], f.normal);
      ++numFaces[f.ids[i]];
    }
  for (size_t i = 0; i < cnt; i++)
    if (numFaces[i]) {
      double d = numFaces[i];
      normals[i] = Vec3(normals[i][0] / d, normals[i][1] / d, normals[i][2] / d);
    }
  vertNorms = true;
}

// ---------------------------------------------------------------------- Scene
void Scene::add(std::unique_ptr<Geometry> g) {
  g->computeBoundingBox();
  sceneBounds.merge(g->bounds);
  objects.push_back(std::move(g));
}

namespace {
// yz/xz/xyPlaneCompareFunction (src/scene/scene.cpp:205-299): order by the
// min (flag 0) or max (flag 1) coordinate; on equal coordinates a min event
// sorts before a max event; otherwise equal.
struct EventLess {
  const double* key;  // key[2*item + flag]
  typedef std::pair<int, int> Event;
  bool operator()(const Event& a, const Event& b) const {
    double ka = key[2 * a.first + a.second], kb = key[2 * b.first + b.second];
    if (ka == kb) {
      if (a.second < b.second) return true;
      if (a.second > b.second) return false;
    }
    return ka < kb;
  }
};
}  // namespace

// One call of Scene::buildMainKdTree (src/scene/scene.cpp:389-567).  `planes`
// holds, per axis, the (item, min/max flag) events of this node in the order
// the parent appended them; they are sorted here with std::sort, exactly as the
// reference does, because the sweep below evaluates the cost after every event
// and the position of tied events decides which cost values are seen.
void Scene::buildNode(KdTree& tree, int nodeIdx, int depth, int leafSize,
                      std::vector<Event> (&planes)[3], const BuildBoxes& bx, bool sceneLevel) {
  for (int d = 0; d < 3; d++) {
    std::sort(planes[d].begin(), planes[d].end(), EventLess{bx.key[d].data()});
    kdBuildSorts++;
  }
  if (depth == 0) return;
  if (sceneLevel) buildMeshesIn(tree.nodes[nodeIdx]);

  double finalCost = std::numeric_limits<double>::max();
  const double tTime = 15, iTime = 80 * tTime;
  const double totalArea = tree.nodes[nodeIdx].bb.area();
  int finalDim = 0;
  double finalSplit = 0.0;
  for (int d = 0; d < 3; d++) {
    double rightArea = 0, rightObjects = 0, leftArea = 0, leftObjects = 0;
    for (const Event& e : planes[d])
      if (e.second == 0) {
        rightArea = rightArea + bx.area[e.first];
        rightObjects++;
      }
    for (const Event& e : planes[d]) {
      if (e.second == 0) {
        leftObjects++;
        leftArea = leftArea + bx.area[e.first];
      } else {
        rightObjects--;
        rightArea = rightArea - bx.area[e.first];
      }
      double cost = tTime + (leftArea / totalArea) * leftObjects * iTime +
                    (rightArea / totalArea) * rightObjects * iTime;
      if (cost < finalCost) {
        finalCost = cost;
        finalDim = d;
        finalSplit = bx.key[d][2 * e.first + e.second];
      }
    }
  }
  for (int d = 0; d < 3; d++) std::vector<Event>().swap(planes[d]);

  int li = (int)tree.nodes.size();
  tree.nodes.emplace_back();
  int ri = (int)tree.nodes.size();
  tree.nodes.emplace_back();
  tree.nodes[nodeIdx].split = finalSplit;
  tree.nodes[nodeIdx].dim = finalDim;
  tree.nodes[nodeIdx].left = li;
  tree.nodes[nodeIdx].right = ri;

  std::vector<Event> lp[3], rp[3];
  {
    const std::vector<int>& its = tree.nodes[nodeIdx].items;
    KdNode& L = tree.nodes[li];
    KdNode& R = tree.nodes[ri];
    const std::vector<double>& key = bx.key[finalDim];
    for (int it : its) {
      const double lo = key[2 * it], hi = key[2 * it + 1];
      bool toL, toR;
      if (lo < finalSplit && hi < finalSplit) {
        toL = true; toR = false;
      } else if (lo > finalSplit) {
        toL = false; toR = true;
      } else {
        toL = toR = true;
      }
      if (toL) {
        L.items.push_back(it);
        L.bb.merge(bx.box[it]);
        for (int d = 0; d < 3; d++) { lp[d].push_back({it, 0}); lp[d].push_back({it, 1}); }
      }
      if (toR) {
        R.items.push_back(it);
        R.bb.merge(bx.box[it]);
        for (int d = 0; d < 3; d++) { rp[d].push_back({it, 0}); rp[d].push_back({it, 1}); }
      }
    }
  }
  if ((int)tree.nodes[li].items.size() > leafSize)
    buildNode(tree, li, depth - 1, leafSize, lp, bx, sceneLevel);
  else if (sceneLevel)
    buildMeshesIn(tree.nodes[li]);
  for (int d = 0; d < 3; d++) std::vector<Event>().swap(lp[d]);
  if ((int)tree.nodes[ri].items.size() > leafSize)
    buildNode(tree, ri, depth - 1, leafSize, rp, bx, sceneLevel);
  else if (sceneLevel)
    buildMeshesIn(tree.nodes[ri]);
}

void Scene::BuildBoxes::add(const BoundingBox& b) {
  box.push_back(b);
  area.push_back(b.area());
  for (int d = 0; d < 3; d++) {
    key[d].push_back(b.bmin[d]);
    key[d].push_back(b.bmax[d]);
  }
}

// The reference builds a mesh's own tree the first time a scene-tree node that
// contains the mesh is processed (scene.cpp:400-410,537-565).
void Scene::buildMeshesIn(const KdNode& node) {
  std::vector<int> its = node.items;
  for (int it : its) {
    Geometry& g = *objects[it];
    if (g.kind == RT_OBJ_TRIMESH && !g.mesh->kd.built()) buildTrimeshKdTree(*g.mesh, g.bounds);
  }
}

// src/scene/scene.cpp:569-602
void Scene::buildTrimeshKdTree(Trimesh& mesh, const BoundingBox& worldBounds) {
  KdTree& t = mesh.kd;
  t.nodes.emplace_back();
  t.nodes[0].bb = worldBounds;
  BuildBoxes bx;
  std::vector<Event> planes[3];
  for (int i = 0; i < (int)mesh.faces.size(); i++) {
    t.nodes[0].items.push_back(i);
    bx.add(mesh.faces[i].bounds);
    for (int d = 0; d < 3; d++) { planes[d].push_back({i, 0}); planes[d].push_back({i, 1}); }
  }
  // The reference sorts the three lists once here and buildMainKdTree sorts the
  // sorted lists again; std::sort is not stable, so both passes are kept.
  for (int d = 0; d < 3; d++) {
    std::sort(planes[d].begin(), planes[d].end(), EventLess{bx.key[d].data()});
    kdBuildSorts++;
  }
  buildNode(t, 0, kdTreeDepth, kdTreeLeafSize, planes, bx, false);
}

// src/scene/scene.cpp:328-387
void Scene::buildKdTree(int depth, int leafSize) {
  kdTreeDepth = depth;
  kdTreeLeafSize = leafSize;
  kd.nodes.clear();
  kd.nodes.emplace_back();
  kd.nodes[0].bb = sceneBounds;
  BuildBoxes bx;
  std::vector<Event> planes[3];
  for (int i = 0; i < (int)objects.size(); i++) {
    kd.nodes[0].items.push_back(i);
    bx.add(objects[i]->bounds);
    for (int d = 0; d < 3; d++) { planes[d].push_back({i, 0}); planes[d].push_back({i, 1}); }
  }
  buildNode(kd, 0, depth, leafSize, planes, bx, true);
}

// src/RayTracer.cpp:306-317
void applyOptions(Scene& scene, const SceneOptions& opt, const std::string& fileName) {
  if (opt.kdTree && fileName.find("turtle.ray") == std::string::npos && !scene.objects.empty()) {
    scene.buildKdTree(opt.kdMaxDepth, opt.kdLeafSize);
    scene.useKdTree = opt.kdTree;
  }
  scene.backFaceCulling = opt.backFaceCulling;
  scene.smoothShading = opt.smoothShading;
}

// ------------------------------------------------------------------- Material
RtMaterial Material::flatten() const {
  RtMaterial m;
  std::memset(&m, 0, sizeof m);
  auto put = [](double* dst, const Vec3& v) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; };
  put(m.ke, ke.eval()); put(m.ka, ka.eval()); put(m.ks, ks.eval());
  put(m.kd, kd.eval()); put(m.kr, kr.eval()); put(m.kt, kt.eval());
  put(m.shininess, shininess.eval());
  put(m.index, index.eval());
  m.shininess_mapped = shininess.mapped ? 1 : 0;
  return m;
}

// ------------------------------------------------------------------ flattener
void FlatScene::fixPointers() {
  desc.lights = lights.data();           desc.n_lights = (int)lights.size();
  desc.materials = materials.data();     desc.n_materials = (int)materials.size();
  desc.objects = objects.data();         desc.n_objects = (int)objects.size();
  desc.meshes = meshes.data();           desc.n_meshes = (int)meshes.size();
  desc.vertices = vertices.data();       desc.n_vertices = (int)(vertices.size() / 3);
  desc.vnormals = hasVertexNormals ? vnormals.data() : nullptr;
  desc.vmaterials = hasVertexMaterials ? vmaterials.data() : nullptr;
  desc.faces = faces.data();             desc.n_faces = (int)faces.size();
  desc.n_nodes = (int)nodeSplit.size();
  desc.node_split = nodeSplit.data();
  desc.node_dim = nodeDim.data();
  desc.node_left = nodeLeft.data();
  desc.node_right = nodeRight.data();
  desc.node_item_start = nodeItemStart.data();
  desc.node_item_count = nodeItemCount.data();
  desc.leaf_items = leafItems.data();    desc.n_leaf_items = (int)leafItems.size();
  for (int f = 0; f < 6; f++)
    desc.cube_rgb[f] = (cubeFaces.size() == 6 && !cubeFaces[f].empty()) ? cubeFaces[f].data() : nullptr;
}

namespace {
// Appends a tree in depth-first preorder (left subtree right after its parent)
// and returns the index of its root in the shared node arrays.
int appendTree(FlatScene& fs, const KdTree& t, int itemOffset) {
  if (!t.built()) return -1;
  int base = (int)fs.nodeSplit.size();
  std::vector<int> order, newIndex(t.nodes.size(), -1), stack{0};
  while (!stack.empty()) {
    int n = stack.back();
    stack.pop_back();
    newIndex[n] = base + (int)order.size();
    order.push_back(n);
    if (t.nodes[n].right >= 0) stack.push_back(t.nodes[n].right);
    if (t.nodes[n].left >= 0) stack.push_back(t.nodes[n].left);
  }
  for (int n : order) {
    const KdNode& k = t.nodes[n];
    bool leaf = k.left < 0 && k.right < 0;
    fs.nodeSplit.push_back(k.split);
    fs.nodeDim.push_back(k.dim);
    fs.nodeLeft.push_back(k.left >= 0 ? newIndex[k.left] : -1);
    fs.nodeRight.push_back(k.right >= 0 ? newIndex[k.right] : -1);
    fs.nodeItemStart.push_back((int)fs.leafItems.size());
    fs.nodeItemCount.push_back(leaf ? (int)k.items.size() : 0);
    if (leaf)
      for (int it : k.items) fs.leafItems.push_back(it + itemOffset);
  }
  return base;
}
}  // namespace

std::unique_ptr<FlatScene> flattenScene(const Scene& scene) {
  std::unique_ptr<FlatScene> fsp(new FlatScene);
  FlatScene& fs = *fsp;
  RtSceneDesc& d = fs.desc;
  std::memset(&d, 0, sizeof d);
  d.abi_version = RT_AMD_ABI_VERSION;
  for (int i = 0; i < 3; i++) {
    d.eye[i] = scene.camera.eye[i];
    d.look[i] = scene.camera.look[i];
    d.u[i] = scene.camera.u[i];
    d.v[i] = scene.camera.v[i];
    d.ambient[i] = scene.ambientIntensity[i];
    d.scene_bb_min[i] = scene.sceneBounds.bmin[i];
    d.scene_bb_max[i] = scene.sceneBounds.bmax[i];
  }
  d.use_kdtree = (scene.useKdTree && scene.kd.built()) ? 1 : 0;
  d.backface_cull = scene.backFaceCulling;
  d.smooth_shading = scene.smoothShading;

  for (const Light& l : scene.lights) {
    RtLight r;
    std::memset(&r, 0, sizeof r);
    r.type = l.type;
    for (int i = 0; i < 3; i++) {
      r.color[i] = l.color[i];
      r.position[i] = l.position[i];
      r.orientation[i] = l.orientation[i];
    }
    r.constant_term = l.constantTerm;
    r.linear_term = l.linearTerm;
    r.quadratic_term = l.quadraticTerm;
    r.atten_angle = l.attenAngle;
    r.fall_rate = l.fallRate;
    fs.lights.push_back(r);
  }

  for (size_t oi = 0; oi < scene.objects.size(); oi++) {
    const Geometry& g = *scene.objects[oi];
    RtObject o;
    std::memset(&o, 0, sizeof o);
    o.kind = g.kind;
    o.object_id = g.objectID;
    o.material = (int)fs.materials.size();
    fs.materials.push_back(g.material.flatten());
    o.mesh = -1;
    std::memcpy(o.inverse, g.transform->inverse.n, sizeof o.inverse);
    std::memcpy(o.normi, g.transform->normi.n, sizeof o.normi);
    for (int i = 0; i < 3; i++) {
      o.bbox_min[i] = g.bounds.bmin[i];
      o.bbox_max[i] = g.bounds.bmax[i];
    }
    o.cone_height = g.height;
    o.cone_b_radius = g.bRadius;
    o.cone_t_radius = g.tRadius;
    o.cone_beta_squared = g.betaSquared;
    o.cone_gamma = g.gamma;
    o.capped = g.capped ? 1 : 0;
    if (g.kind == RT_OBJ_TRIMESH) {
      const Trimesh& tm = *g.mesh;
      RtMesh m;
      std::memset(&m, 0, sizeof m);
      o.mesh = (int)fs.meshes.size();
      m.object = (int)oi;
      m.first_vertex = (int)(fs.vertices.size() / 3);
      m.n_vertices = (int)tm.vertices.size();
      m.first_face = (int)fs.faces.size();
      m.n_faces = (int)tm.faces.size();
      m.has_vertex_normals = tm.vertNorms ? 1 : 0;
      m.has_vertex_materials = tm.materials.empty() ? 0 : 1;
      for (int i = 0; i < 3; i++) {
        m.root_bb_min[i] = g.bounds.bmin[i];
        m.root_bb_max[i] = g.bounds.bmax[i];
      }
      for (size_t vi = 0; vi < tm.vertices.size(); vi++) {
        for (int i = 0; i < 3; i++) {
          fs.vertices.push_back(tm.vertices[vi][i]);
          fs.vnormals.push_back(tm.vertNorms ? tm.normals[vi][i] : 0.0);
        }
        if (!tm.materials.empty()) {
          fs.vmaterials.push_back((int)fs.materials.size());
          fs.materials.push_back(tm.materials[vi].flatten());
        } else {
          fs.vmaterials.push_back(-1);
        }
      }
      if (tm.vertNorms) fs.hasVertexNormals = true;
      if (!tm.materials.empty()) fs.hasVertexMaterials = true;
      for (const TrimeshFace& f : tm.faces) {
        RtFace rf;
        std::memset(&rf, 0, sizeof rf);
        for (int i = 0; i < 3; i++) {
          rf.ids[i] = f.ids[i];
          rf.normal[i] = f.normal[i];
        }
        rf.object_id = f.objectID;
        fs.faces.push_back(rf);
      }
      m.kd_root = -2;  // patched below once the scene tree has been appended
      fs.meshes.push_back(m);
    }
    fs.objects.push_back(o);
  }

  d.scene_kd_root = appendTree(fs, scene.kd, 0);
  for (RtMesh& m : fs.meshes) {
    const Trimesh& tm = *scene.objects[m.object]->mesh;
    m.kd_root = appendTree(fs, tm.kd, m.first_face);
  }
  d.cube_w = d.cube_h = 0;
  fs.fixPointers();
  return fsp;
}

// ------------------------------------------------------------------ AA jitter
std::vector<double> makeJitterTable(unsigned seed, int n) {
  std::vector<double> j((size_t)2 * n * n);
  srand(seed);
  // RAND_MAX+1 overflows int in the reference's expression and evaluates to
  // INT_MIN with g++ on x86-64, hence the negative divisor.
  const double divisor = -2147483648.0;
  for (size_t k = 0; k < j.size(); k++) j[k] = (double)rand() / divisor;
  return j;
}

}  // namespace rta
