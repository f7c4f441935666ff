// Host scene model: the same vocabulary as the reference's Scene / SceneObjects
// API (src/scene/scene.h, src/SceneObjects/*.h), kept as plain data so it can
// be flattened into the C-ABI arrays of include/rt_amd.h.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "vecmath.h"

namespace rta {

const double RAY_EPSILON = 0.00000001;  // src/scene/ray.h:114

// Axis-aligned box with the reference's empty/merge/area rules (src/scene/bbox.h:3-127).
struct BoundingBox {
  bool empty = true;
  Vec3 bmin, bmax;
  BoundingBox() {}
  BoundingBox(const Vec3& lo, const Vec3& hi) : empty(false), bmin(lo), bmax(hi) {}
  void merge(const BoundingBox& o);
  double area() const;
};

// One node of the transform hierarchy (scene.h:134-187): xform is the product
// of all ancestors, inverse and normi are derived once at construction.
struct TransformNode {
  Mat4 xform, inverse;
  Mat3 normi;
  std::vector<std::unique_ptr<TransformNode>> children;
  TransformNode();                                         // root: identity
  TransformNode(const TransformNode& parent, const Mat4& local);
  TransformNode* createChild(const Mat4& local);
};

// One material parameter: constant Vec3, or "mapped" (map("file") in the scene
// file).  The reference's TextureMap::getMappedValue is a stub returning white
// (src/scene/material.cpp:113-127), so a mapped parameter evaluates to (1,1,1).
struct MaterialParameter {
  Vec3 value;
  bool mapped = false;
  Vec3 eval() const { return mapped ? Vec3(1.0, 1.0, 1.0) : value; }
};

struct Material {                      // src/scene/material.h:158-274
  MaterialParameter ke, ka, ks, kd, kr, kt, shininess, index;
  Material() { index.value = Vec3(1.0, 1.0, 1.0); }
  RtMaterial flatten() const;
};

struct Light {                         // src/scene/light.h:13-115
  int type = RT_LIGHT_POINT;
  Vec3 color, position, orientation;
  float constantTerm = 0.0f, linearTerm = 0.0f, quadraticTerm = 1.0f;
  double attenAngle = 0.0, fallRate = 0.0;
};

struct Camera {                        // src/scene/camera.{h,cpp}
  Mat3 m;
  double normalizedHeight = 1.0, aspectRatio = 1.0;
  Vec3 eye, look, u, v;
  Camera();
  void setEye(const Vec3& e) { eye = e; }
  void setLook(double r, double i, double j, double k);
  void setLook(const Vec3& viewDir, const Vec3& upDir);
  void setFOV(double fovDegrees);
  void setAspectRatio(double ar);
  void update();
};

struct Trimesh;

// Any entry of Scene::objects.  kind selects the primitive (RT_OBJ_*).
struct Geometry {
  int kind = RT_OBJ_SPHERE;
  int objectID = 0;
  Material material;
  const TransformNode* transform = nullptr;
  BoundingBox bounds;                  // world space, Geometry::ComputeBoundingBox
  // Cone parameters after the constructor's adjustments (Cone.h:10-37)
  double height = 1.0, bRadius = 1.0, tRadius = 0.0, betaSquared = 0.0, gamma = 0.0;
  bool capped = true;
  std::unique_ptr<Trimesh> mesh;       // RT_OBJ_TRIMESH only
  BoundingBox localBounds() const;     // ComputeLocalBoundingBox of each primitive
  void computeBoundingBox();           // scene.h:214-252
};

struct TrimeshFace {                   // trimesh.h:82-149
  int ids[3];
  int objectID;
  Vec3 normal;
  BoundingBox bounds;                  // local-space box of the three vertices
};

struct KdNode {                        // KdTree<Geometry>, scene.h:30-115
  int left = -1, right = -1;
  std::vector<int> items;              // object indices (scene tree) or face indices (mesh tree)
  BoundingBox bb;
  double split = 0.0;
  int dim = 4;
};

struct KdTree {
  std::vector<KdNode> nodes;           // nodes[0] is the root
  bool built() const { return !nodes.empty(); }
};

struct Trimesh {                       // trimesh.h:13-80
  std::vector<Vec3> vertices, normals;
  std::vector<Material> materials;     // per-vertex, optional
  std::vector<TrimeshFace> faces;      // non-degenerate faces in insertion order
  bool vertNorms = false;
  KdTree kd;
  void generateNormals();              // trimesh.cpp:210-238
};

struct SceneOptions {                  // TraceUI defaults, src/ui/TraceUI.h:21-28
  bool kdTree = true;
  int kdMaxDepth = 15;
  int kdLeafSize = 10;
  bool backFaceCulling = true;
  bool smoothShading = true;
};

class Scene {                          // src/scene/scene.h:313-407
 public:
  TransformNode transformRoot;
  std::vector<std::unique_ptr<Geometry>> objects;
  std::vector<Light> lights;
  Camera camera;
  Vec3 ambientIntensity;
  BoundingBox sceneBounds;
  KdTree kd;
  bool useKdTree = false, backFaceCulling = false, smoothShading = false;
  int kdTreeDepth = 0, kdTreeLeafSize = 0;
  int nextObjectID = 1;                // Geometry::idGen, restarted per scene

  void add(std::unique_ptr<Geometry> g);            // scene.h:339-347
  void buildKdTree(int depth, int leafSize);        // scene.cpp:328-387
  long long kdBuildSorts = 0;                       // diagnostics
 private:
  typedef std::pair<int, int> Event;   // (item, 0 = box minimum | 1 = box maximum)
  struct BuildBoxes {                  // per-tree item boxes, sort keys and areas
    std::vector<BoundingBox> box;
    std::vector<double> key[3];        // key[d][2*item + flag]
    std::vector<double> area;
    void add(const BoundingBox& b);
  };
  void buildNode(KdTree& tree, int nodeIdx, int depth, int leafSize,
                 std::vector<Event> (&planes)[3], const BuildBoxes& bx, bool sceneLevel);
  void buildTrimeshKdTree(Trimesh& mesh, const BoundingBox& worldBounds);
  void buildMeshesIn(const KdNode& node);
};

// Owns the arrays an RtSceneDesc points into.
struct FlatScene {
  RtSceneDesc desc;
  std::vector<RtLight> lights;
  std::vector<RtMaterial> materials;
  std::vector<RtObject> objects;
  std::vector<RtMesh> meshes;
  std::vector<double> vertices, vnormals;
  std::vector<int> vmaterials;
  std::vector<RtFace> faces;
  std::vector<double> nodeSplit;
  std::vector<int> nodeDim, nodeLeft, nodeRight, nodeItemStart, nodeItemCount, leafItems;
  std::vector<std::vector<unsigned char>> cubeFaces;
  bool hasVertexNormals = false, hasVertexMaterials = false;
  void fixPointers();
};

std::unique_ptr<FlatScene> flattenScene(const Scene& scene);

// .ray parser (src/parser/*). Throws std::runtime_error with line information.
std::unique_ptr<Scene> parseSceneText(const std::string& text);
std::unique_ptr<Scene> loadSceneFile(const std::string& path, const SceneOptions& opt,
                                     std::string* error);
void applyOptions(Scene& scene, const SceneOptions& opt, const std::string& fileName);

// 24-bit BMP I/O (src/fileio/bitmap.cpp:17-155): bottom-up rows, BGR on disk.
bool writeBMP(const std::string& path, int width, int height, const unsigned char* rgb);
bool readBMP(const std::string& path, int& width, int& height, std::vector<unsigned char>& rgb);

// 2*n*n jitter values in the order RayTracer::tracePixelAntiAlias draws them
// after srand(seed) (src/RayTracer.cpp:70,87-92), including the signed
// overflow of RAND_MAX+1 that makes every value fall in (-1, 0].
std::vector<double> makeJitterTable(unsigned seed, int n);

}  // namespace rta
