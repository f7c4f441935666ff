/*
 * rt_amd.h -- C-ABI of the MI355X (gfx950) render path.
 *
 * The reference ray tracer (kaivanwadia/RayTracer, CPU-only C++) has no FFI
 * seam.  Its per-pixel render path sits behind these C++ methods:
 *     RayTracer::traceSetup(w,h)            src/RayTracer.cpp:323-335
 *     RayTracer::tracePixel(i,j)            src/RayTracer.cpp:45-62
 *     RayTracer::tracePixelAntiAlias(i,j)   src/RayTracer.cpp:64-103
 *     RayTracer::getBuffer(buf,w,h)         src/RayTracer.cpp:238-243
 * which CommandLineUI::run fans out over std::threads on column strips
 * (src/ui/CommandLineUI.cpp:83-100).  This library cuts at frame granularity:
 * the host flattens its Scene once (rt_scene_create), and one rt_render call
 * replaces the whole thread fan-out + tracePixel double loop.
 * rt_intersect_batch exposes the per-ray seam Scene::intersect
 * (src/scene/scene.cpp:133-159) for tests.
 *
 * Conventions: plain C, POD only, no exceptions cross the boundary.  Every
 * function returns 0 on success or a negative RtError; rt_last_error() returns
 * a thread-local message for the last failure.  All real values are IEEE
 * binary64 unless marked float.  Host pointers inside RtSceneDesc need only
 * stay valid for the duration of rt_scene_create (the library copies).
 * One host thread per RtScene handle; a handle is bound to the HIP device
 * that was current at rt_scene_create.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define RT_AMD_ABI_VERSION 3

typedef enum RtError {
  RT_OK = 0,
  RT_ERR_INVALID = -1,   /* bad argument / inconsistent scene description */
  RT_ERR_DEVICE = -2,    /* HIP runtime failure (message has the HIP error) */
  RT_ERR_NOMEM = -3,     /* device workspace could not be allocated */
  RT_ERR_UNSUPPORTED = -4
} RtError;

/* ray::RayType, src/scene/ray.h:25-32 (same numeric values) */
enum { RT_RAY_VISIBILITY = 0, RT_RAY_REFLECTION = 1, RT_RAY_REFRACTION = 2,
       RT_RAY_REFRACTIONOUT = 3, RT_RAY_SHADOW = 4 };

/* Light kinds: DirectionalLight / PointLight / SpotLight, src/scene/light.h:32-115 */
enum { RT_LIGHT_DIRECTIONAL = 0, RT_LIGHT_POINT = 1, RT_LIGHT_SPOT = 2 };

typedef struct RtLight {
  int    type;
  int    _pad0;
  double color[3];
  double position[3];        /* point, spot */
  double orientation[3];     /* directional, spot: already normalised by the ctor (light.h:37,56) */
  float  constant_term;      /* point light a,b,c are float in the reference (light.h:105-107) */
  float  linear_term;
  float  quadratic_term;
  float  _pad1;
  double atten_angle;        /* spot: degrees */
  double fall_rate;          /* spot */
} RtLight;

/* Material, src/scene/material.h:158-274.  shininess/index keep the Vec3
 * _value so that MaterialParameter::intensityValue (material.cpp:158-167:
 * 0.299 v0 + 0.587 v1 + 0.114 v2) and per-vertex interpolation
 * (material.h:183-195,277-289) behave as in the reference. */
typedef struct RtMaterial {
  double ke[3], ka[3], ks[3], kd[3], kr[3], kt[3];
  double shininess[3];
  double index[3];
  int    shininess_mapped;   /* 1: shininess() multiplies by 128 (material.h:209-215) */
  int    _pad;
} RtMaterial;

/* Object kinds: SceneObjects/{Sphere,Box,Square,Cylinder,Cone,trimesh} */
enum { RT_OBJ_SPHERE = 0, RT_OBJ_BOX = 1, RT_OBJ_SQUARE = 2, RT_OBJ_CYLINDER = 3,
       RT_OBJ_CONE = 4, RT_OBJ_TRIMESH = 5 };

/* One entry of Scene::objects (scene.h:386), in scene order. */
typedef struct RtObject {
  int    kind;
  int    object_id;          /* Geometry::objectID (scene.cpp:12, scene.h:260-263) */
  int    material;           /* index into RtSceneDesc.materials */
  int    mesh;               /* index into RtSceneDesc.meshes for RT_OBJ_TRIMESH, else -1 */
  double inverse[16];        /* TransformNode::inverse, row-major (scene.h:140,184) */
  double normi[9];           /* TransformNode::normi  (scene.h:141,185) */
  double bbox_min[3];        /* Geometry::bounds, world space (scene.h:214-252) */
  double bbox_max[3];
  /* Cone only (SceneObjects/Cone.h:10-37), values after the ctor's fix-ups */
  double cone_height, cone_b_radius, cone_t_radius, cone_beta_squared, cone_gamma;
  int    capped;             /* Cone::capped; Cylinder is always capped (Cylinder.h:11) */
  int    _pad;
} RtObject;

/* One Trimesh (SceneObjects/trimesh.h:13-80). Vertex data live in the shared
 * arrays of RtSceneDesc at [first_vertex, first_vertex + n_vertices). */
typedef struct RtMesh {
  int    object;             /* index into objects[] of the owning Trimesh */
  int    first_vertex, n_vertices;
  int    first_face, n_faces;           /* range in faces[] (non-degenerate faces only) */
  int    has_vertex_normals;            /* Trimesh::vertNorms */
  int    has_vertex_materials;          /* Trimesh::materials.size() > 0 */
  int    kd_root;                       /* node index of Trimesh::kdtreeRoot, -1 if not built */
  double root_bb_min[3], root_bb_max[3];/* kdtreeRoot->bb == Trimesh world bounds (scene.cpp:578) */
} RtMesh;

/* One TrimeshFace (trimesh.h:82-149). ids are mesh-local vertex indices. */
typedef struct RtFace {
  int    ids[3];
  int    object_id;          /* Geometry::objectID of the face */
  double normal[3];          /* TrimeshFace::normal (trimesh.h:110-111) */
} RtFace;

/* Flattened scene.  KD nodes of the scene tree and of every mesh tree share one
 * node array (KdTree<Geometry>, scene.h:30-115).  A node is a leaf iff
 * left < 0 && right < 0 (scene.cpp:68).  Leaf items of the scene tree index
 * objects[]; leaf items of a mesh tree index faces[]. */
typedef struct RtSceneDesc {
  int    abi_version;        /* RT_AMD_ABI_VERSION */
  /* Camera state after Camera::update (camera.cpp:91-97) */
  double eye[3], look[3], u[3], v[3];
  double ambient[3];         /* Scene::ambient() */
  int    use_kdtree;         /* Scene::useKdTree && kdtreeRoot != nullptr */
  int    backface_cull;      /* Scene::backFaceCulling */
  int    smooth_shading;     /* Scene::smoothShading */
  int    n_lights;     const RtLight*    lights;
  int    n_materials;  const RtMaterial* materials;
  int    n_objects;    const RtObject*   objects;
  int    n_meshes;     const RtMesh*     meshes;
  int    n_vertices;   const double*     vertices;    /* 3 per vertex, mesh-local space */
                       const double*     vnormals;    /* 3 per vertex or NULL */
                       const int*        vmaterials;  /* per-vertex material index or NULL */
  int    n_faces;      const RtFace*     faces;
  int    n_nodes;
  const double* node_split;      /* splitting coordinate (scene.cpp:476) */
  const int*    node_dim;        /* 0,1,2 for internal nodes */
  const int*    node_left;       /* child node index or -1 */
  const int*    node_right;
  const int*    node_item_start; /* leaf: range in leaf_items */
  const int*    node_item_count;
  int    n_leaf_items; const int* leaf_items;
  int    scene_kd_root;          /* node index of Scene::kdtreeRoot, -1 if none */
  double scene_bb_min[3], scene_bb_max[3]; /* kdtreeRoot->bb == Scene::bounds() (scene.cpp:349) */
  /* Optional cube map: six W x H RGB8 images, row 0 first, in CubeMap::tMap
   * slot order 0..5 (cubeMap.h:15-38). NULL faces => no cube map. */
  int    cube_w, cube_h;
  const unsigned char* cube_rgb[6];
} RtSceneDesc;

/* Traversal flavour of Scene::intersect.
 *  RT_TRAVERSE_REFERENCE: walks the caller's KD-trees with the reference's push
 *    rules, far-child-first order, no early exit, per-candidate bounding-box
 *    cull (scene.cpp:59-129, 13-36); results equal the reference's, including
 *    the hits its split-plane rules miss.
 *  RT_TRAVERSE_FAST: walks a bounding volume hierarchy that rt_scene_create
 *    builds itself over all triangles and objects (4-wide nodes, quantised
 *    child boxes), nearest child first, culling subtrees behind the closest
 *    hit; shadow rays of scenes without transmissive materials stop at the
 *    first occluder.  The primitive arithmetic is the same, so the closest hit
 *    equals the reference's linear scan over all primitives (m_bKdTree off). */
enum { RT_TRAVERSE_REFERENCE = 0, RT_TRAVERSE_FAST = 1 };

enum { RT_AA_NONE = 0, RT_AA_FULL = 1, RT_AA_ADAPTIVE = 2 };

typedef struct RtRenderParams {
  int width, height;
  int depth;                 /* TraceUI::m_nDepth (TraceUI.h:21) */
  int aa_mode;               /* RT_AA_* */
  int aa_samples;            /* TraceUI::m_nPixelSamples: n x n samples per pixel */
  int aa_threshold;          /* TraceUI::m_nSupersampleThreshold (adaptive) */
  const double* jitter;      /* 2*n*n values, order of RayTracer.cpp:87-92; NULL => zeros */
  int use_cubemap;           /* TraceUI::m_usingCubeMap */
  int filter_width;          /* TraceUI::m_nFilterWidth */
  int traversal;             /* RT_TRAVERSE_* */
  int collect_stats;         /* 1: count kd nodes visited / primitives tested (slower) */
  /* Screen-tile sharding: the frame is cut into tile_w x tile_h tiles numbered
   * row-major; this call renders tiles t with t % shard_count == shard_index. */
  int tile_w, tile_h;
  int shard_index, shard_count;
  /* Output placement. 0: rgb8_out/hit_id_out are host pointers to full frames
   * (w*h*3 and w*h; only this shard's pixels are written).  1: they are device
   * pointers to the packed tile buffer of this shard (rt_shard_bytes). */
  int out_is_device;
  int max_batch_samples;     /* 0 = library default */
} RtRenderParams;

typedef struct RtStats {
  unsigned long long rays_primary, rays_reflect, rays_refract, rays_shadow;
  unsigned long long nodes_visited, prims_tested;   /* collect_stats only: tree nodes visited, triangles tested */
  unsigned long long shade_points;
  double ms_total;           /* host wall time of the call */
  double ms_device;          /* HIP-event time of all kernels of the call */
  double ms_trace;           /* HIP-event time of the closest-hit launches (k_trace); 0 unless collect_stats is set
                                or the scene was created under RT_AMD_LAUNCH_EVENTS=1 (the events cost a little) */
  int    trace_launches;     /* number of those launches */
  int    batches;
  double ms_lights;          /* HIP-event time of the shadow-query launches (k_shadow); as ms_trace */
  int    lights_launches;
  int    _pad;
  unsigned long long shadow_nodes_visited, shadow_prims_tested;  /* shadow rays' share of the two totals */
  int    batches_redone;     /* batches repeated because a recursion level outgrew the grid estimated for it */
  int    _pad2;
  unsigned long long objects_tested, shadow_objects_tested;      /* collect_stats only: non-mesh objects tested */
} RtStats;

/* Result of Scene::intersect for one ray (isect, src/scene/ray.h:57-112). */
typedef struct RtHit {
  double t;                  /* 1000.0 on a miss (scene.cpp:155) */
  double N[3];
  double bary[3];
  int    object_id;          /* 0 on a miss */
  int    _pad;
} RtHit;

typedef struct RtScene RtScene;

int rt_scene_create(const RtSceneDesc* desc, RtScene** out);
void rt_scene_destroy(RtScene* scene);

/* Renders one frame (or this shard's tiles of it).  rgb8_out: RGB8, row 0 =
 * bottom image row as in RayTracer::buffer (RayTracer.cpp:54, bitmap.cpp:140).
 * hit_id_out (optional, may be NULL): objectID of the primary hit per pixel,
 * 0 on a miss (first sub-sample when anti-aliasing).  stats may be NULL. */
int rt_render(RtScene* scene, const RtRenderParams* params,
              unsigned char* rgb8_out, int* hit_id_out, RtStats* stats);

/* Scene::intersect for n rays given as 6 doubles (p, d) each. Host pointers.
 * Stages through a device buffer kept in the scene handle, which grows (one
 * hipMalloc) when a call brings more rays than any before it.  Like every call
 * that takes a handle: one caller at a time per handle (calls on the same
 * handle from two threads share its staging buffer, streams and ray pool). */
int rt_intersect_batch(RtScene* scene, int traversal, int n, const double* rays_pd,
                       const int* ray_types, RtHit* hits);

/* rt_intersect_batch for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid it).  It has no
 * counterpart in the reference.  Same arguments and results as
 * rt_intersect_batch, but rays, types and hit records are staged in an array
 * that is part of the library's code object instead of a buffer the handle
 * may have to grow: the call allocates and frees nothing.  At most 65536 rays per call
 * (more: RT_ERR_INVALID).  Every GPU the library runs on holds its own copy of
 * the array (7.25 MiB) until the process ends; all calls on one GPU share it,
 * so calls that overlap on one GPU -- from different threads or scenes -- race. */
int rt_intersect_batch_static(RtScene* scene, int traversal, int n, const double* rays_pd,
                              const int* ray_types, RtHit* hits);

/* Number of pixels / bytes of the packed tile buffer of one shard, and the
 * scatter of packed tiles back into a full frame on the device.  Packed layout:
 * the shard's tiles in increasing tile number, each tile row-major, clipped at
 * the frame border.  rt_unpack_tiles is ordered on hip_stream and returns
 * without waiting for it: the caller synchronises that stream before it reads
 * frame_dev from elsewhere.  The offset table of a (frame, tiling, shard) is
 * built on its first use and kept for the life of the process. */
long long rt_shard_pixels(int width, int height, int tile_w, int tile_h,
                          int shard_index, int shard_count);
int rt_unpack_tiles(const unsigned char* packed_dev, int bytes_per_pixel,
                    int width, int height, int tile_w, int tile_h,
                    int shard_index, int shard_count, unsigned char* frame_dev, void* hip_stream);

/* Renders the listed pixels only: tracePixel (aa_mode RT_AA_NONE,
 * src/RayTracer.cpp:45-62) or tracePixelAntiAlias with aa_samples x aa_samples
 * jittered samples each (any other aa_mode, src/RayTracer.cpp:64-103).  This is
 * the second pass of GraphicalUI::doAntiAliasing (src/ui/GraphicalUI.cpp:377-435)
 * for a caller that holds the flagged pixels itself -- e.g. one rank of a
 * multi-GPU frame that takes its share of them.  xy: n_pixels (x, y) pairs;
 * rgb8_out: 3 bytes per listed pixel, in list order (both host memory).  Tile and
 * shard fields of the parameters are ignored. */
int rt_render_pixels(RtScene* scene, const RtRenderParams* params, int n_pixels, const int* xy,
                     unsigned char* rgb8_out, RtStats* stats_out);

/* The second pass of GraphicalUI::doAntiAliasing (src/ui/GraphicalUI.cpp:377-435)
 * for one rank of a frame that is sharded over several GPUs, without the frame
 * leaving the device: base_rgb_dev is the COMPLETE base image (width * height *
 * 3 bytes, row 0 = bottom row) in this device's memory -- every rank holds the
 * same one after the all-gather of the base pass.  The call builds the edge
 * mask (applyFilter, :437-486, cutoff params->aa_threshold) and the list of
 * flagged pixels in the reference's row-by-row order on the device, takes
 * entries share_index, share_index + share_count, ... of it, and renders them
 * with aa_samples x aa_samples jittered samples (tracePixelAntiAlias) into
 * rows_dev: 3 bytes per pixel of the share, in list order (device memory, room
 * for rows_capacity pixels).  *n_flagged: pixels flagged in the whole frame (the
 * same on every rank); *n_rendered: this share's part.  The caller makes sure
 * that whatever wrote base_rgb_dev has finished.  Mask and list live in buffers
 * the handle keeps; the list stays valid for rt_patch_flagged until the next
 * call that builds one (this one, or rt_render in adaptive mode). */
int rt_render_flagged(RtScene* scene, const RtRenderParams* params, const unsigned char* base_rgb_dev,
                      int share_index, int share_count, unsigned char* rows_dev, long long rows_capacity,
                      long long* n_flagged, long long* n_rendered, RtStats* stats_out);

/* Writes the re-rendered pixels of share share_index of share_count (rows_dev,
 * as rt_render_flagged of that share left them -- on this or another rank) over
 * their places in frame_dev, a full frame in this device's memory, using the
 * flagged list the handle built last.  Ordered on hip_stream, returns without
 * waiting (GraphicalUI.cpp:357-375 overwrites the buffer's pixels in place). */
int rt_patch_flagged(RtScene* scene, int share_index, int share_count, const unsigned char* rows_dev,
                     unsigned char* frame_dev, void* hip_stream);

/* Laplacian edge mask of GraphicalUI::applyFilter (ui/GraphicalUI.cpp:437-486)
 * on a host RGB8 frame; mask_out gets 255 / 0 per pixel. */
int rt_edge_mask(const unsigned char* rgb8, int width, int height, int cutoff,
                 unsigned char* mask_out);

/* Test hook: out[i] = x[i] ^ (n + delta) as the shading kernels compute the
 * powers of Material::shade and SpotLight::shadowAttenuation
 * (src/scene/material.cpp:72, src/scene/light.cpp:147) when every exponent of a
 * scene is a small non-negative integer n up to a rounding residue delta
 * (|delta| <= 1e-12).  Host pointers. */
int rt_test_pow_near_int(int count, const double* x, int n, double delta, double* out);

/* Test hook, host only (no device call): builds the acceleration structure of
 * RT_TRAVERSE_FAST over n_tris world-space triangles (vertices: 9 doubles each;
 * normals: the face normal TrimeshFace::intersectLocal would use, 3 doubles
 * each, src/SceneObjects/trimesh.cpp:93-104) by the steps rt_scene_create takes
 * for a scene of identity-transform meshes -- binary SAH tree, 4-wide nodes
 * with 8-bit planes, binary32 child boxes, a normal cone per child, eight
 * sign-ordered copies per node -- and checks what the traversal relies on.
 * report[0..15]: [0] 4-wide nodes, [1] their depth, [2] binary nodes,
 * [3] triangles not found exactly once below the root (or unreachable nodes,
 * or a child whose triangles are not one contiguous run), [4] / [5] (triangle,
 * ancestor child) pairs with a vertex outside the child's binary32 / decoded
 * 8-bit box, [6] pairs with the triangle's normal outside the child's cone,
 * [7] sampled directions that the visit's binary32 cone test turns away from a
 * child although a triangle below it has normal . direction <= 0 (the reference
 * would have tested it), [8] sampled directions turned away in all, [9] node
 * copies that are not a reordering of the node's first copy, [10] copies whose
 * children are not sorted near to far along the copy's diagonal, [11] unused
 * slots without an inverted box, [12] children with a cone narrow enough to
 * cull by, [13] leaves with more than max_leaf (1..8) triangles.  A correct
 * build has zeros in [3..7], [9..11] and [13]. */
int rt_test_bvh_invariants(int n_tris, const double* vertices, const double* normals, int max_leaf,
                           long long* report);

const char* rt_last_error(void);
int rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
