// C entry points of the host front-end (scene file -> flattened C-ABI scene),
// used by the Python harness through ctypes and by the `ray` CLI.
#include "host_api.h"

#include <cstring>
#include <string>

#include "scene.h"

using namespace rta;

struct RthScene {
  std::unique_ptr<Scene> scene;
  std::unique_ptr<FlatScene> flat;
};

namespace {
void setErr(char* err, int errLen, const std::string& msg) {
  if (!err || errLen <= 0) return;
  std::strncpy(err, msg.c_str(), (size_t)errLen - 1);
  err[errLen - 1] = 0;
}
SceneOptions toOptions(const RthOptions* o) {
  SceneOptions so;
  if (o) {
    so.kdTree = o->kd_tree != 0;
    so.kdMaxDepth = o->kd_max_depth;
    so.kdLeafSize = o->kd_leaf_size;
    so.backFaceCulling = o->backface_culling != 0;
    so.smoothShading = o->smooth_shading != 0;
  }
  return so;
}
}  // namespace

extern "C" {

void rth_default_options(RthOptions* o) {
  SceneOptions so;
  o->kd_tree = so.kdTree;
  o->kd_max_depth = so.kdMaxDepth;
  o->kd_leaf_size = so.kdLeafSize;
  o->backface_culling = so.backFaceCulling;
  o->smooth_shading = so.smoothShading;
}

RthScene* rth_scene_parse(const char* text, const char* file_name, const RthOptions* opt,
                          char* err, int err_len) {
  try {
    std::unique_ptr<RthScene> h(new RthScene);
    h->scene = parseSceneText(text);
    applyOptions(*h->scene, toOptions(opt), file_name ? file_name : "");
    h->flat = flattenScene(*h->scene);
    return h.release();
  } catch (const std::exception& e) {
    setErr(err, err_len, e.what());
    return nullptr;
  }
}

RthScene* rth_scene_load(const char* path, const RthOptions* opt, char* err, int err_len) {
  std::string msg;
  std::unique_ptr<RthScene> h(new RthScene);
  h->scene = loadSceneFile(path, toOptions(opt), &msg);
  if (!h->scene) {
    setErr(err, err_len, msg);
    return nullptr;
  }
  try {
    h->flat = flattenScene(*h->scene);
  } catch (const std::exception& e) {
    setErr(err, err_len, e.what());
    return nullptr;
  }
  return h.release();
}

const RtSceneDesc* rth_scene_desc(const RthScene* h) { return h ? &h->flat->desc : nullptr; }

double rth_scene_aspect_ratio(const RthScene* h) { return h->scene->camera.aspectRatio; }

int rth_scene_set_cubemap(RthScene* h, int w, int hgt, const unsigned char* const faces[6]) {
  if (!h || w <= 0 || hgt <= 0) return -1;
  h->flat->cubeFaces.assign(6, std::vector<unsigned char>());
  for (int f = 0; f < 6; f++) {
    if (!faces[f]) return -1;
    h->flat->cubeFaces[f].assign(faces[f], faces[f] + (size_t)w * hgt * 3);
  }
  h->flat->desc.cube_w = w;
  h->flat->desc.cube_h = hgt;
  h->flat->fixPointers();
  return 0;
}

int rth_scene_load_cubemap_bmp(RthScene* h, const char* const paths[6]) {
  if (!h) return -1;
  std::vector<std::vector<unsigned char>> imgs(6);
  int w0 = 0, h0 = 0;
  for (int f = 0; f < 6; f++) {
    int w, hh;
    if (!readBMP(paths[f], w, hh, imgs[f])) return -2;
    if (f == 0) { w0 = w; h0 = hh; }
    else if (w != w0 || hh != h0) return -3;
  }
  h->flat->cubeFaces = imgs;
  h->flat->desc.cube_w = w0;
  h->flat->desc.cube_h = h0;
  h->flat->fixPointers();
  return 0;
}

void rth_scene_free(RthScene* h) { delete h; }

int rth_jitter_table(unsigned seed, int n, double* out) {
  if (n <= 0 || !out) return -1;
  std::vector<double> j = makeJitterTable(seed, n);
  std::memcpy(out, j.data(), j.size() * sizeof(double));
  return 0;
}

int rth_write_bmp(const char* path, int w, int h, const unsigned char* rgb) {
  return writeBMP(path, w, h, rgb) ? 0 : -1;
}

int rth_read_bmp_size(const char* path, int* w, int* h) {
  std::vector<unsigned char> px;
  return readBMP(path, *w, *h, px) ? 0 : -1;
}

int rth_read_bmp(const char* path, int w, int h, unsigned char* rgb) {
  std::vector<unsigned char> px;
  int ww, hh;
  if (!readBMP(path, ww, hh, px) || ww != w || hh != h) return -1;
  std::memcpy(rgb, px.data(), px.size());
  return 0;
}

// sizeof of each C-ABI struct, so that foreign-language bindings can verify
// their mirrored layouts: 0 RtLight, 1 RtMaterial, 2 RtObject, 3 RtMesh, 4 RtFace,
// 5 RtSceneDesc, 6 RtRenderParams, 7 RtStats, 8 RtHit, 9 RthOptions.
int rth_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(RtLight);
    case 1: return (int)sizeof(RtMaterial);
    case 2: return (int)sizeof(RtObject);
    case 3: return (int)sizeof(RtMesh);
    case 4: return (int)sizeof(RtFace);
    case 5: return (int)sizeof(RtSceneDesc);
    case 6: return (int)sizeof(RtRenderParams);
    case 7: return (int)sizeof(RtStats);
    case 8: return (int)sizeof(RtHit);
    case 9: return (int)sizeof(RthOptions);
  }
  return -1;
}

}  // extern "C"
