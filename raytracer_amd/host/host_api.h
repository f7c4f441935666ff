/* C entry points of the host front-end: parse a .ray scene, build its KD-trees
 * with the reference's builder, flatten it into an RtSceneDesc (include/rt_amd.h).
 * Pure CPU code; no HIP dependency. */
#ifndef RT_HOST_API_H
#define RT_HOST_API_H
#include "../../include/rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Scene-load options = the TraceUI switches RayTracer::loadScene reads
 * (src/RayTracer.cpp:306-317; defaults src/ui/TraceUI.h:21-28). */
typedef struct RthOptions {
  int kd_tree;
  int kd_max_depth;
  int kd_leaf_size;
  int backface_culling;
  int smooth_shading;
} RthOptions;

typedef struct RthScene RthScene;

void rth_default_options(RthOptions* o);
RthScene* rth_scene_parse(const char* text, const char* file_name, const RthOptions* opt,
                          char* err, int err_len);
RthScene* rth_scene_load(const char* path, const RthOptions* opt, char* err, int err_len);
const RtSceneDesc* rth_scene_desc(const RthScene* h);
double rth_scene_aspect_ratio(const RthScene* h);
/* Installs six RGB8 faces in CubeMap::tMap slot order (src/scene/cubeMap.h:15-38). */
int rth_scene_set_cubemap(RthScene* h, int w, int hgt, const unsigned char* const faces[6]);
int rth_scene_load_cubemap_bmp(RthScene* h, const char* const paths[6]);
void rth_scene_free(RthScene* h);
int rth_jitter_table(unsigned seed, int n, double* out);
int rth_write_bmp(const char* path, int w, int h, const unsigned char* rgb);
int rth_read_bmp_size(const char* path, int* w, int* h);
int rth_read_bmp(const char* path, int w, int h, unsigned char* rgb);
int rth_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif
