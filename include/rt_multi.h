/* rt_multi.h -- one frame over several GPUs of one node, from a C or C++ caller.
 *
 * Replaces the fan-out of CommandLineUI::run (src/ui/CommandLineUI.cpp:68-122:
 * m_nThreads std::threads, each tracing a strip of columns of the shared
 * buffer, then join and writeBMP) by: one host thread per GPU, each rendering
 * its interleaved share of the frame's 32x32 tiles through rt_render
 * (rt_amd.h) into a packed device buffer, ONE exchange per frame that brings
 * the packed tiles to the first GPU (RCCL point-to-point sends over xGMI,
 * grouped), rt_unpack_tiles there, and one copy of the finished frame to the
 * caller.  The scene is replicated on every GPU when the handle is created.
 *
 * librt_multi.so links librt_amd.so and librccl.so.  POD only, int error codes
 * as in rt_amd.h; rt_multi_last_error() holds the message of the last failure
 * of the calling thread.
 */
#ifndef RT_MULTI_H
#define RT_MULTI_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtMulti RtMulti;

/* Uploads the scene to each of the n_devices HIP devices listed in device_ids
 * (NULL: devices 0 .. n_devices-1) and sets up one RCCL communicator over
 * them (ncclCommInitAll).  The description is only read during the call. */
int rt_multi_create(const RtSceneDesc* desc, int n_devices, const int* device_ids, RtMulti** out);

/* Renders one frame: params as for rt_render (tile and shard fields, the
 * output location and the adaptive anti-aliasing mode are set by the library:
 * 32x32 tiles, shard r of n on device r, device output; aa_mode RT_AA_NONE or
 * RT_AA_FULL).  rgb8_out: host memory, width * height * 3 bytes, row 0 = bottom
 * row as in RayTracer::buffer.  stats (may be NULL): ray and shade-point counts
 * summed over the devices, times of the slowest device. */
int rt_multi_render(RtMulti* handle, const RtRenderParams* params, unsigned char* rgb8_out, RtStats* stats);

void rt_multi_destroy(RtMulti* handle);

int rt_multi_device_count(const RtMulti* handle);
const char* rt_multi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_MULTI_H */
