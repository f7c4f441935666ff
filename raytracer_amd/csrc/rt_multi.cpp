// One frame over several GPUs of one node (include/rt_multi.h): the
// multi-device counterpart of CommandLineUI::run's thread fan-out
// (src/ui/CommandLineUI.cpp:68-122).  One host thread per GPU renders that
// GPU's interleaved tiles through the C-ABI of rt_amd.h; one grouped RCCL
// exchange per frame brings the packed tiles to the first GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_multi.h"

namespace {

constexpr int kTile = 32;

thread_local std::string g_multiError;

int failMulti(int code, const std::string& msg) {
  g_multiError = msg;
  return code;
}

struct Rank {
  int device = 0;
  RtScene* scene = nullptr;
  hipStream_t stream = nullptr;      // exchange and unpack of this device
  ncclComm_t comm = nullptr;
  unsigned char* packed = nullptr;   // this device's tiles of the frame, packed
  size_t packedBytes = 0;
  // what one frame left behind (written by the device's thread)
  int rc = RT_OK;
  std::string error;
  RtStats stats;
};

}  // namespace

struct RtMulti {
  std::vector<Rank> ranks;
  // on the first device: one landing buffer per other device, and the frame
  std::vector<unsigned char*> landing;
  std::vector<size_t> landingBytes;
  unsigned char* frame = nullptr;
  size_t frameBytes = 0;
  bool commsReady = false;
};

namespace {

#define MULTI_HIP(expr)                                                                               \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) return failMulti(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Grows a device buffer of the current device to at least `bytes`.
int ensureBytes(unsigned char** buf, size_t* have, size_t bytes) {
  if (*have >= bytes) return RT_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *have = 0;
  MULTI_HIP(hipMalloc((void**)buf, bytes));
  *have = bytes;
  return RT_OK;
}

// One device's part of a frame: render its tiles into the device's packed buffer.
void renderShard(RtMulti* m, int r, const RtRenderParams* params, const std::vector<size_t>* shardBytes) {
  Rank& k = m->ranks[(size_t)r];
  const int n = (int)m->ranks.size();
  k.rc = RT_OK;
  k.error.clear();
  std::memset(&k.stats, 0, sizeof k.stats);
  auto bail = [&](int code, const std::string& what) {
    k.rc = code;
    k.error = what;
  };
  if (hipSetDevice(k.device) != hipSuccess) return bail(RT_ERR_DEVICE, "hipSetDevice failed");
  RtRenderParams p = *params;
  p.tile_w = p.tile_h = kTile;
  p.shard_index = r;
  p.shard_count = n;
  p.out_is_device = 1;
  const size_t bytes = (*shardBytes)[(size_t)r];
  if (bytes > 0) {
    if (ensureBytes(&k.packed, &k.packedBytes, bytes) != RT_OK) return bail(RT_ERR_NOMEM, g_multiError);
    // (returns when the shard is rendered: the buffer is complete for the exchange)
    if (rt_render(k.scene, &p, k.packed, nullptr, &k.stats) != RT_OK) return bail(RT_ERR_DEVICE, rt_last_error());
  }
}

// The frame's one exchange, after every device has rendered its shard: every
// other device sends its packed tiles to the first, which posts one receive per
// sender -- all inside one group, so the transfers of the seven links run side
// by side.  (A device never sends to itself: the first device's tiles are
// unpacked from where they were rendered.)  Issued by the calling thread for
// all devices, and only when every shard was rendered: a device that failed
// cannot leave the others waiting for a transfer that never comes.
int exchangeShards(RtMulti* m, const std::vector<size_t>& shardBytes) {
  const int n = (int)m->ranks.size();
  if (n < 2) return RT_OK;
  ncclResult_t nr = ncclGroupStart();
  for (int q = 1; q < n && nr == ncclSuccess; q++) {
    const size_t bytes = shardBytes[(size_t)q];
    if (bytes == 0) continue;
    Rank& k0 = m->ranks[0];
    Rank& kq = m->ranks[(size_t)q];
    // (each call with its communicator's device current, as for any other work queued on that device's stream)
    if (hipSetDevice(k0.device) != hipSuccess) { nr = ncclUnhandledCudaError; break; }
    nr = ncclRecv(m->landing[(size_t)q], bytes, ncclUint8, q, k0.comm, k0.stream);
    if (nr != ncclSuccess) break;
    if (hipSetDevice(kq.device) != hipSuccess) { nr = ncclUnhandledCudaError; break; }
    nr = ncclSend(kq.packed, bytes, ncclUint8, 0, kq.comm, kq.stream);
  }
  const ncclResult_t ne = ncclGroupEnd();   // (always closed, also after a failed call inside the group)
  if (nr != ncclSuccess || ne != ncclSuccess)
    return failMulti(RT_ERR_DEVICE, std::string("RCCL exchange: ") + ncclGetErrorString(nr != ncclSuccess ? nr : ne));
  for (int q = 0; q < n; q++) {
    Rank& k = m->ranks[(size_t)q];
    MULTI_HIP(hipSetDevice(k.device));
    MULTI_HIP(hipStreamSynchronize(k.stream));
  }
  return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_multi_last_error(void) { return g_multiError.c_str(); }

int rt_multi_device_count(const RtMulti* m) { return m ? (int)m->ranks.size() : 0; }

void rt_multi_destroy(RtMulti* m) {
  if (!m) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (size_t r = 0; r < m->ranks.size(); r++) {
    Rank& k = m->ranks[r];
    (void)hipSetDevice(k.device);
    if (k.stream) (void)hipStreamSynchronize(k.stream);
    if (m->commsReady && k.comm) (void)ncclCommDestroy(k.comm);
    if (k.scene) rt_scene_destroy(k.scene);
    if (k.packed) (void)hipFree(k.packed);
    if (r == 0) {
      for (unsigned char* b : m->landing)
        if (b) (void)hipFree(b);
      if (m->frame) (void)hipFree(m->frame);
    }
    if (k.stream) (void)hipStreamDestroy(k.stream);
  }
  (void)hipSetDevice(cur);
  delete m;
}

int rt_multi_create(const RtSceneDesc* desc, int n_devices, const int* device_ids, RtMulti** out) {
  if (!desc || !out || n_devices < 1) return failMulti(RT_ERR_INVALID, "bad argument");
  *out = nullptr;
  int visible = 0;
  MULTI_HIP(hipGetDeviceCount(&visible));
  std::vector<int> devs((size_t)n_devices);
  for (int r = 0; r < n_devices; r++) {
    devs[(size_t)r] = device_ids ? device_ids[r] : r;
    if (devs[(size_t)r] < 0 || devs[(size_t)r] >= visible) return failMulti(RT_ERR_INVALID, "device id out of range");
    for (int q = 0; q < r; q++)
      if (devs[(size_t)q] == devs[(size_t)r]) return failMulti(RT_ERR_INVALID, "a device is listed twice");
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  RtMulti* m = new RtMulti;
  m->ranks.resize((size_t)n_devices);
  m->landing.assign((size_t)n_devices, nullptr);
  m->landingBytes.assign((size_t)n_devices, 0);
  // The scene on every device: the uploads and BVH builds run side by side.
  std::vector<std::thread> th;
  for (int r = 0; r < n_devices; r++) {
    m->ranks[(size_t)r].device = devs[(size_t)r];
    th.emplace_back([m, r, desc]() {
      Rank& k = m->ranks[(size_t)r];
      if (hipSetDevice(k.device) != hipSuccess) { k.rc = RT_ERR_DEVICE; k.error = "hipSetDevice failed"; return; }
      if (hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess) {
        k.rc = RT_ERR_DEVICE; k.error = "hipStreamCreate failed"; return;
      }
      k.rc = rt_scene_create(desc, &k.scene);
      if (k.rc != RT_OK) k.error = rt_last_error();
    });
  }
  for (std::thread& t : th) t.join();
  for (const Rank& k : m->ranks)
    if (k.rc != RT_OK) {
      const int rc = k.rc;
      const std::string msg = "device " + std::to_string(k.device) + ": " + k.error;
      rt_multi_destroy(m);
      (void)hipSetDevice(cur);
      return failMulti(rc, msg);
    }
  if (n_devices > 1) {
    std::vector<ncclComm_t> comms((size_t)n_devices);
    const ncclResult_t nr = ncclCommInitAll(comms.data(), n_devices, devs.data());
    if (nr != ncclSuccess) {
      rt_multi_destroy(m);
      (void)hipSetDevice(cur);
      return failMulti(RT_ERR_DEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(nr));
    }
    for (int r = 0; r < n_devices; r++) m->ranks[(size_t)r].comm = comms[(size_t)r];
    m->commsReady = true;
  }
  (void)hipSetDevice(cur);
  *out = m;
  return RT_OK;
}

int rt_multi_render(RtMulti* m, const RtRenderParams* params, unsigned char* rgb8_out, RtStats* stats) {
  if (!m || !params || !rgb8_out) return failMulti(RT_ERR_INVALID, "null argument");
  if (params->width <= 0 || params->height <= 0) return failMulti(RT_ERR_INVALID, "bad frame size");
  if (params->aa_mode == RT_AA_ADAPTIVE)
    return failMulti(RT_ERR_UNSUPPORTED, "adaptive anti-aliasing needs the whole base frame on one device");
  const int n = (int)m->ranks.size();
  const int w = params->width, h = params->height;
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::vector<size_t> shardBytes((size_t)n);
  for (int r = 0; r < n; r++) {
    const long long px = rt_shard_pixels(w, h, kTile, kTile, r, n);
    if (px < 0) return failMulti(RT_ERR_INVALID, "bad frame geometry");
    shardBytes[(size_t)r] = (size_t)px * 3;
  }
  // landing buffers and the frame live on the first device
  const size_t frameBytes = (size_t)w * h * 3;
  MULTI_HIP(hipSetDevice(m->ranks[0].device));
  for (int q = 1; q < n; q++) {
    const int rc = ensureBytes(&m->landing[(size_t)q], &m->landingBytes[(size_t)q], std::max<size_t>(shardBytes[(size_t)q], 1));
    if (rc != RT_OK) { (void)hipSetDevice(cur); return rc; }
  }
  {
    const int rc = ensureBytes(&m->frame, &m->frameBytes, frameBytes);
    if (rc != RT_OK) { (void)hipSetDevice(cur); return rc; }
  }
  std::vector<std::thread> th;
  for (int r = 1; r < n; r++) th.emplace_back(renderShard, m, r, params, &shardBytes);
  renderShard(m, 0, params, &shardBytes);   // the calling thread takes the first device, like the reference's main thread its strip
  for (std::thread& t : th) t.join();
  for (const Rank& k : m->ranks)
    if (k.rc != RT_OK) {
      (void)hipSetDevice(cur);
      return failMulti(k.rc, "device " + std::to_string(k.device) + ": " + k.error);
    }
  {
    const int rc = exchangeShards(m, shardBytes);
    if (rc != RT_OK) { (void)hipSetDevice(cur); return rc; }
  }
  // packed tiles -> frame, on the first device, then the frame to the caller
  Rank& k0 = m->ranks[0];
  MULTI_HIP(hipSetDevice(k0.device));
  for (int q = 0; q < n; q++) {
    if (shardBytes[(size_t)q] == 0) continue;
    const unsigned char* src = q == 0 ? k0.packed : m->landing[(size_t)q];
    if (rt_unpack_tiles(src, 3, w, h, kTile, kTile, q, n, m->frame, (void*)k0.stream) != RT_OK) {
      (void)hipSetDevice(cur);
      return failMulti(RT_ERR_DEVICE, rt_last_error());
    }
  }
  MULTI_HIP(hipMemcpyAsync(rgb8_out, m->frame, frameBytes, hipMemcpyDeviceToHost, k0.stream));
  MULTI_HIP(hipStreamSynchronize(k0.stream));
  (void)hipSetDevice(cur);
  if (stats) {
    RtStats s;
    std::memset(&s, 0, sizeof s);
    for (const Rank& k : m->ranks) {
      s.rays_primary += k.stats.rays_primary;
      s.rays_reflect += k.stats.rays_reflect;
      s.rays_refract += k.stats.rays_refract;
      s.rays_shadow += k.stats.rays_shadow;
      s.nodes_visited += k.stats.nodes_visited;
      s.prims_tested += k.stats.prims_tested;
      s.shade_points += k.stats.shade_points;
      s.shadow_nodes_visited += k.stats.shadow_nodes_visited;
      s.shadow_prims_tested += k.stats.shadow_prims_tested;
      s.objects_tested += k.stats.objects_tested;
      s.shadow_objects_tested += k.stats.shadow_objects_tested;
      s.batches += k.stats.batches;
      s.batches_redone += k.stats.batches_redone;
      s.trace_launches += k.stats.trace_launches;
      s.lights_launches += k.stats.lights_launches;
      s.ms_total = std::max(s.ms_total, k.stats.ms_total);
      s.ms_device = std::max(s.ms_device, k.stats.ms_device);
      s.ms_trace = std::max(s.ms_trace, k.stats.ms_trace);
      s.ms_lights = std::max(s.ms_lights, k.stats.ms_lights);
    }
    *stats = s;
  }
  return RT_OK;
}

}  // extern "C"
