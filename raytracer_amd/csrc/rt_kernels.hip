// HIP kernels and C-ABI of the MI355X (gfx950) render path.
//
// The reference's recursive RayTracer::traceRay (src/RayTracer.cpp:132-226)
// runs here as an iterative wavefront over recursion levels.  For one batch of
// primary samples:
//
//   level L = 0 .. depth                              (main stream)
//     k_trace     closest hit of every ray of the level (Scene::intersect);
//                 level 0 generates its camera rays in place
//     k_hit       hits -> compacted shade-point slots with Q, N and the material
//                 word; their reflected rays, compacted, into the next level's
//                 queue; a secondary ray that hits nothing is struck from its
//                 parent (scenes of plain triangles in which nothing transmits)
//       or k_classify (the compaction), k_surface / k_surface_rest (the shade
//       point: triangles / other primitives and interpolated materials) and
//       k_shade (reflected and refracted rays) for every other scene
//     k_background  cube map for the rays that hit nothing, where one is shown
//     k_advance   hands the level's counts on to the next level, on the device
//                                                     (side stream, beside the
//                                                      main stream's next level)
//     k_shadow    one thread per (light, shade point): the shadow query
//     k_lights    Material::shade: Light::{distance,shadow}Attenuation from the
//                 query's outcome and the Phong terms, summed in light order
//   level L = depth .. 1
//     k_combine   I = shade + kr*I(reflected) + kt*I(refracted), i.e. the
//                 additions of traceRay in the reference's order, into the ray's slot
//   k_resolve     the same sum for the level-0 shade point of every sample, clamp,
//                 mean over sub-samples, truncate to RGB8
//
// Every ray of a batch owns one slot of a pool; a shade point stores the pool
// slots of its two children, so the bottom-up pass reproduces the recursion's
// floating-point evaluation order exactly and is independent of the order in
// which waves appended to the queues.  The host queues a whole batch -- every
// level, the bottom-up pass, the resolve -- and waits once (renderBatch).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <limits>
#include <array>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_bvh.h"
#include "rt_shade.h"

using namespace rtd;

// ============================================================== device side
namespace {

constexpr int kBlock = 256;
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 128
#endif
constexpr int kTraceBlock = RT_TRACE_BLOCK;   // threads per block of the traversal kernels

struct Counters {           // device-resident, one block per recursion level
  unsigned int nsp;         // shade points appended at this level
  unsigned int nnext;       // rays appended to the next level
  unsigned int overflow;    // RT_OVERFLOW_* seen at this level or one before it: the batch has to be redone
  // The level's own extent, written on the device by the level before it
  // (k_begin / k_advance): the kernels of a level read how many rays it has and
  // where its rays and shade points start in the pool from here, so the host
  // can queue a whole batch without waiting for any count.
  unsigned int n;           // rays of this level
  unsigned int rayOff;      // pool slot of its first ray
  unsigned int spOff;       // pool index of its first shade point
  unsigned int _pad[2];
  unsigned long long refl, refr, litsp, nodes, prims, objs;
  unsigned long long shadowNodes, shadowPrims, shadowObjs;   // of the shadow rays (k_shadow)
};

// One counter block per recursion level (and one past the last, which the last
// level's k_advance writes), zeroed together when a batch starts.
constexpr int kCounterBlocks = 128;
constexpr int kMaxDepth = kCounterBlocks - 2;

struct Pool {
  size_t cap;                                    // slots
  // level-transient, double buffered by level parity
  // (a secondary ray starts at the shade point it leaves, RayTracer.cpp:165,200:
  // the queue holds that point's pool index instead of a copy of its position)
  int* rayParent[2]; double* rayD[2]; int* rayType[2];
  double* hitT; int* hitPrim;
  // persistent for the whole batch (pool slot / pool shade-point index)
  double* spQ; double* spN;
  double* rayI;
  double* spShade;
  // shadow queries of the current level ([n_lights][cap], indexed 0..nsp): written by
  // k_shadow, read by k_lights
  int* shPrim;                                   // outcome of the query per light
  double* shT;                                   // its hit parameter; NULL when the scene is opaque
  int* spMat;                                    // RT_SPM_* word: material index, lit flag
  int* spChildRefl; int* spChildRefr; int* spRayGlobal;
  // only for shade points whose material is interpolated from vertex materials
  // (RT_SPM_INTERP); one-element stubs when the scene has none
  double* spKd; double* spKs; double* spShin;
  double* spKr; double* spKt; double* spIor;
  int* primaryId;                                // objectID of the level-0 hit per sample
  int* primarySp;                                // pool index of the level-0 shade point per sample, -1: no hit
  Counters* counters;
};

// Shade-point material word: index into DevScene::devMats, or RT_SPM_INTERP when
// the material was interpolated and lives in the per-point arrays.
#define RT_PRIM_ANY (-2)                         // shadow query: some occluder in front of the light
#define RT_PRIM_OUTSIDE (-3)                     // shadow query not run: the point lies outside the spot light's cone
#define RT_SPM_PENDING 0xffffffffu               // k_surface left the point to k_surface_rest
#define RT_SPM_INTERP 0x80000000u
#define RT_SPM_LIT 0x40000000u                   // kd or ks non-zero: lights contribute
#define RT_SPM_INDEX 0x3fffffffu

// The grid of every per-level kernel is sized on the host from an ESTIMATE of
// the level's item count (the exact count lives on the device).  A grid that
// turns out too small leaves items undone: its first thread says so in the
// level's counter block, the levels below are skipped (k_advance) and the host
// redoes the batch with exact sizes.
#define RT_OVERFLOW_POOL 1u
#define RT_OVERFLOW_GRID 2u
__device__ __forceinline__ void checkCoverage(Counters* c, long long covered, long long items) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && covered < items) c->overflow = RT_OVERFLOW_GRID;
}

// Rays of the level whose closest hit k_trace has written: all of them, unless
// its grid (`traced` threads, sized from an estimate) fell short of the level.
// The kernels that read hitPrim / hitT stop there: what lies beyond was never
// written for this level (the batch is redone anyway, k_trace has flagged it).
__device__ __forceinline__ long long tracedRays(const Counters* c, long long traced) {
  const long long n = c->n;
  return n < traced ? n : traced;
}

// Division of a 31-bit unsigned number by a divisor that is fixed for a launch
// (tile size, tiles per row, samples per pixel): one multiply-high and a shift
// instead of the few dozen instructions of a run-time division -- the sample ->
// pixel mapping below runs once per primary ray in four kernels.
// Granlund & Montgomery: with l = ceil(log2 d) and m = floor(2^(31+l) / d) + 1,
// floor(n / d) = (m n) >> (31 + l) for every 0 <= n < 2^31.
struct FastDiv {
  unsigned int mul, shift, d;
};
inline FastDiv makeFastDiv(unsigned int d) {
  FastDiv f{0u, 0u, d};
  if (d <= 1u) return f;
  unsigned int l = 0;
  while ((1ull << l) < d) l++;
  f.mul = (unsigned int)(((1ull << (31 + l)) / d) + 1ull);
  f.shift = l - 1u;        // the multiply-high has already dropped 32 bits
  return f;
}
__device__ __forceinline__ unsigned int fastDiv(unsigned int n, const FastDiv& f) {
  return f.d <= 1u ? n : (__umulhi(n, f.mul) >> f.shift);
}

struct FrameMap {           // pixel slot -> pixel, for tile-ordered rendering
  int width, height;
  int tileW, tileH, tilesX;
  int shardIndex, shardCount;
  const int* pixelList;     // optional explicit (x,y) list (adaptive AA), else NULL
  int listOutput;           // with pixelList: 1 = the output is indexed by list position, 0 = by frame position
  int listFirst, listStride;   // pixel slot k is entry listFirst + k * listStride of the list (a share of it)
  int ns;                   // samples per pixel
  int aaN;                  // n (ns = n*n) or 0
  const double* jitter;     // device copy, 2*ns values or NULL
  // x = i / width for every column i, y = j / height for every row j
  // (RayTracer.cpp:51-52), divided once per frame size on the host -- IEEE
  // division, the same bits -- instead of twice per primary ray
  const double* colX;
  const double* rowY;
  FastDiv dPer, dTilesX, dBlocksPerRow, dNs, dAaN;   // by tileW*tileH, tilesX, tileW/8, ns, aaN (setDivisors)
  // sub-sample pitch of tracePixelAntiAlias (RayTracer.cpp:84-85), the same for
  // every sample: divided once on the host (IEEE division, the same bits)
  double deltaX, deltaY;
};

// Fills the divisors from the sizes above (after every change of those).
inline void setDivisors(FrameMap& fm) {
  fm.dPer = makeFastDiv((unsigned int)(fm.tileW * fm.tileH));
  fm.dTilesX = makeFastDiv((unsigned int)fm.tilesX);
  fm.dBlocksPerRow = makeFastDiv((unsigned int)(fm.tileW >> 3));
  fm.dNs = makeFastDiv((unsigned int)fm.ns);
  fm.dAaN = makeFastDiv((unsigned int)std::max(1, fm.aaN));
  const int n = std::max(1, fm.aaN);
  fm.deltaX = (1.0 / double(fm.width)) / n;
  fm.deltaY = (1.0 / double(fm.height)) / n;
}

// Where a batch (or a launch) starts in the slot order, split on the host so
// that the device only divides 32-bit numbers: slot = tile * (tileW*tileH) + rem.
struct SlotBase {
  long long slot;           // first pixel slot
  int tile;                 // slot / (tileW*tileH)
  unsigned int rem;         // slot % (tileW*tileH)
};
inline SlotBase makeSlotBase(const FrameMap& fm, long long slot) {
  const long long per = (long long)fm.tileW * fm.tileH;
  return SlotBase{slot, (int)(slot / per), (unsigned int)(slot % per)};
}

// Pixel slot -> (x, y).  Slots enumerate this shard's tiles in increasing tile
// number; inside a tile, 8x8 blocks row-major, pixels row-major in the block,
// so one wavefront works on one 8x8 block.  Slots outside the frame are inactive.
__device__ __forceinline__ bool slotToPixel(const FrameMap& fm, const SlotBase& base, unsigned int rel, int& x, int& y,
                                            int& tileK, int& lx, int& ly) {
  if (fm.pixelList) {
    const long long slot = fm.listFirst + (base.slot + rel) * (long long)fm.listStride;
    x = fm.pixelList[2 * slot];
    y = fm.pixelList[2 * slot + 1];
    tileK = 0; lx = ly = 0;
    return true;
  }
  const unsigned int u = base.rem + rel;                  // < 2^31: a batch has fewer than 2^30 samples
  const unsigned int dk = fastDiv(u, fm.dPer);
  const int rem = (int)(u - dk * fm.dPer.d);
  tileK = base.tile + (int)dk;
  const unsigned int t = (unsigned int)(fm.shardIndex + tileK * fm.shardCount);
  const unsigned int ty = fastDiv(t, fm.dTilesX);
  const int tx = (int)(t - ty * fm.dTilesX.d);
  const unsigned int blk = (unsigned int)(rem >> 6);
  const int in = rem & 63;
  const unsigned int by = fastDiv(blk, fm.dBlocksPerRow);
  const int bx = (int)(blk - by * fm.dBlocksPerRow.d);
  lx = bx * 8 + (in & 7);
  ly = (int)by * 8 + (in >> 3);
  x = tx * fm.tileW + lx;
  y = (int)ty * fm.tileH + ly;
  return x < fm.width && y < fm.height;
}

// Where the rays of a level come from: the queue written by k_shade, or, for
// level 0, the camera -- primary rays are generated where they are needed
// instead of being written to and read from memory (every kernel that asks gets
// the same bits, the function is deterministic).
struct RaySrc {
  const int* parent;      // pool index of the shade point a queued ray starts at
  const double* spQ;      // shade-point positions (Pool::spQ)
  const double* d;
  const int* type;        // NULL: every ray of the queue is a reflected one (k_hit's queues)
  size_t stride;
  int primary;            // 1: level 0
  FrameMap fm;
  SlotBase first;         // pixel slot of sample 0 of the batch
};

// A ray of a level; type < 0: no ray in this slot.  (Handed around by value:
// with output references the compiler merges the stores of the two sources of
// a ray behind a selected address, and the ray then lives in scratch memory.)
struct RayPD {
  V3 p, d;
  int type;
};

// Camera::rayThrough (camera.cpp:20-31) at the coordinates of tracePixel /
// tracePixelAntiAlias (RayTracer.cpp:51-52,84-92) for sample i of the batch;
// type -1 for a sample of a pixel slot outside the frame.
__device__ __forceinline__ RayPD primaryRay(const DevScene& S, const FrameMap& fm, const SlotBase& first, int i) {
  RayPD r;
  r.type = -1;
  const unsigned int rel = fastDiv((unsigned int)i, fm.dNs);
  const int k = (int)((unsigned int)i - rel * fm.dNs.d);
  int px, py, tk, lx, ly;
  if (!slotToPixel(fm, first, rel, px, py, tk, lx, ly)) return r;
  double x = fm.colX[px];
  double y = fm.rowY[py];
  if (fm.aaN > 0) {
    const int n = fm.aaN;
    const double deltaX = fm.deltaX, deltaY = fm.deltaY;
    const int sc = (int)fastDiv((unsigned int)k, fm.dAaN), sr = k - sc * n;   // column outer, row inner (RayTracer.cpp:87-89)
    const double jx = fm.jitter ? fm.jitter[2 * k] : 0.0, jy = fm.jitter ? fm.jitter[2 * k + 1] : 0.0;
    x = x + sc * deltaX + jx * deltaX;
    y = y + sr * deltaY + jy * deltaY;
  }
  x -= 0.5;
  y -= 0.5;
  r.d = normalized(V3(S.look) + x * V3(S.u) + y * V3(S.v));
  r.p = V3(S.eye);
  r.type = RT_RAY_VISIBILITY;
  return r;
}

// A secondary ray that hits nothing where the background is black returns
// (0, 0, 0) (RayTracer.cpp:213-224), and kr (*) 0 or kt (*) 0 adds nothing to
// its parent (finite coefficients: checked when the scene is created).  Instead
// of storing that colour for k_combine to read, the ray is struck from its
// parent: 4 bytes written instead of 24, nothing read later.
__device__ __forceinline__ void dropChild(const RaySrc& R, int* spChildRefl, int* spChildRefr, long long i) {
  const int g = R.parent[i];
  if (R.type[i] == RT_RAY_REFRACTION) spChildRefr[g] = -1;
  else spChildRefl[g] = -1;
}

// Sample i of a batch is the first of its pixel: the one whose hit id the frame
// shows (k_resolve), the only one whose id is stored.
__device__ __forceinline__ bool firstSampleOfPixel(const FrameMap& fm, long long i) {
  return (unsigned int)i == fastDiv((unsigned int)i, fm.dNs) * fm.dNs.d;
}

// Ray i of the level.
__device__ __forceinline__ RayPD rayAt(const DevScene& S, const RaySrc& R, int i) {
  if (R.primary) return primaryRay(S, R.fm, R.first, i);
  RayPD r;
  r.type = R.type ? R.type[i] : RT_RAY_REFLECTION;
  if (r.type >= 0) {
    r.p = loadV3(R.spQ, R.stride, (size_t)R.parent[i]);
    r.d = loadV3(R.d, R.stride, i);
  }
  return r;
}
__device__ __forceinline__ int rayOf(const DevScene& S, const RaySrc& R, int i, V3& p, V3& d) {
  const RayPD r = rayAt(S, R, i);
  p = r.p;
  d = r.d;
  return r.type;
}

__device__ __forceinline__ void flushCounters(const TraceCounters& c, Counters* g, bool shadow) {
  // wave-level sum, one atomic per counter and wavefront
  unsigned int n = c.nodes, t = c.prims, o = c.objs;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n += __shfl_down(n, off);
    t += __shfl_down(t, off);
    o += __shfl_down(o, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(shadow ? &g->shadowNodes : &g->nodes, (unsigned long long)n);
    atomicAdd(shadow ? &g->shadowPrims : &g->prims, (unsigned long long)t);
    if (o) atomicAdd(shadow ? &g->shadowObjs : &g->objs, (unsigned long long)o);
  }
}

// Resident wavefronts per SIMD the triangle-only BVH traversal kernels are
// compiled for.  Five is what the closest-hit kernel gets by itself (87
// registers); the shadow kernel lands one register past that step (97) and
// gains 12 % from being held to it.  (Six: both kernels spill and lose more
// than the extra wavefront hides.)  The BVH kernels with every primitive kind
// are held to four (125-129 registers by themselves, on either side of that
// step); the reference's KD traversal keeps what the compiler gives it.
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 5
#endif
#define RT_TRACE_BOUNDS(FAST, TRI) __launch_bounds__(kTraceBlock, (FAST) ? ((TRI) ? RT_TRACE_WAVES : 4) : 1)

// Closest hit of every ray of a level, one ray per thread.  (A streaming form in
// which a wavefront refills finished lanes from a run of rays was measured 1.6x
// slower on coherent rays: refilled lanes restart at the root and put the
// wavefront's lanes out of phase between node and leaf work.)
// Workgroups are dealt to the 8 XCDs (each with its own L2) round-robin by
// their index.  Consecutive rays look at neighbouring parts of the tree, so the
// traversal kernels renumber their blocks in chunks: the kXcdChunk blocks that
// XCD x receives in a row work on kXcdChunk consecutive pieces of the queue.
// (Giving each XCD one contiguous eighth of the queue was 1.5x slower: sky and
// terrain parts of a frame cost very different amounts.)
#ifndef RT_XCD_CHUNK
#define RT_XCD_CHUNK 32
#endif
__device__ __forceinline__ long long xcdBlock() {
  constexpr unsigned int C = RT_XCD_CHUNK;
  if (C <= 1) return blockIdx.x;
  const unsigned int nb = gridDim.x, b = blockIdx.x, span = 8u * C;
  const unsigned int full = nb / span * span;
  if (b >= full) return b;
  const unsigned int g = b / span, r = b % span;     // group of 8*C blocks, position in it
  return (long long)g * span + (r & 7u) * C + (r >> 3);
}

// Dynamic LDS of the kernels that walk the BVH: the per-lane traversal stacks
// (RT_LDS_STACK_BYTES_* per thread, set at launch).
extern __shared__ __align__(16) uint32_t rtStackLds[];

template <bool FAST, bool STATS, bool TRI>
__global__ void RT_TRACE_BOUNDS(FAST, TRI)
k_trace(DevScene S, RaySrc R, double* hitT, int* hitPrim, Counters* counters) {
  const long long n = counters->n;
  checkCoverage(counters, (long long)gridDim.x * blockDim.x, n);
  const long long i = xcdBlock() * blockDim.x + threadIdx.x;
  TraceCounters tc{0, 0, 0};
  if (i < n) {
    V3 p, d;
    const int type = rayOf(S, R, (int)i, p, d);
    int prim = RT_PRIM_NONE;
    double t = 0.0;
    if (type >= 0) {
      if (FAST) prim = traceBvh<false, STATS, TRI>(S, p, d, type, 1.0e308, -1.0, t, tc, (rt_lds_u32*)rtStackLds);
      else prim = traceKdReference<STATS>(S, p, d, type, t, tc, (rt_kd_lds_u32*)rtStackLds);
    }
    hitT[i] = t;
    hitPrim[i] = prim;
  }
  if (STATS) flushCounters(tc, counters, false);
}

// Sets up level 0 of a batch / hands the counts of level l on to level l+1.
// One thread: the next level's kernels are ordered behind it on the stream.
// (k_begin also clears the counter blocks of the batch's levels: one launch
// instead of a fill and a launch)
__global__ void k_begin(Counters* c, unsigned int n, int blocks) {
  unsigned int* w = reinterpret_cast<unsigned int*>(c);
  const int words = blocks * (int)(sizeof(Counters) / sizeof(unsigned int));
  for (int k = threadIdx.x; k < words; k += blockDim.x) w[k] = (k == (int)(offsetof(Counters, n) / sizeof(unsigned int))) ? n : 0u;
}
__global__ void k_advance(Counters* c, unsigned int cap) {
  Counters* next = c + 1;
  const unsigned int rayOff = c->rayOff + c->n, spOff = c->spOff + c->nsp;
  // an exhausted pool ends the batch here: nothing of the levels below is touched
  const bool noRoom = rayOff > cap || spOff > cap || c->nnext > cap - rayOff;
  const unsigned int flags = c->overflow | (noRoom ? RT_OVERFLOW_POOL : 0u);
  const bool dead = flags != 0;
  next->overflow = flags;
  next->n = dead ? 0u : c->nnext;
  next->rayOff = dead ? 0u : rayOff;
  next->spOff = dead ? 0u : spOff;
}

// ---- queue compaction without a hot atomic ----------------------------------
// A block owns a tile of kBlock * items consecutive queue entries (thread t
// handles entries tileBase + k*kBlock + t, k < items <= kMaxItems).  Pass 1
// counts the entries that want a slot per (k, wave), one thread turns the counts
// into exclusive offsets and reserves the tile's total with ONE atomic, pass 2
// hands out the slots.  A single device-wide counter saturates near 90 M
// atomics/s on this chip, so one atomic per wavefront (what a ballot-append
// does) would cost tens of milliseconds per 4K frame.
constexpr int kMaxItems = 16;
constexpr int kWaves = kBlock / 64;

struct TileScan {
  unsigned int cnt[2][kMaxItems][kWaves];   // [class][item][wave]
  unsigned int base[2];                     // global offset of the tile per class
};

// Pass 1 for one item: record how many lanes of this wave raise flag(s).
__device__ __forceinline__ void tileCount(TileScan& ts, int cls, int k, bool flag) {
  const unsigned long long m = __ballot(flag);
  if ((threadIdx.x & 63) == 0) ts.cnt[cls][k][threadIdx.x >> 6] = (unsigned int)__popcll(m);
}

// Between the passes: exclusive scan of the counts (class 0 first, then class
// 1), one atomicAdd of the total on *counter.  Returns the totals per class (in
// thread 0 only).  The first wavefront scans: lane l owns the counter of item
// l / kWaves, wave l % kWaves (kMaxItems * kWaves = 64 counters per class).
__device__ __forceinline__ void tileReserve(TileScan& ts, int items, int classes, unsigned int* counter,
                                            unsigned int& total0, unsigned int& total1) {
  static_assert(kMaxItems * kWaves == 64, "one lane per counter");
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int k = lane / kWaves, w = lane % kWaves;
    unsigned int run = 0, tot[2] = {0, 0};
    for (int c = 0; c < classes; c++) {
      const unsigned int v = (k < items) ? ts.cnt[c][k][w] : 0u;
      unsigned int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (k < items) ts.cnt[c][k][w] = run + incl - v;
      tot[c] = __shfl(incl, 63);
      run += tot[c];
    }
    if (lane == 0) {
      const unsigned int g = run ? atomicAdd(counter, run) : 0;
      ts.base[0] = g;
      ts.base[1] = g;
      total0 = tot[0];
      total1 = tot[1];
    }
  }
  __syncthreads();
}

// Pass 2 for one item: slot of this lane if it raises the flag.
__device__ __forceinline__ unsigned int tileSlot(const TileScan& ts, int cls, int k, bool flag) {
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63;
  return ts.base[cls] + ts.cnt[cls][k][threadIdx.x >> 6] + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
}

// Misses take the black background (RayTracer.cpp:213-224; a cube map is
// added by k_background); hits get a shade-point slot (tile compaction) and are
// handed to the surface kernels as RT_SPM_PENDING.
__global__ void __launch_bounds__(kBlock)
k_classify(int items, int level, RaySrc R, int blackMiss, long long traced, Pool P) {
  __shared__ TileScan ts;
  const long long n = tracedRays(P.counters, traced);
  const size_t rayOff = P.counters->rayOff, spOff = P.counters->spOff;
  const long long perTile = (long long)kBlock * items;
  checkCoverage(P.counters, gridDim.x * perTile, n);
  const long long tileBase = blockIdx.x * perTile;
  {
    unsigned int hitMask = 0;
    for (int k = 0; k < items; k++) {
      const long long i = tileBase + k * kBlock + threadIdx.x;
      const bool hit = i < n && P.hitPrim[i] != RT_PRIM_NONE;   // k_trace leaves none in an empty slot
      hitMask |= (hit ? 1u : 0u) << k;
      tileCount(ts, 0, k, hit);
    }
    unsigned int tot0 = 0, tot1 = 0;
    tileReserve(ts, items, 1, &P.counters->nsp, tot0, tot1);
    for (int k = 0; k < items; k++) {
      const long long i = tileBase + k * kBlock + threadIdx.x;
      const bool isHit = (hitMask >> k) & 1u;
      const unsigned int s = tileSlot(ts, 0, k, isHit);
      if (i >= n) continue;
      if (level == 0) P.primarySp[i] = (isHit && spOff + s < P.cap) ? (int)(spOff + s) : -1;
      if (!isHit) {
        if (blackMiss) {
          // level 0: k_resolve paints a sample without a shade point black
          if (level > 0 && R.type[i] >= 0) dropChild(R, P.spChildRefl, P.spChildRefr, i);
        } else {
          storeV3(P.rayI, P.cap, rayOff + i, V3(0.0, 0.0, 0.0));
        }
        if (level == 0 && firstSampleOfPixel(R.fm, i)) P.primaryId[i] = 0;
      } else if (spOff + s >= P.cap) {
        P.counters->overflow = RT_OVERFLOW_POOL;
      } else {
        P.spRayGlobal[spOff + s] = (int)(rayOff + i);
        P.spMat[spOff + s] = (int)RT_SPM_PENDING;
      }
    }
  }
}

// One thread per shade point: the common case -- a triangle whose material is
// one of the scene's materials -- is finished here.  Hits on non-mesh objects
// and on meshes that interpolate per-vertex materials stay pending for
// k_surface_rest: the code of those cases more than doubles the registers of
// this kernel, which waits on gathers and needs its wavefronts.
// Shade points of the level that exist AND lie inside the pool: an exhausted
// pool leaves the count above what was stored (k_classify), and every kernel
// that walks the level's shade points stops at the pool's end.
__device__ __forceinline__ long long shadePointsOf(const Pool& P) {
  const long long nsp = P.counters->nsp, room = (long long)P.cap - (long long)P.counters->spOff;
  return nsp < room ? nsp : room;
}

__global__ void __launch_bounds__(kBlock)
k_surface(DevScene S, RaySrc R, int level, Pool P) {
  const long long nsp = shadePointsOf(P);
  checkCoverage(P.counters, (long long)gridDim.x * blockDim.x, nsp);
  const size_t rayOff = P.counters->rayOff, spOff = P.counters->spOff;
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nsp) {
    const size_t g = spOff + s;
    const int i = (int)((size_t)P.spRayGlobal[g] - rayOff);
    const int prim = P.hitPrim[i];
    if (!(prim & RT_PRIM_FACE)) return;
    const int fi = prim & ~RT_PRIM_FACE;
    const int info = S.faceInfo[fi];
    if (S.smooth && (info & RT_FI_VMATERIALS)) return;
    V3 p, d;
    rayOf(S, R, i, p, d);
    Surface sf;
    surfaceNormalOfFace(S, fi, info, p, d, sf);
    P.spMat[g] = (int)((unsigned int)(info & RT_FI_MATERIAL) | ((info & RT_FI_LIT) ? RT_SPM_LIT : 0u));   // material.cpp:50-53
    if (level == 0 && firstSampleOfPixel(R.fm, i)) P.primaryId[i] = sf.objectID;
    const V3 Q = p + (P.hitT[i] * d);                                    // ray::at (ray.h:42-43)
    storeV3(P.spQ, P.cap, g, Q);
    storeV3(P.spN, P.cap, g, sf.N);
  }
}

// The shade points k_surface left pending: hits on spheres, boxes, squares,
// cylinders and cones, and hits whose material is interpolated from per-vertex
// materials.  One thread per shade point of the level.
__global__ void __launch_bounds__(kBlock)
k_surface_rest(DevScene S, RaySrc R, int level, Pool P) {
  const long long nsp = shadePointsOf(P);
  checkCoverage(P.counters, (long long)gridDim.x * blockDim.x, nsp);
  const size_t rayOff = P.counters->rayOff, spOff = P.counters->spOff;
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nsp) {
    const size_t g = spOff + s;
    if ((unsigned int)P.spMat[g] == RT_SPM_PENDING) {
      const int i = (int)((size_t)P.spRayGlobal[g] - rayOff);
      const int prim = P.hitPrim[i];
      V3 p, d;
      const int type = rayOf(S, R, i, p, d);
      const double t = P.hitT[i];
      Surface sf;
      bool lit;
      const bool interp = (prim & RT_PRIM_FACE) != 0;   // pending faces are the interpolated ones
      if (!interp) {
        const int tableMat = S.objects[prim].material;
        surfaceAt(S, prim, p, d, type, sf, nullptr, false);
        lit = (S.devMats[tableMat].flags & (RT_MATF_KD | RT_MATF_KS)) != 0;   // material.cpp:50-53
        P.spMat[g] = (int)((unsigned int)tableMat | (lit ? RT_SPM_LIT : 0u));
      } else {
        MatD m;
        surfaceAt(S, prim, p, d, type, sf, &m, false);
        lit = !(iszero(m.kd) && iszero(m.ks));
        P.spMat[g] = (int)(RT_SPM_INTERP | (lit ? RT_SPM_LIT : 0u));
        storeV3(P.spKd, P.cap, g, m.kd);
        storeV3(P.spKs, P.cap, g, m.ks);
        P.spShin[g] = shininessOf(m);
        storeV3(P.spShade, P.cap, g, m.ke + prod(m.ka, V3(S.ambient)));   // material.cpp:46
        storeV3(P.spKr, P.cap, g, m.kr);
        storeV3(P.spKt, P.cap, g, m.kt);
        P.spIor[g] = intensityValue(m.index);                              // Material::index (material.h:217)
      }
      if (level == 0 && firstSampleOfPixel(R.fm, i)) P.primaryId[i] = sf.objectID;
      const V3 Q = p + (t * d);                                            // ray::at (ray.h:42-43)
      storeV3(P.spQ, P.cap, g, Q);
      storeV3(P.spN, P.cap, g, sf.N);
    }
  }
}

// Cube-map background of the rays that hit nothing (RayTracer.cpp:213-224).
__global__ void __launch_bounds__(kBlock)
k_background(DevScene S, RaySrc R, int filterWidth, long long traced, Pool P) {
  const long long n = tracedRays(P.counters, traced);
  checkCoverage(P.counters, (long long)gridDim.x * blockDim.x, n);
  const size_t rayOff = P.counters->rayOff;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || P.hitPrim[i] != RT_PRIM_NONE) return;
  V3 p, d;
  if (rayOf(S, R, (int)i, p, d) >= 0) storeV3(P.rayI, P.cap, rayOff + i, cubeColor(S, d, filterWidth));
}

// Shadow query of one light from shade point Q (light.cpp:14-31,66-98,112-149):
// the closest occluder `prim` at parameter `th`, or, when the scene is opaque
// and the traversal may stop at any occluder in front of the light, just
// whether there is one (`anyOccluder`).
template <bool FAST, bool STATS, bool TRI>
__device__ __forceinline__ void shadowQuery(const DevScene& S, const RtLight& l, const double* cone, const V3& Q, int& prim,
                                            double& th, bool& anyOccluder, TraceCounters& tc) {
  anyOccluder = false;
  if (l.type == RT_LIGHT_DIRECTIONAL) {
    const V3 sd = normalized(-V3(l.orientation));
    // any hit blacks a directional light out (light.cpp:21-24)
    if (FAST) prim = traceBvh<true, STATS, TRI>(S, Q, sd, RT_RAY_SHADOW, 1.0e308, -1.0, th, tc, (rt_lds_u32*)rtStackLds);
    else prim = traceKdReference<STATS>(S, Q, sd, RT_RAY_SHADOW, th, tc, (rt_kd_lds_u32*)rtStackLds);
    return;
  }
  const V3 position(l.position);
  const V3 sd = normalized(position - Q);
  if (FAST && l.type == RT_LIGHT_SPOT) {
    // A point outside the cone gets (0, 0, 0) from SpotLight::shadowAttenuation
    // whether its shadow ray hits something or not (light.cpp:127-133 and
    // 141-145 test the same angle on both sides of the occlusion test): the
    // query's outcome cannot reach the colour, so it is not run.  (The ray
    // still counts as cast: shadow rays are counted as lit points x lights.)
    // The same decision as lightIntensityFrom's, taken only where it needs no
    // acos; in the hair around the cone's edge the query runs.
    const double x = dot(V3(l.orientation), sd * -1);
    if (x < cone[0] && x >= -1.0) {
      prim = RT_PRIM_OUTSIDE;
      th = 0.0;
      return;
    }
  }
  const double lightDistance = length2(position - Q);
  // FAST: an occluder beyond the light is irrelevant; bound the search a little
  // past the light so that the strict < of the caller decides
  // (only ever a cull bound: a binary32 square root, widened past its rounding, will do)
  const double limit = (double)(sqrtf((float)lightDistance) * 1.000001f) + 1e-7;
  if (FAST && S.opaque) {
    // Every material has kt == 0, so the filter colour of ANY occluder in
    // front of the light is the closest occluder's: stop at the first one.
    anyOccluder = traceBvh<true, STATS, TRI>(S, Q, sd, RT_RAY_SHADOW, limit, lightDistance, th, tc,
                                             (rt_lds_u32*)rtStackLds) != RT_PRIM_NONE;
    prim = RT_PRIM_NONE;
  } else if (FAST) {
    prim = traceBvh<false, STATS, TRI>(S, Q, sd, RT_RAY_SHADOW, limit, -1.0, th, tc, (rt_lds_u32*)rtStackLds);
  } else {
    prim = traceKdReference<STATS>(S, Q, sd, RT_RAY_SHADOW, th, tc, (rt_kd_lds_u32*)rtStackLds);
  }
}

// distanceAttenuation * shadowAttenuation of one light (material.cpp:59-60,
// light.cpp:14-149) from the outcome of its shadow query, and the direction L
// to the light as Material::shade normalises it.
template <bool INTPOW>
__device__ __forceinline__ V3 lightIntensityFrom(const DevScene& S, const RtLight& l, const double* cone, const V3& Q,
                                                 int prim, double th, bool anyOccluder, V3& L) {
  const V3 color(l.color);
  V3 atten;
  if (l.type == RT_LIGHT_DIRECTIONAL) {
    L = normalized(-V3(l.orientation));
    atten = (prim != RT_PRIM_NONE) ? V3(0, 0, 0) : color;
  } else {
    const V3 position(l.position);
    V3 sd = normalized(position - Q);
    L = normalized(sd);  // getDirection normalises, shade normalises again
    const double lightDistance = length2(position - Q);
    bool occluded = anyOccluder;
    V3 kt;
    if (prim != RT_PRIM_NONE) {
      const V3 Qh = Q + (th * sd);
      const double distanceSq = length2(Qh - Q);
      if (distanceSq < lightDistance) {
        occluded = true;
        if ((prim & RT_PRIM_FACE) && S.smooth && S.vmaterials &&
            S.meshes[S.objects[S.tris[prim & ~RT_PRIM_FACE].owner].mesh].has_vertex_materials) {
          Surface sf;
          MatD m;
          surfaceAt(S, prim, Q, sd, RT_RAY_SHADOW, sf, &m, true);
          kt = m.kt;
        } else {
          const int oi = (prim & RT_PRIM_FACE) ? S.tris[prim & ~RT_PRIM_FACE].owner : prim;
          kt = V3(S.materials[S.objects[oi].material].kt);
        }
      }
    }
    if (l.type == RT_LIGHT_POINT) {
      atten = occluded ? prod(kt, color) : color;
    } else {
      sd = sd * -1;
      const V3 orientation(l.orientation);
      const double x = dot(orientation, sd);
      // acos is decreasing: x below cone[0] is outside the cone, x above cone[1]
      // (and at most 1: acos of more is NaN, which compares false) inside it; a
      // NaN x fails both tests and takes the reference's own expression
      bool outside;
      if (x < cone[0] && x >= -1.0) outside = true;
      else if (x > cone[1] && x <= 1.0) outside = false;
      else outside = acos(x) * 180 / M_PI > l.atten_angle;
      if (outside) {
        atten = V3(0.0, 0.0, 0.0);
      } else {
        const double fallFactor = INTPOW ? powNearInt(x, (int)cone[2], cone[3]) : pow(x, l.fall_rate);
        atten = occluded ? prod((color * fallFactor), kt) : (color * fallFactor);
      }
    }
  }
  return distanceAttenuation(l, Q) * atten;
}

// One thread per (light, shade point): the shadow query, nothing else, so that
// the kernel's register budget is the traversal's (the Phong terms need pow and
// acos, whose double-precision library code would nearly double it and halve
// the wavefronts that hide the traversal's memory latency).  (Measured: a thread
// that reads its point once and asks 2 .. 16 lights in a loop -- 64 lights:
// 10.2 -> 9.9 ms at 8 per thread; 2 lights: 20.4 -> 20.6 ms.  Not kept.)
template <bool FAST, bool STATS, bool TRI>
__global__ void RT_TRACE_BOUNDS(FAST, TRI)
k_shadow(DevScene S, Pool P) {
  // grid: x over the shade points of the level, y = light
  const long long nsp = shadePointsOf(P);
  checkCoverage(P.counters, (long long)gridDim.x * blockDim.x, nsp);
  const size_t spOff = P.counters->spOff;
  const long long s = xcdBlock() * blockDim.x + threadIdx.x;
  TraceCounters tc{0, 0, 0};
  if (s < nsp) {
    const int li = (int)blockIdx.y;
    // (Measured: asking for the position together with the material word instead of
    // behind the test of the word: no change with two lights, 4 % slower with 64.)
    if ((unsigned int)P.spMat[spOff + s] & RT_SPM_LIT) {
      const V3 Q = loadV3(P.spQ, P.cap, spOff + s);
      int prim;
      double th;
      bool anyOccluder;
      const RtLight& l = S.lights[li];
      shadowQuery<FAST, STATS, TRI>(S, l, S.spotCone + 4 * li, Q, prim, th, anyOccluder, tc);
      if (!P.shT && !anyOccluder && prim != RT_PRIM_NONE && prim != RT_PRIM_OUTSIDE && l.type != RT_LIGHT_DIRECTIONAL) {
        // The pool of an opaque scene keeps no hit parameter, so the closest
        // occluder is held against the light here, with the expressions of
        // Point/SpotLight::shadowAttenuation (light.cpp:74-80,120-126): it
        // counts only when it lies in front of the light.
        const V3 position(l.position);
        const V3 sd = normalized(position - Q);
        const V3 Qh = Q + (th * sd);
        anyOccluder = length2(Qh - Q) < length2(position - Q);
        prim = RT_PRIM_NONE;
      }
      const size_t o = (size_t)li * P.cap + s;
      P.shPrim[o] = anyOccluder ? RT_PRIM_ANY : prim;
      if (P.shT) P.shT[o] = th;
    }
  }
  if (STATS) flushCounters(tc, P.counters, true);
}

// Material::shade of shade point s (material.cpp:41-75): the emissive + ambient
// base, then per light in scene order the attenuation from its shadow query and
// the diffuse and specular terms, added in the reference's order.
template <bool INTPOW>
__device__ __forceinline__ void shadePoint(const DevScene& S, const Pool& P, int s, size_t g, unsigned int mw,
                                           const V3& Q, const V3& N, int firstOutcome) {
  const bool interp = (mw & RT_SPM_INTERP) != 0;
  if (!(mw & RT_SPM_LIT)) {
    // an interpolated material's base is already in spShade (k_surface)
    if (!interp) storeV3(P.spShade, P.cap, g, V3(S.devMats[mw & RT_SPM_INDEX].base));
    return;
  }
  V3 kd, ks, I;
  double shin, shinD = 0.0;
  int shinN = 0;
  if (interp) {
    kd = loadV3(P.spKd, P.cap, g);
    ks = loadV3(P.spKs, P.cap, g);
    shin = P.spShin[g];
    I = loadV3(P.spShade, P.cap, g);
  } else {
    const DevMat& dm = S.devMats[mw & RT_SPM_INDEX];
    kd = V3(dm.kd);
    ks = V3(dm.ks);
    shin = dm.shin;
    shinN = dm.shinN;
    shinD = dm.shinD;
    I = V3(dm.base);
  }
  const bool hasKd = !iszero(kd), hasKs = !iszero(ks);
  // the same for every light (material.cpp:66-67)
  const V3 view = hasKs ? normalized(V3(S.eye) - Q) : V3(0.0, 0.0, 0.0);
  // (a light's query outcome is asked for one light ahead: read where it is
  // used, every turn of the loop starts with a wait for memory)
  int spNext = firstOutcome;
  for (int li = 0; li < S.n_lights; li++) {
    const size_t o = (size_t)li * P.cap + s;
    int sp = spNext;
    if (li + 1 < S.n_lights) spNext = P.shPrim[o + P.cap];
    if (sp == RT_PRIM_OUTSIDE) {
      // k_shadow found the point outside this spot light's cone (with the
      // comparison lightIntensityFrom would make again): the light's intensity
      // is (0, 0, 0), and unless a NaN or a negative shininess turns 0 * x into
      // something (see below) the light adds nothing -- without a square root
      // taken.  (Q is finite here: the comparison that found the point outside
      // the cone was true.)
      if (!anynan(N) && !(hasKs && shin < 0.0)) continue;
      sp = RT_PRIM_NONE;
    }
    const double th = P.shT ? P.shT[o] : 0.0;
    V3 L;
    const V3 lightIntensity = lightIntensityFrom<INTPOW>(S, S.lights[li], S.spotCone + 4 * li, Q,
                                                 sp == RT_PRIM_ANY ? RT_PRIM_NONE : sp, th, sp == RT_PRIM_ANY, L);
    // A light that contributes exactly zero (occluded by an opaque surface,
    // outside a spot cone) adds +0 to every component when the Phong factors are
    // finite: skip their sqrt / pow.  A NaN shade point (the hit of a NaN ray)
    // makes the factors NaN, and the reference's 0 * NaN then poisons the colour.
    // So does a negative shininess: pow(0, negative) is infinite and 0 * inf is NaN.
    if (iszero(lightIntensity) && !anynan(Q) && !anynan(N) && !anynan(L) && !(hasKs && shin < 0.0)) continue;
    // std::max(x, 0.0) of material.cpp:61,72 is (x < 0.0) ? 0.0 : x -- it hands a NaN on
    if (hasKd) {
      const double c = dot(N, L);
      I = I + prod(kd, lightIntensity) * ((c < 0.0) ? 0.0 : c);
    }
    if (hasKs) {
      const V3 cosVector = N * dot(L, N);
      const V3 sinVector = cosVector - L;
      const V3 refl = normalized(cosVector + sinVector);
      const double c = dot(refl, view);
      const double base = (c < 0.0) ? 0.0 : c;
      // pow(0, y) is 0 for y > 0: the highlight's far side needs no pow
      const double spec = (base == 0.0 && shin > 0.0) ? 0.0 : INTPOW ? powNearInt(base, shinN, shinD) : pow(base, shin);
      I = I + prod(ks, lightIntensity) * spec;
    }
  }
  storeV3(P.spShade, P.cap, g, I);
}

// One thread per shade point: its colour.  (Folding this into k_shade's first
// pass was measured slower: a thread there walks up to 16 points one after the
// other, and pow / acos want the parallelism of a dense launch.)
// INTPOW: every exponent of the scene (shininess of each material, fall_rate of
// each spot light) is a small non-negative integer up to rounding, and no
// material is interpolated: powNearInt instead of pow.
template <bool INTPOW>
__global__ void __launch_bounds__(kBlock, INTPOW ? 4 : 1)   // (without pow the kernel sits at the 128-register step: held to it)
k_lights(DevScene S, long long queried, Pool P) {
  long long nsp = shadePointsOf(P);
  checkCoverage(P.counters, (long long)gridDim.x * blockDim.x, nsp);
  if (nsp > queried) nsp = queried;   // k_shadow's grid fell short (it has flagged the level): no outcomes beyond
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsp) return;
  const size_t g = (size_t)P.counters->spOff + s;
  const int firstOutcome = S.n_lights > 0 ? P.shPrim[s] : RT_PRIM_NONE;   // (with the point's own data, before its material is known)
  shadePoint<INTPOW>(S, P, (int)s, g, (unsigned int)P.spMat[g], loadV3(P.spQ, P.cap, g), loadV3(P.spN, P.cap, g), firstOutcome);
}

// The secondary rays of RayTracer::traceRay (RayTracer.cpp:152-201).  Reflected rays of a tile are
// queued before its refracted rays.
// REFR false: no material of the scene transmits (DevScene::opaque), so the
// refraction code -- acos, a second queue class -- is left out; it costs this
// latency-bound kernel a third of its resident wavefronts even when it never runs.
template <bool REFR>
__global__ void __launch_bounds__(kBlock)
k_shade(DevScene S, int items, int par, RaySrc R, int depthLeft, Pool P) {
  __shared__ TileScan ts;
  __shared__ unsigned int litCount;
  const long long nsp = shadePointsOf(P);
  const size_t spOff = P.counters->spOff, rayOff = P.counters->rayOff;
  const size_t nextRayOff = rayOff + P.counters->n;
  const long long perTile = (long long)kBlock * items;
  checkCoverage(P.counters, gridDim.x * perTile, nsp);
  const long long tileBase = blockIdx.x * perTile;
  if (tileBase >= nsp) return;   // whole block, uniform
  {
    // pass 1: which children each shade point spawns (and how many points are lit)
    unsigned int reflMask = 0, refrMask = 0, myLit = 0;
    for (int k = 0; k < items; k++) {
      const long long s = tileBase + k * kBlock + threadIdx.x;
      bool wantRefl = false, wantRefr = false;
      if (s < nsp) {
        const size_t g = spOff + s;
        const unsigned int mw = (unsigned int)P.spMat[g];
        const bool interp = (mw & RT_SPM_INTERP) != 0;
        const DevMat* dm = interp ? nullptr : &S.devMats[mw & RT_SPM_INDEX];
        if (mw & RT_SPM_LIT) myLit++;
        if (depthLeft > 0) {
          wantRefl = interp ? !iszero(loadV3(P.spKr, P.cap, g)) : (dm->flags & RT_MATF_KR) != 0;
          if (REFR && (interp ? !iszero(loadV3(P.spKt, P.cap, g)) : (dm->flags & RT_MATF_KT) != 0)) {
            V3 p, d;
            rayOf(S, R, (int)((size_t)P.spRayGlobal[g] - rayOff), p, d);
            const double cosineAngle = acos(dot(loadV3(P.spN, P.cap, g), d)) * 180 / M_PI;
            wantRefr = cosineAngle < 360;  // RayTracer.cpp:173,195: false only for a NaN angle
          }
        } else {
          P.spChildRefl[g] = -1;
          if (REFR) P.spChildRefr[g] = -1;
        }
      }
      reflMask |= (wantRefl ? 1u : 0u) << k;
      refrMask |= (wantRefr ? 1u : 0u) << k;
      tileCount(ts, 0, k, wantRefl);
      if (REFR) tileCount(ts, 1, k, wantRefr);
    }
    if (threadIdx.x == 0) litCount = 0;
    unsigned int nRefl = 0, nRefr = 0;
    tileReserve(ts, items, REFR ? 2 : 1, &P.counters->nnext, nRefl, nRefr);   // has barriers before and after the scan
    if (myLit) atomicAdd(&litCount, myLit);                         // LDS atomic
    __syncthreads();
    if (threadIdx.x == 0) {
      if (nRefl) atomicAdd(&P.counters->refl, (unsigned long long)nRefl);
      if (nRefr) atomicAdd(&P.counters->refr, (unsigned long long)nRefr);
      if (litCount) atomicAdd(&P.counters->litsp, (unsigned long long)litCount);
    }
    if (depthLeft <= 0) return;
    // pass 2: generate the rays into their slots
    const int np = par ^ 1;
    for (int k = 0; k < items; k++) {
      const long long s = tileBase + k * kBlock + threadIdx.x;
      const bool wantRefl = (reflMask >> k) & 1u, wantRefr = REFR && ((refrMask >> k) & 1u);
      const unsigned int a = tileSlot(ts, 0, k, wantRefl);
      const unsigned int b = REFR ? tileSlot(ts, 1, k, wantRefr) : 0u;
      if (s >= nsp) continue;
      const size_t g = spOff + s;
      int childRefl = -1, childRefr = -1;
      if (wantRefl || wantRefr) {
        V3 p, d;
        rayOf(S, R, (int)((size_t)P.spRayGlobal[g] - rayOff), p, d);
        const V3 N = loadV3(P.spN, P.cap, g);
        const V3 minusD = -1 * d;
        const V3 cosVector = N * dot(minusD, N);
        const V3 sinVector = cosVector + d;
        if (wantRefl) {
          if (nextRayOff + a >= P.cap) {
            P.counters->overflow = RT_OVERFLOW_POOL;
          } else {
            const V3 reflDir = normalized(cosVector + sinVector);
            P.rayParent[np][a] = (int)g;
            storeV3(P.rayD[np], P.cap, a, reflDir);
            P.rayType[np][a] = RT_RAY_REFLECTION;
            childRefl = (int)(nextRayOff + a);
          }
        }
        if (wantRefr) {
          if (nextRayOff + b >= P.cap) {
            P.counters->overflow = RT_OVERFLOW_POOL;
          } else {
            const unsigned int mw = (unsigned int)P.spMat[g];
            const double ior = (mw & RT_SPM_INTERP) ? P.spIor[g] : S.devMats[mw & RT_SPM_INDEX].ior;
            const double cosineAngle = acos(dot(N, d)) * 180 / M_PI;
            double n_i, n_r;
            int iDirection;
            if (cosineAngle > 90) { n_i = 1; n_r = ior; iDirection = 1; }
            else { n_i = ior; n_r = 1; iDirection = -1; }
            const V3 sinT = (n_i / n_r) * sinVector;
            const V3 cosT = (-1 * N) * sqrt(1 - dot(sinT, sinT));
            const V3 rd = normalized(cosT + (double)iDirection * sinT);
            P.rayParent[np][b] = (int)g;
            storeV3(P.rayD[np], P.cap, b, (double)iDirection * rd);
            P.rayType[np][b] = RT_RAY_REFRACTION;
            childRefr = (int)(nextRayOff + b);
          }
        }
      }
      P.spChildRefl[g] = childRefl;
      if (REFR) P.spChildRefr[g] = childRefr;   // (an opaque scene has no such children: k_combine does not look)
    }
  }
}

// k_classify + k_surface + k_shade in one pass over the level's rays, for scenes
// in which every hit is a triangle with one of the scene's materials and no
// material transmits (no k_surface_rest cases, no refraction): a thread that
// has just looked its hit up goes on to finish the shade point and to spawn the
// reflected ray, instead of three kernels handing the same hit to each other
// through memory (hit -> slot list -> position / normal / material -> ray).
// Shade-point slots and next-level ray slots come from two tile scans.
// (the kernel waits for its gathers three quarters of the time: five wavefronts
// per SIMD, with 44 bytes of spill per lane, beat four without and six with 88.
// Measured and dropped: holding an entry's stores back until the next entry's
// loads are issued -- a load behind a store waits for the store too -- needs
// the next entry's 50 loaded registers beside the 23 waiting to be stored:
// 8.8 ms instead of 4.4 at five wavefronts per SIMD, 4.5 at four.)
#ifndef RT_HIT_WAVES
#define RT_HIT_WAVES 5
#endif
// Entries per thread of k_hit's tiles: the primitive and the face word of every
// entry wait in LDS between the two passes (2 x 4 bytes x kBlock per entry).
constexpr int kHitItems = 8;

__global__ void __launch_bounds__(kBlock, RT_HIT_WAVES)
k_hit(DevScene S, int items, int par, RaySrc R, int level, int depthLeft, int blackMiss, long long traced, Pool P) {
  __shared__ TileScan tsHit, tsRefl;
  __shared__ unsigned int litCount;
  __shared__ int heldPrim[kHitItems][kBlock], heldInfo[kHitItems][kBlock];
  const long long n = tracedRays(P.counters, traced);
  const size_t rayOff = P.counters->rayOff, spOff = P.counters->spOff;
  const size_t nextRayOff = rayOff + (size_t)n;
  const long long perTile = (long long)kBlock * items;
  checkCoverage(P.counters, gridDim.x * perTile, n);
  const long long tileBase = blockIdx.x * perTile;
  if (tileBase >= n) return;   // whole block, uniform
  // pass 1: which rays hit, which of the hits reflect, how many are lit.  Every
  // entry's loads are independent of the other entries': all primitives are
  // asked for, then all face words, two waits instead of two per entry (left
  // to itself the compiler keeps each entry's pair behind the LDS writes of
  // the entry before).
  unsigned int hitMask = 0, reflMask = 0, myLit = 0;
  int prims[kHitItems], infos[kHitItems];
#pragma unroll
  for (int k = 0; k < kHitItems; k++) {
    const long long i = tileBase + k * kBlock + threadIdx.x;
    prims[k] = (k < items && i < n) ? P.hitPrim[i] : RT_PRIM_NONE;
  }
#pragma unroll
  for (int k = 0; k < kHitItems; k++)
    infos[k] = prims[k] != RT_PRIM_NONE ? S.faceInfo[prims[k] & ~RT_PRIM_FACE] : 0;
#pragma unroll
  for (int k = 0; k < kHitItems; k++) {
    if (k >= items) break;
    const int prim = prims[k], info = infos[k];
    const bool hit = prim != RT_PRIM_NONE;
    heldPrim[k][threadIdx.x] = prim;
    heldInfo[k][threadIdx.x] = info;
    if (info & RT_FI_LIT) myLit++;
    const bool refl = depthLeft > 0 && ((unsigned int)info & RT_FI_REFLECTS) != 0;
    hitMask |= (hit ? 1u : 0u) << k;
    reflMask |= (refl ? 1u : 0u) << k;
    tileCount(tsHit, 0, k, hit);
    tileCount(tsRefl, 0, k, refl);
  }
  if (threadIdx.x == 0) litCount = 0;
  unsigned int nHit = 0, nRefl = 0, unused = 0;
  tileReserve(tsHit, items, 1, &P.counters->nsp, nHit, unused);
  tileReserve(tsRefl, items, 1, &P.counters->nnext, nRefl, unused);
  if (myLit) atomicAdd(&litCount, myLit);   // LDS atomic
  __syncthreads();
  if (threadIdx.x == 0) {
    if (nRefl) atomicAdd(&P.counters->refl, (unsigned long long)nRefl);
    if (litCount) atomicAdd(&P.counters->litsp, (unsigned long long)litCount);
  }
  // pass 2: misses turn black, hits become shade points and spawn their reflected ray
  const int np = par ^ 1;
  for (int k = 0; k < items; k++) {
    const long long i = tileBase + k * kBlock + threadIdx.x;
    const bool hit = (hitMask >> k) & 1u, refl = (reflMask >> k) & 1u;
    const unsigned int s = tileSlot(tsHit, 0, k, hit);
    const unsigned int a = tileSlot(tsRefl, 0, k, refl);
    if (i >= n) continue;
    if (level == 0) P.primarySp[i] = (hit && spOff + s < P.cap) ? (int)(spOff + s) : -1;
    if (!hit) {
      if (blackMiss) {
        // level 0: k_resolve paints a sample without a shade point black
        if (level > 0) P.spChildRefl[R.parent[i]] = -1;   // see dropChild; every ray of such a scene is a reflected one
      } else {
        storeV3(P.rayI, P.cap, rayOff + i, V3(0.0, 0.0, 0.0));   // RayTracer.cpp:213-224 (k_background adds a cube map)
      }
      if (level == 0 && firstSampleOfPixel(R.fm, i)) P.primaryId[i] = 0;
      continue;
    }
    const size_t g = spOff + s;
    if (g >= P.cap) {
      P.counters->overflow = RT_OVERFLOW_POOL;
      continue;
    }
    const int fi = heldPrim[k][threadIdx.x] & ~RT_PRIM_FACE;
    const int info = heldInfo[k][threadIdx.x];
    // (Measured: asking for everything the entry reads of its face in one go, ahead of
    // the ray's arithmetic, instead of where each value is first used: 43 registers
    // held across that arithmetic spill, and the kernel takes twice as long.)
    const double t = P.hitT[i];
    const RayPD ray = rayAt(S, R, (int)i);
    Surface sf;
    surfaceNormalOfFace(S, fi, info, ray.p, ray.d, sf);
    const V3 Q = ray.p + (t * ray.d);                            // ray::at (ray.h:42-43)
    if (level > 0) P.spRayGlobal[g] = (int)(rayOff + i);   // for k_combine, which level 0 does not go through
    P.spMat[g] = (int)((unsigned int)(info & RT_FI_MATERIAL) | ((info & RT_FI_LIT) ? RT_SPM_LIT : 0u));   // material.cpp:50-53
    if (level == 0 && firstSampleOfPixel(R.fm, i)) P.primaryId[i] = sf.objectID;
    storeV3(P.spQ, P.cap, g, Q);
    storeV3(P.spN, P.cap, g, sf.N);
    int childRefl = -1;
    if (refl) {
      if (nextRayOff + a >= P.cap) {
        P.counters->overflow = RT_OVERFLOW_POOL;
      } else {
        // RayTracer.cpp:156-164
        const V3 minusD = -1 * ray.d;
        const V3 cosVector = sf.N * dot(minusD, sf.N);
        const V3 sinVector = cosVector + ray.d;
        const V3 reflDir = normalized(cosVector + sinVector);
        P.rayParent[np][a] = (int)g;
        storeV3(P.rayD[np], P.cap, a, reflDir);
        childRefl = (int)(nextRayOff + a);
      }
    }
    P.spChildRefl[g] = childRefl;
  }
}

// What traceRay returns for the ray that made shade point g, once the colours of
// its child rays are known: intensity = shade; += kr*traceRay(reflected);
// += kt*traceRay(refracted) (RayTracer.cpp:150-201).
__device__ __forceinline__ V3 combinedColour(const DevScene& S, const Pool& P, size_t g) {
  V3 I = loadV3(P.spShade, P.cap, g);
  const int cr = P.spChildRefl[g], ct = S.opaque ? -1 : P.spChildRefr[g];
  if (cr >= 0 || ct >= 0) {
    const unsigned int mw = (unsigned int)P.spMat[g];
    const bool interp = (mw & RT_SPM_INTERP) != 0;
    const DevMat* dm = interp ? nullptr : &S.devMats[mw & RT_SPM_INDEX];
    if (cr >= 0) I = I + prod(interp ? loadV3(P.spKr, P.cap, g) : V3(dm->kr), loadV3(P.rayI, P.cap, (size_t)cr));
    if (ct >= 0) I = I + prod(interp ? loadV3(P.spKt, P.cap, g) : V3(dm->kt), loadV3(P.rayI, P.cap, (size_t)ct));
  }
  return I;
}

// Levels 1 and deeper, bottom-up: the colour of every ray that hit something,
// into the ray's slot, where the level above finds it.  (Level 0 needs no such
// pass: k_resolve works a primary sample's colour out where it uses it.)
// (P.counters: the block of the level that is being combined)
__global__ void __launch_bounds__(kBlock)
k_combine(DevScene S, Pool P) {
  const long long nsp = shadePointsOf(P);
  checkCoverage(P.counters, (long long)gridDim.x * blockDim.x, nsp);
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nsp) {
    const size_t g = (size_t)P.counters->spOff + s;
    storeV3(P.rayI, P.cap, (size_t)P.spRayGlobal[g], combinedColour(S, P, g));
  }
}

// RayTracer::trace's clamp, tracePixel / tracePixelAntiAlias's byte conversion
// (RayTracer.cpp:41,58-60,94-101).  tileOffset: packed-buffer pixel offset of
// each of this shard's tiles, or NULL for a full-frame destination.
//
// A block resolves `ppb` consecutive pixels: all threads read the block's
// ppb * ns sample colours with coalesced loads, clamp them and park them in LDS
// (row pitch odd, so the column walk below is free of bank conflicts); then
// one thread per pixel adds its ns samples in sample order, as the reference's
// loop does.
constexpr int kResolveSamples = 1024;   // LDS rows * pitch per colour channel

__global__ void __launch_bounds__(kBlock)
k_resolve(DevScene S, FrameMap fm, SlotBase first, int nPixels, int ppb, int blackMiss, Pool P, unsigned char* rgb,
          int* hitId, const long long* tileOffset) {
  __shared__ double lds[3][kResolveSamples + kBlock];
  __shared__ int ldsId[kBlock];
  const int ns = fm.ns, pitch = ns | 1;
  const int pix0 = blockIdx.x * ppb;
  const int npix = min(ppb, nPixels - pix0);
  const size_t base = (size_t)pix0 * ns;
  // A thread's samples (at most kResolveSamples / kBlock of them) go through the
  // chain sample -> shade point -> its colour and child ray -> the child's colour
  // side by side: each link is asked for all samples before the next link is.
  constexpr int kPer = kResolveSamples / kBlock;
  const int nSamples = npix * ns;
  int sp[kPer], cr[kPer];
  V3 c[kPer];
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int e = threadIdx.x + j * kBlock;
    sp[j] = e < nSamples ? P.primarySp[base + e] : -1;
  }
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int e = threadIdx.x + j * kBlock;
    cr[j] = -1;
    if (sp[j] >= 0) {
      c[j] = loadV3(P.spShade, P.cap, (size_t)sp[j]);
      cr[j] = P.spChildRefl[sp[j]];
    } else if (e < nSamples && !blackMiss) {
      // the background's colour: what k_background left in the ray's slot
      c[j] = loadV3(P.rayI, P.cap, base + e);
    }
  }
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int e = threadIdx.x + j * kBlock;
    if (e >= nSamples) continue;
    if (sp[j] >= 0 && !(S.opaque && cr[j] < 0)) {
      // intensity = shade; += kr * traceRay(reflected); += kt * traceRay(refracted) (combinedColour, from its colour on)
      const size_t g = (size_t)sp[j];
      const int ct = S.opaque ? -1 : P.spChildRefr[g];
      if (cr[j] >= 0 || ct >= 0) {
        const unsigned int mw = (unsigned int)P.spMat[g];
        const bool interp = (mw & RT_SPM_INTERP) != 0;
        const DevMat* dm = interp ? nullptr : &S.devMats[mw & RT_SPM_INDEX];
        if (cr[j] >= 0) c[j] = c[j] + prod(interp ? loadV3(P.spKr, P.cap, g) : V3(dm->kr), loadV3(P.rayI, P.cap, (size_t)cr[j]));
        if (ct >= 0) c[j] = c[j] + prod(interp ? loadV3(P.spKt, P.cap, g) : V3(dm->kt), loadV3(P.rayI, P.cap, (size_t)ct));
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int e = threadIdx.x + j * kBlock;
    if (e >= nSamples) continue;
    V3 col = c[j];   // (a sample that hit nothing under a black background: (0, 0, 0), as V3 starts)
    if (col.x < 0) col.x = 0.0;
    if (col.x > 1) col.x = 1.0;
    if (col.y < 0) col.y = 0.0;
    if (col.y > 1) col.y = 1.0;
    if (col.z < 0) col.z = 0.0;
    if (col.z > 1) col.z = 1.0;
    const int pl = (int)fastDiv((unsigned int)e, fm.dNs), k = e - pl * ns;
    const int o = pl * pitch + k;
    lds[0][o] = col.x;
    lds[1][o] = col.y;
    lds[2][o] = col.z;
    if (k == 0 && hitId) ldsId[pl] = P.primaryId[base + e];
  }
  __syncthreads();
  const int pl = threadIdx.x;
  if (pl >= npix) return;
  const unsigned int rel = (unsigned int)(pix0 + pl);
  int px, py, tk, lx, ly;
  if (!slotToPixel(fm, first, rel, px, py, tk, lx, ly)) return;
  V3 col(0, 0, 0);
  const int r = pl * pitch;
  if (fm.aaN > 0) {
    for (int k = 0; k < ns; k++) col = V3(col.x + lds[0][r + k], col.y + lds[1][r + k], col.z + lds[2][r + k]);
    const double nn = double(fm.aaN * fm.aaN);
    col = V3(col.x / nn, col.y / nn, col.z / nn);
  } else {
    col = V3(lds[0][r + ns - 1], lds[1][r + ns - 1], lds[2][r + ns - 1]);
  }
  size_t o;
  if (fm.pixelList && fm.listOutput) {
    o = (size_t)(first.slot + rel);
  } else if (tileOffset) {
    const unsigned int t = (unsigned int)(fm.shardIndex + tk * fm.shardCount);
    const int tx = (int)(t - fastDiv(t, fm.dTilesX) * fm.dTilesX.d);
    const int wt = min(fm.tileW, fm.width - tx * fm.tileW);
    o = (size_t)tileOffset[tk] + (size_t)ly * wt + lx;
  } else {
    o = (size_t)px + (size_t)py * fm.width;
  }
  // (int)(255.0*c) then narrowing to a byte; a NaN converts to 0 here and to
  // INT_MIN -> byte 0 on the reference's x86-64
  rgb[3 * o + 0] = (unsigned char)(int)(255.0 * col.x);
  rgb[3 * o + 1] = (unsigned char)(int)(255.0 * col.y);
  rgb[3 * o + 2] = (unsigned char)(int)(255.0 * col.z);
  if (hitId) hitId[o] = ldsId[pl];
}

// Scene::intersect seam for tests: full isect of n rays.
template <bool FAST>
__global__ void __launch_bounds__(kBlock)
k_intersect(DevScene S, int n, const double* rays, const int* types, RtHit* hits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 p(rays + 6 * (size_t)i), d(rays + 6 * (size_t)i + 3);
  const int type = types ? types[i] : RT_RAY_VISIBILITY;
  TraceCounters tc{0, 0, 0};
  double t;
  const int prim = FAST ? traceBvh<false, false>(S, p, d, type, 1.0e308, -1.0, t, tc, (rt_lds_u32*)rtStackLds)
                        : traceKdReference<false>(S, p, d, type, t, tc, (rt_kd_lds_u32*)rtStackLds);
  RtHit h;
  h.t = 1000.0;                      // scene.cpp:155
  h.N[0] = h.N[1] = h.N[2] = 0.0;
  h.bary[0] = h.bary[1] = h.bary[2] = 0.0;
  h.object_id = 0;
  h._pad = 0;
  if (prim != RT_PRIM_NONE) {
    Surface sf;
    surfaceAt(S, prim, p, d, type, sf, nullptr, false);
    h.t = t;
    h.N[0] = sf.N.x; h.N[1] = sf.N.y; h.N[2] = sf.N.z;
    h.bary[0] = sf.bary.x; h.bary[1] = sf.bary.y; h.bary[2] = sf.bary.z;
    h.object_id = sf.objectID;
  }
  hits[i] = h;
}

// powNearInt on a list of bases (test hook, rt_test_pow_near_int)
__global__ void __launch_bounds__(kBlock)
k_pow_near_int(int count, const double* x, int n, double delta, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = powNearInt(x[i], n, delta);
}

// Packed tiles -> full frame
__global__ void __launch_bounds__(kBlock)
k_unpack(const unsigned char* packed, int bpp, FrameMap fm, const long long* tileOffset, int nTiles,
         unsigned char* frame) {
  const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= (long long)nTiles * fm.tileW * fm.tileH) return;   // < 2^31, checked by the caller
  int px, py, tk, lx, ly;
  if (!slotToPixel(fm, SlotBase{0, 0, 0u}, (unsigned int)slot, px, py, tk, lx, ly)) return;
  const unsigned int t = (unsigned int)(fm.shardIndex + tk * fm.shardCount);
  const int tx = (int)(t - fastDiv(t, fm.dTilesX) * fm.dTilesX.d);
  const int wt = min(fm.tileW, fm.width - tx * fm.tileW);
  const size_t src = (size_t)tileOffset[tk] + (size_t)ly * wt + lx;
  const size_t dst = (size_t)px + (size_t)py * fm.width;
  for (int b = 0; b < bpp; b++) frame[dst * bpp + b] = packed[src * bpp + b];
}

// GraphicalUI::applyFilter (ui/GraphicalUI.cpp:437-486) on the device: one
// thread per pixel.  Column == width reads the next row's first pixel like the
// reference; anything past the end of the image reads as 0.
__global__ void __launch_bounds__(kBlock)
k_edge_mask(const unsigned char* rgb, int w, int h, int cutOff, unsigned char* mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int r = i / w, c = i % w;
  const int kernel[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};
  double sum = 0;
  for (int fr = 0; fr < 3; fr++)
    for (int fc = 0; fc < 3; fc++) {
      const int tr = r - 1 + fr, tc = c - 1 + fc;
      if (tr < 0 || tc < 0 || tr > h || tc > w) continue;
      const long long idx = (long long)tr * w + tc;
      double g = 0;
      if (idx < (long long)w * h) {
        const unsigned char* p = rgb + 3 * idx;
        g = (double)(unsigned char)(0.299 * p[0] + 0.587 * p[1] + 0.114 * p[2]);
      }
      sum = sum + kernel[3 * fr + fc] * g;
    }
  mask[i] = (sum > cutOff) ? 255 : 0;
}

// ---- flagged pixels of an edge mask, as an (x, y) list in row-major order ----
// GraphicalUI::doAntiAliasing walks the mask row by row (ui/GraphicalUI.cpp:
// 405-425); the list keeps that order: a block counts the flagged pixels of its
// kBlock consecutive pixels, one workgroup turns the counts into offsets (and
// tells the host the total through mapped host memory), a block writes its
// pixels behind its offset in pixel order.
__global__ void __launch_bounds__(kBlock)
k_flag_count(const unsigned char* mask, long long n, unsigned int* blockCount) {
  __shared__ unsigned int waveCount[kWaves];
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  const unsigned long long m = __ballot(i < n && mask[i] == 255);
  if ((threadIdx.x & 63) == 0) waveCount[threadIdx.x >> 6] = (unsigned int)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int c = 0;
    for (int k = 0; k < kWaves; k++) c += waveCount[k];
    blockCount[blockIdx.x] = c;
  }
}

// Exclusive scan of nBlocks counts in place by ONE workgroup of 1024 threads
// (a 4K frame has 32 400 of them); *totalHost is device-visible host memory.
__global__ void __launch_bounds__(1024)
k_flag_scan(unsigned int* blockCount, int nBlocks, unsigned int* totalHost) {
  __shared__ unsigned int part[1024];
  const int per = (nBlocks + 1023) / 1024;
  const int lo = min(nBlocks, (int)threadIdx.x * per), hi = min(nBlocks, lo + per);
  unsigned int mine = 0;
  for (int k = lo; k < hi; k++) mine += blockCount[k];
  part[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the 1024 partial sums
    const unsigned int o = threadIdx.x >= (unsigned int)off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += o;
    __syncthreads();
  }
  unsigned int run = part[threadIdx.x] - mine;
  for (int k = lo; k < hi; k++) {
    const unsigned int c = blockCount[k];
    blockCount[k] = run;
    run += c;
  }
  if (threadIdx.x == 1023) *totalHost = part[1023];
}

__global__ void __launch_bounds__(kBlock)
k_flag_write(const unsigned char* mask, int w, long long n, const unsigned int* blockOffset, int* xy) {
  __shared__ unsigned int waveCount[kWaves];
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  const bool flag = i < n && mask[i] == 255;
  const unsigned long long m = __ballot(flag);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) waveCount[wave] = (unsigned int)__popcll(m);
  __syncthreads();
  if (!flag) return;
  unsigned int slot = blockOffset[blockIdx.x] + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
  for (int k = 0; k < wave; k++) slot += waveCount[k];
  xy[2 * (size_t)slot] = (int)(i % w);
  xy[2 * (size_t)slot + 1] = (int)(i / w);
}

// Entries first, first + stride, ... of an (x, y) list take the colours of
// rows 0, 1, ... (the re-rendered pixels of one share, GraphicalUI.cpp:405-425).
__global__ void __launch_bounds__(kBlock)
k_patch_listed(const int* xy, long long first, long long stride, long long n, const unsigned char* rows, int w,
               unsigned char* frame) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const long long e = first + k * stride;
  const size_t o = (size_t)xy[2 * e] + (size_t)xy[2 * e + 1] * w;
  frame[3 * o + 0] = rows[3 * k + 0];
  frame[3 * o + 1] = rows[3 * k + 1];
  frame[3 * o + 2] = rows[3 * k + 2];
}

}  // namespace

// ================================================================ host side
namespace {

thread_local std::string g_lastError;

int fail(int code, const std::string& msg) {
  g_lastError = msg;
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));         \
  } while (0)

template <class T>
hipError_t upload(const T* host, size_t count, T** dev) {
  *dev = nullptr;
  if (count == 0) return hipSuccess;
  hipError_t e = hipMalloc((void**)dev, count * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemcpy(*dev, host, count * sizeof(T), hipMemcpyHostToDevice);
}

inline int gridFor(long long n) { return (int)((n + kBlock - 1) / kBlock); }

// Plain Gauss-Jordan inverse of a row-major 4x4 (only used to box transformed
// meshes for the BVH, never for shading arithmetic).
void invert4(const double* m, double* out) {
  double a[4][8];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) { a[r][c] = m[4 * r + c]; a[r][4 + c] = (r == c) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; c++) {
    int piv = c;
    for (int r = c + 1; r < 4; r++) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    for (int k = 0; k < 8; k++) std::swap(a[c][k], a[piv][k]);
    const double s = a[c][c];
    if (s == 0.0) continue;
    for (int k = 0; k < 8; k++) a[c][k] /= s;
    for (int r = 0; r < 4; r++) {
      if (r == c) continue;
      const double f = a[r][c];
      for (int k = 0; k < 8; k++) a[r][k] -= f * a[c][k];
    }
  }
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) out[4 * r + c] = a[r][4 + c];
}

}  // namespace

// ---- staging memory of rt_intersect_batch_static ------------------------------
// Rays, ray types and hit records of one call, kRaysStaticMax rays at most, in
// an array that is part of the code object: every GPU the library runs on holds
// its own copy (7.25 MiB) until the process ends, and a call that uses it
// allocates nothing.  In host code the array's name is only a handle for
// hipGetSymbolAddress; the address it yields for the current device is what
// kernels and copies are given.
constexpr int kRaysStaticMax = 65536;
constexpr size_t kStaticRaysBytes = (size_t)kRaysStaticMax * 6 * sizeof(double);
constexpr size_t kStaticTypesBytes = (size_t)kRaysStaticMax * sizeof(int);
constexpr size_t kStaticHitsBytes = (size_t)kRaysStaticMax * sizeof(RtHit);
static __device__ __align__(256) unsigned char scratch[kStaticRaysBytes + kStaticTypesBytes + kStaticHitsBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratchMutex;
unsigned char* g_scratchPtr[kMaxDevices] = {};   // device address of `scratch` per HIP device

// Looks the array's address up on the CURRENT device, once per device.
hipError_t scratchOfCurrentDevice(int device, unsigned char** out) {
  *out = nullptr;
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratchMutex);
  if (!g_scratchPtr[device]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratchPtr[device] = static_cast<unsigned char*>(ptr);
  }
  *out = g_scratchPtr[device];
  return hipSuccess;
}
}  // namespace

struct RtScene {
  int device = 0;
  hipStream_t stream = nullptr;
  // Shadow queries and shading of recursion level l run here beside the main
  // stream's spawn / trace / surface kernels of level l+1 (see renderBatch).
  hipStream_t side = nullptr;
  hipEvent_t evSurface = nullptr, evSide = nullptr;   // main -> side, side -> main
  int overlapMode = -1;            // RT_AMD_OVERLAP=0: everything on one stream
  DevScene dev;
  std::vector<void*> allocs;       // scene arrays
  int nLights = 0;
  int maxStackNeed = 0;
  bool triOnly = true;             // nothing but triangles of identity-transform meshes: lean traversal kernels
  bool hasFaces = false;           // some mesh has triangles (else k_surface has nothing to do)
  bool surfaceRest = false;        // some hits need k_surface_rest (non-mesh objects, interpolated materials)
  // workspace
  Pool pool;
  std::vector<void*> poolAllocs;
  size_t poolLights = 0;
  // pool slots one sample is expected to need (rays or shade points over all
  // recursion levels); measured by the batches rendered so far at growthDepth
  double growth = 6.0;
  bool growthMeasured = false;
  int growthDepth = -1;
  unsigned char* frameRgb = nullptr; int* frameHit = nullptr; size_t framePixels = 0;
  double* jitterDev = nullptr; size_t jitterCap = 0;
  std::vector<double> jitterHost;  // what jitterDev holds
  double* pixelXY = nullptr; int pixelXYw = 0, pixelXYh = 0;   // i / width per column, then j / height per row
  std::vector<double> pixelXYHost;
  unsigned char* isectBuf = nullptr; size_t isectCap = 0;   // rt_intersect_batch's staging, in rays
  // adaptive anti-aliasing: edge mask, flagged pixels per run of kBlock pixels, the (x, y) list of the flagged
  // pixels of the last base image (and whose it was)
  unsigned char* flagMask = nullptr; size_t flagMaskCap = 0;
  unsigned int* flagRuns = nullptr; size_t flagRunsCap = 0;
  int* flagList = nullptr; size_t flagListCap = 0;
  long long flagCount = 0; int flagW = 0, flagH = 0;
  unsigned char* listStage = nullptr; size_t listStageCap = 0;   // rt_render_pixels: the caller's list and its colours
  std::vector<hipEvent_t> events;  // pairs around traversal kernels
  size_t eventsUsed = 0;
  std::vector<unsigned char> eventKind;   // per pair: 0 k_trace, 1 k_lights
  hipEvent_t evStart = nullptr, evStop = nullptr;
  Counters* countersHost = nullptr;   // pinned: the read-back of the counter blocks (kCounterBlocks of them)
  // Rays per recursion level of the batches rendered so far, by what the batch
  // was (pixel range, sampling, depth, frame and shard geometry): a batch that
  // has been rendered before sizes its grids from these counts.  The scene and
  // its camera never change under a handle, so they only miss when the caller
  // changes what the key does not hold (another pixel list of the same length).
  typedef std::array<long long, 12> BatchKey;
  std::map<BatchKey, std::vector<unsigned int>> levelHist;
  std::vector<unsigned int> levelN = std::vector<unsigned int>(kCounterBlocks, 0u);   // level-sync batches: rays per level
  int levelSyncMode = 0;              // RT_AMD_LEVEL_SYNC=1: read every level's count back before the next
  bool finiteCoefficients = true;     // kr and kt of every material are finite (k_classify / k_hit: dropChild)
  bool intPow = true;                 // every pow exponent of the scene is near a small integer: k_lights<true>
  bool launchEvents = false;          // RT_AMD_LAUNCH_EVENTS=1: time every traversal launch (RtStats::ms_trace, ms_lights)
  bool fusedHit = true;               // RT_AMD_FUSED_HIT=0: k_classify / k_surface / k_shade even where k_hit would do

  ~RtScene() {
    if (stream) hipStreamSynchronize(stream);
    if (side) hipStreamSynchronize(side);
    for (void* p : allocs) hipFree(p);
    freePool();
    if (frameRgb) hipFree(frameRgb);
    if (frameHit) hipFree(frameHit);
    if (jitterDev) hipFree(jitterDev);
    if (pixelXY) hipFree(pixelXY);
    if (isectBuf) hipFree(isectBuf);
    if (flagMask) hipFree(flagMask);
    if (flagRuns) hipFree(flagRuns);
    if (flagList) hipFree(flagList);
    if (listStage) hipFree(listStage);
    for (hipEvent_t e : events) hipEventDestroy(e);
    if (countersHost) hipHostFree(countersHost);
    if (evStart) hipEventDestroy(evStart);
    if (evStop) hipEventDestroy(evStop);
    if (evSurface) hipEventDestroy(evSurface);
    if (evSide) hipEventDestroy(evSide);
    if (side) hipStreamDestroy(side);
    if (stream) hipStreamDestroy(stream);
  }
  void freePool() {
    for (void* p : poolAllocs) hipFree(p);
    poolAllocs.clear();
    pool.cap = 0;
  }
};

namespace {

// v = n + delta with n a non-negative integer below 2^20 and |delta| <= 1e-12:
// what powNearInt takes for an exponent.
bool nearSmallInt(double v, int& n, double& delta) {
  if (!(v >= 0.0 && v < 1048576.0)) return false;
  const double r = std::floor(v + 0.5);
  if (std::fabs(v - r) > 1.0e-12) return false;
  if (r == 0.0 && v != 0.0) return false;   // a positive exponent, however small, sends 0 to 0 and not to 1
  n = (int)r;
  delta = v - r;
  return true;
}

// Depth of a subtree and validation of child / item indices.
int treeDepth(const RtSceneDesc& d, int root, int itemLimit, std::string& err) {
  if (root < 0) return 0;
  int best = 0;
  std::vector<std::pair<int, int>> st{{root, 1}};
  size_t visited = 0;
  while (!st.empty()) {
    auto [n, dep] = st.back();
    st.pop_back();
    if (n < 0 || n >= d.n_nodes) { err = "kd node index out of range"; return -1; }
    if (++visited > (size_t)d.n_nodes) { err = "kd tree has a cycle"; return -1; }
    best = std::max(best, dep);
    const int l = d.node_left[n], r = d.node_right[n];
    if (l < 0 && r < 0) {
      const int s = d.node_item_start[n], c = d.node_item_count[n];
      if (c < 0 || s < 0 || (long long)s + c > d.n_leaf_items) { err = "kd leaf item range out of bounds"; return -1; }
      for (int k = 0; k < c; k++)
        if (d.leaf_items[s + k] < 0 || d.leaf_items[s + k] >= itemLimit) {
          err = "kd leaf item out of range";
          return -1;
        }
    } else {
      if (l < 0 || r < 0) { err = "kd node with a single child"; return -1; }
      if (d.node_dim[n] < 0 || d.node_dim[n] > 2) { err = "kd node with bad split dimension"; return -1; }
      st.push_back({l, dep + 1});
      st.push_back({r, dep + 1});
    }
  }
  return best;
}

// Appends the null triangle (normal zero: every ray is parallel to it and
// rejected) and points every unused child slot at a leaf that holds it.  False:
// the array has outgrown the leaf encoding.
bool addNullLeaf(std::vector<DevTri>& ordered, std::vector<rtbvh::Node4>& nodes4) {
  DevTri nullTri;
  std::memset(&nullTri, 0, sizeof nullTri);
  nullTri.axes = 0 | (1 << 2) | 16;
  ordered.push_back(nullTri);
  if (ordered.size() >= (1u << 27)) return false;
  const int32_t nullLeaf = ~(int32_t)(((int32_t)(ordered.size() - 1) << 3) | 0);
  for (rtbvh::Node4& n4 : nodes4)
    for (int c = 0; c < 4; c++)
      if (n4.child[c] == rtbvh::kEmptyChild) n4.child[c] = nullLeaf;
  return true;
}

// What the kernels of triangle-only scenes read instead of the quantised nodes
// (DevBvhNodeF): per 4-wide node its child boxes as binary32 planes, the normal
// cone of every child, and all of it eight times over, once per combination of
// direction signs with the children sorted near to far.  `ordered` is the
// leaf-ordered triangle array with the null triangle of the unused slots at its
// end; anyCone: some child has a cone narrow enough to cull by.
void buildNodeCopies(const std::vector<rtbvh::Node4>& nodes4, const std::vector<rtbvh::Collapser::Tight>& tightBoxes,
                     const std::vector<DevTri>& ordered, std::vector<DevBvhNodeF>& fn, bool& anyCone) {
  struct {
    const std::vector<rtbvh::Node4>& nodes;
    const std::vector<rtbvh::Collapser::Tight>& tight;
  } wide{nodes4, tightBoxes};
  // Normal cone of every child of every node (see DevBvhNodeF::cone).  The
  // binary builder partitions the triangle array in place, so the
  // triangles below a child are a contiguous run of the leaf-ordered
  // array: axis = the normalised sum of the run's unit normals, opening =
  // the widest angle any of them makes with it.
  struct Run { int lo, hi; };
  std::vector<std::array<float, 16>> cones(wide.nodes.size());
  {
    std::vector<std::array<double, 3>> unit(ordered.size());
    std::vector<char> valid(ordered.size(), 0);
    for (size_t t = 0; t < ordered.size(); t++) {
      const double* nn = ordered[t].n;
      const double len = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
      // a triangle without a direction (null or NaN normal) is never hit, whatever the ray
      if (len > 0.0 && std::isfinite(len)) {
        unit[t] = {nn[0] / len, nn[1] / len, nn[2] / len};
        valid[t] = 1;
      }
    }
    auto coneOf = [&](Run r, float* out) {
      double ax[3] = {0.0, 0.0, 0.0};
      for (int t = r.lo; t < r.hi; t++)
        if (valid[(size_t)t]) for (int a = 0; a < 3; a++) ax[a] += unit[(size_t)t][a];
      const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      out[0] = out[1] = out[2] = 0.0f;
      out[3] = 2.0f;
      if (!(al > 1e-9)) return;
      for (int a = 0; a < 3; a++) { ax[a] /= al; out[a] = (float)ax[a]; }
      // The widest angle a normal of the run makes with the axis AS STORED (binary32, so only nearly of unit
      // length), by its sine from the cross product: the sine of a narrow cone cannot be had from the cosine
      // (1 - cos is below the rounding of the axis once the angle is under a few 1e-4, which coplanar
      // neighbours of a terrain are).
      const double sx = out[0], sy = out[1], sz = out[2];
      const double sl = std::sqrt(sx * sx + sy * sy + sz * sz);
      double cosMin = 1.0, sinMax = 0.0;
      for (int t = r.lo; t < r.hi; t++)
        if (valid[(size_t)t]) {
          const double* u = unit[(size_t)t].data();
          cosMin = std::min(cosMin, (sx * u[0] + sy * u[1] + sz * u[2]) / sl);
          const double cx = sy * u[2] - sz * u[1], cy = sz * u[0] - sx * u[2], cz = sx * u[1] - sy * u[0];
          sinMax = std::max(sinMax, std::sqrt(cx * cx + cy * cy + cz * cz) / sl);
        }
      if (cosMin <= 0.05) return;   // wider than 87 degrees: nothing to gain
      // sin(theta), and room for the binary32 roundings of this bound, of the direction and of the visit's dot product
      out[3] = (float)(sinMax * (1.0 + 1.0e-4) + 1.0e-5);
      if (out[3] >= 1.0f) out[3] = 2.0f;
    };
    // runs bottom-up: the children of a node come after it in `wide.nodes` (the collapser appends them).
    // An empty run marks a child without triangles of its own (an unused slot, which points at the
    // null triangle at the end of the array; a non-mesh object).
    const int nullTriangle = (int)ordered.size() - 1;
    std::vector<std::array<Run, 4>> runs(wide.nodes.size());
    std::vector<Run> whole(wide.nodes.size());
    for (size_t i = wide.nodes.size(); i-- > 0;) {
      Run all{INT32_MAX, -1};
      for (int c = 0; c < 4; c++) {
        const int32_t ref = wide.nodes[i].child[c];
        Run r{0, -1};
        if (ref >= 0) {
          r = whole[(size_t)ref];
        } else {
          const int x = ~ref;
          if (!(x & rtbvh::kObjectBit) && (x >> 3) != nullTriangle) r = Run{x >> 3, (x >> 3) + (x & 7) + 1};
        }
        runs[i][(size_t)c] = r;
        if (r.hi > r.lo) { all.lo = std::min(all.lo, r.lo); all.hi = std::max(all.hi, r.hi); }
      }
      whole[i] = all;
    }
    bool any = false;
    for (size_t i = 0; i < wide.nodes.size(); i++)
      for (int c = 0; c < 4; c++) {
        float* out = &cones[i][(size_t)(4 * c)];
        out[0] = out[1] = out[2] = 0.0f;
        out[3] = 2.0f;
        const Run r = runs[i][(size_t)c];
        if (r.hi > r.lo) coneOf(r, out);
        any = any || out[3] < 1.0f;
      }
    anyCone = any;
  }
  // Eight copies per node, copy o for rays whose direction is negative on
  // the axes of o's set bits, its children sorted near to far along that
  // diagonal (box centres; unused slots, whose boxes are inverted, last).
  fn.resize(wide.nodes.size() * 8);
  for (size_t i = 0; i < wide.nodes.size(); i++) {
    const rtbvh::Node4& n4 = wide.nodes[i];
    // (the boxes as the builder made them, not their 8-bit quantised form:
    // this path has room for them, and tighter boxes cull more)
    const rtbvh::Collapser::Tight& tb = wide.tight[i];
    float blo[3][4], bhi[3][4];
    for (int c = 0; c < 4; c++)
      for (int a = 0; a < 3; a++) { blo[a][c] = tb.lo[a][c]; bhi[a][c] = tb.hi[a][c]; }
    for (int o = 0; o < 8; o++) {
      double key[4];
      int order[4] = {0, 1, 2, 3};
      for (int c = 0; c < 4; c++) {
        bool inverted = false;
        double k = 0.0;
        for (int a = 0; a < 3; a++) {
          inverted |= blo[a][c] > bhi[a][c];
          const double centre = 0.5 * ((double)blo[a][c] + (double)bhi[a][c]);
          k += ((o >> a) & 1) ? -centre : centre;
        }
        key[c] = inverted ? std::numeric_limits<double>::infinity() : k;
      }
      std::stable_sort(order, order + 4, [&](int x, int y) { return key[x] < key[y]; });
      DevBvhNodeF& F = fn[i * 8 + (size_t)o];
      std::memset(&F, 0, sizeof F);
      float* lo[3] = {F.lo_x, F.lo_y, F.lo_z};
      float* hi[3] = {F.hi_x, F.hi_y, F.hi_z};
      for (int j = 0; j < 4; j++) {
        const int c = order[j];
        F.child[j] = n4.child[c];
        for (int a = 0; a < 3; a++) { lo[a][j] = blo[a][c]; hi[a][j] = bhi[a][c]; }
        for (int a = 0; a < 4; a++) F.cone[j][a] = cones[i][(size_t)(4 * c + a)];
      }
    }
  }
}

int ensurePool(RtScene* sc, size_t cap) {
  const size_t nl = (size_t)std::max(1, sc->nLights);
  if (sc->pool.cap >= cap && sc->poolLights >= nl) return RT_OK;
  sc->freePool();
  Pool& P = sc->pool;
  bool outOfMemory = false;
  auto A = [&](size_t bytes, void** out) -> hipError_t {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) sc->poolAllocs.push_back(*out);
    else if (e == hipErrorOutOfMemory) outOfMemory = true;
    return e;
  };
  // per-point material arrays only where some mesh interpolates materials
  const size_t icap = sc->dev.interp_materials ? cap : 1;
#define POOL_ALLOC(field, type, count)                                                          \
  do {                                                                                          \
    if (A(sizeof(type) * (count), (void**)&(field)) != hipSuccess) {                            \
      (void)hipGetLastError();                                                                  \
      sc->freePool();                                                                           \
      return fail(outOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, "cannot allocate the ray pool");  \
    }                                                                                           \
  } while (0)
  for (int b = 0; b < 2; b++) {
    POOL_ALLOC(P.rayParent[b], int, cap);
    POOL_ALLOC(P.rayD[b], double, 3 * cap);
    POOL_ALLOC(P.rayType[b], int, cap);
  }
  POOL_ALLOC(P.hitT, double, cap);
  POOL_ALLOC(P.hitPrim, int, cap);
  POOL_ALLOC(P.spQ, double, 3 * cap);
  POOL_ALLOC(P.spN, double, 3 * cap);
  POOL_ALLOC(P.shPrim, int, nl * cap);
  if (sc->dev.opaque) P.shT = nullptr;
  else POOL_ALLOC(P.shT, double, nl * cap);
  POOL_ALLOC(P.rayI, double, 3 * cap);
  POOL_ALLOC(P.spShade, double, 3 * cap);
  POOL_ALLOC(P.spMat, int, cap);
  POOL_ALLOC(P.spChildRefl, int, cap);
  POOL_ALLOC(P.spChildRefr, int, cap);
  POOL_ALLOC(P.spRayGlobal, int, cap);
  POOL_ALLOC(P.primaryId, int, cap);
  POOL_ALLOC(P.primarySp, int, cap);
  POOL_ALLOC(P.spKd, double, 3 * icap);
  POOL_ALLOC(P.spKs, double, 3 * icap);
  POOL_ALLOC(P.spShin, double, icap);
  POOL_ALLOC(P.spKr, double, 3 * icap);
  POOL_ALLOC(P.spKt, double, 3 * icap);
  POOL_ALLOC(P.spIor, double, icap);
  POOL_ALLOC(P.counters, Counters, kCounterBlocks);
#undef POOL_ALLOC
  // (no kernel reads a hit it was not given, see tracedRays; a fresh pool still starts as "no hit" everywhere)
  if (hipMemset(P.hitPrim, 0xff, sizeof(int) * cap) != hipSuccess) {
    (void)hipGetLastError();
    sc->freePool();
    return fail(RT_ERR_DEVICE, "cannot clear the ray pool");
  }
  P.cap = cap;
  sc->poolLights = nl;
  return RT_OK;
}

size_t poolBytesPerSlot(int nLights, bool interp, bool opaque) {
  const size_t nl = (size_t)std::max(1, nLights);
  return 2 * (24 + 4 + 4) + 12 + 24 * 2 + 24 * 2 + 4 * 6 + nl * (opaque ? 4 : 12) + (interp ? 24 * 4 + 8 * 2 : 0);
}

struct TileLayout {
  int tilesX = 0, tilesY = 0, nTiles = 0;       // of this shard
  std::vector<long long> offsets;                // packed pixel offset per shard tile
  long long pixels = 0;
};

TileLayout tileLayout(int w, int h, int tw, int th, int shardIndex, int shardCount) {
  TileLayout L;
  L.tilesX = (w + tw - 1) / tw;
  L.tilesY = (h + th - 1) / th;
  const int total = L.tilesX * L.tilesY;
  for (int t = shardIndex; t < total; t += shardCount) {
    const int tx = t % L.tilesX, ty = t / L.tilesX;
    const int wt = std::min(tw, w - tx * tw), ht = std::min(th, h - ty * th);
    L.offsets.push_back(L.pixels);
    L.pixels += (long long)wt * ht;
    L.nTiles++;
  }
  return L;
}

// Device copies of packed-tile offset tables, one per (device, frame, tiling,
// shard): rendering and unpacking a frame with the same geometry again reuses
// the table instead of allocating and uploading it per call.  The tables live
// until the process ends (a 4K frame's is 64 KiB).
const long long* cachedTileOffsets(const TileLayout& TL, int w, int h, int tw, int th, int shardIndex,
                                   int shardCount) {
  static std::mutex mu;
  static std::map<std::array<int, 7>, long long*> cache;
  if (TL.offsets.empty()) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const std::array<int, 7> key{dev, w, h, tw, th, shardIndex, shardCount};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  long long* p = nullptr;
  if (hipMalloc((void**)&p, TL.offsets.size() * sizeof(long long)) != hipSuccess) return nullptr;
  if (hipMemcpy(p, TL.offsets.data(), TL.offsets.size() * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(p);
    return nullptr;
  }
  cache[key] = p;
  return p;
}

// Device allocation that lives as long as the call that made it, whichever
// way the call returns.
struct DevBuf {
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  hipError_t allocCopy(const void* host, size_t bytes) {
    hipError_t e = alloc(bytes);
    return e == hipSuccess ? hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) : e;
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

struct LevelInfo { size_t rayOff, spOff; int n, nsp; };

struct RenderCtx {
  RtScene* sc;
  const RtRenderParams* rp;
  FrameMap fm;
  unsigned char* rgbDev;
  int* hitDev;
  const long long* tileOffDev;
  RtStats stats;
  bool fast, wantStats;
  double usedSlots = 0.0, usedSamples = 0.0;   // pool slots the frame's batches needed, for their samples
};

// Blocks of `threads` for an estimated `est` items (at least one, so that a
// level the estimate takes for empty still checks its coverage).
inline unsigned int blocksFor(long long est, int threads) {
  return (unsigned int)std::max<long long>(1, (est + threads - 1) / threads);
}

template <bool FAST, bool STATS>
long long launchTrace(RtScene* sc, const Pool& P, long long est, const RaySrc& R) {
  // (Measured and dropped: giving the wavefronts of a thin level 16 or 32 rays
  // instead of 64, so that more wavefronts share its rays: 10-70 % slower.  A
  // thin level lasts as long as its longest ray, not as long as a wavefront's
  // lanes take one after the other.)
  const unsigned int blocks = blocksFor(est, kTraceBlock);
  const size_t lds = (size_t)(FAST ? RT_LDS_STACK_BYTES_CLOSEST : RT_KD_LDS_STACK_BYTES) * kTraceBlock;
  if (FAST && sc->triOnly)
    k_trace<FAST, STATS, true><<<blocks, kTraceBlock, lds, sc->stream>>>(sc->dev, R, P.hitT, P.hitPrim, P.counters);
  else
    k_trace<FAST, STATS, false><<<blocks, kTraceBlock, lds, sc->stream>>>(sc->dev, R, P.hitT, P.hitPrim, P.counters);
  return (long long)blocks * kTraceBlock;   // rays the grid covers
}
template <bool FAST, bool STATS>
void launchShadow(RtScene* sc, hipStream_t st, const Pool& P, long long est) {
  const dim3 grid(blocksFor(est, kTraceBlock), (unsigned int)sc->dev.n_lights);
  // opaque scenes: occlusion queries (4-byte stack entries) except for directional
  // lights, which use them anyway; otherwise closest-hit queries bounded by the light
  const size_t lds = (size_t)(!FAST ? RT_KD_LDS_STACK_BYTES
                                    : sc->dev.opaque ? RT_LDS_STACK_BYTES_ANYHIT : RT_LDS_STACK_BYTES_CLOSEST) * kTraceBlock;
  if (FAST && sc->triOnly)
    k_shadow<FAST, STATS, true><<<grid, kTraceBlock, lds, st>>>(sc->dev, P);
  else
    k_shadow<FAST, STATS, false><<<grid, kTraceBlock, lds, st>>>(sc->dev, P);
}

// HIP events around the traversal kernels (RtStats::ms_trace / ms_lights): only
// where somebody asked for them (RT_AMD_LAUNCH_EVENTS=1 when the scene is
// created, or a statistics pass) -- an event record between two kernels of a
// stream costs the back-to-back dispatch of the second.
bool timedBegin(RtScene* sc, hipStream_t st, bool wanted) {
  if (!wanted) return false;
  if (sc->eventsUsed + 2 > sc->events.size()) {
    for (int k = 0; k < 64; k++) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return false;
      sc->events.push_back(e);
    }
  }
  hipEventRecord(sc->events[sc->eventsUsed], st);
  return true;
}
void timedEnd(RtScene* sc, hipStream_t st, int kind) {
  hipEventRecord(sc->events[sc->eventsUsed + 1], st);
  if (sc->eventKind.size() < sc->eventsUsed / 2 + 1) sc->eventKind.resize(sc->eventsUsed / 2 + 1);
  sc->eventKind[sc->eventsUsed / 2] = (unsigned char)kind;
  sc->eventsUsed += 2;
}

// Renders pixel slots [first, first+count) of the frame map.  Returns
// RT_OK, an error, +1 when the ray pool overflowed (caller splits the batch) or
// +2 when an estimated grid was too small (caller repeats the batch with
// forceSync).
//
// The whole batch -- every recursion level, the bottom-up pass and the resolve
// -- is queued without a host round trip: a level's ray and shade-point counts
// stay on the device (Counters::n / rayOff / spOff, handed on by k_advance) and
// the grids are sized from what each level held when the same batch was
// rendered before (RtScene::levelHist).  One read-back of all counter blocks
// and ONE stream synchronisation end the batch.  A batch that is new to the
// scene handle reads each level's count back before it sizes the next, as does
// every batch under RT_AMD_LEVEL_SYNC=1.
int renderBatch(RenderCtx& cx, long long first, int count, long long* slotsUsed, bool forceSync) {
  RtScene* sc = cx.sc;
  const Pool& pool = sc->pool;
  const RtRenderParams& rp = *cx.rp;
  const int ns = cx.fm.ns;
  const long long nSamples = (long long)count * ns;
  if ((size_t)nSamples > pool.cap) return 1;
  const int nLevels = rp.depth + 1;
  const RtScene::BatchKey key{cx.fm.pixelList ? 1 : 0, first, count, ns, rp.depth, cx.fm.width, cx.fm.height,
                              cx.fm.tileW, cx.fm.tileH, cx.fm.shardIndex, cx.fm.shardCount, cx.fm.aaN};
  std::vector<unsigned int> hist;
  {
    auto it = sc->levelHist.find(key);
    if (it != sc->levelHist.end()) hist = it->second;
  }
  const bool haveHist = (int)hist.size() == nLevels;
  const bool levelSync = !haveHist || forceSync || sc->levelSyncMode == 1;
  k_begin<<<1, kBlock, 0, sc->stream>>>(pool.counters, (unsigned int)nSamples, nLevels + 1);
  // A level's shadow queries and shading depend on its shade points only, the
  // next level's rays on neither: they go to the side stream and run beside the
  // main stream's k_shade of this level and k_trace / k_classify / k_surface of
  // the next.  A thin level spends much of its time waiting for the dependent
  // loads of its longest rays, and the shadow rays of the level above fill those
  // gaps: 3 % of a 4K frame, 8 % of an eighth of it (one rank's share on 8 GPUs).
  // RT_AMD_OVERLAP=0 keeps one stream (kernel durations without a neighbour).
  const bool overlap = sc->dev.n_lights > 0 && sc->overlapMode != 0;
  hipStream_t lightStream = overlap ? sc->side : sc->stream;
  bool sidePending = false;
  int par = 0, levelsRun = 0;
  long long est = nSamples;     // rays the level is sized for (exact with levelSync)
  for (int level = 0; level < nLevels; level++) {
    if (level > 0) {
      if (levelSync) {
        HIP_TRY(hipMemcpyAsync(sc->countersHost, pool.counters + level, sizeof(Counters), hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        if (sc->countersHost[0].overflow) {
          if (sidePending) HIP_TRY(hipStreamSynchronize(sc->side));
          return (sc->countersHost[0].overflow & RT_OVERFLOW_POOL) ? 1 : 2;
        }
        est = (long long)sc->countersHost[0].n;
        if (est == 0) break;
      } else {
        // what the level held when this batch was rendered before (and a little more)
        est = (long long)hist[(size_t)level] + (long long)(hist[(size_t)level] >> 4) + 256;
      }
    }
    levelsRun = level + 1;
    sc->levelN[(size_t)level] = (unsigned int)std::min<long long>(est, 0x7fffffff);
    Pool P = pool;   // this level's view of the pool: its own counter block
    P.counters = pool.counters + level;
    RaySrc R;
    // every hit a plain triangle and nothing refracts: k_hit does what k_classify,
    // k_surface and k_shade otherwise share out, and its queues hold reflected rays only
    const bool fused = sc->fusedHit && sc->hasFaces && !sc->surfaceRest && sc->dev.opaque;
    R.parent = P.rayParent[par]; R.spQ = P.spQ; R.d = P.rayD[par]; R.type = fused ? nullptr : P.rayType[par];
    R.stride = P.cap;
    R.primary = level == 0 ? 1 : 0;
    R.fm = cx.fm;
    R.first = makeSlotBase(cx.fm, first);
    const bool timed = timedBegin(sc, sc->stream, sc->launchEvents || cx.wantStats);
    long long traced;   // rays k_trace's grid covers
    if (cx.fast) traced = cx.wantStats ? launchTrace<true, true>(sc, P, est, R) : launchTrace<true, false>(sc, P, est, R);
    else traced = cx.wantStats ? launchTrace<false, true>(sc, P, est, R) : launchTrace<false, false>(sc, P, est, R);
    if (timed) timedEnd(sc, sc->stream, 0);
    cx.stats.trace_launches++;
    // entries per thread: enough tiles to fill the chip, few enough atomics
    const int items = (int)std::min<long long>(kMaxItems, std::max<long long>(1, est / (kBlock * 2048)));
    const unsigned int tiles = blocksFor(est, kBlock * items);
    const unsigned int dense = blocksFor(est, kBlock);
    // every hit a plain triangle and nothing refracts: one kernel does what
    // k_classify, k_surface and k_shade otherwise share out
    // nothing to show behind a ray that hits nothing, and no coefficient that would turn a zero into a NaN
    const int blackMiss = (!(rp.use_cubemap && sc->dev.cube_w > 0) && sc->finiteCoefficients) ? 1 : 0;
    if (fused) {
      const int hitItems = std::min(items, kHitItems);
      k_hit<<<blocksFor(est, kBlock * hitItems), kBlock, 0, sc->stream>>>(sc->dev, hitItems, par, R, level, rp.depth - level,
                                                                        blackMiss, traced, P);
    } else {
      k_classify<<<tiles, kBlock, 0, sc->stream>>>(items, level, R, blackMiss, traced, P);
      if (sc->hasFaces) k_surface<<<dense, kBlock, 0, sc->stream>>>(sc->dev, R, level, P);
      if (sc->surfaceRest) k_surface_rest<<<dense, kBlock, 0, sc->stream>>>(sc->dev, R, level, P);
    }
    if (rp.use_cubemap && sc->dev.cube_w > 0)
      k_background<<<dense, kBlock, 0, sc->stream>>>(sc->dev, R, rp.filter_width, traced, P);
    if (overlap) {
      HIP_TRY(hipEventRecord(sc->evSurface, sc->stream));
      HIP_TRY(hipStreamWaitEvent(sc->side, sc->evSurface, 0));
    }
    if (sc->dev.n_lights > 0) {
      const bool timed2 = timedBegin(sc, lightStream, sc->launchEvents || cx.wantStats);
      if (cx.fast) { if (cx.wantStats) launchShadow<true, true>(sc, lightStream, P, est); else launchShadow<true, false>(sc, lightStream, P, est); }
      else { if (cx.wantStats) launchShadow<false, true>(sc, lightStream, P, est); else launchShadow<false, false>(sc, lightStream, P, est); }
      if (timed2) timedEnd(sc, lightStream, 1);
      cx.stats.lights_launches++;
    }
    // shade points whose shadow queries k_shadow's grid answers (no lights: no queries to wait for)
    const long long queried = sc->dev.n_lights > 0 ? (long long)blocksFor(est, kTraceBlock) * kTraceBlock
                                                   : std::numeric_limits<long long>::max();
    if (sc->intPow) k_lights<true><<<dense, kBlock, 0, lightStream>>>(sc->dev, queried, P);
    else k_lights<false><<<dense, kBlock, 0, lightStream>>>(sc->dev, queried, P);
    if (overlap) {
      HIP_TRY(hipEventRecord(sc->evSide, sc->side));
      sidePending = true;
    }
    if (fused) {
      // its rays are spawned already
    } else if (sc->dev.opaque)
      k_shade<false><<<tiles, kBlock, 0, sc->stream>>>(sc->dev, items, par, R, rp.depth - level, P);
    else
      k_shade<true><<<tiles, kBlock, 0, sc->stream>>>(sc->dev, items, par, R, rp.depth - level, P);
    k_advance<<<1, 1, 0, sc->stream>>>(P.counters, (unsigned int)pool.cap);
    par ^= 1;
  }
  // the colours of every level's shade points are needed from here on
  if (sidePending) HIP_TRY(hipStreamWaitEvent(sc->stream, sc->evSide, 0));
  for (int level = levelsRun - 1; level >= 1; level--) {   // (level 0: in k_resolve)
    Pool P = pool;
    P.counters = pool.counters + level;
    k_combine<<<blocksFor((long long)sc->levelN[(size_t)level], kBlock), kBlock, 0, sc->stream>>>(sc->dev, P);
  }
  {
    const int pitch = ns | 1;
    const int ppb = std::max(1, std::min(kBlock, kResolveSamples / pitch));
    const int blackMiss = (!(rp.use_cubemap && sc->dev.cube_w > 0) && sc->finiteCoefficients) ? 1 : 0;
    k_resolve<<<(count + ppb - 1) / ppb, kBlock, 0, sc->stream>>>(sc->dev, cx.fm, makeSlotBase(cx.fm, first), count, ppb, blackMiss,
                                                                    pool, cx.rgbDev, cx.hitDev, cx.tileOffDev);
  }
  HIP_TRY(hipGetLastError());
  // the batch's one read-back: every level's counter block, and the block the
  // last level's k_advance wrote (overflow seen anywhere ends up there)
  HIP_TRY(hipMemcpyAsync(sc->countersHost, pool.counters, sizeof(Counters) * (size_t)(levelsRun + 1), hipMemcpyDeviceToHost,
                         sc->stream));
  HIP_TRY(hipStreamSynchronize(sc->stream));
  const Counters* c = sc->countersHost;
  {
    unsigned int flags = 0;
    for (int l = 0; l <= levelsRun; l++) flags |= c[l].overflow;
    if (flags & RT_OVERFLOW_POOL) return 1;
    if (flags) return 2;
  }
  unsigned long long litTotal = 0, nodes = 0, prims = 0, objs = 0, refl = 0, refr = 0, snodes = 0, sprims = 0, sobjs = 0;
  size_t rayEnd = 0, spEnd = 0;
  hist.assign((size_t)nLevels, 0u);
  for (int l = 0; l < levelsRun; l++) {
    litTotal += c[l].litsp;
    nodes += c[l].nodes;
    prims += c[l].prims;
    objs += c[l].objs;
    sobjs += c[l].shadowObjs;
    snodes += c[l].shadowNodes;
    sprims += c[l].shadowPrims;
    refl += c[l].refl;
    refr += c[l].refr;
    cx.stats.shade_points += (unsigned long long)c[l].nsp;
    rayEnd = (size_t)c[l].rayOff + c[l].n;
    spEnd = (size_t)c[l].spOff + c[l].nsp;
    hist[(size_t)l] = c[l].n;
  }
  if (sc->levelHist.size() >= 64 && !sc->levelHist.count(key)) sc->levelHist.clear();
  sc->levelHist[key] = hist;
  // stats: dead (out-of-frame) slots are not rays
  cx.stats.rays_reflect += refl;
  cx.stats.rays_refract += refr;
  cx.stats.rays_shadow += litTotal * (unsigned long long)sc->dev.n_lights;
  cx.stats.nodes_visited += nodes + snodes;
  cx.stats.prims_tested += prims + sprims;
  cx.stats.shadow_nodes_visited += snodes;
  cx.stats.shadow_prims_tested += sprims;
  cx.stats.objects_tested += objs + sobjs;
  cx.stats.shadow_objects_tested += sobjs;
  cx.stats.batches++;
  *slotsUsed = (long long)std::max(rayEnd, spEnd);
  return RT_OK;
}

// Pixel slots per batch such that the expected number of pool slots
// (samples x growth) fits the pool.
long long batchFor(const RenderCtx& cx) {
  const RtScene* sc = cx.sc;
  const int ns = cx.fm.ns;
  long long b = (long long)((double)sc->pool.cap / (sc->growth * ns));
  if (cx.rp->max_batch_samples > 0) b = std::min<long long>(b, cx.rp->max_batch_samples / ns);
  return std::max<long long>(64, b / 64 * 64);
}

// Renders pixel slots [first, first+count) in batches.  The batch size follows
// the growth (pool slots per sample) that the batches before it really had:
// deep recursion levels hold few rays but each costs the latency of its longest
// ray, so few large batches beat many small ones.
int renderRange(RenderCtx& cx, long long first, long long count) {
  RtScene* sc = cx.sc;
  const long long end = first + count;
  long long limit = std::numeric_limits<long long>::max();
  long long b = first;
  bool forceSync = false;
  while (b < end) {
    const int c = (int)std::min(std::min(batchFor(cx), limit), end - b);
    long long used = 0;
    int rc = renderBatch(cx, b, c, &used, forceSync);
    if (rc == 2 && !forceSync) {   // a level outgrew its estimate: once more, sized level by level
      forceSync = true;
      cx.stats.batches_redone++;
      continue;
    }
    forceSync = false;
    if (rc == 2) return fail(RT_ERR_DEVICE, "a level outgrew a grid sized from its exact count");
    if (rc == 1) {
      if (c <= 64) return fail(RT_ERR_NOMEM, "ray pool too small for a 64-pixel batch");
      HIP_TRY(hipStreamSynchronize(sc->stream));
      limit = std::max<long long>(64, (c / 2 + 63) / 64 * 64);
      sc->growth = std::max(sc->growth, 1.5 * (double)sc->pool.cap / ((double)c * cx.fm.ns));
      continue;
    }
    if (rc != RT_OK) return rc;
    const double seen = 1.15 * (double)used / ((double)c * cx.fm.ns) + 0.05;
    if (!sc->growthMeasured) { sc->growth = seen; sc->growthMeasured = true; }
    else sc->growth = std::max(sc->growth, seen);
    cx.usedSlots += (double)used;
    cx.usedSamples += (double)c * cx.fm.ns;
    b += c;
  }
  return RT_OK;
}

// Bytes the ray pool may take: half of the free device memory up to 64 GiB
// (RT_AMD_WORKSPACE_MB overrides the ceiling).
size_t workspaceBudgetBytes(size_t alreadyHeld) {
  size_t mb = 65536;
  if (const char* e = std::getenv("RT_AMD_WORKSPACE_MB")) {
    long v = std::atol(e);
    if (v >= 64) mb = (size_t)v;
  }
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB > 0)
    mb = std::min(mb, ((freeB + alreadyHeld) / (1024 * 1024)) / 2);
  return mb * 1024 * 1024;
}

// Sizes (and if needed reallocates) the ray pool for a frame of `slots` samples.
int sizePool(RtScene* sc, const RtRenderParams* rp, long long slots) {
  if (sc->growthDepth != rp->depth) {
    // a chain of rays per bounce, a tree where surfaces also refract
    sc->growth = (double)std::min<long long>((1ll << std::min(rp->depth + 1, 20)) - 1, 6);
    sc->growthMeasured = false;
    sc->growthDepth = rp->depth;
  }
  const bool interp = sc->dev.interp_materials != 0;
  const bool opaque = sc->dev.opaque != 0;
  const size_t perSlot = poolBytesPerSlot(sc->nLights, interp, opaque);
  const size_t want = (size_t)std::max<double>((double)slots * sc->growth * 1.1, 1 << 16);
  // the pool in hand already holds what the frame wants and is not grossly larger: no question to the driver
  if (rp->max_batch_samples <= 0 && sc->pool.cap >= std::min<size_t>(want, 0x3fffffff) && sc->pool.cap <= 4 * want &&
      sc->poolLights >= (size_t)std::max(1, sc->nLights))
    return RT_OK;
  size_t cap = workspaceBudgetBytes(sc->pool.cap * poolBytesPerSlot((int)sc->poolLights, interp, opaque)) / perSlot;
  cap = std::min(cap, want);
  cap = std::min<size_t>(cap, 0x3fffffff);
  if (rp->max_batch_samples > 0) cap = std::min<size_t>(cap, (size_t)rp->max_batch_samples * 8);
  cap = std::max<size_t>(cap, 1 << 14);
  if (sc->pool.cap < cap || sc->pool.cap > 4 * cap) return ensurePool(sc, cap);
  return RT_OK;
}

// Device copy of the caller's jitter table (2 * ns doubles).
int uploadJitter(RtScene* sc, const RtRenderParams* rp, int ns) {
  if (ns <= 1 || !rp->jitter) return RT_OK;
  const size_t need = 2 * (size_t)ns;
  if (need > sc->jitterCap) {
    if (sc->jitterDev) hipFree(sc->jitterDev);
    sc->jitterDev = nullptr;
    sc->jitterCap = 0;
    sc->jitterHost.clear();
    HIP_TRY(hipMalloc((void**)&sc->jitterDev, need * sizeof(double)));
    sc->jitterCap = need;
  }
  // (a caller renders frame after frame with one table: copy it when it differs from the one last copied)
  if (sc->jitterHost.size() == need && std::memcmp(sc->jitterHost.data(), rp->jitter, need * sizeof(double)) == 0) return RT_OK;
  HIP_TRY(hipStreamSynchronize(sc->stream));   // the staging vector below may still be read by an earlier copy
  sc->jitterHost.assign(rp->jitter, rp->jitter + need);
  HIP_TRY(hipMemcpyAsync(sc->jitterDev, sc->jitterHost.data(), need * sizeof(double), hipMemcpyHostToDevice, sc->stream));
  return RT_OK;
}

// Grows a device buffer that lives in the scene handle (contents are not kept).
template <class T>
int growBuffer(T** buf, size_t* cap, size_t need) {
  if (*cap >= need) return RT_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *cap = 0;
  const size_t take = need + need / 4;
  const hipError_t e = hipMalloc((void**)buf, take * sizeof(T));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, "cannot allocate a work buffer");
  }
  *cap = take;
  return RT_OK;
}

// GraphicalUI::applyFilter + the walk of doAntiAliasing over its mask
// (src/ui/GraphicalUI.cpp:437-486,405-425) for a base image in device memory:
// the (x, y) list of the flagged pixels, row by row, in RtScene::flagList.
// Mask, counts and list stay on the device, in buffers the handle keeps; the
// host learns one number -- how many pixels were flagged -- through mapped host
// memory, because the second pass is sized by it.
int buildFlagList(RtScene* sc, const unsigned char* rgbDev, int w, int h, int cutoff) {
  const long long npx = (long long)w * h;
  const int nBlocks = gridFor(npx);
  sc->flagCount = 0;
  int rc = growBuffer(&sc->flagMask, &sc->flagMaskCap, (size_t)npx);
  if (rc != RT_OK) return rc;
  rc = growBuffer(&sc->flagRuns, &sc->flagRunsCap, (size_t)nBlocks);
  if (rc != RT_OK) return rc;
  unsigned int* totalHost = reinterpret_cast<unsigned int*>(sc->countersHost);   // pinned, device-visible
  *totalHost = 0;
  k_edge_mask<<<nBlocks, kBlock, 0, sc->stream>>>(rgbDev, w, h, cutoff, sc->flagMask);
  k_flag_count<<<nBlocks, kBlock, 0, sc->stream>>>(sc->flagMask, npx, sc->flagRuns);
  k_flag_scan<<<1, 1024, 0, sc->stream>>>(sc->flagRuns, nBlocks, totalHost);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(sc->stream));
  const long long nFlag = (long long)*totalHost;
  if (nFlag > 0) {
    rc = growBuffer(&sc->flagList, &sc->flagListCap, (size_t)nFlag * 2);
    if (rc != RT_OK) return rc;
    k_flag_write<<<nBlocks, kBlock, 0, sc->stream>>>(sc->flagMask, w, npx, sc->flagRuns, sc->flagList);
    HIP_TRY(hipGetLastError());
  }
  sc->flagCount = nFlag;
  sc->flagW = w;
  sc->flagH = h;
  return RT_OK;
}

// Pixel -> image-plane coordinate tables of a frame size (FrameMap::colX / rowY).
int uploadPixelCoordinates(RtScene* sc, int w, int h, FrameMap& fm) {
  if (sc->pixelXYw != w || sc->pixelXYh != h) {
    HIP_TRY(hipStreamSynchronize(sc->stream));   // nothing in flight reads the old tables
    if (sc->pixelXY) hipFree(sc->pixelXY);
    sc->pixelXY = nullptr;
    sc->pixelXYw = sc->pixelXYh = 0;
    sc->pixelXYHost.resize((size_t)w + h);
    for (int i = 0; i < w; i++) sc->pixelXYHost[(size_t)i] = double(i) / double(w);
    for (int j = 0; j < h; j++) sc->pixelXYHost[(size_t)w + j] = double(j) / double(h);
    HIP_TRY(hipMalloc((void**)&sc->pixelXY, sc->pixelXYHost.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(sc->pixelXY, sc->pixelXYHost.data(), sc->pixelXYHost.size() * sizeof(double), hipMemcpyHostToDevice));
    sc->pixelXYw = w;
    sc->pixelXYh = h;
  }
  fm.colX = sc->pixelXY;
  fm.rowY = sc->pixelXY + w;
  return RT_OK;
}

}  // namespace

// ================================================================== C-ABI
extern "C" {

int rt_abi_version(void) { return RT_AMD_ABI_VERSION; }

const char* rt_last_error(void) { return g_lastError.c_str(); }

int rt_scene_create(const RtSceneDesc* d, RtScene** out) {
  if (!d || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  if (d->abi_version != RT_AMD_ABI_VERSION) return fail(RT_ERR_INVALID, "ABI version mismatch");
  if (d->n_objects < 0 || d->n_lights < 0 || d->n_meshes < 0 || d->n_faces < 0 || d->n_nodes < 0 ||
      d->n_materials < 0 || d->n_vertices < 0 || d->n_leaf_items < 0)
    return fail(RT_ERR_INVALID, "negative count in scene description");
  if (d->n_lights > 65535) return fail(RT_ERR_UNSUPPORTED, "more than 65535 lights");   // the shadow kernel's grid has one row per light
  // an array that a count says is there has to be there
  if ((d->n_objects > 0 && !d->objects) || (d->n_lights > 0 && !d->lights) || (d->n_meshes > 0 && !d->meshes) ||
      (d->n_materials > 0 && !d->materials) || (d->n_vertices > 0 && !d->vertices) || (d->n_faces > 0 && !d->faces) ||
      (d->n_leaf_items > 0 && !d->leaf_items) ||
      (d->n_nodes > 0 && (!d->node_split || !d->node_dim || !d->node_left || !d->node_right || !d->node_item_start ||
                          !d->node_item_count)))
    return fail(RT_ERR_INVALID, "null array with a non-zero count in scene description");
  // ---- validate indices so that no kernel can read out of bounds
  for (int i = 0; i < d->n_objects; i++) {
    const RtObject& o = d->objects[i];
    if (o.kind < RT_OBJ_SPHERE || o.kind > RT_OBJ_TRIMESH) return fail(RT_ERR_INVALID, "bad object kind");
    if (o.material < 0 || o.material >= d->n_materials) return fail(RT_ERR_INVALID, "object material out of range");
    if (o.kind == RT_OBJ_TRIMESH && (o.mesh < 0 || o.mesh >= d->n_meshes))
      return fail(RT_ERR_INVALID, "object mesh index out of range");
  }
  for (int i = 0; i < d->n_lights; i++)
    if (d->lights[i].type < RT_LIGHT_DIRECTIONAL || d->lights[i].type > RT_LIGHT_SPOT)
      return fail(RT_ERR_INVALID, "bad light type");
  std::string err;
  int sceneDepth = 0, meshDepth = 0;
  if (d->use_kdtree && d->scene_kd_root >= 0) {
    sceneDepth = treeDepth(*d, d->scene_kd_root, d->n_objects, err);
    if (sceneDepth < 0) return fail(RT_ERR_INVALID, err);
  }
  for (int m = 0; m < d->n_meshes; m++) {
    const RtMesh& M = d->meshes[m];
    if (M.object < 0 || M.object >= d->n_objects || d->objects[M.object].mesh != m)
      return fail(RT_ERR_INVALID, "mesh/object cross reference broken");
    if (M.first_vertex < 0 || M.n_vertices < 0 || (long long)M.first_vertex + M.n_vertices > d->n_vertices ||
        M.first_face < 0 || M.n_faces < 0 || (long long)M.first_face + M.n_faces > d->n_faces)
      return fail(RT_ERR_INVALID, "mesh vertex/face range out of bounds");
    if (M.has_vertex_normals && !d->vnormals) return fail(RT_ERR_INVALID, "vertex normals missing");
    if (M.has_vertex_materials && !d->vmaterials) return fail(RT_ERR_INVALID, "vertex materials missing");
    for (int f = 0; f < M.n_faces; f++)
      for (int k = 0; k < 3; k++) {
        const int id = d->faces[M.first_face + f].ids[k];
        if (id < 0 || id >= M.n_vertices) return fail(RT_ERR_INVALID, "face vertex index out of range");
      }
    if (M.has_vertex_materials)
      for (int v = 0; v < M.n_vertices; v++) {
        const int mi = d->vmaterials[M.first_vertex + v];
        if (mi < 0 || mi >= d->n_materials) return fail(RT_ERR_INVALID, "vertex material out of range");
      }
    if (d->use_kdtree && M.kd_root >= 0) {
      // leaf items of a mesh tree must be faces of that mesh
      int dep = treeDepth(*d, M.kd_root, d->n_faces, err);
      if (dep < 0) return fail(RT_ERR_INVALID, err);
      meshDepth = std::max(meshDepth, dep);
    }
  }
  if (sceneDepth + meshDepth + 4 > RT_STACK_DEPTH)
    return fail(RT_ERR_UNSUPPORTED, "kd trees deeper than the traversal stack allows");

  std::unique_ptr<RtScene> sc(new RtScene);
  std::memset(&sc->dev, 0, sizeof sc->dev);
  std::memset(&sc->pool, 0, sizeof sc->pool);
  HIP_TRY(hipGetDevice(&sc->device));
  {
    // the static staging array's address on this device (rt_intersect_batch_static)
    unsigned char* staging = nullptr;
    HIP_TRY(scratchOfCurrentDevice(sc->device, &staging));
  }
  // Both streams at the default priority.  (Measured: a higher priority for the
  // main stream, which carries the chain of levels a frame cannot finish
  // before, changes nothing for a handle that renders alone -- and when two
  // handles render on one device at a time, frames in flight on host threads
  // of their own, a low-priority side stream starves behind the other handle's
  // main stream until that handle's frame is over: the frames then follow each
  // other instead of filling each other's thin levels, 3.98 instead of 3.22 ms
  // per frame for an eighth of the 4K benchmark frame.)
  HIP_TRY(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&sc->side, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&sc->evSurface, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&sc->evSide, hipEventDisableTiming));
  if (const char* e = std::getenv("RT_AMD_OVERLAP")) sc->overlapMode = std::atoi(e) != 0 ? 1 : 0;
  HIP_TRY(hipHostMalloc((void**)&sc->countersHost, sizeof(Counters) * kCounterBlocks, hipHostMallocDefault));
  if (const char* e = std::getenv("RT_AMD_LEVEL_SYNC")) sc->levelSyncMode = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("RT_AMD_FUSED_HIT")) sc->fusedHit = std::atoi(e) != 0;
  if (const char* e = std::getenv("RT_AMD_LAUNCH_EVENTS")) sc->launchEvents = std::atoi(e) != 0;
  HIP_TRY(hipEventCreate(&sc->evStart));
  HIP_TRY(hipEventCreate(&sc->evStop));
  DevScene& D = sc->dev;
  auto keep = [&](void* p) { if (p) sc->allocs.push_back(p); };

  // nodes
  {
    std::vector<DevNode> nodes((size_t)d->n_nodes);
    for (int n = 0; n < d->n_nodes; n++) {
      DevNode& N = nodes[n];
      N.split = d->node_split[n];
      if (d->node_left[n] < 0 && d->node_right[n] < 0) {
        N.a = ((uint32_t)d->node_item_start[n] << 2) | 3u;
        N.b = (uint32_t)d->node_item_count[n];
      } else {
        N.a = ((uint32_t)d->node_left[n] << 2) | (uint32_t)(d->node_dim[n] & 3);
        N.b = (uint32_t)d->node_right[n];
      }
    }
    DevNode* p; HIP_TRY(upload(nodes.data(), nodes.size(), &p)); keep(p); D.nodes = p;
    int* it; HIP_TRY(upload(d->leaf_items, (size_t)d->n_leaf_items, &it)); keep(it); D.items = it;
  }
  // triangles with the ray-independent terms of TrimeshFace::intersectLocal
  {
    std::vector<char> ownerIdentity((size_t)d->n_objects, 0);
    for (int i = 0; i < d->n_objects; i++) {
      bool id = true;
      for (int k = 0; k < 16; k++) id = id && d->objects[i].inverse[k] == ((k % 5 == 0) ? 1.0 : 0.0);
      ownerIdentity[i] = id;
      if (d->objects[i].kind != RT_OBJ_TRIMESH || !id) sc->triOnly = false;
    }
    std::vector<DevTri> tris((size_t)d->n_faces);
    std::vector<int> ids((size_t)d->n_faces * 3);
    for (int m = 0; m < d->n_meshes; m++) {
      const RtMesh& M = d->meshes[m];
      for (int f = 0; f < M.n_faces; f++) {
        const int fi = M.first_face + f;
        const RtFace& F = d->faces[fi];
        DevTri& T = tris[fi];
        const double* vb = d->vertices + 3 * (size_t)M.first_vertex;
        for (int k = 0; k < 3; k++) {
          T.v0[k] = vb[3 * F.ids[0] + k];
          T.v1[k] = vb[3 * F.ids[1] + k];
          T.v2[k] = vb[3 * F.ids[2] + k];
          T.n[k] = F.normal[k];
          ids[3 * fi + k] = M.first_vertex + F.ids[k];
        }
        T.dconst = -(T.v0[0] * T.n[0] + T.v0[1] * T.n[1] + T.v0[2] * T.n[2]);
        const double na0 = std::fabs(T.n[0]), na1 = std::fabs(T.n[1]), na2 = std::fabs(T.n[2]);
        int ax, ay;
        if (na0 >= na1 && na0 >= na2) { ax = 1; ay = 2; }
        else if (na1 >= na0 && na1 >= na2) { ax = 0; ay = 2; }
        else { ax = 0; ay = 1; }
        T.axes = ax | (ay << 2) | (ownerIdentity[M.object] ? 16 : 0);
        const double z = (T.v1[ax] - T.v0[ax]) * (T.v2[ay] - T.v0[ay]) -
                         (T.v1[ay] - T.v0[ay]) * (T.v2[ax] - T.v0[ax]);
        T.abc = std::fabs(z) / 2;
        T.object_id = F.object_id;
        T.owner = M.object;
        T.face = fi;
      }
    }
    DevTri* p; HIP_TRY(upload(tris.data(), tris.size(), &p)); keep(p); D.tris = p;
    int* q; HIP_TRY(upload(ids.data(), ids.size(), &q)); keep(q); D.face_ids = q;
    // the three vertex normals of a face side by side: one 72-byte gather per
    // hit instead of an index triple and three scattered normals
    D.face_normals = nullptr;
    if (d->smooth_shading && d->vnormals) {
      bool any = false;
      for (int mi = 0; mi < d->n_meshes; mi++) any = any || d->meshes[mi].has_vertex_normals;
      if (any) {
        std::vector<double> fn(9 * (size_t)d->n_faces, 0.0);
        for (int mi = 0; mi < d->n_meshes; mi++) {
          const RtMesh& M = d->meshes[mi];
          if (!M.has_vertex_normals) continue;
          for (int f = 0; f < M.n_faces; f++) {
            const size_t fi = (size_t)M.first_face + f;
            for (int k = 0; k < 3; k++)
              for (int a = 0; a < 3; a++) fn[9 * fi + 3 * k + a] = d->vnormals[3 * (size_t)ids[3 * fi + k] + a];
          }
        }
        double* fp; HIP_TRY(upload(fn.data(), fn.size(), &fp)); keep(fp); D.face_normals = fp;
      }
    }

    // ---- the library's own BVH over world-space triangle and object boxes
    if ((size_t)d->n_faces >= (1u << 27)) return fail(RT_ERR_UNSUPPORTED, "too many triangles for the BVH leaf encoding");
    std::vector<rtbvh::Prim> prims;
    prims.reserve((size_t)d->n_faces + d->n_objects);
    for (int oi = 0; oi < d->n_objects; oi++) {
      const RtObject& o = d->objects[oi];
      if (o.kind != RT_OBJ_TRIMESH) {
        rtbvh::Prim pr;
        for (int a = 0; a < 3; a++) {
          pr.box.lo[a] = o.bbox_min[a];
          pr.box.hi[a] = o.bbox_max[a];
          pr.c[a] = 0.5 * (o.bbox_min[a] + o.bbox_max[a]);
        }
        pr.id = rtbvh::kObjectBit | oi;
        prims.push_back(pr);
        continue;
      }
      const RtMesh& M = d->meshes[o.mesh];
      double xf[16];
      if (!ownerIdentity[oi]) invert4(o.inverse, xf);
      for (int f = 0; f < M.n_faces; f++) {
        const DevTri& T = tris[M.first_face + f];
        rtbvh::Prim pr;
        pr.box.reset();
        const double* vs[3] = {T.v0, T.v1, T.v2};
        for (int k = 0; k < 3; k++) {
          double w[3] = {vs[k][0], vs[k][1], vs[k][2]};
          if (!ownerIdentity[oi])
            for (int r = 0; r < 3; r++)
              w[r] = xf[4 * r] * vs[k][0] + xf[4 * r + 1] * vs[k][1] + xf[4 * r + 2] * vs[k][2] + xf[4 * r + 3];
          pr.box.grow(w);
        }
        if (!ownerIdentity[oi]) {
          // the forward transform is a numerical inverse of the given inverse: pad
          for (int a = 0; a < 3; a++) {
            const double e = 1e-9 * (std::fabs(pr.box.lo[a]) + std::fabs(pr.box.hi[a]) + 1e-30);
            pr.box.lo[a] -= e;
            pr.box.hi[a] += e;
          }
        }
        for (int a = 0; a < 3; a++) pr.c[a] = 0.5 * (pr.box.lo[a] + pr.box.hi[a]);
        pr.id = M.first_face + f;
        prims.push_back(pr);
      }
    }
    rtbvh::Builder bvh;
    rtbvh::Collapser wide;
    if (const char* e = std::getenv("RT_AMD_BVH_LEAF")) {   // tuning knob, 1..8 triangles per leaf
      const int v = std::atoi(e);
      if (v >= 1 && v <= 8) bvh.maxLeaf = v;
    }
    D.bvh_valid = bvh.build(prims) ? 1 : 0;
    if (D.bvh_valid) {
      wide.run(bvh.nodes);
      if (3 * wide.maxDepth + 2 > RT_BVH_STACK) {
        // pathological SAH depth: rebuild with median splits (depth <= log2 n + 1)
        bvh.balanced = true;
        bvh.build(prims);
        wide.run(bvh.nodes);
        if (3 * wide.maxDepth + 2 > RT_BVH_STACK)
          return fail(RT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack allows");
      }
      std::vector<DevTri> ordered(bvh.triOrder.size());
      for (size_t i = 0; i < ordered.size(); i++) {
        if (bvh.triOrder[i] & rtbvh::kObjectBit) std::memset(&ordered[i], 0, sizeof(DevTri));
        else ordered[i] = tris[bvh.triOrder[i]];
      }
      // Unused child slots of a node point at a leaf of one null triangle (normal
      // zero: every ray is parallel to it and rejected), so that the traversal
      // needs no "is this slot used" test next to the slab test of its
      // inverted box.
      if (!addNullLeaf(ordered, wide.nodes)) return fail(RT_ERR_UNSUPPORTED, "too many triangles for the BVH leaf encoding");
      DevTri* bt; HIP_TRY(upload(ordered.data(), ordered.size(), &bt)); keep(bt); D.bvhTris = bt;
      if (sc->triOnly) {
        std::vector<DevTriFast> fastTris(ordered.size());
        for (size_t i = 0; i < ordered.size(); i++) {
          const DevTri& T = ordered[i];
          DevTriFast& F = fastTris[i];
          const int ax = T.axes & 3, ay = (T.axes >> 2) & 3;
          for (int a = 0; a < 3; a++) F.n[a] = T.n[a];
          F.dconst = T.dconst;
          F.a0 = T.v0[ax]; F.a1 = T.v0[ay];
          F.b0 = T.v1[ax]; F.b1 = T.v1[ay];
          F.c0 = T.v2[ax]; F.c1 = T.v2[ay];
          F.abc2 = T.abc * 2;            // exact
          F.axes = (ax == 1 ? 1 : 0) | (ay == 2 ? 2 : 0);
          F.face = T.face;
        }
        DevTriFast* ft; HIP_TRY(upload(fastTris.data(), fastTris.size(), &ft)); keep(ft); D.bvhTrisFast = ft;
      }
      static_assert(sizeof(rtbvh::Node4) == sizeof(DevBvhNode), "BVH node layouts differ");
      DevBvhNode* bn;
      HIP_TRY(upload(reinterpret_cast<const DevBvhNode*>(wide.nodes.data()), wide.nodes.size(), &bn));
      keep(bn);
      D.bvh = bn;
      D.bvhF = nullptr;
      D.bvhLaneF = 0;
      if (sc->triOnly) {
        // (only the kernels of triangle-only scenes read this form: 1 KiB per
        // node, sixteen times the quantised node array)
        // the same child boxes decoded to binary32 planes, outward rounded
        static_assert(sizeof(DevBvhNodeF) == RT_BVHF_BYTES, "float BVH node size");
        std::vector<DevBvhNodeF> fn;
        {
          bool anyCone = false;
          buildNodeCopies(wide.nodes, wide.tight, ordered, fn, anyCone);
          D.bvhCones = anyCone ? 1 : 0;
          if (const char* e = std::getenv("RT_AMD_CONES")) {   // 0: off, 1: all rays but camera rays, 2: camera rays too
            const int v = std::atoi(e);
            if (v >= 0 && v <= 2 && D.bvhCones) D.bvhCones = v;
          }
        }
        DevBvhNodeF* fp;
        HIP_TRY(upload(fn.data(), fn.size(), &fp));
        keep(fp);
        D.bvhF = fp;
        D.bvhLaneF = fn.size() * sizeof(DevBvhNodeF) < (size_t)0x7fffffff ? 1 : 0;
        if (const char* e = std::getenv("RT_AMD_LANE_F32")) { if (std::atoi(e) == 0) D.bvhLaneF = 0; }
      }
      for (int a = 0; a < 3; a++) {
        D.bvh_bb_min[a] = bvh.rootBox.lo[a];
        D.bvh_bb_max[a] = bvh.rootBox.hi[a];
        D.bvh_bb_f[a] = rtbvh::down(bvh.rootBox.lo[a]);
        D.bvh_bb_f[3 + a] = rtbvh::up(bvh.rootBox.hi[a]);
      }
      // 255 * scale + |origin| of every node, bounded from the nodes themselves
      {
        double g[3] = {0.0, 0.0, 0.0};
        for (int a = 0; a < 3; a++) g[a] = std::max(std::fabs(bvh.rootBox.lo[a]), std::fabs(bvh.rootBox.hi[a]));
        for (const rtbvh::Node4& n4 : wide.nodes) {
          const double sc[3] = {n4.scaleX, n4.scaleY, n4.scaleZ};
          for (int a = 0; a < 3; a++) g[a] = std::max(g[a], 255.0 * sc[a] + std::fabs((double)n4.origin[a]));
        }
        for (int a = 0; a < 3; a++) D.bvhErrG[a] = rtbvh::up(g[a] * (1.0 + 1e-6));
      }
    }
  }
  { RtObject* p; HIP_TRY(upload(d->objects, (size_t)d->n_objects, &p)); keep(p); D.objects = p; }
  { RtMesh* p; HIP_TRY(upload(d->meshes, (size_t)d->n_meshes, &p)); keep(p); D.meshes = p; }
  { RtMaterial* p; HIP_TRY(upload(d->materials, (size_t)d->n_materials, &p)); keep(p); D.materials = p; }
  { RtLight* p; HIP_TRY(upload(d->lights, (size_t)d->n_lights, &p)); keep(p); D.lights = p; }
  {
    std::vector<double> cone(4 * (size_t)std::max(1, d->n_lights), 0.0);
    for (int li = 0; li < d->n_lights; li++) {
      // outside: angle > atten + hair  <=>  x < cos(that); inside: angle < atten - hair.
      // The hair (1.7e-11 rad) is far above the rounding of cos and acos as long
      // as the cone's edge is not within 0.01 degrees of 0 or 180, where acos is
      // steep; such cones (and angles outside [0, 180]) keep the acos for every x.
      const double a = d->lights[li].atten_angle, hair = 1e-9;
      const bool plain = a > 0.01 && a < 179.99;
      cone[4 * li] = plain ? std::cos((a + hair) * M_PI / 180.0) : -2.0;
      cone[4 * li + 1] = plain ? std::cos((a - hair) * M_PI / 180.0) : 2.0;
      if (d->lights[li].type == RT_LIGHT_SPOT) {
        int n;
        double delta;
        if (nearSmallInt(d->lights[li].fall_rate, n, delta)) { cone[4 * li + 2] = n; cone[4 * li + 3] = delta; }
        else sc->intPow = false;
      }
    }
    double* p; HIP_TRY(upload(cone.data(), cone.size(), &p)); keep(p); D.spotCone = p;
  }
  {
    std::vector<DevMat> dm((size_t)d->n_materials);
    for (int m = 0; m < d->n_materials; m++) {
      const RtMaterial& M = d->materials[m];
      DevMat& o = dm[m];
      std::memset(&o, 0, sizeof o);
      auto nz = [](const double* v) { return !(v[0] == 0.0 && v[1] == 0.0 && v[2] == 0.0); };
      for (int k = 0; k < 3; k++) {
        o.kd[k] = M.kd[k]; o.ks[k] = M.ks[k]; o.kr[k] = M.kr[k]; o.kt[k] = M.kt[k];
        const double amb = M.ka[k] * d->ambient[k];
        o.base[k] = M.ke[k] + amb;
      }
      const double si = (0.299 * M.shininess[0]) + (0.587 * M.shininess[1]) + (0.114 * M.shininess[2]);
      o.shin = M.shininess_mapped ? 128.0 * si : si;
      o.ior = (0.299 * M.index[0]) + (0.587 * M.index[1]) + (0.114 * M.index[2]);
      for (int k = 0; k < 3; k++)
        if (!std::isfinite(M.kr[k]) || !std::isfinite(M.kt[k])) sc->finiteCoefficients = false;
      if (!nearSmallInt(o.shin, o.shinN, o.shinD)) {
        o.shinN = -1;
        o.shinD = 0.0;
        if (nz(M.ks)) sc->intPow = false;   // (a material without a highlight never reaches pow)
      }
      o.flags = (nz(M.kd) ? RT_MATF_KD : 0) | (nz(M.ks) ? RT_MATF_KS : 0) | (nz(M.kr) ? RT_MATF_KR : 0) |
                (nz(M.kt) ? RT_MATF_KT : 0);
    }
    DevMat* p;
    HIP_TRY(upload(dm.data(), dm.size(), &p));
    keep(p);
    D.devMats = p;
  }
  D.interp_materials = 0;
  if (d->smooth_shading)
    for (int m = 0; m < d->n_meshes; m++)
      if (d->meshes[m].has_vertex_materials) D.interp_materials = 1;
  if (D.interp_materials) sc->intPow = false;   // interpolated shininess is anything
  if (const char* e = std::getenv("RT_AMD_INT_POW")) { if (std::atoi(e) == 0) sc->intPow = false; }
  D.faceInfo = nullptr;
  if (d->n_faces > 0) {
    if (d->n_materials > RT_FI_MATERIAL) return fail(RT_ERR_UNSUPPORTED, "too many materials");
    std::vector<int> info((size_t)d->n_faces, 0);
    for (int mi = 0; mi < d->n_meshes; mi++) {
      const RtMesh& M = d->meshes[mi];
      const RtMaterial& mat = d->materials[d->objects[M.object].material];
      auto nz = [](const double* v) { return !(v[0] == 0.0 && v[1] == 0.0 && v[2] == 0.0); };
      const int word = (int)((unsigned int)d->objects[M.object].material | ((nz(mat.kd) || nz(mat.ks)) ? RT_FI_LIT : 0) |
                             (M.has_vertex_normals ? RT_FI_VNORMALS : 0) | (M.has_vertex_materials ? RT_FI_VMATERIALS : 0) |
                             (nz(mat.kr) ? RT_FI_REFLECTS : 0u));
      for (int f = 0; f < M.n_faces; f++) info[(size_t)M.first_face + f] = word;
    }
    int* p;
    HIP_TRY(upload(info.data(), info.size(), &p));
    keep(p);
    D.faceInfo = p;
  }
  if (d->vnormals) { double* p; HIP_TRY(upload(d->vnormals, 3 * (size_t)d->n_vertices, &p)); keep(p); D.vnormals = p; }
  if (d->vmaterials) { int* p; HIP_TRY(upload(d->vmaterials, (size_t)d->n_vertices, &p)); keep(p); D.vmaterials = p; }
  D.cube_w = D.cube_h = 0;
  if (d->cube_w > 0 && d->cube_h > 0 && d->cube_rgb[0]) {
    for (int f = 0; f < 6; f++) {
      if (!d->cube_rgb[f]) return fail(RT_ERR_INVALID, "cube map needs six faces");
      unsigned char* p;
      HIP_TRY(upload(d->cube_rgb[f], (size_t)d->cube_w * d->cube_h * 3, &p));
      keep(p);
      D.cube[f] = p;
    }
    D.cube_w = d->cube_w;
    D.cube_h = d->cube_h;
  }
  for (int k = 0; k < 3; k++) {
    D.eye[k] = d->eye[k]; D.look[k] = d->look[k]; D.u[k] = d->u[k]; D.v[k] = d->v[k];
    D.ambient[k] = d->ambient[k];
    D.scene_bb_min[k] = d->scene_bb_min[k];
    D.scene_bb_max[k] = d->scene_bb_max[k];
  }
  D.opaque = 1;
  for (int m = 0; m < d->n_materials; m++)
    if (d->materials[m].kt[0] != 0 || d->materials[m].kt[1] != 0 || d->materials[m].kt[2] != 0) D.opaque = 0;
  D.n_objects = d->n_objects;
  D.n_lights = d->n_lights;
  D.scene_root = d->scene_kd_root;
  D.use_kd = d->use_kdtree;
  D.cull = d->backface_cull;
  D.smooth = d->smooth_shading;
  sc->nLights = d->n_lights;
  sc->hasFaces = d->n_faces > 0;
  sc->surfaceRest = D.interp_materials != 0;
  for (int oi = 0; oi < d->n_objects; oi++)
    if (d->objects[oi].kind != RT_OBJ_TRIMESH) sc->surfaceRest = true;
  *out = sc.release();
  return RT_OK;
}

void rt_scene_destroy(RtScene* scene) {
  if (!scene) return;
  int cur = 0;
  hipGetDevice(&cur);
  hipSetDevice(scene->device);
  delete scene;
  hipSetDevice(cur);
}

long long rt_shard_pixels(int width, int height, int tile_w, int tile_h, int shard_index,
                          int shard_count) {
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || shard_count <= 0 ||
      shard_index < 0 || shard_index >= shard_count)
    return -1;
  return tileLayout(width, height, tile_w, tile_h, shard_index, shard_count).pixels;
}

int rt_render(RtScene* sc, const RtRenderParams* rp, unsigned char* rgb8_out, int* hit_id_out,
              RtStats* stats_out) {
  if (!sc || !rp || !rgb8_out) return fail(RT_ERR_INVALID, "null argument");
  if (rp->width <= 0 || rp->height <= 0 || rp->depth < 0) return fail(RT_ERR_INVALID, "bad frame size or depth");
  if (rp->depth > kMaxDepth) return fail(RT_ERR_UNSUPPORTED, "recursion depth above 126");
  if (rp->tile_w <= 0 || rp->tile_h <= 0 || (rp->tile_w & 7) || (rp->tile_h & 7))
    return fail(RT_ERR_INVALID, "tile_w and tile_h must be positive multiples of 8");
  if (rp->shard_count <= 0 || rp->shard_index < 0 || rp->shard_index >= rp->shard_count)
    return fail(RT_ERR_INVALID, "bad shard index/count");
  if (rp->aa_mode != RT_AA_NONE && (rp->aa_samples < 1 || rp->aa_samples > 8))
    return fail(RT_ERR_INVALID, "aa_samples must be 1..8");
  if (rp->aa_mode == RT_AA_ADAPTIVE && (rp->shard_count != 1 || rp->out_is_device))
    return fail(RT_ERR_UNSUPPORTED, "adaptive anti-aliasing needs the whole frame on one device");
  if (rp->traversal != RT_TRAVERSE_REFERENCE && rp->traversal != RT_TRAVERSE_FAST)
    return fail(RT_ERR_INVALID, "bad traversal mode");
  const auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(sc->device));

  const int w = rp->width, h = rp->height;
  TileLayout TL = tileLayout(w, h, rp->tile_w, rp->tile_h, rp->shard_index, rp->shard_count);
  const int nsFull = (rp->aa_mode == RT_AA_NONE) ? 1 : rp->aa_samples * rp->aa_samples;

  {
    const int rc = sizePool(sc, rp, (long long)TL.nTiles * rp->tile_w * rp->tile_h * nsFull);
    if (rc != RT_OK) return rc;
  }
  const long long* tileOffsets = nullptr;
  if (rp->out_is_device && !TL.offsets.empty()) {
    tileOffsets = cachedTileOffsets(TL, w, h, rp->tile_w, rp->tile_h, rp->shard_index, rp->shard_count);
    if (!tileOffsets) return fail(RT_ERR_NOMEM, "cannot allocate the tile offset table");
  }
  {
    const int rc = uploadJitter(sc, rp, nsFull);
    if (rc != RT_OK) return rc;
  }
  if (!rp->out_is_device) {
    const size_t px = (size_t)w * h;
    if (px > sc->framePixels) {
      if (sc->frameRgb) hipFree(sc->frameRgb);
      if (sc->frameHit) hipFree(sc->frameHit);
      sc->frameRgb = nullptr; sc->frameHit = nullptr; sc->framePixels = 0;
      HIP_TRY(hipMalloc((void**)&sc->frameRgb, px * 3));
      HIP_TRY(hipMalloc((void**)&sc->frameHit, px * sizeof(int)));
      sc->framePixels = px;
    }
    HIP_TRY(hipMemsetAsync(sc->frameRgb, 0, px * 3, sc->stream));
    HIP_TRY(hipMemsetAsync(sc->frameHit, 0, px * sizeof(int), sc->stream));
  }

  RenderCtx cx;
  cx.sc = sc;
  cx.rp = rp;
  std::memset(&cx.stats, 0, sizeof cx.stats);
  cx.fast = rp->traversal == RT_TRAVERSE_FAST;
  cx.wantStats = rp->collect_stats != 0;
  cx.rgbDev = rp->out_is_device ? rgb8_out : sc->frameRgb;
  cx.hitDev = rp->out_is_device ? hit_id_out : sc->frameHit;
  cx.tileOffDev = tileOffsets;
  FrameMap& fm = cx.fm;
  fm.width = w; fm.height = h;
  fm.tileW = rp->tile_w; fm.tileH = rp->tile_h; fm.tilesX = TL.tilesX;
  fm.shardIndex = rp->shard_index; fm.shardCount = rp->shard_count;
  fm.pixelList = nullptr;
  fm.listOutput = 0;
  fm.listFirst = 0;
  fm.listStride = 1;
  fm.ns = (rp->aa_mode == RT_AA_FULL) ? nsFull : 1;
  fm.aaN = (rp->aa_mode == RT_AA_FULL) ? rp->aa_samples : 0;
  fm.jitter = (rp->aa_mode != RT_AA_NONE && rp->jitter) ? sc->jitterDev : nullptr;
  setDivisors(fm);
  {
    const int rc = uploadPixelCoordinates(sc, w, h, fm);
    if (rc != RT_OK) return rc;
  }

  sc->eventsUsed = 0;
  HIP_TRY(hipEventRecord(sc->evStart, sc->stream));
  const long long pixelSlots = (long long)TL.nTiles * rp->tile_w * rp->tile_h;
  int rc = renderRange(cx, 0, pixelSlots);
  if (rc != RT_OK) return rc;
  cx.stats.rays_primary = (unsigned long long)TL.pixels * fm.ns;

  if (rp->aa_mode == RT_AA_ADAPTIVE) {
    // GraphicalUI::doAntiAliasing (ui/GraphicalUI.cpp:377-435): edge mask on the
    // finished base image, then tracePixelAntiAlias on the flagged pixels only.
    rc = buildFlagList(sc, sc->frameRgb, w, h, rp->aa_threshold);
    if (rc != RT_OK) return rc;
    const long long nFlag = sc->flagCount;
    if (nFlag > 0) {
      fm.pixelList = sc->flagList;
      fm.listFirst = 0;
      fm.listStride = 1;
      fm.ns = nsFull;
      fm.aaN = rp->aa_samples;
      setDivisors(fm);
      int* keepHit = cx.hitDev;
      cx.hitDev = nullptr;  // the base pass already recorded the primary hits
      rc = renderRange(cx, 0, nFlag);
      cx.hitDev = keepHit;
      cx.stats.rays_primary += (unsigned long long)nFlag * nsFull;
      if (rc != RT_OK) return rc;
    }
  }
  HIP_TRY(hipEventRecord(sc->evStop, sc->stream));

  if (!rp->out_is_device) {
    const size_t px = (size_t)w * h;
    if (rp->shard_count == 1) {
      HIP_TRY(hipMemcpyAsync(rgb8_out, sc->frameRgb, px * 3, hipMemcpyDeviceToHost, sc->stream));
      if (hit_id_out) HIP_TRY(hipMemcpyAsync(hit_id_out, sc->frameHit, px * sizeof(int), hipMemcpyDeviceToHost, sc->stream));
      HIP_TRY(hipStreamSynchronize(sc->stream));
    } else {
      // only this shard's pixels may be written in the caller's frame
      std::vector<unsigned char> tmp(px * 3);
      std::vector<int> tmpHit(hit_id_out ? px : 0);
      HIP_TRY(hipMemcpyAsync(tmp.data(), sc->frameRgb, px * 3, hipMemcpyDeviceToHost, sc->stream));
      if (hit_id_out) HIP_TRY(hipMemcpyAsync(tmpHit.data(), sc->frameHit, px * sizeof(int), hipMemcpyDeviceToHost, sc->stream));
      HIP_TRY(hipStreamSynchronize(sc->stream));
      const int total = TL.tilesX * TL.tilesY;
      for (int t = rp->shard_index; t < total; t += rp->shard_count) {
        const int tx = t % TL.tilesX, ty = t / TL.tilesX;
        const int x0 = tx * rp->tile_w, y0 = ty * rp->tile_h;
        const int x1 = std::min(w, x0 + rp->tile_w), y1 = std::min(h, y0 + rp->tile_h);
        for (int y = y0; y < y1; y++) {
          std::memcpy(rgb8_out + ((size_t)y * w + x0) * 3, tmp.data() + ((size_t)y * w + x0) * 3, (size_t)(x1 - x0) * 3);
          if (hit_id_out) std::memcpy(hit_id_out + (size_t)y * w + x0, tmpHit.data() + (size_t)y * w + x0, (size_t)(x1 - x0) * sizeof(int));
        }
      }
    }
  } else {
    HIP_TRY(hipStreamSynchronize(sc->stream));
  }
  // the next frame starts from this frame's overall growth, not from its worst batch
  if (cx.usedSamples > 0.0) sc->growth = 1.1 * cx.usedSlots / cx.usedSamples + 0.05;

  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, sc->evStart, sc->evStop));
  cx.stats.ms_device = ms;
  double traceMs = 0.0, lightsMs = 0.0;
  for (size_t e = 0; e + 1 < sc->eventsUsed; e += 2) {
    float m = 0.f;
    if (hipEventElapsedTime(&m, sc->events[e], sc->events[e + 1]) == hipSuccess)
      (sc->eventKind[e / 2] ? lightsMs : traceMs) += m;
  }
  cx.stats.ms_trace = traceMs;
  cx.stats.ms_lights = lightsMs;
  cx.stats.ms_total =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (stats_out) *stats_out = cx.stats;
  return RT_OK;
}

}  // extern "C"

namespace {

// tracePixel / tracePixelAntiAlias on entries first, first + stride, ... (count
// of them) of an (x, y) list in device memory; 3 bytes per entry, in that
// order, to outDev.  Queued on the scene's stream and waited for.
int renderListed(RtScene* sc, const RtRenderParams* rp, const int* listDev, long long first, long long stride,
                 long long count, unsigned char* outDev, RtStats* stats_out) {
  const auto t0 = std::chrono::steady_clock::now();
  RenderCtx cx;
  cx.sc = sc;
  cx.rp = rp;
  std::memset(&cx.stats, 0, sizeof cx.stats);
  const int ns = (rp->aa_mode == RT_AA_NONE) ? 1 : rp->aa_samples * rp->aa_samples;
  int rc = sizePool(sc, rp, count * ns);
  if (rc != RT_OK) return rc;
  rc = uploadJitter(sc, rp, ns);
  if (rc != RT_OK) return rc;
  cx.fast = rp->traversal == RT_TRAVERSE_FAST;
  cx.wantStats = rp->collect_stats != 0;
  cx.rgbDev = outDev;
  cx.hitDev = nullptr;
  cx.tileOffDev = nullptr;
  FrameMap& fm = cx.fm;
  fm.width = rp->width; fm.height = rp->height;
  fm.tileW = fm.tileH = 8; fm.tilesX = 1;
  // (the share of the list stands where the shard of a tiled frame stands: it is part of what a batch is)
  fm.shardIndex = (int)first; fm.shardCount = (int)stride;
  fm.pixelList = listDev;
  fm.listOutput = 1;
  fm.listFirst = (int)first;
  fm.listStride = (int)stride;
  fm.ns = ns;
  fm.aaN = (rp->aa_mode == RT_AA_NONE) ? 0 : rp->aa_samples;
  fm.jitter = (rp->aa_mode != RT_AA_NONE && rp->jitter) ? sc->jitterDev : nullptr;
  setDivisors(fm);
  rc = uploadPixelCoordinates(sc, rp->width, rp->height, fm);
  if (rc != RT_OK) return rc;
  sc->eventsUsed = 0;
  HIP_TRY(hipEventRecord(sc->evStart, sc->stream));
  rc = renderRange(cx, 0, count);
  if (rc != RT_OK) {
    hipStreamSynchronize(sc->stream);
    return rc;
  }
  HIP_TRY(hipEventRecord(sc->evStop, sc->stream));
  HIP_TRY(hipStreamSynchronize(sc->stream));
  if (cx.usedSamples > 0.0) sc->growth = 1.1 * cx.usedSlots / cx.usedSamples + 0.05;
  cx.stats.rays_primary = (unsigned long long)count * ns;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, sc->evStart, sc->evStop) == hipSuccess) cx.stats.ms_device = ms;
  for (size_t e = 0; e + 1 < sc->eventsUsed; e += 2) {
    float m = 0.f;
    if (hipEventElapsedTime(&m, sc->events[e], sc->events[e + 1]) == hipSuccess)
      (sc->eventKind[e / 2] ? cx.stats.ms_lights : cx.stats.ms_trace) += m;
  }
  cx.stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (stats_out) *stats_out = cx.stats;
  return RT_OK;
}

int checkListedParams(const RtRenderParams* rp) {
  if (rp->width <= 0 || rp->height <= 0 || rp->depth < 0) return fail(RT_ERR_INVALID, "bad frame size or depth");
  if (rp->depth > kMaxDepth) return fail(RT_ERR_UNSUPPORTED, "recursion depth above 126");
  if (rp->aa_mode != RT_AA_NONE && (rp->aa_samples < 1 || rp->aa_samples > 8))
    return fail(RT_ERR_INVALID, "aa_samples must be 1..8");
  if (rp->traversal != RT_TRAVERSE_REFERENCE && rp->traversal != RT_TRAVERSE_FAST)
    return fail(RT_ERR_INVALID, "bad traversal mode");
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_render_pixels(RtScene* sc, const RtRenderParams* rp, int n_pixels, const int* xy, unsigned char* rgb8_out,
                     RtStats* stats_out) {
  if (!sc || !rp || n_pixels < 0 || (n_pixels > 0 && (!xy || !rgb8_out))) return fail(RT_ERR_INVALID, "null argument");
  int rc = checkListedParams(rp);
  if (rc != RT_OK) return rc;
  for (int k = 0; k < n_pixels; k++)
    if (xy[2 * k] < 0 || xy[2 * k] >= rp->width || xy[2 * k + 1] < 0 || xy[2 * k + 1] >= rp->height)
      return fail(RT_ERR_INVALID, "listed pixel outside the frame");
  HIP_TRY(hipSetDevice(sc->device));
  if (n_pixels == 0) {
    if (stats_out) std::memset(stats_out, 0, sizeof *stats_out);
    return RT_OK;
  }
  // the caller's list and the colours of its pixels, staged in a buffer the handle keeps
  const size_t listBytes = (size_t)n_pixels * 2 * sizeof(int);
  rc = growBuffer(&sc->listStage, &sc->listStageCap, listBytes + (size_t)n_pixels * 3);
  if (rc != RT_OK) return rc;
  int* listDev = reinterpret_cast<int*>(sc->listStage);
  unsigned char* outDev = sc->listStage + listBytes;
  HIP_TRY(hipMemcpyAsync(listDev, xy, listBytes, hipMemcpyHostToDevice, sc->stream));
  rc = renderListed(sc, rp, listDev, 0, 1, n_pixels, outDev, stats_out);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipMemcpyAsync(rgb8_out, outDev, (size_t)n_pixels * 3, hipMemcpyDeviceToHost, sc->stream));
  HIP_TRY(hipStreamSynchronize(sc->stream));
  return RT_OK;
}

int rt_render_flagged(RtScene* sc, const RtRenderParams* rp, const unsigned char* base_rgb_dev, int share_index,
                      int share_count, unsigned char* rows_dev, long long rows_capacity, long long* n_flagged,
                      long long* n_rendered, RtStats* stats_out) {
  if (!sc || !rp || !base_rgb_dev) return fail(RT_ERR_INVALID, "null argument");
  if (share_count <= 0 || share_index < 0 || share_index >= share_count) return fail(RT_ERR_INVALID, "bad share index/count");
  int rc = checkListedParams(rp);
  if (rc != RT_OK) return rc;
  if (rp->aa_mode == RT_AA_NONE) return fail(RT_ERR_INVALID, "the second pass of adaptive anti-aliasing needs aa_samples");
  HIP_TRY(hipSetDevice(sc->device));
  rc = buildFlagList(sc, base_rgb_dev, rp->width, rp->height, rp->aa_threshold);
  if (rc != RT_OK) return rc;
  const long long total = sc->flagCount;
  const long long mine = total > share_index ? (total - share_index + share_count - 1) / share_count : 0;
  if (n_flagged) *n_flagged = total;
  if (n_rendered) *n_rendered = mine;
  if (stats_out) std::memset(stats_out, 0, sizeof *stats_out);
  if (mine == 0) {
    HIP_TRY(hipStreamSynchronize(sc->stream));
    return RT_OK;
  }
  if (!rows_dev || rows_capacity < mine) return fail(RT_ERR_INVALID, "rows buffer smaller than this share of the flagged pixels");
  RtRenderParams p = *rp;
  p.aa_mode = RT_AA_FULL;
  return renderListed(sc, &p, sc->flagList, share_index, share_count, mine, rows_dev, stats_out);
}

int rt_patch_flagged(RtScene* sc, int share_index, int share_count, const unsigned char* rows_dev,
                     unsigned char* frame_dev, void* hip_stream) {
  if (!sc || !frame_dev) return fail(RT_ERR_INVALID, "null argument");
  if (share_count <= 0 || share_index < 0 || share_index >= share_count) return fail(RT_ERR_INVALID, "bad share index/count");
  const long long total = sc->flagCount;
  const long long n = total > share_index ? (total - share_index + share_count - 1) / share_count : 0;
  if (n == 0) return RT_OK;
  if (!rows_dev) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(sc->device));
  k_patch_listed<<<gridFor(n), kBlock, 0, (hipStream_t)hip_stream>>>(sc->flagList, share_index, share_count, n, rows_dev,
                                                                      sc->flagW, frame_dev);
  HIP_TRY(hipGetLastError());
  return RT_OK;   // stream-ordered: the caller synchronises hip_stream
}

int rt_intersect_batch(RtScene* sc, int traversal, int n, const double* rays_pd, const int* ray_types,
                       RtHit* hits) {
  if (!sc || n < 0 || (n > 0 && (!rays_pd || !hits))) return fail(RT_ERR_INVALID, "bad argument");
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(sc->device));
  // staging for rays, hit records and ray types: kept in the handle, grown when a call needs more
  if ((size_t)n > sc->isectCap) {
    if (sc->isectBuf) (void)hipFree(sc->isectBuf);
    sc->isectBuf = nullptr;
    sc->isectCap = 0;
    const size_t cap = std::max<size_t>((size_t)n + (size_t)n / 2, 4096);
    const hipError_t e = hipMalloc((void**)&sc->isectBuf, cap * (6 * sizeof(double) + sizeof(RtHit) + sizeof(int)));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, "cannot allocate the ray staging buffer");
    }
    sc->isectCap = cap;
  }
  double* raysDev = reinterpret_cast<double*>(sc->isectBuf);
  RtHit* hitsDev = reinterpret_cast<RtHit*>(sc->isectBuf + sc->isectCap * 6 * sizeof(double));
  int* typesDev = reinterpret_cast<int*>(sc->isectBuf + sc->isectCap * (6 * sizeof(double) + sizeof(RtHit)));
  HIP_TRY(hipMemcpyAsync(raysDev, rays_pd, 6 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, sc->stream));
  if (ray_types) HIP_TRY(hipMemcpyAsync(typesDev, ray_types, (size_t)n * sizeof(int), hipMemcpyHostToDevice, sc->stream));
  const int* typesArg = ray_types ? typesDev : nullptr;
  if (traversal == RT_TRAVERSE_FAST)
    k_intersect<true><<<gridFor(n), kBlock, (size_t)RT_LDS_STACK_BYTES_CLOSEST * kBlock, sc->stream>>>(
        sc->dev, n, raysDev, typesArg, hitsDev);
  else
    k_intersect<false><<<gridFor(n), kBlock, (size_t)RT_KD_LDS_STACK_BYTES * kBlock, sc->stream>>>(
        sc->dev, n, raysDev, typesArg, hitsDev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(hits, hitsDev, (size_t)n * sizeof(RtHit), hipMemcpyDeviceToHost, sc->stream));
  HIP_TRY(hipStreamSynchronize(sc->stream));
  return RT_OK;
}

int rt_intersect_batch_static(RtScene* sc, int traversal, int n, const double* rays_pd, const int* ray_types,
                              RtHit* hits) {
  if (!sc || n < 0 || (n > 0 && (!rays_pd || !hits))) return fail(RT_ERR_INVALID, "bad argument");
  if (n > kRaysStaticMax) return fail(RT_ERR_INVALID, "rt_intersect_batch_static takes at most 65536 rays per call");
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(sc->device));
  unsigned char* base = nullptr;
  HIP_TRY(scratchOfCurrentDevice(sc->device, &base));   // looked up at rt_scene_create: a table read here
  double* raysDev = reinterpret_cast<double*>(base);
  int* typesDev = reinterpret_cast<int*>(base + kStaticRaysBytes);
  RtHit* hitsDev = reinterpret_cast<RtHit*>(base + kStaticRaysBytes + kStaticTypesBytes);
  // every byte a launch reads is written by this call first and the kernel writes
  // every hit record it hands back, so nothing of an earlier call can show through
  HIP_TRY(hipMemcpyAsync(raysDev, rays_pd, 6 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, sc->stream));
  if (ray_types) HIP_TRY(hipMemcpyAsync(typesDev, ray_types, (size_t)n * sizeof(int), hipMemcpyHostToDevice, sc->stream));
  const int* typesArg = ray_types ? typesDev : nullptr;
  if (traversal == RT_TRAVERSE_FAST)
    k_intersect<true><<<gridFor(n), kBlock, (size_t)RT_LDS_STACK_BYTES_CLOSEST * kBlock, sc->stream>>>(
        sc->dev, n, raysDev, typesArg, hitsDev);
  else
    k_intersect<false><<<gridFor(n), kBlock, (size_t)RT_KD_LDS_STACK_BYTES * kBlock, sc->stream>>>(sc->dev, n, raysDev, typesArg,
                                                                                                  hitsDev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(hits, hitsDev, (size_t)n * sizeof(RtHit), hipMemcpyDeviceToHost, sc->stream));
  HIP_TRY(hipStreamSynchronize(sc->stream));
  return RT_OK;
}

int rt_unpack_tiles(const unsigned char* packed_dev, int bytes_per_pixel, int width, int height,
                    int tile_w, int tile_h, int shard_index, int shard_count,
                    unsigned char* frame_dev, void* hip_stream) {
  if (!packed_dev || !frame_dev || bytes_per_pixel <= 0 || width <= 0 || height <= 0 || tile_w <= 0 ||
      tile_h <= 0 || (tile_w & 7) || (tile_h & 7) || shard_count <= 0 || shard_index < 0 ||
      shard_index >= shard_count)
    return fail(RT_ERR_INVALID, "bad argument");
  TileLayout TL = tileLayout(width, height, tile_w, tile_h, shard_index, shard_count);
  if (TL.nTiles == 0) return RT_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  const long long* off = cachedTileOffsets(TL, width, height, tile_w, tile_h, shard_index, shard_count);
  if (!off) return fail(RT_ERR_NOMEM, "cannot allocate the tile offset table");
  FrameMap fm;
  std::memset(&fm, 0, sizeof fm);
  fm.width = width; fm.height = height; fm.tileW = tile_w; fm.tileH = tile_h; fm.tilesX = TL.tilesX;
  fm.shardIndex = shard_index; fm.shardCount = shard_count; fm.ns = 1;
  setDivisors(fm);
  const long long slots = (long long)TL.nTiles * tile_w * tile_h;
  if (slots >= (1ll << 31)) return fail(RT_ERR_UNSUPPORTED, "frame too large for one unpack launch");
  k_unpack<<<gridFor(slots), kBlock, 0, st>>>(packed_dev, bytes_per_pixel, fm, off, TL.nTiles, frame_dev);
  HIP_TRY(hipGetLastError());
  return RT_OK;   // stream-ordered: the caller synchronises hip_stream
}

int rt_test_pow_near_int(int count, const double* x, int n, double delta, double* out) {
  if (count < 0 || (count > 0 && (!x || !out)) || n < 0) return fail(RT_ERR_INVALID, "bad argument");
  if (count == 0) return RT_OK;
  DevBuf in, res;
  HIP_TRY(in.allocCopy(x, (size_t)count * sizeof(double)));
  HIP_TRY(res.alloc((size_t)count * sizeof(double)));
  k_pow_near_int<<<gridFor(count), kBlock, 0, 0>>>(count, in.as<double>(), n, delta, res.as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, res.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_test_bvh_invariants(int n_tris, const double* vertices, const double* normals, int max_leaf, long long* report) {
  if (n_tris < 1 || !vertices || !normals || !report || max_leaf < 1 || max_leaf > 8)
    return fail(RT_ERR_INVALID, "bad argument");
  for (int k = 0; k < 16; k++) report[k] = 0;
  // the structure, by the steps of rt_scene_create for a scene of identity-transform meshes
  std::vector<rtbvh::Prim> prims((size_t)n_tris);
  for (int f = 0; f < n_tris; f++) {
    rtbvh::Prim& pr = prims[(size_t)f];
    pr.box.reset();
    for (int k = 0; k < 3; k++) pr.box.grow(vertices + 9 * (size_t)f + 3 * k);
    for (int a = 0; a < 3; a++) pr.c[a] = 0.5 * (pr.box.lo[a] + pr.box.hi[a]);
    pr.id = f;
  }
  rtbvh::Builder bvh;
  rtbvh::Collapser wide;
  bvh.maxLeaf = max_leaf;
  if (!bvh.build(prims)) return fail(RT_ERR_INVALID, "empty scene");
  wide.run(bvh.nodes);
  std::vector<DevTri> ordered(bvh.triOrder.size());
  for (size_t i = 0; i < ordered.size(); i++) {
    DevTri& T = ordered[i];
    std::memset(&T, 0, sizeof T);
    const size_t f = (size_t)bvh.triOrder[i];
    for (int a = 0; a < 3; a++) {
      T.v0[a] = vertices[9 * f + a];
      T.v1[a] = vertices[9 * f + 3 + a];
      T.v2[a] = vertices[9 * f + 6 + a];
      T.n[a] = normals[3 * f + a];
    }
    T.face = (int)f;
  }
  if (!addNullLeaf(ordered, wide.nodes)) return fail(RT_ERR_UNSUPPORTED, "too many triangles for the BVH leaf encoding");
  std::vector<DevBvhNodeF> fn;
  bool anyCone = false;
  buildNodeCopies(wide.nodes, wide.tight, ordered, fn, anyCone);
  const int nullTriangle = (int)ordered.size() - 1;
  const size_t nNodes = wide.nodes.size();
  report[0] = (long long)nNodes;
  report[1] = wide.maxDepth;
  report[2] = (long long)bvh.nodes.size();

  // ---- the checks, written against what the traversal assumes, not against the builder
  struct Anc { size_t node; int slot; };          // a child (slot of copy 0) on the way down to a leaf
  std::vector<int> seen((size_t)n_tris, 0);
  std::vector<Anc> chain;
  // triangles below each child of copy 0, as positions in `ordered` (a contiguous run by construction; held
  // against the walk: first, last and count)
  struct Span { int lo = INT32_MAX, hi = -1, count = 0; };
  std::vector<std::array<Span, 4>> spans(nNodes);
  const auto planesOf = [&](const DevBvhNodeF& F, int j, float lo[3], float hi[3]) {
    lo[0] = F.lo_x[j]; hi[0] = F.hi_x[j];
    lo[1] = F.lo_y[j]; hi[1] = F.hi_y[j];
    lo[2] = F.lo_z[j]; hi[2] = F.hi_z[j];
  };
  const auto leafTriangle = [&](int t) {
    const DevTri& T = ordered[(size_t)t];
    seen[(size_t)T.face]++;
    const double* vs[3] = {T.v0, T.v1, T.v2};
    const double nl = std::sqrt(T.n[0] * T.n[0] + T.n[1] * T.n[1] + T.n[2] * T.n[2]);
    for (const Anc& an : chain) {
      const DevBvhNodeF& F = fn[an.node * 8];
      float lo[3], hi[3];
      planesOf(F, an.slot, lo, hi);
      bool outF = false, outQ = false;
      // the quantised node's slot that holds the same child
      const rtbvh::Node4& n4 = wide.nodes[an.node];
      int c4 = -1;
      for (int c = 0; c < 4; c++) if (n4.child[c] == F.child[an.slot]) c4 = c;
      const double sc[3] = {n4.scaleX, n4.scaleY, n4.scaleZ};
      for (int k = 0; k < 3; k++)
        for (int a = 0; a < 3; a++) {
          outF |= !(vs[k][a] >= (double)lo[a] && vs[k][a] <= (double)hi[a]);
          if (c4 >= 0) {
            const double ql = (double)n4.origin[a] + (double)((n4.qlo[a] >> (8 * c4)) & 255u) * sc[a];
            const double qh = (double)n4.origin[a] + (double)((n4.qhi[a] >> (8 * c4)) & 255u) * sc[a];
            outQ |= !(vs[k][a] >= ql && vs[k][a] <= qh);
          }
        }
      if (outF) report[4]++;
      if (outQ || c4 < 0) report[5]++;
      const float* cn = F.cone[an.slot];
      if (cn[3] < 1.0f && nl > 0.0 && std::isfinite(nl)) {
        // every direction the cone test turns away makes an angle below 90 degrees - asin(bound) with the
        // axis: the normal has to lie within asin(bound) of the axis
        // (the sine from the cross product: a narrow cone's cosine says nothing)
        const double ax = cn[0], ay = cn[1], az = cn[2];
        const double al = std::sqrt(ax * ax + ay * ay + az * az);
        const double cosAngle = (ax * T.n[0] + ay * T.n[1] + az * T.n[2]) / (al * nl);
        const double cx = ay * T.n[2] - az * T.n[1], cy = az * T.n[0] - ax * T.n[2], cz = ax * T.n[1] - ay * T.n[0];
        const double sinAngle = std::sqrt(cx * cx + cy * cy + cz * cz) / (al * nl);
        if (!(cosAngle > 0.0 && sinAngle <= (double)cn[3])) report[6]++;
      }
      Span& sp = spans[an.node][(size_t)an.slot];
      sp.lo = std::min(sp.lo, t);
      sp.hi = std::max(sp.hi, t + 1);
      sp.count++;
    }
  };
  {
    // depth-first over copy 0 of every node, explicit stack of (node, next slot)
    std::vector<std::pair<size_t, int>> st{{0, 0}};
    std::vector<char> visited(nNodes, 0);
    visited[0] = 1;
    while (!st.empty()) {
      const size_t i = st.back().first;
      const int j = st.back().second;
      if (j == 4) {
        st.pop_back();
        if (!chain.empty()) chain.pop_back();
        continue;
      }
      st.back().second++;
      const DevBvhNodeF& F = fn[i * 8];
      const int ref = F.child[j];
      if (ref >= 0) {
        if ((size_t)ref >= nNodes || visited[(size_t)ref]) return fail(RT_ERR_DEVICE, "child index out of range or visited twice");
        visited[(size_t)ref] = 1;
        chain.push_back(Anc{i, j});
        st.push_back({(size_t)ref, 0});
        continue;
      }
      const int x = ~ref, first = x >> 3, count = (x & 7) + 1;
      float lo[3], hi[3];
      planesOf(F, j, lo, hi);
      if (first == nullTriangle) {
        // an unused slot: an inverted box in both forms, so that no ray enters it
        bool inverted = false;
        for (int a = 0; a < 3; a++) inverted |= lo[a] > hi[a];
        if (!inverted || count != 1) report[11]++;
        continue;
      }
      if ((x & RT_BVH_OBJECT_BIT) || first < 0 || first + count > nullTriangle) return fail(RT_ERR_DEVICE, "leaf range out of bounds");
      if (count > max_leaf) report[13]++;
      chain.push_back(Anc{i, j});
      for (int t = first; t < first + count; t++) leafTriangle(t);
      chain.pop_back();
    }
    for (size_t i = 0; i < nNodes; i++)
      if (!visited[i]) report[3]++;   // a node nothing points at
  }
  for (int f = 0; f < n_tris; f++)
    if (seen[(size_t)f] != 1) report[3]++;
  for (size_t i = 0; i < nNodes; i++)
    for (int j = 0; j < 4; j++) {
      const Span& sp = spans[i][(size_t)j];
      if (sp.count > 0 && sp.hi - sp.lo != sp.count) report[3]++;   // not a contiguous run
      if (fn[i * 8].cone[j][3] < 1.0f) report[12]++;
    }

  // ---- culls as the kernels take them (binary32, the visit's own expression) on directions spread around
  // every cone's rim: whatever is turned away must hold only triangles the reference rejects one by one
  // (trimesh.cpp:93-104: normal . direction > 0 with back faces culled)
  {
    unsigned long long rng = 0x9e3779b97f4a7c15ull;
    const auto rnd = [&]() {   // uniform in [0, 1)
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      return (double)(rng >> 11) * (1.0 / 9007199254740992.0);
    };
    const size_t step = std::max<size_t>(1, nNodes / 4096);
    for (size_t i = 0; i < nNodes; i += step)
      for (int j = 0; j < 4; j++) {
        const float* cn = fn[i * 8].cone[j];
        const Span& sp = spans[i][(size_t)j];
        if (!(cn[3] < 1.0f) || sp.count == 0) continue;
        // two unit vectors across the axis
        const double A[3] = {cn[0], cn[1], cn[2]};
        double u[3] = {0.0, 0.0, 0.0};
        u[std::fabs(A[0]) < 0.6 ? 0 : (std::fabs(A[1]) < 0.6 ? 1 : 2)] = 1.0;
        const double ua = u[0] * A[0] + u[1] * A[1] + u[2] * A[2];
        for (int a = 0; a < 3; a++) u[a] -= ua * A[a];
        const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int a = 0; a < 3; a++) u[a] /= ul;
        const double w[3] = {A[1] * u[2] - A[2] * u[1], A[2] * u[0] - A[0] * u[2], A[0] * u[1] - A[1] * u[0]};
        for (int k = 0; k < 12; k++) {
          // the direction's component along the axis: just above the bound (the rim, where a cull is
          // least safe), or anywhere above it
          const double along = (k < 8) ? std::min(1.0, (double)cn[3] * (1.0 + 1.0e-6 * k) + 1.0e-7 * k) : (double)cn[3] + (1.0 - (double)cn[3]) * rnd();
          const double across = std::sqrt(std::max(0.0, 1.0 - along * along)), phi = 6.283185307179586 * rnd();
          double d[3];
          for (int a = 0; a < 3; a++) d[a] = along * A[a] + across * (std::cos(phi) * u[a] + std::sin(phi) * w[a]);
          const double dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
          for (int a = 0; a < 3; a++) d[a] /= dl;
          const float dx = (float)d[0], dy = (float)d[1], dz = (float)d[2];
          if (!(fmaf(cn[2], dz, fmaf(cn[1], dy, cn[0] * dx)) > cn[3])) continue;   // as RT_DESCEND's cone test
          report[8]++;
          for (int t = sp.lo; t < sp.hi; t++) {
            const double* nn = ordered[(size_t)t].n;
            const double nd = nn[0] * d[0] + nn[1] * d[1] + nn[2] * d[2];
            const double nl = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
            // (a triangle without a direction is rejected by every ray: nd == 0, or NaN through every later test)
            if (nl > 0.0 && std::isfinite(nl) && !(nd > 0.0)) { report[7]++; break; }
          }
        }
      }
  }

  // ---- the eight copies of a node: the same children, reordered near to far along the copy's diagonal
  for (size_t i = 0; i < nNodes; i++) {
    const DevBvhNodeF& F0 = fn[i * 8];
    for (int o = 0; o < 8; o++) {
      const DevBvhNodeF& F = fn[i * 8 + (size_t)o];
      bool used0[4] = {false, false, false, false};
      bool same = true, sorted = true;
      double lastKey = -std::numeric_limits<double>::infinity();
      bool pastUsed = false;
      for (int j = 0; j < 4; j++) {
        float lo[3], hi[3];
        planesOf(F, j, lo, hi);
        int match = -1;
        for (int m = 0; m < 4 && match < 0; m++) {
          if (used0[m] || F0.child[m] != F.child[j]) continue;
          float lo0[3], hi0[3];
          planesOf(F0, m, lo0, hi0);
          if (std::memcmp(lo, lo0, sizeof lo) == 0 && std::memcmp(hi, hi0, sizeof hi) == 0 &&
              std::memcmp(F.cone[j], F0.cone[m], sizeof F.cone[j]) == 0) match = m;
        }
        if (match < 0) same = false; else used0[match] = true;
        bool inverted = false;
        double key = 0.0;
        for (int a = 0; a < 3; a++) {
          inverted |= lo[a] > hi[a];
          const double centre = 0.5 * ((double)lo[a] + (double)hi[a]);
          key += ((o >> a) & 1) ? -centre : centre;
        }
        if (inverted) { pastUsed = true; continue; }
        if (pastUsed || key < lastKey) sorted = false;
        lastKey = key;
      }
      if (!same) report[9]++;
      if (!sorted) report[10]++;
    }
  }
  return RT_OK;
}

int rt_edge_mask(const unsigned char* rgb8, int width, int height, int cutoff, unsigned char* mask_out) {
  if (!rgb8 || !mask_out || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad argument");
  const size_t px = (size_t)width * height;
  DevBuf src, dst;
  HIP_TRY(src.allocCopy(rgb8, px * 3));
  HIP_TRY(dst.alloc(px));
  k_edge_mask<<<gridFor((long long)px), kBlock, 0, 0>>>(src.as<unsigned char>(), width, height, cutoff, dst.as<unsigned char>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mask_out, dst.p, px, hipMemcpyDeviceToHost));
  return RT_OK;
}

}  // extern "C"
