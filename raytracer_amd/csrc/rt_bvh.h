// Host-side builder of the library's own acceleration structure for
// RT_TRAVERSE_FAST: a binary bounding-volume hierarchy over every triangle of
// every mesh (world-space boxes) and every non-mesh object, built top-down with
// a binned surface-area heuristic.
//
// Why not the reference's KD-tree for throughput: Scene::buildMainKdTree
// (src/scene/scene.cpp:415-461) prices a split by the SUM of the object boxes'
// areas on either side, not by the areas of the two child cells, so it never
// cuts empty space away and it duplicates primitives freely (a 1M-triangle
// heightfield becomes 4.8M nodes holding 25.7M references; a ray over the field
// tests ~1200 triangles).  Closest-hit results do not depend on the shape of
// the acceleration structure, so the fast path builds its own; the reference's
// tree is still traversed verbatim by RT_TRAVERSE_REFERENCE.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <future>
#include <vector>

namespace rtbvh {

struct Box {
  double lo[3], hi[3];
  void reset() { for (int a = 0; a < 3; a++) { lo[a] = DBL_MAX; hi[a] = -DBL_MAX; } }
  void grow(const Box& b) {
    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
  }
  void grow(const double* p) {
    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
  }
  double area() const {
    double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  }
};

// 64-byte device node: the boxes of both children as outward-rounded floats and
// the children themselves.  child >= 0: node index.  child < 0: leaf, x = ~child;
// bit 30 set: a single non-mesh object, index x & 0x3fffffff; else triangles
// [first, first+count) of the leaf-ordered triangle array with first = x >> 3,
// count = (x & 7) + 1.
struct Node {
  float lo[2][3], hi[2][3];
  int32_t child[2];
  int32_t pad[2];
};
static_assert(sizeof(Node) == 64, "BVH node must be 64 bytes");

constexpr int kBins = 16;
constexpr int32_t kObjectBit = 0x40000000;

inline float down(double x) {
  float f = (float)x;
  return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f;
}
inline float up(double x) {
  float f = (float)x;
  return ((double)f < x) ? std::nextafterf(f, INFINITY) : f;
}

struct Prim {
  Box box;
  double c[3];     // centroid
  int32_t id;      // triangle: global face index; object: kObjectBit | object index
};

class Builder {
 public:
  std::vector<Node> nodes;
  std::vector<int32_t> triOrder;   // leaf order -> global face index
  Box rootBox;
  int maxDepth = 0;
  bool balanced = false;   // median splits only: depth <= log2(n) + 1
  int maxLeaf = 2;         // triangles per leaf, 1..8 (the leaf reference holds count - 1 in 3 bits)

  // prims is consumed (reordered).  Returns false for an empty scene.
  bool build(std::vector<Prim>& prims) {
    nodes.clear();
    triOrder.clear();
    rootBox.reset();
    if (prims.empty()) return false;
    for (const Prim& p : prims) rootBox.grow(p.box);
    prims_ = &prims;
    nodes.resize(std::max<size_t>(1, 2 * prims.size()));
    nodeCount_ = 1;
    triSlots_.assign(prims.size(), -1);
    if (prims.size() == 1) {
      // a single primitive: node 0 with one real child and one empty child
      Node& n = nodes[0];
      setChildBox(n, 0, prims[0].box);
      n.child[0] = makeLeaf(0, 1);
      Box e; e.reset();
      for (int a = 0; a < 3; a++) { n.lo[1][a] = INFINITY; n.hi[1][a] = -INFINITY; }
      n.child[1] = n.child[0];
      n.pad[0] = n.pad[1] = 0;
      maxDepth = 1;
    } else {
      depthSeen_ = 0;
      buildNode(0, 0, (int)prims.size(), 1, 0);
      maxDepth = depthSeen_;
    }
    nodes.resize(nodeCount_);
    // leaf-ordered triangle list: position in prims == slot
    triOrder.resize(prims.size());
    for (size_t i = 0; i < prims.size(); i++) triOrder[i] = prims[i].id;
    return true;
  }

 private:
  std::vector<Prim>* prims_ = nullptr;
  std::atomic<int> nodeCount_{0};
  std::atomic<int> depthSeen_{0};
  std::vector<int> triSlots_;

  static void setChildBox(Node& n, int c, const Box& b) {
    for (int a = 0; a < 3; a++) { n.lo[c][a] = down(b.lo[a]); n.hi[c][a] = up(b.hi[a]); }
  }

  int32_t makeLeaf(int first, int count) const {
    const Prim& p = (*prims_)[first];
    if (p.id & kObjectBit) return ~(int32_t)(kObjectBit | (p.id & 0x3fffffff));
    return ~(int32_t)(((int32_t)first << 3) | (count - 1));
  }

  bool leafAllowed(int first, int count) const {
    if (count == 1) return true;
    if (count > maxLeaf) return false;
    for (int i = first; i < first + count; i++)
      if ((*prims_)[i].id & kObjectBit) return false;  // objects sit alone in a leaf
    return true;
  }

  // Fills node `ni` (its two child boxes and children) for prims [first, first+count), count >= 2.
  void buildNode(int ni, int first, int count, int depth, int parallelDepth) {
    std::vector<Prim>& P = *prims_;
    int prev = depthSeen_.load();
    while (depth > prev && !depthSeen_.compare_exchange_weak(prev, depth)) {}
    Box cb; cb.reset();
    for (int i = first; i < first + count; i++) cb.grow(P[i].c);
    int axis = 0;
    double ext = -1;
    for (int a = 0; a < 3; a++) if (cb.hi[a] - cb.lo[a] > ext) { ext = cb.hi[a] - cb.lo[a]; axis = a; }
    int mid = first + count / 2;
    if (ext > 0 && balanced) {
      std::nth_element(P.begin() + first, P.begin() + mid, P.begin() + first + count,
                       [axis](const Prim& x, const Prim& y) { return x.c[axis] < y.c[axis]; });
    } else if (ext > 0) {
      // binned SAH over the three axes
      double bestCost = DBL_MAX; int bestAxis = -1, bestBin = -1;
      for (int a = 0; a < 3; a++) {
        const double lo = cb.lo[a], e = cb.hi[a] - cb.lo[a];
        if (!(e > 0)) continue;
        Box bb[kBins]; int bc[kBins];
        for (int b = 0; b < kBins; b++) { bb[b].reset(); bc[b] = 0; }
        const double k = kBins * (1.0 - 1e-9) / e;
        for (int i = first; i < first + count; i++) {
          int b = (int)((P[i].c[a] - lo) * k);
          b = std::min(std::max(b, 0), kBins - 1);
          bb[b].grow(P[i].box); bc[b]++;
        }
        double leftArea[kBins]; int leftCnt[kBins];
        Box acc; acc.reset(); int n = 0;
        for (int b = 0; b < kBins - 1; b++) { acc.grow(bb[b]); n += bc[b]; leftArea[b] = acc.area(); leftCnt[b] = n; }
        acc.reset(); n = 0;
        for (int b = kBins - 1; b > 0; b--) {
          acc.grow(bb[b]); n += bc[b];
          if (leftCnt[b - 1] == 0 || n == 0) continue;
          double cost = leftArea[b - 1] * leftCnt[b - 1] + acc.area() * n;
          if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = b - 1; }
        }
      }
      if (bestAxis >= 0) {
        const double lo = cb.lo[bestAxis], e = cb.hi[bestAxis] - cb.lo[bestAxis];
        const double k = kBins * (1.0 - 1e-9) / e;
        auto it = std::partition(P.begin() + first, P.begin() + first + count, [&](const Prim& p) {
          int b = (int)((p.c[bestAxis] - lo) * k);
          b = std::min(std::max(b, 0), kBins - 1);
          return b <= bestBin;
        });
        mid = (int)(it - P.begin());
      }
      if (mid == first || mid == first + count) {
        mid = first + count / 2;
        std::nth_element(P.begin() + first, P.begin() + mid, P.begin() + first + count,
                         [axis](const Prim& x, const Prim& y) { return x.c[axis] < y.c[axis]; });
      }
    }
    const int cnt[2] = {mid - first, first + count - mid};
    const int beg[2] = {first, mid};
    int childNode[2] = {-1, -1};
    Node& n = nodes[ni];
    n.pad[0] = n.pad[1] = 0;
    for (int c = 0; c < 2; c++) {
      Box b; b.reset();
      for (int i = beg[c]; i < beg[c] + cnt[c]; i++) b.grow(P[i].box);
      setChildBox(n, c, b);
      if (leafAllowed(beg[c], cnt[c])) {
        n.child[c] = makeLeaf(beg[c], cnt[c]);
      } else {
        childNode[c] = nodeCount_.fetch_add(1);
        n.child[c] = childNode[c];
      }
    }
    // large subtrees are built by two threads each, a few levels deep
    if (childNode[0] >= 0 && childNode[1] >= 0 && parallelDepth < 4 && count > (1 << 16)) {
      auto fut = std::async(std::launch::async, [&, this] {
        buildNode(childNode[0], beg[0], cnt[0], depth + 1, parallelDepth + 1);
      });
      buildNode(childNode[1], beg[1], cnt[1], depth + 1, parallelDepth + 1);
      fut.get();
    } else {
      for (int c = 0; c < 2; c++)
        if (childNode[c] >= 0) buildNode(childNode[c], beg[c], cnt[c], depth + 1, parallelDepth + 1);
    }
  }
};


// 64-byte 4-wide node with 8-bit quantised child boxes.  Child box c spans
// origin + q * scale per axis, scale a power of two (lower planes rounded down, upper planes rounded
// up, so the decoded box always contains the builder's box).  child[c] uses the
// encoding of Node::child; kEmptyChild marks an unused slot.
struct Node4 {
  float origin[3];
  float scaleX;      // 2^exp of the x axis, as the number the traversal multiplies by
  int32_t child[4];
  uint32_t qlo[3];   // byte c of qlo[a]: lower plane of child c on axis a
  uint32_t qhi[3];
  float scaleY, scaleZ;
};
static_assert(sizeof(Node4) == 64, "BVH4 node must be 64 bytes");
constexpr int32_t kEmptyChild = 0x7fffffff;

// Collapses the binary tree into 4-wide nodes: a node adopts the two children
// of a binary node and keeps replacing its largest internal child by that
// child's two children until it has four or only leaves are left.
class Collapser {
 public:
  std::vector<Node4> nodes;
  // The child boxes of every 4-wide node before quantisation (binary32, rounded
  // outward by the binary builder): lo[a][c] / hi[a][c]; an unused slot is inverted.
  struct Tight { float lo[3][4], hi[3][4]; };
  std::vector<Tight> tight;
  int maxDepth = 0;

  void run(const std::vector<Node>& bin) {
    bin_ = &bin;
    nodes.clear();
    nodes.reserve(bin.size() / 2 + 1);
    nodes.emplace_back();
    tight.clear();
    tight.reserve(bin.size() / 2 + 1);
    tight.emplace_back();
    maxDepth = 0;
    // explicit work list instead of recursion: (binary node, output slot, depth)
    struct Job { int binNode; int out; int depth; };
    std::vector<Job> jobs{{0, 0, 1}};
    while (!jobs.empty()) {
      Job j = jobs.back();
      jobs.pop_back();
      maxDepth = std::max(maxDepth, j.depth);
      struct Kid { Box box; int32_t ref; };
      std::vector<Kid> kids;
      auto addKids = [&](int b) {
        const Node& n = (*bin_)[b];
        for (int c = 0; c < 2; c++) {
          if (c == 1 && n.child[1] == n.child[0]) continue;  // single-primitive scene
          Kid k;
          for (int a = 0; a < 3; a++) { k.box.lo[a] = n.lo[c][a]; k.box.hi[a] = n.hi[c][a]; }
          k.ref = n.child[c];
          kids.push_back(k);
        }
      };
      addKids(j.binNode);
      while (kids.size() < 4) {
        int best = -1;
        double bestArea = -1;
        for (size_t i = 0; i < kids.size(); i++)
          if (kids[i].ref >= 0 && kids[i].box.area() > bestArea) { bestArea = kids[i].box.area(); best = (int)i; }
        if (best < 0) break;
        const int b = kids[best].ref;
        kids.erase(kids.begin() + best);
        addKids(b);
      }
      Node4 out;
      std::memset(&out, 0, sizeof out);
      Tight tb;
      Box all; all.reset();
      for (const Kid& k : kids) all.grow(k.box);
      int exps[3];
      for (int a = 0; a < 3; a++) {
        out.origin[a] = down(all.lo[a]);
        const double ext = all.hi[a] - (double)out.origin[a];
        int e = -126;
        if (ext > 0) {
          e = (int)std::ceil(std::log2(ext / 255.0));
          while (std::ldexp(255.0, e) < ext) e++;   // guard the rounding of log2
          e = std::max(-126, std::min(127, e));
        }
        exps[a] = e;
      }
      out.scaleX = (float)std::ldexp(1.0, exps[0]);
      out.scaleY = (float)std::ldexp(1.0, exps[1]);
      out.scaleZ = (float)std::ldexp(1.0, exps[2]);
      for (int c = 0; c < 4; c++) {
        if (c >= (int)kids.size()) {
          out.child[c] = kEmptyChild;
          for (int a = 0; a < 3; a++) out.qlo[a] |= 255u << (8 * c);   // inverted box: upper planes stay 0
          for (int a = 0; a < 3; a++) { tb.lo[a][c] = 1.0f; tb.hi[a][c] = 0.0f; }
          continue;
        }
        const Kid& k = kids[c];
        for (int a = 0; a < 3; a++) { tb.lo[a][c] = down(k.box.lo[a]); tb.hi[a][c] = up(k.box.hi[a]); }
        for (int a = 0; a < 3; a++) {
          const double sc = std::ldexp(1.0, exps[a]);
          double ql = std::floor((k.box.lo[a] - (double)out.origin[a]) / sc);
          double qh = std::ceil((k.box.hi[a] - (double)out.origin[a]) / sc);
          ql = std::min(255.0, std::max(0.0, ql));
          qh = std::min(255.0, std::max(0.0, qh));
          out.qlo[a] |= (uint32_t)ql << (8 * c);
          out.qhi[a] |= (uint32_t)qh << (8 * c);
        }
        if (k.ref >= 0) {
          const int idx = (int)nodes.size();
          nodes.emplace_back();
          tight.emplace_back();
          out.child[c] = idx;
          jobs.push_back({k.ref, idx, j.depth + 1});
        } else {
          out.child[c] = k.ref;
        }
      }
      nodes[j.out] = out;
      tight[j.out] = tb;
    }
  }

 private:
  const std::vector<Node>* bin_ = nullptr;
};

}  // namespace rtbvh
