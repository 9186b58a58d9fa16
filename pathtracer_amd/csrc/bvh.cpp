#include "bvh.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cfloat>

namespace pt {
namespace {

struct Box {
  float mn[3], mx[3];
  void reset() { for (int k = 0; k < 3; k++) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; } }
  void grow(const float* p) { for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); } }
  void grow(const Box& b) { for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); } }
  float halfArea() const {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (dx < 0) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Work { uint32_t node, first, count, depth; };

constexpr int NBINS = 16;
constexpr float COST_PAIR = 1.0f;    // one step that fetches and tests a sibling pair

}  // namespace

void buildBvh(const float* v, uint32_t T, uint32_t maxLeaf, Bvh& out, float triCost) {
  const float COST_TRI = triCost;
  auto t0 = std::chrono::steady_clock::now();
  out = Bvh();
  maxLeaf = std::max(1u, std::min(maxLeaf, BVH_MAX_LEAF));

  // Triangles with v0 == v1 or v0 == v2 have an exactly zero determinant for every ray and can never be
  // hit (|det| < FloatEpsilon); they are left out.  NaN vertices are never hit either.
  std::vector<uint32_t> idx; idx.reserve(T);
  std::vector<Box> tb(T); std::vector<float> cen((size_t)T * 3);
  Box scene; scene.reset();
  for (uint32_t i = 0; i < T; i++) {
    const float* p = v + 9 * (size_t)i;
    bool finite = true;
    for (int k = 0; k < 9; k++) finite = finite && std::isfinite(p[k]);
    bool same01 = p[0] == p[3] && p[1] == p[4] && p[2] == p[5];
    bool same02 = p[0] == p[6] && p[1] == p[7] && p[2] == p[8];
    if (!finite || same01 || same02) continue;
    tb[i].reset(); tb[i].grow(p); tb[i].grow(p + 3); tb[i].grow(p + 6);
    for (int k = 0; k < 3; k++) cen[3 * (size_t)i + k] = 0.5f * (tb[i].mn[k] + tb[i].mx[k]);
    scene.grow(tb[i]);
    idx.push_back(i);
  }
  uint32_t N = (uint32_t)idx.size();

  // Conservative padding: the slab test and the triangle test round differently; boxes are grown by a
  // distance far above either rounding error at scene scale so traversal never culls a triangle that an
  // exhaustive test over all triangles would accept.
  float scale = 1.0f;
  if (N) for (int k = 0; k < 3; k++) scale = std::max(scale, std::max(std::fabs(scene.mn[k]), std::fabs(scene.mx[k])));
  const float pad = scale * 1e-5f;

  auto putNode = [&](uint32_t node, const Box& b, uint32_t ref) {
    if (out.nodes.size() < (size_t)(node + 1) * 8) out.nodes.resize((size_t)(node + 1) * 8, 0.0f);
    float* n = &out.nodes[(size_t)node * 8];
    for (int k = 0; k < 3; k++) { n[k] = b.mn[k] - pad; n[4 + k] = b.mx[k] + pad; }
    std::memcpy(&n[3], &ref, 4); n[7] = 0.0f;
  };

  if (N == 0) {                                   // nothing can be hit: the root is a leaf holding one all-zero
    Box b; for (int k = 0; k < 3; k++) b.mn[k] = b.mx[k] = 0.0f;             // triangle (triOrder 0xFFFFFFFF), det == 0 for every ray
    putNode(0, b, 1u << BVH_COUNT_SHIFT); putNode(1, b, 0u);
    out.rootRef = 1u << BVH_COUNT_SHIFT; out.numNodes = 2; out.depth = 0;
    out.triOrder.assign(1, 0xFFFFFFFFu);
    out.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return;
  }

  out.nodes.reserve((size_t)N * 2 * 8 / std::max(1u, maxLeaf / 2) + 64);
  uint32_t nextPair = 1;                          // pair 0 = {root, padding}
  std::vector<Work> stack; stack.push_back(Work{0, 0, N, 0});
  std::vector<uint32_t> tmp;
  Box zero; for (int k = 0; k < 3; k++) zero.mn[k] = zero.mx[k] = 0.0f;
  putNode(1, zero, 0);

  while (!stack.empty()) {
    Work w = stack.back(); stack.pop_back();
    out.depth = std::max(out.depth, w.depth);
    Box bounds, cb; bounds.reset(); cb.reset();
    for (uint32_t i = w.first; i < w.first + w.count; i++) { bounds.grow(tb[idx[i]]); cb.grow(&cen[3 * (size_t)idx[i]]); }

    bool makeLeaf = false; uint32_t mid = 0;
    if (w.count == 1) makeLeaf = true;
    else {
      float bestCost = FLT_MAX; int bestAxis = -1, bestBin = -1;
      bool balanced = w.depth + 26 >= BVH_MAX_DEPTH;      // deep branch: fall back to median splits so depth stays bounded
      if (!balanced) {
        for (int ax = 0; ax < 3; ax++) {
          float lo = cb.mn[ax], ext = cb.mx[ax] - cb.mn[ax];
          if (!(ext > 0.0f)) continue;
          Box bb[NBINS]; uint32_t bc[NBINS];
          for (int b = 0; b < NBINS; b++) { bb[b].reset(); bc[b] = 0; }
          float k1 = (float)NBINS * (1.0f - 1e-6f) / ext;
          for (uint32_t i = w.first; i < w.first + w.count; i++) {
            uint32_t t = idx[i]; int b = (int)(k1 * (cen[3 * (size_t)t + ax] - lo));
            b = std::max(0, std::min(NBINS - 1, b));
            bb[b].grow(tb[t]); bc[b]++;
          }
          float rightArea[NBINS]; uint32_t rightCnt[NBINS];
          Box acc; acc.reset(); uint32_t cnt = 0;
          for (int b = NBINS - 1; b > 0; b--) { acc.grow(bb[b]); cnt += bc[b]; rightArea[b] = acc.halfArea(); rightCnt[b] = cnt; }
          acc.reset(); cnt = 0;
          for (int b = 0; b < NBINS - 1; b++) {
            acc.grow(bb[b]); cnt += bc[b];
            if (cnt == 0 || rightCnt[b + 1] == 0) continue;
            float cost = acc.halfArea() * (float)cnt + rightArea[b + 1] * (float)rightCnt[b + 1];
            if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = b; }
          }
        }
      }
      float area = bounds.halfArea();
      float leafCost = COST_TRI * (float)w.count;
      float splitCost = (bestAxis >= 0 && area > 0.0f) ? COST_PAIR + COST_TRI * bestCost / area : FLT_MAX;
      if (w.count <= maxLeaf && (bestAxis < 0 || leafCost <= splitCost)) makeLeaf = true;
      else if (bestAxis >= 0) {
        float lo = cb.mn[bestAxis], ext = cb.mx[bestAxis] - cb.mn[bestAxis];
        float k1 = (float)NBINS * (1.0f - 1e-6f) / ext;
        auto it = std::partition(idx.begin() + w.first, idx.begin() + w.first + w.count, [&](uint32_t t) {
          int b = (int)(k1 * (cen[3 * (size_t)t + bestAxis] - lo)); b = std::max(0, std::min(NBINS - 1, b));
          return b <= bestBin;
        });
        mid = (uint32_t)(it - idx.begin());
      } else {
        // all centroids coincide, or depth guard: split at the median of the widest centroid axis
        int ax = 0; float e = -1.0f;
        for (int k = 0; k < 3; k++) { float x = cb.mx[k] - cb.mn[k]; if (x > e) { e = x; ax = k; } }
        mid = w.first + w.count / 2;
        std::nth_element(idx.begin() + w.first, idx.begin() + mid, idx.begin() + w.first + w.count,
                         [&](uint32_t a, uint32_t b) { return cen[3 * (size_t)a + ax] < cen[3 * (size_t)b + ax]; });
      }
      if (!makeLeaf && (mid <= w.first || mid >= w.first + w.count)) mid = w.first + w.count / 2;
    }
    if (makeLeaf) {
      putNode(w.node, bounds, (w.count << BVH_COUNT_SHIFT) | w.first);
    } else {
      uint32_t pair = nextPair++;
      putNode(w.node, bounds, pair);
      putNode(2 * pair + 1, zero, 0);             // reserve both slots before the children are filled in
      stack.push_back(Work{2 * pair + 1, mid, w.first + w.count - mid, w.depth + 1});
      stack.push_back(Work{2 * pair, w.first, mid - w.first, w.depth + 1});
    }
  }
  out.numNodes = (uint32_t)(out.nodes.size() / 8);
  out.triOrder = idx;
  uint32_t rootRef; std::memcpy(&rootRef, &out.nodes[3], 4);
  out.rootRef = rootRef;
  out.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void treeletLayout(Bvh& b) {
  const uint32_t P = b.numNodes / 2;
  if (P <= 2) return;
  constexpr uint32_t NONE = 0xFFFFFFFFu, LINE = 4;
  auto refOf = [&](uint32_t node) { uint32_t r; std::memcpy(&r, &b.nodes[(size_t)node * 8 + 3], 4); return r; };
  auto internal = [](uint32_t ref) { return (ref >> BVH_COUNT_SHIFT) == 0; };
  std::vector<uint32_t> newIdx(P, NONE), order;          // old pair -> new pair; new pair -> old pair (NONE: padding)
  order.reserve((size_t)P + P / 4 + LINE);
  std::vector<uint32_t> open[LINE];                     // open[f]: first free slot of lines with f slots left
  order.assign(LINE, NONE); order[0] = 0; newIdx[0] = 0; open[LINE - 1].push_back(1);
  std::vector<uint32_t> roots;                          // pairs that start a treelet, depth first
  if (internal(b.rootRef)) roots.push_back(b.rootRef & BVH_INDEX_MASK);
  std::vector<uint32_t> frontier;
  while (!roots.empty()) {
    const uint32_t r = roots.back(); roots.pop_back();
    uint32_t g[LINE], n = 0;
    frontier.clear(); frontier.push_back(r);
    for (size_t at = 0; at < frontier.size() && n < LINE; at++) {
      const uint32_t p = frontier[at];
      g[n++] = p;
      for (uint32_t k = 0; k < 2; k++) { uint32_t ref = refOf(2 * p + k); if (internal(ref)) frontier.push_back(ref & BVH_INDEX_MASK); }
    }
    for (size_t at = frontier.size(); at > n; at--) roots.push_back(frontier[at - 1]);     // left to right off the stack
    uint32_t slot = NONE;
    for (uint32_t f = n; f < LINE && slot == NONE; f++)
      if (!open[f].empty()) {
        slot = open[f].back(); open[f].pop_back();
        if (f > n) open[f - n].push_back(slot + n);
      }
    if (slot == NONE) {
      slot = (uint32_t)order.size(); order.resize(order.size() + LINE, NONE);
      if (n < LINE) open[LINE - n].push_back(slot + n);
    }
    for (uint32_t i = 0; i < n; i++) { order[slot + i] = g[i]; newIdx[g[i]] = slot + i; }
  }
  std::vector<float> out(order.size() * 16);
  for (size_t np = 0; np < order.size(); np++) {
    float* dst = &out[np * 16];
    if (order[np] == NONE) {
      std::memcpy(dst, &b.nodes[0], 32); std::memcpy(dst + 8, &b.nodes[0], 32);
      std::memcpy(dst + 3, &NONE, 4); std::memcpy(dst + 11, &NONE, 4);
      continue;
    }
    std::memcpy(dst, &b.nodes[(size_t)order[np] * 16], 64);
    if (np == 0) continue;                               // root box + padding: the root reference is set below
    for (uint32_t k = 0; k < 2; k++) {
      uint32_t ref; std::memcpy(&ref, dst + 8 * k + 3, 4);
      if (internal(ref)) { ref = newIdx[ref & BVH_INDEX_MASK]; std::memcpy(dst + 8 * k + 3, &ref, 4); }
    }
  }
  if (internal(b.rootRef)) { b.rootRef = newIdx[b.rootRef & BVH_INDEX_MASK]; std::memcpy(&out[3], &b.rootRef, 4); }
  b.nodes.swap(out);
  b.numNodes = (uint32_t)(b.nodes.size() / 8);
}

void quantizeNodes(const float* nodes, uint32_t numNodes, uint32_t* out, float qBase[3], float qScale[3]) {
  double cell[3], base[3];
  for (int k = 0; k < 3; k++) {
    double lo = numNodes ? nodes[k] : 0.0, hi = numNodes ? nodes[4 + k] : 0.0;
    if (!(hi >= lo)) { lo = 0.0; hi = 0.0; }                            // empty scene: the root box is inverted
    cell[k] = std::max((hi - lo) / 65529.0, 1e-30);                     // three spare cells at either end
    base[k] = lo - 3.0 * cell[k];
    qBase[k] = (float)base[k]; qScale[k] = (float)cell[k];
  }
  for (uint32_t n = 0; n < numNodes; n++) {
    const float* p = nodes + (size_t)n * 8;
    uint32_t w[3];
    for (int k = 0; k < 3; k++) {
      float mn = p[k], mx = p[4 + k];
      long qlo, qhi;
      if (!(mx >= mn)) { qlo = 65535; qhi = 0; }                        // an empty box (padding node) stays empty
      else {
        qlo = (long)std::floor(((double)mn - base[k]) / cell[k]) - 1;
        qhi = (long)std::ceil(((double)mx - base[k]) / cell[k]) + 1;
        qlo = std::min(std::max(qlo, 0l), 65535l); qhi = std::min(std::max(qhi, 0l), 65535l);
        // containment under the float evaluation the kernels and dequantizeNodes use
        while (qlo > 0 && !(fmaf((float)qlo, qScale[k], qBase[k]) <= mn)) qlo--;
        while (qhi < 65535 && !(fmaf((float)qhi, qScale[k], qBase[k]) >= mx)) qhi++;
      }
      w[k] = (uint32_t)qlo | ((uint32_t)qhi << 16);
    }
    uint32_t ref; std::memcpy(&ref, p + 3, 4);
    uint32_t* o = out + (size_t)n * 4;
    o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; o[3] = ref;
  }
}

void dequantizeNodes(const uint32_t* q, uint32_t numNodes, const float qBase[3], const float qScale[3], float* nodes) {
  for (uint32_t n = 0; n < numNodes; n++) {
    const uint32_t* w = q + (size_t)n * 4; float* p = nodes + (size_t)n * 8;
    for (int k = 0; k < 3; k++) {
      p[k] = fmaf((float)(w[k] & 0xFFFFu), qScale[k], qBase[k]);
      p[4 + k] = fmaf((float)(w[k] >> 16), qScale[k], qBase[k]);
    }
    std::memcpy(p + 3, w + 3, 4); p[7] = 0.0f;
  }
}

}  // namespace pt
