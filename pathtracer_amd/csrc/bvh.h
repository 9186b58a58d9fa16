// Flattened acceleration structure for the traversal kernels.  Replaces the reference's pointer-linked
// bounding-sphere tree (Scene.Node, scene.d:145-156; O(T^2) build at scene.d:457-500) with an
// axis-aligned BVH2 laid out for the GPU: 32-byte nodes, sibling pairs adjacent, triangles in leaf order.
#pragma once
#include <stdint.h>
#include <vector>

namespace pt {

// node = { min.xyz, ref, max.xyz, 0 } as 8 x 32-bit words.
// ref: bits 31..28 = triangle count (0 => internal), bits 27..0 = pair index of the two children
// (children are nodes 2*pair and 2*pair+1) or index of the leaf's first triangle in leaf order.
constexpr uint32_t BVH_COUNT_SHIFT = 28;
constexpr uint32_t BVH_INDEX_MASK = (1u << BVH_COUNT_SHIFT) - 1;
constexpr uint32_t BVH_MAX_LEAF = 15;
constexpr uint32_t BVH_MAX_DEPTH = 60;       // the kernels' traversal stack holds 64 entries

struct Bvh {
  std::vector<float> nodes;          // numNodes x 8 (node 1 is padding so that pairs stay aligned)
  std::vector<uint32_t> triOrder;    // original triangle id per leaf-order slot (degenerate triangles dropped)
  uint32_t rootRef = 0;
  uint32_t depth = 0;
  uint32_t numNodes = 0;
  double buildSeconds = 0;
};

// The tree for traversal from global memory: 16 B per node instead of 32.  Box planes on a 16-bit grid over the root
// box (node 0), per node 4 words: x: min | max << 16, y and z likewise, w: ref.  plane = fmaf(q, qScale, qBase) in
// float, and every quantised box contains its original under exactly that evaluation (quantizeNodes rounds outwards,
// one grid cell beyond the float rounding of the evaluation).  out = numNodes x 4 words.
void quantizeNodes(const float* nodes, uint32_t numNodes, uint32_t* out, float qBase[3], float qScale[3]);
// The boxes the quantised tree stands for, in the layout of Bvh::nodes.
void dequantizeNodes(const uint32_t* q, uint32_t numNodes, const float qBase[3], const float qScale[3], float* nodes);

// Re-orders the sibling pairs for traversal from global memory: a 128-byte cache line holds four pairs of 16-byte
// nodes, and after this every line holds a pair together with descendants of it (a treelet of up to four pairs, grown
// breadth-first), so that the step after a line was fetched usually finds its pair in the same line.  Small treelets
// share lines; slots left over hold unreferenced padding (ref = 0xFFFFFFFF).  Pair 0 (root box + padding) stays first.
// Changes only where pairs lie and the references between them: the tree, and so every traversal result, is the same.
void treeletLayout(Bvh& bvh);

// v = T x 9 floats (v0, v1, v2).  Binned-SAH top-down build, leaves of at most maxLeaf triangles.  triCost: cost of a
// triangle test relative to a step (one sibling pair): below 1 the tree gets fewer nodes and fuller leaves.
void buildBvh(const float* v, uint32_t T, uint32_t maxLeaf, Bvh& out, float triCost = 1.0f);

}  // namespace pt
