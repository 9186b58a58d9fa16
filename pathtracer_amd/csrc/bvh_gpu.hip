// BVH construction on the GPU for scenes whose host build would take seconds (the 10 M-triangle soup: 10.8 s of
// binned SAH on one host core).  Replaces, like bvh.cpp, the reference's O(T^2) bounding-sphere tree build
// (scene.d:329-637) -- the step before the path.
//
// Linear BVH: 63-bit Morton codes of the triangle centroids, radix sort (hipCUB), the binary radix tree of
// Karras ("Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", 2012) with one thread per
// internal node, boxes fitted bottom-up (the second thread to arrive at a node merges its children), subtrees of
// at most maxLeaf triangles collapsed into leaves, and the result written in the layout of bvh.h: sibling pairs
// adjacent, pair 0 = {root, padding}, triangles of a leaf contiguous in (Morton) leaf order.  The traversal
// result does not depend on the tree (conservative boxes, lowest-id tie-break), so hits are the same bits as with
// the SAH tree; only the number of steps differs.
#include "bvh_gpu.h"
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>

namespace pt {
namespace {

#define GPUCHECK(expr)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) { why = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
  } while (0)

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
  template <typename T> T* as() const { return (T*)p; }
};

// order-preserving map float -> uint32 (for atomicMin / atomicMax)
__host__ __device__ inline uint32_t encF(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float decF(uint32_t e) { uint32_t u = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e; float f; memcpy(&f, &u, 4); return f; }

constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint64_t KEY_INVALID = ~0ull;

// bounds[0..2] = centroid min, [3..5] = centroid max, [6] = max |coordinate| (all encoded), [7] = number of valid triangles
__global__ void k_prepare(const float* __restrict__ v, uint32_t T, float* __restrict__ cen, uint8_t* __restrict__ valid,
                          uint32_t* bounds) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float cmn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, cmx[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, amax = 0.0f;
  uint32_t ok = 0;
  if (i < T) {
    const float* p = v + 9 * (size_t)i;
    bool finite = true;
    for (int k = 0; k < 9; k++) finite = finite && isfinite(p[k]);
    // the same triangles the host builder drops: never hit (determinant exactly zero for every ray), or NaN
    bool same01 = p[0] == p[3] && p[1] == p[4] && p[2] == p[5];
    bool same02 = p[0] == p[6] && p[1] == p[7] && p[2] == p[8];
    ok = finite && !same01 && !same02;
    if (ok) {
      for (int k = 0; k < 3; k++) {
        float lo = fminf(p[k], fminf(p[3 + k], p[6 + k])), hi = fmaxf(p[k], fmaxf(p[3 + k], p[6 + k]));
        float c = 0.5f * (lo + hi);
        cen[3 * (size_t)i + k] = c; cmn[k] = c; cmx[k] = c;
        amax = fmaxf(amax, fmaxf(fabsf(lo), fabsf(hi)));
      }
    }
    valid[i] = (uint8_t)ok;
  }
  for (int off = 32; off > 0; off >>= 1) {
    for (int k = 0; k < 3; k++) { cmn[k] = fminf(cmn[k], __shfl_down(cmn[k], off)); cmx[k] = fmaxf(cmx[k], __shfl_down(cmx[k], off)); }
    amax = fmaxf(amax, __shfl_down(amax, off));
    ok += __shfl_down(ok, off);
  }
  if ((threadIdx.x & 63) == 0 && ok) {
    for (int k = 0; k < 3; k++) { atomicMin(&bounds[k], encF(cmn[k])); atomicMax(&bounds[3 + k], encF(cmx[k])); }
    atomicMax(&bounds[6], encF(amax));
    atomicAdd(&bounds[7], ok);
  }
}

__device__ inline uint64_t spread21(uint32_t x) {       // 21 bits -> every third bit of 63
  uint64_t v = x & 0x1FFFFFu;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

__global__ void k_morton(const float* __restrict__ cen, const uint8_t* __restrict__ valid, uint32_t T, const uint32_t* bounds,
                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  vals[i] = i;
  if (!valid[i]) { keys[i] = KEY_INVALID; return; }
  uint32_t q[3];
  for (int k = 0; k < 3; k++) {
    float lo = decF(bounds[k]), hi = decF(bounds[3 + k]);
    float ext = hi - lo;
    float t = ext > 0.0f ? (cen[3 * (size_t)i + k] - lo) / ext : 0.0f;
    t = fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f);
    q[k] = (uint32_t)t;
  }
  keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);   // < 2^63: never KEY_INVALID
}

// length of the common prefix of the keys of sorted positions i and j (-1 outside the array); equal keys are told
// apart by their positions
__device__ inline int delta(const uint64_t* __restrict__ keys, int N, int i, int j) {
  if (j < 0 || j >= N) return -1;
  uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clzll((long long)(a ^ b));
}

// one thread per internal node of the radix tree over N sorted keys (N - 1 internal nodes, node 0 = root)
__global__ void k_radix_tree(const uint64_t* __restrict__ keys, int N, uint32_t* __restrict__ childL, uint32_t* __restrict__ childR,
                             uint32_t* __restrict__ first, uint32_t* __restrict__ last, uint32_t* __restrict__ parentOfInternal,
                             uint32_t* __restrict__ parentOfLeaf) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N - 1) return;
  int d = (delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, N, i, i - d);
  int lmax = 2;
  while (lmax < 2 * N && delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, N, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) / 2;
    if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int gamma = i + s * d + (d < 0 ? -1 : 0);
  int lo = i < j ? i : j, hi = i < j ? j : i;
  first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
  if (lo == gamma) { childL[i] = (uint32_t)gamma | LEAF_BIT; parentOfLeaf[gamma] = (uint32_t)i; }
  else { childL[i] = (uint32_t)gamma; parentOfInternal[gamma] = (uint32_t)i; }
  if (hi == gamma + 1) { childR[i] = (uint32_t)(gamma + 1) | LEAF_BIT; parentOfLeaf[gamma + 1] = (uint32_t)i; }
  else { childR[i] = (uint32_t)(gamma + 1); parentOfInternal[gamma + 1] = (uint32_t)i; }
}

// one thread per sorted triangle: its box, then up the tree; the second thread to reach a node merges the
// children's boxes and goes on, the first one stops.  Boxes: 6 floats, leaves first [0, N), then internals.
__global__ void k_fit(const float* __restrict__ v, const uint32_t* __restrict__ vals, int N, const uint32_t* __restrict__ childL,
                      const uint32_t* __restrict__ childR, const uint32_t* __restrict__ parentOfInternal,
                      const uint32_t* __restrict__ parentOfLeaf, float* boxes, uint32_t* arrived) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const float* p = v + 9 * (size_t)vals[k];
  float* b = boxes + 6 * (size_t)k;
  for (int a = 0; a < 3; a++) {
    b[a] = fminf(p[a], fminf(p[3 + a], p[6 + a]));
    b[3 + a] = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
  }
  if (N < 2) return;
  uint32_t cur = parentOfLeaf[k];
  for (int guard = 0; guard < 200; guard++) {            // the tree is at most 64 + 32 levels deep
    __threadfence();
    if (atomicAdd(&arrived[cur], 1u) == 0u) return;
    __threadfence();
    uint32_t l = childL[cur], r = childR[cur];
    const volatile float* bl = boxes + 6 * (size_t)((l & LEAF_BIT) ? (l & ~LEAF_BIT) : (uint32_t)N + l);
    const volatile float* br = boxes + 6 * (size_t)((r & LEAF_BIT) ? (r & ~LEAF_BIT) : (uint32_t)N + r);
    float* bc = boxes + 6 * ((size_t)N + cur);
    for (int a = 0; a < 3; a++) { bc[a] = fminf(bl[a], br[a]); bc[3 + a] = fmaxf(bl[3 + a], br[3 + a]); }
    if (cur == 0) return;
    cur = parentOfInternal[cur];
  }
}

__global__ void k_flags(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, int N, uint32_t maxLeaf,
                        uint32_t* __restrict__ real) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N - 1) return;
  real[i] = (last[i] - first[i] + 1u > maxLeaf) ? 1u : 0u;
}

// reference to child c (radix-tree encoding) in the layout of bvh.h
__device__ inline uint32_t refOf(uint32_t c, const uint32_t* first, const uint32_t* last, const uint32_t* real, const uint32_t* rank) {
  if (c & LEAF_BIT) return (1u << BVH_COUNT_SHIFT) | (c & ~LEAF_BIT);
  if (real[c]) return 1u + rank[c];                                    // pair that holds c's children
  return ((last[c] - first[c] + 1u) << BVH_COUNT_SHIFT) | first[c];     // collapsed subtree = one leaf
}

__device__ inline void putNode(float* nodes, uint32_t node, const float* box, float pad, uint32_t ref) {
  float* n = nodes + 8 * (size_t)node;
  n[0] = box[0] - pad; n[1] = box[1] - pad; n[2] = box[2] - pad; n[3] = __uint_as_float(ref);
  n[4] = box[3] + pad; n[5] = box[4] + pad; n[6] = box[5] + pad; n[7] = 0.0f;
}

// one thread per internal node that survives the collapse: writes the pair of its two children
__global__ void k_emit(int N, const uint32_t* __restrict__ childL, const uint32_t* __restrict__ childR,
                       const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, const uint32_t* __restrict__ real,
                       const uint32_t* __restrict__ rank, const float* __restrict__ boxes, const uint32_t* bounds,
                       float* __restrict__ nodes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N - 1 || !real[i]) return;
  float scale = fmaxf(1.0f, decF(bounds[6]));
  float pad = scale * 1e-5f;                                           // as the host builder (bvh.cpp)
  uint32_t pair = 1u + rank[i];
  uint32_t c[2] = {childL[i], childR[i]};
  for (int s = 0; s < 2; s++) {
    const float* b = boxes + 6 * (size_t)((c[s] & LEAF_BIT) ? (c[s] & ~LEAF_BIT) : (uint32_t)N + c[s]);
    putNode(nodes, 2 * pair + s, b, pad, refOf(c[s], first, last, real, rank));
  }
  if (i == 0) {                                                        // the root itself, and the padding next to it
    putNode(nodes, 0, boxes + 6 * (size_t)N, pad, pair);
    const float zero[6] = {0, 0, 0, 0, 0, 0};
    putNode(nodes, 1, zero, 0.0f, 0u);
  }
}

}  // namespace

bool buildBvhGpu(const float* v, uint32_t T, uint32_t maxLeaf, Bvh& out, std::string& why) {
  auto t0 = std::chrono::steady_clock::now();
  out = Bvh();
  maxLeaf = std::max(1u, std::min(maxLeaf, BVH_MAX_LEAF));
  if (T < 2) { why = "fewer than two triangles"; return false; }
  if (T >= (1u << 28)) { why = "too many triangles"; return false; }

  Buf dV, dCen, dValid, dBounds, dKeys, dKeys2, dVals, dVals2, dTmp;
  GPUCHECK(dV.alloc((size_t)T * 36)); GPUCHECK(dCen.alloc((size_t)T * 12)); GPUCHECK(dValid.alloc(T));
  GPUCHECK(dBounds.alloc(32)); GPUCHECK(dKeys.alloc((size_t)T * 8)); GPUCHECK(dKeys2.alloc((size_t)T * 8));
  GPUCHECK(dVals.alloc((size_t)T * 4)); GPUCHECK(dVals2.alloc((size_t)T * 4));
  GPUCHECK(hipMemcpy(dV.p, v, (size_t)T * 36, hipMemcpyHostToDevice));
  const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
  GPUCHECK(hipMemcpy(dBounds.p, init, 32, hipMemcpyHostToDevice));
  const int B = 256;
  const unsigned gT = (T + B - 1) / B;
  hipLaunchKernelGGL(k_prepare, dim3(gT), dim3(B), 0, 0, dV.as<float>(), T, dCen.as<float>(), dValid.as<uint8_t>(), dBounds.as<uint32_t>());
  hipLaunchKernelGGL(k_morton, dim3(gT), dim3(B), 0, 0, dCen.as<float>(), dValid.as<uint8_t>(), T, dBounds.as<uint32_t>(),
                     dKeys.as<uint64_t>(), dVals.as<uint32_t>());
  GPUCHECK(hipGetLastError());
  size_t tmpBytes = 0;
  GPUCHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, dKeys.as<uint64_t>(), dKeys2.as<uint64_t>(), dVals.as<uint32_t>(),
                                              dVals2.as<uint32_t>(), (int)T, 0, 64, (hipStream_t)0));
  GPUCHECK(dTmp.alloc(tmpBytes));
  GPUCHECK(hipcub::DeviceRadixSort::SortPairs(dTmp.p, tmpBytes, dKeys.as<uint64_t>(), dKeys2.as<uint64_t>(), dVals.as<uint32_t>(),
                                              dVals2.as<uint32_t>(), (int)T, 0, 64, (hipStream_t)0));
  uint32_t hb[8];
  GPUCHECK(hipMemcpy(hb, dBounds.p, 32, hipMemcpyDeviceToHost));
  const int N = (int)hb[7];                               // valid triangles; invalid ones sorted to the end
  if (N < 2) { why = "fewer than two hittable triangles"; return false; }

  const uint64_t* keys = dKeys2.as<uint64_t>();
  const uint32_t* vals = dVals2.as<uint32_t>();
  Buf dL, dR, dFirst, dLast, dPI, dPL, dBoxes, dArr, dReal, dRank, dNodes;
  const size_t NI = (size_t)N - 1;
  GPUCHECK(dL.alloc(NI * 4)); GPUCHECK(dR.alloc(NI * 4)); GPUCHECK(dFirst.alloc(NI * 4)); GPUCHECK(dLast.alloc(NI * 4));
  GPUCHECK(dPI.alloc(NI * 4)); GPUCHECK(dPL.alloc((size_t)N * 4)); GPUCHECK(dBoxes.alloc((2 * (size_t)N - 1) * 24));
  GPUCHECK(dArr.alloc(NI * 4)); GPUCHECK(dReal.alloc(NI * 4)); GPUCHECK(dRank.alloc(NI * 4));
  GPUCHECK(hipMemset(dArr.p, 0, NI * 4));
  GPUCHECK(hipMemset(dPI.p, 0, NI * 4));
  const unsigned gN = ((unsigned)N + B - 1) / B;
  hipLaunchKernelGGL(k_radix_tree, dim3(gN), dim3(B), 0, 0, keys, N, dL.as<uint32_t>(), dR.as<uint32_t>(), dFirst.as<uint32_t>(),
                     dLast.as<uint32_t>(), dPI.as<uint32_t>(), dPL.as<uint32_t>());
  hipLaunchKernelGGL(k_fit, dim3(gN), dim3(B), 0, 0, dV.as<float>(), vals, N, dL.as<uint32_t>(), dR.as<uint32_t>(), dPI.as<uint32_t>(),
                     dPL.as<uint32_t>(), dBoxes.as<float>(), dArr.as<uint32_t>());
  hipLaunchKernelGGL(k_flags, dim3(gN), dim3(B), 0, 0, dFirst.as<uint32_t>(), dLast.as<uint32_t>(), N, maxLeaf, dReal.as<uint32_t>());
  GPUCHECK(hipGetLastError());
  size_t scanBytes = 0;
  GPUCHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, dReal.as<uint32_t>(), dRank.as<uint32_t>(), (int)NI, (hipStream_t)0));
  Buf dScanTmp; GPUCHECK(dScanTmp.alloc(scanBytes));
  GPUCHECK(hipcub::DeviceScan::ExclusiveSum(dScanTmp.p, scanBytes, dReal.as<uint32_t>(), dRank.as<uint32_t>(), (int)NI, (hipStream_t)0));
  uint32_t lastRank = 0, lastReal = 0, rootReal = 0;
  GPUCHECK(hipMemcpy(&lastRank, dRank.as<uint32_t>() + (NI - 1), 4, hipMemcpyDeviceToHost));
  GPUCHECK(hipMemcpy(&lastReal, dReal.as<uint32_t>() + (NI - 1), 4, hipMemcpyDeviceToHost));
  GPUCHECK(hipMemcpy(&rootReal, dReal.p, 4, hipMemcpyDeviceToHost));
  if (!rootReal) { why = "all triangles fit one leaf"; return false; }        // tiny scene: the host builder handles it
  const uint32_t numPairs = 1u + lastRank + lastReal;
  if (numPairs >= (1u << 28)) { why = "too many nodes"; return false; }
  GPUCHECK(dNodes.alloc((size_t)numPairs * 64));
  hipLaunchKernelGGL(k_emit, dim3(gN), dim3(B), 0, 0, N, dL.as<uint32_t>(), dR.as<uint32_t>(), dFirst.as<uint32_t>(), dLast.as<uint32_t>(),
                     dReal.as<uint32_t>(), dRank.as<uint32_t>(), dBoxes.as<float>(), dBounds.as<uint32_t>(), dNodes.as<float>());
  GPUCHECK(hipGetLastError());
  out.numNodes = 2 * numPairs;
  out.nodes.resize((size_t)out.numNodes * 8);
  out.triOrder.resize((size_t)N);
  GPUCHECK(hipMemcpy(out.nodes.data(), dNodes.p, out.nodes.size() * 4, hipMemcpyDeviceToHost));
  GPUCHECK(hipMemcpy(out.triOrder.data(), vals, (size_t)N * 4, hipMemcpyDeviceToHost));
  std::memcpy(&out.rootRef, &out.nodes[3], 4);

  // depth of the emitted tree (the traversal stack is sized for BVH_MAX_DEPTH): iterative walk on the host
  {
    std::vector<std::pair<uint32_t, uint32_t>> st;      // (pair, depth)
    st.push_back({out.rootRef, 1});
    uint32_t depth = 0;
    size_t visited = 0;
    while (!st.empty()) {
      auto [pair, dpt] = st.back(); st.pop_back();
      depth = std::max(depth, dpt);
      if (++visited > numPairs) { why = "cycle in the emitted tree"; return false; }
      for (int s = 0; s < 2; s++) {
        uint32_t ref; std::memcpy(&ref, &out.nodes[((size_t)2 * pair + s) * 8 + 3], 4);
        if ((ref >> BVH_COUNT_SHIFT) == 0) {
          if (ref == 0 || ref >= numPairs) { why = "bad child reference in the emitted tree"; return false; }
          st.push_back({ref, dpt + 1});
        } else if ((ref & BVH_INDEX_MASK) + (ref >> BVH_COUNT_SHIFT) > (uint32_t)N) { why = "bad leaf range in the emitted tree"; return false; }
      }
    }
    out.depth = depth;
    if (depth > BVH_MAX_DEPTH) { why = "radix tree deeper than the traversal stack (" + std::to_string(depth) + ")"; return false; }
  }
  out.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return true;
}

}  // namespace pt
