"""Experiment (not part of the library): patches kernels.hip so that the shadow queues of a scene in global memory are
sorted over the whole launch with rocPRIM's radix sort (key: Morton code of the origin's cell at PT_SORT_BITS bits per
axis, then the direction's octant) before k_connect_refill / k_occluded_refill read them.  Build with
scripts/build_variant.sh <tag> -DPT_SORT_BITS=<n> and undo with `git checkout pathtracer_amd/csrc/kernels.hip`."""
p = 'pathtracer_amd/csrc/kernels.hip'; s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new, 1)
rep("""                                                     uint32_t* taskCounter, float area, float4* __restrict__ Lacc,
                                                     unsigned long long* stats) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  __syncthreads();""", """                                                     uint32_t* taskCounter, float area, float4* __restrict__ Lacc,
                                                     unsigned long long* stats, const uint32_t* __restrict__ order) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  __syncthreads();""")
rep("""        e = physIndexHint(q, j, sup.seg, segCap, cur);
        const float4 ro = shO[e], rd = shD[e];""", """        e = order ? order[j] : physIndexHint(q, j, sup.seg, segCap, cur);
        const float4 ro = shO[e], rd = shD[e];""")
rep("""                                                      uint32_t* taskCounter, float4* __restrict__ Lacc, unsigned long long* stats) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  __syncthreads();""", """                                                      uint32_t* taskCounter, float4* __restrict__ Lacc, unsigned long long* stats,
                                                      const uint32_t* __restrict__ order) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  __syncthreads();""")
rep("""        e = physIndexHint(q, j, sup.seg, segCap, cur);
        const float4 ro = shO[e], a = shD[e], b = shW[e];
        const bool eSun""", """        e = order ? order[j] : physIndexHint(q, j, sup.seg, segCap, cur);
        const float4 ro = shO[e], a = shD[e], b = shW[e];
        const bool eSun""")
helper = r'''
#ifndef PT_SORT_BITS
#define PT_SORT_BITS 0
#endif
#ifndef PT_SORT_WHICH
#define PT_SORT_WHICH 3
#endif
#ifndef PT_SORT_OCTANT
#define PT_SORT_OCTANT 1
#endif
#if PT_SORT_BITS
namespace {
__device__ __forceinline__ uint32_t spread3(uint32_t x) {
  x &= 0x3FFu; x = (x | x << 16) & 0x030000FFu; x = (x | x << 8) & 0x0300F00Fu; x = (x | x << 4) & 0x030C30C3u; x = (x | x << 2) & 0x09249249u;
  return x;
}
__global__ void k_shadow_keys(DevScene sc, QMeta q, uint32_t segCap, uint32_t cap, const float4* __restrict__ shO,
                              const float4* __restrict__ dirs, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= cap) return;
  const uint32_t seg = e / segCap, i = e - seg * segCap;
  uint32_t key = 1u << (3 * PT_SORT_BITS + 3);
  if (i < q.prefix[seg + 1] - q.prefix[seg]) {
    const float4 o = shO[e], d = dirs[e];
    const float g = (float)(1u << PT_SORT_BITS) / 65536.0f;
    uint32_t c[3]; const float ov[3] = {o.x, o.y, o.z};
    for (int k = 0; k < 3; ++k) {
      float f = (ov[k] - sc.qBase[k]) / sc.qScale[k] * g;
      c[k] = (uint32_t)fminf(fmaxf(f, 0.0f), (float)((1u << PT_SORT_BITS) - 1u));
    }
    const uint32_t oct = PT_SORT_OCTANT ? (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u) : 0u;
    key = (spread3(c[0]) | spread3(c[1]) << 1 | spread3(c[2]) << 2) << 3 | oct;
  }
  keys[e] = key; vals[e] = e;
}
struct SortBufs { uint32_t* k[2] = {nullptr, nullptr}; uint32_t* v[2] = {nullptr, nullptr}; void* tmp = nullptr; size_t tmpBytes = 0; uint32_t cap = 0; };
SortBufs gSort;
const uint32_t* sortShadowQueue(const LaunchCfg& lc, const DevScene& sc, const QMeta& q, uint32_t cap, const float4* shO, const float4* dirs) {
  if (gSort.cap < cap) {
    for (int i = 0; i < 2; ++i) { (void)hipFree(gSort.k[i]); (void)hipFree(gSort.v[i]); (void)hipMalloc((void**)&gSort.k[i], (size_t)cap * 4); (void)hipMalloc((void**)&gSort.v[i], (size_t)cap * 4); }
    (void)hipFree(gSort.tmp);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, gSort.tmpBytes, gSort.k[0], gSort.k[1], gSort.v[0], gSort.v[1], (int)cap, 0, 3 * PT_SORT_BITS + 4, lc.stream);
    (void)hipMalloc(&gSort.tmp, gSort.tmpBytes);
    gSort.cap = cap;
  }
  hipLaunchKernelGGL(k_shadow_keys, dim3((cap + 255u) / 256u), dim3(256), 0, lc.stream, sc, q, cap / PT_SEGS, cap, shO, dirs, gSort.k[0], gSort.v[0]);
  (void)hipcub::DeviceRadixSort::SortPairs(gSort.tmp, gSort.tmpBytes, gSort.k[0], gSort.k[1], gSort.v[0], gSort.v[1], (int)cap, 0, 3 * PT_SORT_BITS + 4, lc.stream);
  return gSort.v[1];
}
}  // namespace
#endif

void launchConnect('''
rep("\nvoid launchConnect(", helper)
rep("""    hipLaunchKernelGGL(k_connect_refill, g.grid, g.block, 0, lc.stream, sc, cfg.eps, wf.sh[kind], wf.capQueue / PT_SEGS,
                       (const float4*)(wf.shO + e), (const float4*)(wf.shD + e), (const float4*)(wf.shW + e),
                       wf.taskCounters + (1 + kind) * PT_TASK_PARTS, area, wf.Lacc, wf.stats);""", """    const uint32_t* order = nullptr;
#if PT_SORT_BITS
    if (PT_SORT_WHICH & 1) order = sortShadowQueue(lc, sc, wf.sh[kind], wf.capQueue, wf.shO + e, wf.shD + e);
#endif
    hipLaunchKernelGGL(k_connect_refill, g.grid, g.block, 0, lc.stream, sc, cfg.eps, wf.sh[kind], wf.capQueue / PT_SEGS,
                       (const float4*)(wf.shO + e), (const float4*)(wf.shD + e), (const float4*)(wf.shW + e),
                       wf.taskCounters + (1 + kind) * PT_TASK_PARTS, area, wf.Lacc, wf.stats, order);""")
rep("""    hipLaunchKernelGGL(k_occluded_refill, g.grid, g.block, 0, lc.stream, sc, cfg.eps, cfg.sunRadiance, cfg.skyRadiance,""", """    const uint32_t* order = nullptr;
#if PT_SORT_BITS
    if (PT_SORT_WHICH & 2) order = sortShadowQueue(lc, sc, wf.sh[kind], wf.capQueue, wf.shO + e, wf.shW + e);
#endif
    hipLaunchKernelGGL(k_occluded_refill, g.grid, g.block, 0, lc.stream, sc, cfg.eps, cfg.sunRadiance, cfg.skyRadiance,""")
rep("""                       (const float4*)(wf.shD + e), (const float4*)(wf.shW + e), (const float4*)wf.shX,
                       wf.taskCounters + (1 + kind) * PT_TASK_PARTS, wf.Lacc, wf.stats);
    return;""", """                       (const float4*)(wf.shD + e), (const float4*)(wf.shW + e), (const float4*)wf.shX,
                       wf.taskCounters + (1 + kind) * PT_TASK_PARTS, wf.Lacc, wf.stats, order);
    return;""")
rep("namespace pt {\n", "#if defined(PT_SORT_BITS) && PT_SORT_BITS\n#include <hipcub/hipcub.hpp>\n#endif\nnamespace pt {\n") if s.count("namespace pt {\n") == 1 else None
open(p, 'w').write(s)
