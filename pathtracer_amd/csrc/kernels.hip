// Wavefront path tracer kernels for MI355X (gfx950, wave64).
//
// One sample of the reference's computeL (rendering.d:464-484) is split into stages that each run over a
// whole batch of paths; the stages talk through structure-of-arrays queues in HBM:
//
//   raygen             getViewRay (rendering.d:278-285) + the part of Scene.trace that can be decided per pixel
//                      (coverage mask of the triangles' projections, else the scene's bounding box): rays
//                      that cannot hit take the `else` branch of computeL right away (sampleSky, rendering.d:480);
//                      the rest are compacted into a ray queue
//   extend             Scene.trace (scene.d:681-743) of a ray queue: view rays, later the indirect rays
//   shade              computeL / computeLdirect / computeLindirect bodies between the trace calls
//                                                           rendering.d:376-462, 472-481
//   connect            Scene.trace of a light-sample ray + the G term   rendering.d:414-425
//   occluded           Scene.hitsNothing of the sun and sky rays (one queue entry per vertex carries both)
//                                                           rendering.d:440-443, 455-457
//   accumulate         sum += L; n += 1                                  rendering.d:477-483
//
// Live paths are compacted when they are appended to the next queue.  Queues are segmented (kernels.h, QMeta):
// a lane's rank inside its wave comes from a wave64 ballot + mbcnt, the wave's base from an LDS counter of its
// workgroup, the workgroup's base is the fixed start of its segment -- no global atomic is on the data path (a
// single 4-byte word sustains < 10^8 atomics/s, which at these ray rates is less than one per workgroup).
// k_scan turns the per-segment counts into the prefix sums the next stage indexes with.  raygen and shade run as
// PT_SEGS static block-tasks (uniform work per entry); the traversal kernels are persistent and hand a queue out
// dynamically, 64 entries per wave at a time (WorkPool), because traversal times vary by an order of magnitude.
//
// Traversal: stack-based, near child first, over the BVH2 of bvh.h.  When the scene fits, nodes and
// triangles are staged in LDS once per (persistent) workgroup; otherwise they are read from HBM/L2, the nodes
// in a 16-byte form with the box planes on a 16-bit grid (bvh.h: quantizeNodes).
// The ray-box test is conservative (boxes are padded at build time, far distance inflated by 2 ulp) and
// the ray-triangle test is evaluated in one fixed order with explicit fma (pt_math.h), so the closest
// hit (t, u, v, triangle id) does not depend on the tree.
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <set>
#include <utility>

namespace pt {

extern __shared__ float4 g_lds[];   // [0, 2*numNodes) nodes, then 3*numTris triangle rows

namespace {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
// Counters are kept in PT_STAT_REPLICAS copies, each on a cache line of its own, and a workgroup adds to the copy of
// its index: thousands of waves adding to ONE word at the end of a launch queue up behind each other (a word takes
// fewer than 10^8 atomic adds a second -- half a millisecond for a launch of 8192 waves with six counters, more than
// the rest of a small pass).  ptgpu_get_stats sums the copies.
__device__ __forceinline__ void statAdd(unsigned long long* stats, int which, unsigned long long v) {
  atomicAdd(&stats[(blockIdx.x % PT_STAT_REPLICAS) * PT_STAT_STRIDE + which], v);
}
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ V3 xyz(float4 a) { return v3(a.x, a.y, a.z); }
__device__ __forceinline__ float4 f4(V3 a, float w) { return make_float4(a.x, a.y, a.z, w); }

template <bool LDSS> __device__ __forceinline__ float4 ldNode(const DevScene& sc, uint32_t i) {
  static_assert(LDSS, "from global memory the tree is read through DevScene::qnodes");
  return g_lds[i];
}
template <bool LDSS> __device__ __forceinline__ float4 ldTri(const DevScene& sc, uint32_t i) {
  if constexpr (LDSS) return g_lds[2 * sc.numNodes + i]; else return sc.tris[i];
}
// Keeps the .w of rows read from LDS live, so that they are read as ds_read_b128: when .w is unused the compiler
// shortens the read to ds_read_b96, which occupies the LDS for 8 cycles per wave where ds_read_b128 takes 4.
// Placed after the last read of a group (the asm waits for its operands).
template <bool LDSS> __device__ __forceinline__ void keepW(const float4& a, const float4& b) {
  if constexpr (LDSS) asm volatile("" ::"v"(a.w), "v"(b.w));
}

template <bool LDSS> __device__ __forceinline__ void stageScene(const DevScene& sc) {
  if constexpr (LDSS) {
    uint32_t nn = 2 * sc.numNodes, nt = 3 * sc.numTris;
    for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) g_lds[i] = sc.nodes[i];
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) g_lds[nn + i] = sc.tris[i];
    __syncthreads();
  }
}

// Reciprocal direction for the slab test.  Only the (conservative) box test uses it, never the triangle test.
// A plane at coordinate c is crossed at t = c * i + a.  Every axis has two addends: one for the plane at the box's
// minimum (a0) and one for the plane at its maximum (a1).  They are -o * i moved apart by a slack towards "earlier" for
// the plane the ray enters through and "later" for the one it leaves through: the triangle test subtracts the origin
// from the vertices in fp32, i.e. it answers for a ray whose origin is off by up to an ulp of the origin's
// coordinates -- of its largest coordinate on every axis, because the test's dot products mix the axes -- and the box
// test must not cull what that test accepts; the padding of the boxes (relative to the scene) covers that only while
// the origin is not much farther away than the scene is large.  The slack is 4 ulp of max|o| * |i| per axis (plus the
// same for the offset of a quantised grid): next to nothing for rays that start inside the scene, and it costs no
// instruction in the box test.  It keeps the traversal's answer equal to the exhaustive search's up to origins about
// 10^3 scene extents away; from about 10^4 on the triangle test itself accepts triangles that the exact ray misses by
// a tenth of the scene, which no box test can follow.
struct RayInv { float ix, iy, iz, a0x, a0y, a0z, a1x, a1y, a1z; };
__device__ __forceinline__ RayInv makeRayInv(V3 o, V3 d, V3 gridBase = v3(0, 0, 0)) {
  float dx = fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x;
  float dy = fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y;
  float dz = fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z;
  RayInv r;
  r.ix = __builtin_amdgcn_rcpf(dx); r.iy = __builtin_amdgcn_rcpf(dy); r.iz = __builtin_amdgcn_rcpf(dz);
  const float ox = -o.x * r.ix, oy = -o.y * r.iy, oz = -o.z * r.iz;
  const float k = 2.4e-7f;                                          // 4 * 2^-24
  const float om = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
  const float sx = (om + fabsf(gridBase.x)) * fabsf(r.ix) * k, sy = (om + fabsf(gridBase.y)) * fabsf(r.iy) * k,
              sz = (om + fabsf(gridBase.z)) * fabsf(r.iz) * k;
  // i > 0: the ray enters through the minimum plane
  r.a0x = r.ix > 0.0f ? ox - sx : ox + sx; r.a1x = r.ix > 0.0f ? ox + sx : ox - sx;
  r.a0y = r.iy > 0.0f ? oy - sy : oy + sy; r.a1y = r.iy > 0.0f ? oy + sy : oy - sy;
  r.a0z = r.iz > 0.0f ? oz - sz : oz + sz; r.a1z = r.iz > 0.0f ? oz + sz : oz - sz;
  return r;
}
// Direction octant: bit k set when component k of the (reciprocal) direction is negative.
__device__ __forceinline__ uint32_t octantOf(const RayInv& r) {
  return (__float_as_uint(r.ix) >> 31) | ((__float_as_uint(r.iy) >> 31) << 1) | ((__float_as_uint(r.iz) >> 31) << 2);
}
// Entry distance of the ray into the box [mn, mx]; hit iff tn <= min(far * (1 + 2 ulp), tmax).
// UNBOUNDED: tmax is not looked at (it is FLT_MAX).
// OCT < 0: any direction (near / far plane of each axis by min / max).  OCT = 0..7: the direction's octant is known
// at compile time, so is which plane of each axis is the near one, and the six min / max disappear.
template <int OCT = -1, bool UNBOUNDED = false>
__device__ __forceinline__ bool slab(const RayInv& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                     float tmax, float& tn) {
  float p0x = fmaf(mnx, r.ix, r.a0x), p1x = fmaf(mxx, r.ix, r.a1x);
  float p0y = fmaf(mny, r.iy, r.a0y), p1y = fmaf(mxy, r.iy, r.a1y);
  float p0z = fmaf(mnz, r.iz, r.a0z), p1z = fmaf(mxz, r.iz, r.a1z);
  float nx, ny, nz, fx, fy, fz;
  if constexpr (OCT < 0) {
    nx = fminf(p0x, p1x); fx = fmaxf(p0x, p1x);
    ny = fminf(p0y, p1y); fy = fmaxf(p0y, p1y);
    nz = fminf(p0z, p1z); fz = fmaxf(p0z, p1z);
  } else {
    nx = (OCT & 1) ? p1x : p0x; fx = (OCT & 1) ? p0x : p1x;
    ny = (OCT & 2) ? p1y : p0y; fy = (OCT & 2) ? p0y : p1y;
    nz = (OCT & 4) ? p1z : p0z; fz = (OCT & 4) ? p0z : p1z;
  }
  tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
  float tf = fminf(fminf(fx, fy), fz) * 1.0000005f;
  if constexpr (UNBOUNDED) return !(tn > tf);      // any-hit rays have no far limit; a NaN keeps the box, as fminf does
  return tn <= fminf(tf, tmax);
}

struct Hit { float t, u, v; uint32_t tri; bool front; };

// Workgroup sizes of the kernels that run as PT_SEGS static block-tasks.
#ifndef PT_RAYGEN_THREADS
#define PT_RAYGEN_THREADS 256
#endif
#ifndef PT_RAYGEN_UNROLL
#define PT_RAYGEN_UNROLL 4
#endif
#ifndef PT_SHADE_THREADS
#define PT_SHADE_THREADS 256
#endif
// The shading variant that traces its own rays is bound by vector instructions, not by its stores: it wants more
// waves per SIMD than 1024 block-tasks of 256 threads give (4).
#ifndef PT_FUSE_THREADS
#define PT_FUSE_THREADS 512
#endif
// Its traversals use the general box test, not the per-octant variants (17 more copies of the loop in one kernel).
#ifndef PT_FUSE_WAVES
#define PT_FUSE_WAVES 6
#endif
#ifndef PT_FUSE_OCTANTS
#define PT_FUSE_OCTANTS false
#endif
// Diagnostic builds (scripts/build_variant.sh <tag> -D...; never defined in the library that ships):
//   PT_STAMP / PT_SLOTS   the counting variants report wave clock ticks / lane slots of the descent loop instead of visits
//   PT_PROBE_HALF=n       extend gives a wave n rays instead of 64 (what part of a launch depends on busy lanes);
//   PT_PROBE_OCC=n        the same for the sun/sky kernel
//   PT_PROBE_TRI2         every triangle test of a scene in global memory fetches a second, random triangle as well
//                         (what triangle traffic costs: +18-20 % on the 10 M-triangle soup's traversal kernels)
//   PT_PROBE_A / _B / _C  the traced shading without its chain traversals / its chain loop / its sun and sky traversals
//   PT_PROBE_D            the chains' directions without angleToDirection
//                         (wrong images: timing only)
#if defined(PT_STAMP) || defined(PT_SLOTS)
#define PT_STAMP_ON 1
#else
#define PT_STAMP_ON 0
#endif


// Traversal kernels are compiled for 8 waves per SIMD: two 1024-thread workgroups per CU when the scene copy in LDS
// leaves room.  Besides 64 VGPRs this bounds the scalar registers: with more than 80 of them a SIMD holds only 7
// waves, and a second 16-wave workgroup no longer fits on the CU.
#define PT_TRAVERSAL_BOUNDS __launch_bounds__(1024, 8)

// One triangle of a leaf: the two-sided Moeller-Trumbore test in the fixed evaluation order of pt_math.h.
// ANY: returns true when the hit is accepted (Scene.hitsNothing stops there).  Otherwise updates the closest
// front-face hit and returns false.  (A straight-line form with one combined predicate instead of the nested
// rejections was measured: the vector work of rejected triangles outweighs the exec-mask regions it removes.)
// BOTH (closest hit only): back faces are accepted too and h.front tells which side was hit -- the texcoord
// variant of Scene.trace called with IgnoreBackfaces.no (scene.d:767-779).
template <bool ANY, bool LDSS, bool BOTH = false>
__device__ __forceinline__ bool testTriangle(const DevScene& sc, uint32_t row, V3 o, V3 d, float eps, Hit& h, bool& found) {
  float4 t0 = ldTri<LDSS>(sc, row), tA = ldTri<LDSS>(sc, row + 1), tB = ldTri<LDSS>(sc, row + 2);
  keepW<LDSS>(tA, tB); keepW<LDSS>(t0, t0);
#ifdef PT_PROBE_TRI2
  if constexpr (!LDSS) {       // as many bytes again from a random other triangle: what triangle traffic costs
    uint32_t r2 = ((row / 3) * 2654435761u) % sc.numTris * 3;
    float4 x0 = sc.tris[r2], x1 = sc.tris[r2 + 1], x2 = sc.tris[r2 + 2];
    asm volatile("" ::"v"(x0.x), "v"(x1.x), "v"(x2.x));
  }
#endif
  V3 A = xyz(tA), B = xyz(tB);
  V3 p = cross(d, B);
  float det = dot(A, p);
  if (!(fabsf(det) < eps)) {
    V3 s = o - xyz(t0);
    float inv = 1.0f / det;
    float u = dot(s, p) * inv;
    if (!(u < 0.0f || u > 1.0f)) {
      V3 q = cross(s, A);
      float v = dot(d, q) * inv;
      if (!(v < 0.0f || u + v > 1.0f)) {
        float t = dot(B, q) * inv;
        uint32_t orig = f2u(t0.w);
        if (ANY) {
          if (t >= eps) { h.t = t; h.u = u; h.v = v; h.tri = orig; return true; }
        } else {
          // pos < rayPos && pos >= FloatEpsilon, then front face; equal t keeps the lower triangle id
          if ((t < h.t || (t == h.t && orig < h.tri)) && t >= eps && (BOTH || det < 0.0f)) {
            h.t = t; h.u = u; h.v = v; h.tri = orig; found = true;
            if (BOTH) h.front = det < 0.0f;
          }
        }
      }
    }
  }
  return false;
}

// One step down the tree: tests the two children of internal node `ref` and returns the next reference -- the
// nearer child that is hit (the other one is pushed), else the top of the stack, else PT_NONE.  Selects instead
// of a three-way branch: the exec-mask bookkeeping of nested branches cost as many scalar instructions as the
// two box tests cost vector instructions.
constexpr uint32_t PT_NONE = 0xFFFFFFFFu;

// Pop with the wait for the load kept inside the branch that pops.  The stack lives in scratch memory and loads and
// stores share one counter (vmcnt): left to itself the compiler waits for the popped value where the branches
// join, and on the path that only pushed, that wait stalls on the push's store instead.
__device__ __forceinline__ uint32_t popRef(const uint32_t* stack, int& sp) {
  uint32_t r = stack[--sp];
  asm volatile("" : "+v"(r));
  return r;
}

// The traversal stack of a lane: an array in scratch memory (uint32_t*), or -- the kernels for scenes in global memory,
// whose LDS is idle and whose vector-memory address unit is what bounds them -- the first PT_LDS_STACK levels in LDS
// (the lane's column of its wave's block: level * 64 + lane, no bank conflicts whatever the lanes' levels) and the
// rest in scratch memory.
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 16
#endif
struct LdsStack { uint32_t* lds; uint32_t* rest; };
__device__ __forceinline__ void stkPush(uint32_t* s, int& sp, uint32_t v) { s[sp++] = v; }
__device__ __forceinline__ uint32_t stkPop(const uint32_t* s, int& sp) { return popRef(s, sp); }
__device__ __forceinline__ void stkPush(const LdsStack& s, int& sp, uint32_t v) {
  if (sp < PT_LDS_STACK) s.lds[sp * 64] = v; else s.rest[sp - PT_LDS_STACK] = v;
  sp++;
}
__device__ __forceinline__ uint32_t stkPop(const LdsStack& s, int& sp) {
  --sp;
  uint32_t r;
  if (sp < PT_LDS_STACK) { r = s.lds[sp * 64]; asm volatile("" : "+v"(r)); }
  else { r = s.rest[sp - PT_LDS_STACK]; asm volatile("" : "+v"(r)); }
  return r;
}
// The LDS part of the stacks of a workgroup of a kernel for scenes in global memory: a block per wave.  Only kernels that
// traverse such a scene refer to it (and so have it); they are launched with PT_GLOBAL_THREADS threads (traversalGeo).
constexpr uint32_t PT_GLOBAL_THREADS = 512;
constexpr uint32_t PT_STACK_REST = PT_LDS_STACK ? 64 - PT_LDS_STACK : 64;      // levels kept in scratch memory
#if PT_LDS_STACK
__shared__ uint32_t sWaveStacks[(PT_GLOBAL_THREADS / 64) * PT_LDS_STACK * 64];
__device__ __forceinline__ LdsStack globalSceneStack(uint32_t* rest) {
  return LdsStack{sWaveStacks + (threadIdx.x >> 6) * (PT_LDS_STACK * 64) + (threadIdx.x & 63u), rest};
}
#else
__device__ __forceinline__ uint32_t* globalSceneStack(uint32_t* rest) { return rest; }
#endif
// The stack of a traversal: scratch memory for a scene in LDS (mem: 64 words), else globalSceneStack (mem: PT_STACK_REST words).
template <bool LDSS> __device__ __forceinline__ auto traversalStack(uint32_t* mem) {
  if constexpr (LDSS) return mem; else return globalSceneStack(mem);
}

template <bool LDSS, int OCT, bool UNBOUNDED = false, class STK = uint32_t*>
__device__ __forceinline__ uint32_t stepDown(const DevScene& sc, const RayInv& ri, float tmax, uint32_t ref, STK stack, int& sp) {
  float tn0, tn1;
  bool h0, h1;
  uint32_t r0, r1;
  if constexpr (LDSS) {
    uint32_t b = ref * 4;
    float4 a0 = ldNode<LDSS>(sc, b), a1 = ldNode<LDSS>(sc, b + 1);
    float4 b0 = ldNode<LDSS>(sc, b + 2), b1 = ldNode<LDSS>(sc, b + 3);
    keepW<LDSS>(a1, b1);
    h0 = slab<OCT, UNBOUNDED>(ri, a0.x, a0.y, a0.z, a1.x, a1.y, a1.z, tmax, tn0);
    h1 = slab<OCT, UNBOUNDED>(ri, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, tmax, tn1);
    r0 = f2u(a0.w); r1 = f2u(b0.w);
  } else {
    // 32 B per sibling pair instead of 64: the vector-memory address unit, which bounds this path, is busy 16 cycles
    // per 16-byte-per-lane load whatever the addresses.  ri maps grid coordinates straight to ray parameters.
    uint4 a = sc.qnodes[ref * 2], b = sc.qnodes[ref * 2 + 1];
    h0 = slab<OCT, UNBOUNDED>(ri, (float)(a.x & 0xFFFFu), (float)(a.y & 0xFFFFu), (float)(a.z & 0xFFFFu),
                              (float)(a.x >> 16), (float)(a.y >> 16), (float)(a.z >> 16), tmax, tn0);
    h1 = slab<OCT, UNBOUNDED>(ri, (float)(b.x & 0xFFFFu), (float)(b.y & 0xFFFFu), (float)(b.z & 0xFFFFu),
                              (float)(b.x >> 16), (float)(b.y >> 16), (float)(b.z >> 16), tmax, tn1);
    r0 = a.w; r1 = b.w;
  }
  // any-hit rays in an LDS-resident scene take the children in storage order: no occluder is nearer than another to
  // a query that stops at the first, and the comparison and selects cost more than the ordering saves (-2 %); with
  // the scene in HBM every extra step is a memory round trip and near-first stays ahead (+2 %)
  bool sw = (UNBOUNDED && LDSS) ? (h1 && !h0) : (h1 && (!h0 || tn1 < tn0));
  uint32_t nearRef = sw ? r1 : r0, farRef = sw ? r0 : r1;
  if (h0 && h1) stkPush(stack, sp, farRef);  // (storing unconditionally above the top was measured: 20 % slower)
  if (h0 || h1) return nearRef;
  if (sp == 0) return PT_NONE;
  return stkPop(stack, sp);
}
__device__ __forceinline__ bool isInternal(uint32_t ref) { return (ref >> BVH_COUNT_SHIFT) == 0; }
__device__ __forceinline__ bool isLeaf(uint32_t ref) { return (ref >> BVH_COUNT_SHIFT) != 0 && ref != PT_NONE; }

// Closest front-face hit (ANY = false: Scene.trace, scene.d:691-740) or first hit of either face
// (ANY = true: Scene.hitsNothing, scene.d:808-835).  Returns true if something was accepted.  COUNT counts box and
// triangle tests.
//
// POSTPONE = false: a leaf is tested as soon as it is reached (the canonical order, and the one the counters use).
// POSTPONE = true: a lane keeps its first leaf back and goes on descending until no lane of the wave is still
// looking for its first leaf; then the wave tests leaves together.  The extra boxes tested with a not yet shortened
// ray do not change the result (the closest hit does not depend on the order in which leaves are visited).
// Measured: with the scene in LDS the bookkeeping costs more instructions than the better lane utilisation saves
// (7-10 % slower); with the scene in HBM, where a step is a memory round trip, it is 25-35 % faster.
// STOP (closest-hit rays, scene in LDS): the traversal ends as soon as a hit at t <= tStop is accepted -- for callers that
// only ask whether the closest hit lies beyond tStop (advanceTrav has the same for scenes in global memory).
template <bool ANY, bool LDSS, bool COUNT, int OCT, bool BOTH = false, bool STOP = false>
__device__ __forceinline__ bool traverseOct(const DevScene& sc, const RayInv& ri, V3 o, V3 d, float eps, Hit& h, uint32_t& cN, uint32_t& cT,
                                            float tStop = 0.0f) {
  constexpr bool POSTPONE = !LDSS && !COUNT;
  h.t = PT_FLT_MAX; h.u = 0.0f; h.v = 0.0f; h.tri = 0xFFFFFFFFu; h.front = true;
  bool found = false;
  uint32_t stackMem[LDSS ? 64 : PT_STACK_REST];
  const auto stack = traversalStack<LDSS>(stackMem);
  int sp = 0;
  uint32_t ref = sc.rootRef;
  if constexpr (!POSTPONE) {
    for (;;) {
#ifdef PT_STAMP
      // diagnostic build: cN / cT count the wave's clock ticks inside the descent loop and inside the leaf phase
      // (whatever lane reports them: the wave spends them as a whole)
      unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef PT_SLOTS
      // diagnostic build: cN = lane steps taken, cT = lane slots the wave spent on them (64 per iteration of the
      // descent loop, charged to the lowest lane in it)
      for (;;) {
        const bool in = isInternal(ref);
        if (!__any(in)) break;
        unsigned long long m = __ballot(1);
        if (COUNT) {
          if (!in) cN += 1;                              // slots a lane spends waiting at a leaf (steps taken: the normal counting build)
          if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0) cT += 64;
        }
        if (in) ref = stepDown<LDSS, OCT, ANY>(sc, ri, h.t, ref, stack, sp);
      }
#else
      while (isInternal(ref)) { if (COUNT && !PT_STAMP_ON) cN += 2; ref = stepDown<LDSS, OCT, ANY>(sc, ri, h.t, ref, stack, sp); }
#endif
#ifdef PT_STAMP
      unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (COUNT) cN += (uint32_t)(t1 - t0);
#endif
      if (ref == PT_NONE) break;
      uint32_t cnt = ref >> BVH_COUNT_SHIFT, first = ref & BVH_INDEX_MASK;
      for (uint32_t i = 0; i < cnt; i++) {
        if (COUNT && !PT_STAMP_ON) cT += 1;
        if (testTriangle<ANY, LDSS, BOTH>(sc, (first + i) * 3, o, d, eps, h, found)) return true;
      }
      if (STOP && h.t <= tStop) break;
#ifdef PT_STAMP
      if (COUNT) cT += (uint32_t)(__builtin_amdgcn_s_memtime() - t1);
#endif
      if (sp == 0) break;
      ref = stkPop(stack, sp);
    }
  } else {
    uint32_t leaf = PT_NONE;
    if (isLeaf(ref)) { leaf = ref; ref = PT_NONE; }                        // the root is a leaf
    for (;;) {
      while (isInternal(ref)) {
        ref = stepDown<LDSS, OCT, ANY>(sc, ri, h.t, ref, stack, sp);
        if (leaf == PT_NONE && isLeaf(ref)) { leaf = ref; ref = sp ? stkPop(stack, sp) : PT_NONE; }
        // on to the leaf tests when fewer than a quarter of the wave's lanes that are still traversing look for their
        // first leaf (waiting for the last of them leaves most lanes idle: advanceTrav, LEAF_WAIT)
        if ((uint32_t)__popcll(__ballot(leaf == PT_NONE && ref != PT_NONE)) * 4u < (uint32_t)__popcll(__ballot(1))) break;
      }
      while (leaf != PT_NONE) {
        uint32_t cnt = leaf >> BVH_COUNT_SHIFT, first = leaf & BVH_INDEX_MASK;
        for (uint32_t i = 0; i < cnt; i++)
          if (testTriangle<ANY, LDSS, BOTH>(sc, (first + i) * 3, o, d, eps, h, found)) return true;
        leaf = PT_NONE;
        if (isLeaf(ref)) { leaf = ref; ref = sp ? stkPop(stack, sp) : PT_NONE; }
      }
      if (ref == PT_NONE) break;
    }
  }
  return found;
}

// The traversal proper.  Closest-hit queries: when every active lane of the wave has its direction in the same
// octant (view rays of a pixel tile), the variant compiled for that octant runs (measured: 5 % on the view-ray
// launch; nothing for any-hit queries, which therefore keep one variant).  Otherwise, and for the counting and
// stand-alone variants, the general one.  Must be called with the lanes that have a ray active and no others.
template <bool ANY, bool LDSS, bool COUNT, bool BY_OCTANT = !COUNT && !ANY, bool BOTH = false, bool STOP = false>
__device__ __forceinline__ bool traverse(const DevScene& sc, V3 o, V3 d, float eps, Hit& h, uint32_t& cN, uint32_t& cT, float tStop = 0.0f) {
  static_assert(!STOP || (LDSS && !ANY && !COUNT), "the early end is in the loop for scenes in LDS");
  RayInv ri = makeRayInv(o, d, LDSS ? v3(0, 0, 0) : v3(sc.qBase[0], sc.qBase[1], sc.qBase[2]));
  if constexpr (!LDSS) {     // box planes come as grid coordinates q: plane = qBase + q * qScale
    ri.a0x = fmaf(sc.qBase[0], ri.ix, ri.a0x); ri.a1x = fmaf(sc.qBase[0], ri.ix, ri.a1x); ri.ix = sc.qScale[0] * ri.ix;
    ri.a0y = fmaf(sc.qBase[1], ri.iy, ri.a0y); ri.a1y = fmaf(sc.qBase[1], ri.iy, ri.a1y); ri.iy = sc.qScale[1] * ri.iy;
    ri.a0z = fmaf(sc.qBase[2], ri.iz, ri.a0z); ri.a1z = fmaf(sc.qBase[2], ri.iz, ri.a1z); ri.iz = sc.qScale[2] * ri.iz;
  }
  if constexpr (BY_OCTANT) {
    uint32_t oct = octantOf(ri), o0 = __builtin_amdgcn_readfirstlane(oct);
    if (__all(oct == o0)) {
      switch (o0) {
        case 0: return traverseOct<ANY, LDSS, COUNT, 0, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
        case 1: return traverseOct<ANY, LDSS, COUNT, 1, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
        case 2: return traverseOct<ANY, LDSS, COUNT, 2, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
        case 3: return traverseOct<ANY, LDSS, COUNT, 3, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
        case 4: return traverseOct<ANY, LDSS, COUNT, 4, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
        case 5: return traverseOct<ANY, LDSS, COUNT, 5, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
        case 6: return traverseOct<ANY, LDSS, COUNT, 6, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
        default: return traverseOct<ANY, LDSS, COUNT, 7, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
      }
    }
  }
  return traverseOct<ANY, LDSS, COUNT, -1, BOTH, STOP>(sc, ri, o, d, eps, h, cN, cT, tStop);
}

template <bool COUNT> __device__ __forceinline__ void flushCounts(unsigned long long* stats, int nodeSlot, uint32_t cN, uint32_t cT) {
  if constexpr (COUNT) {
    for (int off = 32; off > 0; off >>= 1) { cN += __shfl_down(cN, off); cT += __shfl_down(cT, off); }
    if ((threadIdx.x & 63) == 0) { statAdd(stats, nodeSlot, (unsigned long long)cN); statAdd(stats, nodeSlot + 1, (unsigned long long)cT); }
  }
}

// ------------------------------------------------------------------------------------------------ segmented queues
// Append from converged code (every lane of the wave executes the call): rank by ballot + mbcnt, one LDS atomic per wave.
__device__ __forceinline__ uint32_t waveAppend(bool pred, uint32_t* ldsCounter) {
  unsigned long long mask = __ballot(pred);
  uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  uint32_t base = 0;
  if ((threadIdx.x & 63) == 0 && mask) base = atomicAdd(ldsCounter, (uint32_t)__popcll(mask));
  base = __builtin_amdgcn_readfirstlane(base);
  return base + rank;
}

// Flat input range of block-task T and the segment it starts in.
struct Task { uint32_t begin, end, seg; };
__device__ __forceinline__ Task taskOf(const QMeta& q, uint32_t T) {
  uint32_t total = q.prefix[PT_SEGS], P = q.prefix[PT_SEGS + 1];
  Task t; t.begin = T * P; t.end = min(t.begin + P, total); t.seg = q.start[T];
  return t;
}
// Physical index of flat entry j through the lane's cursor (j only grows within a block-task).  The cursor keeps
// the bounds of its segment in registers: the prefix array is read again only when j leaves the segment, not once
// per entry (a read per entry puts one more memory round trip in front of every iteration of shade).
struct SegCursor { uint32_t seg, lo, hi; };
__device__ __forceinline__ SegCursor segCursor(const QMeta& q, uint32_t seg) {
  SegCursor c; c.seg = seg; c.lo = q.prefix[seg]; c.hi = q.prefix[seg + 1];
  return c;
}
__device__ __forceinline__ uint32_t physIndex(const QMeta& q, uint32_t j, SegCursor& c, uint32_t segCap) {
  while (j >= c.hi) { c.seg++; c.lo = c.hi; c.hi = q.prefix[c.seg + 1]; }
  return c.seg * segCap + (j - c.lo);
}
__device__ __forceinline__ uint32_t dense64(uint32_t total) { return ((total + PT_SEGS - 1) / PT_SEGS + 63u) & ~63u; }

// One workgroup per queue: exclusive prefix of the PT_SEGS counts, total, consumers' P and start segments.
__global__ void __launch_bounds__(1024) k_scan(const QMeta* __restrict__ metas, uint32_t* taskCounters) {
  __shared__ uint32_t sp[PT_SEGS + 1];
  __shared__ uint32_t wsum[16];
  QMeta q = metas[blockIdx.x];
  uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && tid < PT_TASK_WORDS) taskCounters[tid] = 0;
  uint32_t c = q.count[tid], x = c;
  for (int off = 1; off < 64; off <<= 1) { uint32_t y = __shfl_up(x, off); if (lane >= (uint32_t)off) x += y; }   // inclusive in wave
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; w++) base += wsum[w];
  uint32_t excl = base + x - c;
  sp[tid] = excl;
  if (tid == PT_SEGS - 1) sp[PT_SEGS] = excl + c;
  __syncthreads();
  uint32_t total = sp[PT_SEGS], P = dense64(total);
  q.prefix[tid] = excl;
  if (tid == 0) { q.prefix[PT_SEGS] = total; q.prefix[PT_SEGS + 1] = P; }
  // segment s with sp[s] <= target < sp[s+1] (skipping empty segments); past the end: last segment
  uint32_t target = tid * P, s = PT_SEGS - 1;
  if (target < total) {
    uint32_t lo = 0, hi = PT_SEGS;                       // invariant: sp[lo] <= target < sp[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (sp[mid] <= target) lo = mid; else hi = mid; }
    s = lo;
  }
  q.start[tid] = s;
}

// Physical index of flat entry j for the traversal kernels' lanes.  The lane keeps the bounds of the segment its
// last entry was in (consecutive chunks of a wave come from the same 2048 entries, i.e. nearly always the same
// segment) and searches again, from the hint `seg` (which may be off in either direction), only when j is outside.
__device__ __forceinline__ uint32_t physIndexHint(const QMeta& q, uint32_t j, uint32_t seg, uint32_t segCap, SegCursor& c) {
  if (j < c.lo || j >= c.hi) {
    uint32_t lo = q.prefix[seg];
    while (j < lo) { seg--; lo = q.prefix[seg]; }
    uint32_t hi = q.prefix[seg + 1];
    while (j >= hi) { seg++; lo = hi; hi = q.prefix[seg + 1]; }
    c.seg = seg; c.lo = lo; c.hi = hi;
  }
  return c.seg * segCap + (j - c.lo);
}

// Dynamic distribution of a flat queue over the waves of the persistent traversal kernels.  The launch's global
// counter hands out PT_SUPER consecutive entries at a time to a workgroup (total / PT_SUPER atomics per launch);
// the workgroup's waves take 64 at a time from the LDS pool: state = (end << 32) | next, advanced by a 64-bit
// compare-and-swap.  Only the wave that holds `lock` refills, and only an empty pool is ever refilled, so a
// successful compare-and-swap always advanced the range it read.  Waves never wait for each other between chunks.
struct WorkPool { unsigned long long state; uint32_t lock, done, seg, out, partsDone; };
#ifndef PT_XCD_PARTS
#define PT_XCD_PARTS PT_TASK_PARTS      // diagnostic builds: 1 = one counter for the whole queue
#endif
// First entry of part p of a queue of `total` entries (whole waves).
__device__ __forceinline__ uint32_t partBegin(uint32_t total, uint32_t p) {
  return p >= PT_XCD_PARTS ? total : (uint32_t)(((uint64_t)total * p / PT_XCD_PARTS) & ~63ull);
}
struct Chunk { uint32_t begin, end, seg; };

__device__ __forceinline__ void initPool(WorkPool* p) {
  if (threadIdx.x == 0) { p->state = 0; p->lock = 0; p->done = 0; p->seg = 0; p->out = 0; p->partsDone = 0; }
}

// Next chunk for the calling wave (wave-uniform result); false when the queue is drained.  outLimit != 0: the
// workgroup appends at most one entry per input entry to a region of outLimit entries counted in p->out, and stops
// taking input when the entries already taken plus a new refill could overflow the region.
__device__ __forceinline__ bool nextChunk(WorkPool* p, uint32_t* counter, const QMeta& q, uint32_t total, uint32_t outLimit, Chunk& c,
                                          uint32_t inFlight = 0) {
  uint32_t b = 0xFFFFFFFFu, e = 0, s = 0;
  if ((threadIdx.x & 63) == 0) {
    volatile WorkPool* vp = p;
    for (;;) {
      unsigned long long st = vp->state;
      uint32_t next = (uint32_t)st, end = (uint32_t)(st >> 32);
      if (next < end) {
        if (atomicCAS(&p->state, st, st + 64) == st) { b = next; e = min(next + 64u, end); s = vp->seg; break; }
        continue;
      }
      if (vp->done) break;
      if (atomicCAS(&p->lock, 0u, 1u) == 0u) {
        st = vp->state;
        if ((uint32_t)st >= (uint32_t)(st >> 32) && !vp->done) {
          // + entries taken from the pool but not yet decided: a chunk per wave, or what the caller says
          bool full = outLimit && vp->out + (inFlight ? inFlight : blockDim.x) + PT_SUPER > outLimit;
          // a short queue is handed out in smaller pieces, so that every workgroup of the launch gets some of it: with
          // PT_SUPER entries a piece, 76 800 rays (a 320 x 240 pass) would keep 38 workgroups busy and leave the rest idle
          const uint32_t piece = min(PT_SUPER, max(blockDim.x, ((total + gridDim.x - 1) / gridDim.x + 63u) & ~63u));
          // the next piece of the part this workgroup is working through (the part of its index first, then the others
          // in turn)
          uint32_t g = 0xFFFFFFFFu, gEnd = 0;
          while (!full && vp->partsDone < PT_XCD_PARTS) {
            const uint32_t part = (blockIdx.x + vp->partsDone) % PT_XCD_PARTS;
            const uint32_t lo = partBegin(total, part), hi = partBegin(total, part + 1);
            const uint32_t off = lo < hi ? atomicAdd(counter + part, piece) : 0xFFFFFFFFu;
            if (off < hi - lo) { g = lo + off; gEnd = min(g + piece, hi); break; }
            vp->partsDone = vp->partsDone + 1;
          }
          if (g == 0xFFFFFFFFu) vp->done = 1;
          else {
            vp->seg = q.start[g / q.prefix[PT_SEGS + 1]];
            __threadfence_block();
            atomicExch(&p->state, ((unsigned long long)gEnd << 32) | g);
          }
        }
        __threadfence_block();
        atomicExch(&p->lock, 0u);
      } else {
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  c.begin = __builtin_amdgcn_readfirstlane(b);
  c.end = __builtin_amdgcn_readfirstlane(e);
  c.seg = __builtin_amdgcn_readfirstlane(s);
  return c.begin != 0xFFFFFFFFu;
}

// (pixel, sample index, view ray) of path slot i: getViewRay, rendering.d:278-285
struct PrimaryRay { uint32_t gp, sampleIdx; V3 d; Rng g; };
// frame pixel and sample number (within the batch) of path slot i
__device__ __forceinline__ uint32_t slotPixel(const DevCfg& cfg, const PixMap& pm, uint32_t i, uint32_t& s) {
  s = fdiv(pm.divNpix, i);
  return mapPixel(pm, cfg.W, i - s * pm.npix);
}
__device__ __forceinline__ bool maskBit(const uint32_t* __restrict__ m, uint32_t p) { return (m[p >> 5] >> (p & 31u)) & 1u; }

// accW: the pixel's sample count so far (Pixel.n), which numbers the sample
__device__ __forceinline__ PrimaryRay primaryRayFrom(const DevCfg& cfg, uint32_t gp, uint32_t s, float accW, bool jitter) {
  PrimaryRay r;
  r.gp = gp;
  uint32_t row = fdiv(cfg.divW, r.gp), col = r.gp - row * cfg.W;
  float x, y;
  if (jitter) {
    r.sampleIdx = cfg.sampleBase + ((uint32_t)accW + s) * cfg.sampleStride;
    r.g.seed(cfg.seed, r.gp, r.sampleIdx, 0);
    viewRayXY(col, row, cfg.W, cfg.H, &r.g, x, y);
  } else {
    r.sampleIdx = 0; r.g.s0 = 1; r.g.s1 = 0;
    viewRayXY(col, row, cfg.W, cfg.H, nullptr, x, y);
  }
  r.d = screenRayDir(cfg.cam, x, y);
  return r;
}
__device__ __forceinline__ PrimaryRay primaryRayAt(const DevCfg& cfg, const float4* __restrict__ acc, uint32_t gp, uint32_t s,
                                                   bool jitter) {
  return primaryRayFrom(cfg, gp, s, jitter ? acc[gp].w : 0.0f, jitter);
}
__device__ __forceinline__ PrimaryRay primaryRay(const DevCfg& cfg, const PixMap& pm, const float4* __restrict__ acc, uint32_t i,
                                                 bool jitter) {
  uint32_t s, gp = slotPixel(cfg, pm, i, s);
  return primaryRayAt(cfg, acc, gp, s, jitter);
}

// ------------------------------------------------------------------------------------------------ raygen
// Block-task T = blockIdx.x covers slots [T*P, (T+1)*P).  A view ray that misses the scene's box cannot hit
// anything: its sample is finished here (sampleSky).  The others are queued as (direction, slot), 16 B: the origin
// is the camera position and everything else is a function of the slot.  The box is a kernel argument, i.e. scalar
// registers: the test costs no memory traffic.
// With a coverage mask (one bit per pixel: some triangle's screen rectangle, grown by the sample jitter, overlaps
// the pixel) the bit decides instead of the box: for an object in front of a background it culls nearly every ray
// that would traverse the tree and come back empty.  Pixels of the constant-sky mask are skipped altogether: each
// of their samples is the sky constant, which accumulate adds without reading anything.
__global__ void __launch_bounds__(PT_RAYGEN_THREADS) k_raygen(SceneBox box, ViewMasks masks, DevCfg cfg, PixMap pm,
                                                const float4* __restrict__ acc, uint32_t nslots, Wavefront wf) {
  __shared__ uint32_t sCnt;
  if (threadIdx.x == 0) sCnt = 0;
  __syncthreads();
  uint32_t T = blockIdx.x, P = dense64(nslots), begin = T * P, end = min(begin + P, nslots);
  // Wavefront::regionPix != 0: block-task T covers the pixels [T * regionPix, (T + 1) * regionPix) of the batch at all k
  // samples, sample by sample -- the queue then lists the view rays region by region, and the workgroups of an XCD,
  // which take a contiguous part of it, trace all samples of a patch of pixels one after the other
  uint32_t lp0 = 0, Rt = 1;
  if (wf.regionPix) {
    lp0 = min(T * wf.regionPix, pm.npix); Rt = min(wf.regionPix, pm.npix - lp0);
    begin = 0; end = Rt * pm.k;
    if (Rt == 0) Rt = 1;
  }
  uint32_t segCap = wf.capSlots / PT_SEGS, culled = 0, nConst = 0;
  // PT_RAYGEN_UNROLL slots per lane and pass, phase by phase: the loads of a phase (mask words, then sample counts)
  // are in flight together, so a pass waits through two memory round trips for four slots, not per slot
  constexpr int U = PT_RAYGEN_UNROLL;
  for (uint32_t base = begin; base < end; base += blockDim.x * U) {
    uint32_t i[U], s[U], gp[U];
    bool valid[U], constant[U], enter[U];
    float accW[U];
    V3 d[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      i[u] = base + u * blockDim.x + threadIdx.x;
      valid[u] = i[u] < end;
      if (wf.regionPix) {
        const uint32_t r = valid[u] ? i[u] : 0u, smp = r / Rt;
        i[u] = smp * pm.npix + lp0 + (r - smp * Rt);
      }
      gp[u] = slotPixel(cfg, pm, valid[u] ? i[u] : begin, s[u]);
      constant[u] = valid[u] && masks.constantSky && maskBit(masks.constantSky, gp[u]);
      enter[u] = false;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      bool live = valid[u] && !constant[u];
      accW[u] = live ? acc[gp[u]].w : 0.0f;
      if (live && masks.covered) enter[u] = maskBit(masks.covered, gp[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      d[u] = v3(0, 0, 0);
      if (constant[u]) { culled++; nConst++; }
      else if (valid[u]) {
        d[u] = primaryRayFrom(cfg, gp[u], s[u], accW[u], true).d;
        if (!masks.covered) {
          RayInv ri = makeRayInv(cfg.cam.pos, d[u]);
          float tn;
          enter[u] = slab(ri, box.mn[0], box.mn[1], box.mn[2], box.mx[0], box.mx[1], box.mx[2], PT_FLT_MAX, tn);
        }
        if (!enter[u]) { wf.Lacc[i[u]] = f4(sampleSky(d[u]), 0.0f); culled++; }        // rendering.d:480
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t idx = waveAppend(enter[u], &sCnt);
      if (enter[u]) wf.slotQ[T * segCap + idx] = f4(d[u], u2f(i[u]));
    }
  }
  for (int off = 32; off > 0; off >>= 1) { culled += __shfl_down(culled, off); nConst += __shfl_down(nConst, off); }
  if ((threadIdx.x & 63) == 0 && culled) statAdd(wf.stats, STAT_RAYS_CULLED, (unsigned long long)culled);
  if ((threadIdx.x & 63) == 0 && nConst) statAdd(wf.stats, STAT_CONSTANT_SKY, (unsigned long long)nConst);
  __syncthreads();
  if (threadIdx.x == 0) wf.rayOut.count[T] = sCnt;
}

// ------------------------------------------------------------------------------------------------ extend
// Scene.trace for every entry of the input queue.  PRIMARY: entries are (view direction, path slot) and a
// miss finishes the sample (sampleSky, rendering.d:480).  Otherwise entries are stored indirect rays and a miss ends
// the chain (rendering.d:393).  Hits -- and only hits -- are appended to the hit queue, with the hit position
// (Ray.get(rayPos), rendering.d:390,472) so that shading never needs the ray again.
template <bool PRIMARY, bool LDSS, bool COUNT>
__global__ void PT_TRAVERSAL_BOUNDS k_extend(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc, Wavefront wf) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  stageScene<LDSS>(sc);
  __syncthreads();
  uint32_t cN = 0, cT = 0;
  const QMeta q = wf.rayIn;
  const uint32_t total = q.prefix[PT_SEGS];
  const uint32_t segCapIn = (PRIMARY ? wf.capSlots : wf.capQueue) / PT_SEGS, segCapOut = wf.capHit / PT_SEGS;
  // this workgroup appends to the adjacent output segments [segFirst, segLast): one contiguous region
  const uint32_t segFirst = (uint32_t)(((uint64_t)blockIdx.x * PT_SEGS) / gridDim.x);
  const uint32_t segLast = (uint32_t)(((uint64_t)(blockIdx.x + 1) * PT_SEGS) / gridDim.x);
  const uint32_t outBase = segFirst * segCapOut, outLimit = (segLast - segFirst) * segCapOut;
  if (!PRIMARY && blockIdx.x == 0 && threadIdx.x == 0) statAdd(wf.stats, STAT_RAYS_INDIRECT, (unsigned long long)total);
  SegCursor cur; cur.seg = 0; cur.lo = 1; cur.hi = 0;      // empty: the first entry searches
  Chunk ck;
  while (nextChunk(&pool, wf.taskCounters, q, total, outLimit, ck)) {
    uint32_t j = ck.begin + (threadIdx.x & 63u);
    bool isHit = false;
    V3 hitPos = v3(0, 0, 0); Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0; uint32_t slot = 0;
    float carry = 0.0f;                       // fourth word of an indirect ray's direction record, handed on to shade
#ifdef PT_PROBE_HALF
    if (j < ck.end && (threadIdx.x & 63u) < PT_PROBE_HALF) {
#else
    if (j < ck.end) {
#endif
      uint32_t e = physIndexHint(q, j, ck.seg, segCapIn, cur);
      V3 o, d;
      if (PRIMARY) {
        float4 sd = wf.slotQ[e];
        o = cfg.cam.pos; d = xyz(sd); slot = f2u(sd.w);
      } else {
        float4 ro = wf.rayO[e], rd = wf.rayD[e];
        o = xyz(ro); d = xyz(rd); slot = f2u(ro.w); carry = rd.w;
      }
      isHit = traverse<false, LDSS, COUNT>(sc, o, d, cfg.eps, h, cN, cT);
      if (isHit) hitPos = o + d * h.t;
      else if (PRIMARY) wf.Lacc[slot] = f4(sampleSky(d), 0.0f);
    }
    uint32_t idx = waveAppend(isHit, &pool.out);
    if (isHit) {
      uint32_t e = outBase + idx;
      wf.hitA[e] = f4(hitPos, u2f(h.tri));
      wf.hitB[e] = make_float4(h.u, h.v, u2f(slot), PRIMARY ? h.t : carry);
    }
  }
  flushCounts<COUNT>(wf.stats, STAT_NODES_CLOSEST, cN, cT);
  __syncthreads();
  uint32_t n = pool.out;
  for (uint32_t s = segFirst + threadIdx.x; s < segLast; s += blockDim.x) {
    uint32_t before = (s - segFirst) * segCapOut;
    wf.hitQ.count[s] = n > before ? min(n - before, segCapOut) : 0u;
  }
}

// Extend for scenes traversed from global memory, with lanes that take a new ray when theirs is decided.  There a step
// is a memory round trip, rays of one wave differ several-fold in the number of steps they take, and a wave that runs
// its 64 rays to the end spends most of its load instructions on a few lanes (18 % of the lanes active per vector
// instruction on the 10 M-triangle soup).  Here a wave keeps its traversal state per lane across rays: whenever fewer
// than PT_REFILL_BELOW lanes are still traversing it writes out the decided rays and hands the free lanes the next
// entries of its chunk (a new chunk from the pool when that is used up), until the queue is drained.  The traversal is
// the postponed-leaf loop of traverseOct, resumable.  A ray's result does not depend on its lane or its companions.
#ifndef PT_REFILL_BELOW
#define PT_REFILL_BELOW 48
#endif
// (advanceTrav: LEAF_WAIT) measured on the 10 M-triangle soup, 1 / 4 / 8 / 16 / 24 / 32 / 44 lanes: view rays 18.9 / 16.9 /
// 16.1 / 15.6 / 15.8 / 16.7 / 19.6 ms, sun and sky rays 11.2 / 10.5 / 10.4 / 10.5 / 10.7 / 10.9 / 12.3, light samples
// 10.1 / 10.1 / 10.3 / 10.2 / 10.4 / 10.3 / 10.4
#ifndef PT_LEAF_WAIT_EXTEND
#define PT_LEAF_WAIT_EXTEND 16
#endif
#ifndef PT_LEAF_WAIT_CONNECT
#define PT_LEAF_WAIT_CONNECT 1
#endif
#ifndef PT_LEAF_WAIT_ANY
#define PT_LEAF_WAIT_ANY 12
#endif

// A lane's ray and where its traversal of the (quantised, global-memory) tree stands.
struct LaneTrav {
  RayInv ri; V3 o, d; Hit h; bool found; uint32_t ref, leaf; int sp;
  __device__ __forceinline__ bool busy() const { return ref != PT_NONE || leaf != PT_NONE; }
};
__device__ __forceinline__ void idleTrav(LaneTrav& t) {
  t.ri = makeRayInv(v3(0, 0, 0), v3(1, 1, 1)); t.o = v3(0, 0, 0); t.d = v3(0, 0, 0);
  t.h.t = PT_FLT_MAX; t.h.u = 0; t.h.v = 0; t.h.tri = 0xFFFFFFFFu; t.h.front = true; t.found = false;
  t.ref = PT_NONE; t.leaf = PT_NONE; t.sp = 0;
}
__device__ __forceinline__ void startTrav(const DevScene& sc, LaneTrav& t, V3 o, V3 d) {
  t.o = o; t.d = d;
  t.ri = makeRayInv(o, d, v3(sc.qBase[0], sc.qBase[1], sc.qBase[2]));
  // box planes come as grid coordinates q: plane = qBase + q * qScale
  t.ri.a0x = fmaf(sc.qBase[0], t.ri.ix, t.ri.a0x); t.ri.a1x = fmaf(sc.qBase[0], t.ri.ix, t.ri.a1x); t.ri.ix = sc.qScale[0] * t.ri.ix;
  t.ri.a0y = fmaf(sc.qBase[1], t.ri.iy, t.ri.a0y); t.ri.a1y = fmaf(sc.qBase[1], t.ri.iy, t.ri.a1y); t.ri.iy = sc.qScale[1] * t.ri.iy;
  t.ri.a0z = fmaf(sc.qBase[2], t.ri.iz, t.ri.a0z); t.ri.a1z = fmaf(sc.qBase[2], t.ri.iz, t.ri.a1z); t.ri.iz = sc.qScale[2] * t.ri.iz;
  t.h.t = PT_FLT_MAX; t.h.u = 0.0f; t.h.v = 0.0f; t.h.tri = 0xFFFFFFFFu; t.found = false;
  t.sp = 0; t.ref = sc.rootRef; t.leaf = PT_NONE;
  if (isLeaf(t.ref)) { t.leaf = t.ref; t.ref = PT_NONE; }
}
// Advances every lane's traversal until fewer than minBusy lanes are still at it (the postponed-leaf loop of
// traverseOct, resumable; ANY: a lane stops at the first triangle it accepts).  Converged code.
// tStop (closest-hit rays): a lane whose ray has an accepted hit at t <= tStop is done -- the caller only asks whether
// the closest hit lies beyond tStop (a light sample: anything nearer than the light point hides it, whichever of the
// occluders is the nearest).
// LEAF_WAIT: the descent goes on while at least that many lanes are still looking for their first leaf (1: until every
// lane has one); the lanes that are left over sit out the leaf tests and descend on afterwards.  Likewise the leaf
// tests go on while at least that many lanes have another leaf at hand (-3 % on top).
template <bool ANY, uint32_t LEAF_WAIT, class STK>
__device__ __forceinline__ void advanceTrav(const DevScene& sc, LaneTrav& t, STK stack, float eps, uint32_t minBusy, float tStop = -PT_FLT_MAX) {
  for (;;) {
    if ((uint32_t)__popcll(__ballot(t.busy())) < minBusy) break;
    while (isInternal(t.ref)) {
      t.ref = stepDown<false, -1, ANY, STK>(sc, t.ri, t.h.t, t.ref, stack, t.sp);
      if (t.leaf == PT_NONE && isLeaf(t.ref)) { t.leaf = t.ref; t.ref = t.sp ? stkPop(stack, t.sp) : PT_NONE; }
      if ((uint32_t)__popcll(__ballot(t.leaf == PT_NONE && t.ref != PT_NONE)) < LEAF_WAIT) break;
    }
    while (t.leaf != PT_NONE) {
      uint32_t cnt = t.leaf >> BVH_COUNT_SHIFT, first = t.leaf & BVH_INDEX_MASK;
      bool stop = false;
      for (uint32_t i = 0; i < cnt && !stop; i++) stop = testTriangle<ANY, false>(sc, (first + i) * 3, t.o, t.d, eps, t.h, t.found);
      t.leaf = PT_NONE;
      if (ANY && stop) { t.found = true; t.ref = PT_NONE; }
      else if (!ANY && t.h.t <= tStop) t.ref = PT_NONE;
      else if (isLeaf(t.ref)) { t.leaf = t.ref; t.ref = t.sp ? stkPop(stack, t.sp) : PT_NONE; }
      if ((uint32_t)__popcll(__ballot(t.leaf != PT_NONE)) < LEAF_WAIT) break;      // the same the other way round
    }
  }
}

// The wave's share of a queue that is not handed to a lane yet, for the kernels whose lanes take a new entry when
// theirs is done.  take(): flat index of the entry the lane gets (need: the lane wants one), or PT_NONE; false when
// the queue is drained.  Converged code.
struct WaveSupply {
  uint32_t cb = 0, ce = 0, seg = 0; bool open = true;
  __device__ __forceinline__ bool take(WorkPool* pool, uint32_t* counter, const QMeta& q, uint32_t total, uint32_t outLimit, bool need, uint32_t& j) {
    j = PT_NONE;
    if (cb == ce) {
      Chunk ck;
      // entries of this workgroup that are taken but not written out: at most a wave of rays and a chunk per wave
      if (!nextChunk(pool, counter, q, total, outLimit, ck, 2u * blockDim.x)) { open = false; return false; }
      cb = ck.begin; ce = ck.end; seg = ck.seg;
    }
    const unsigned long long m = __ballot(need);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t n = min((uint32_t)__popcll(m), ce - cb);
    if (need && rank < n) j = cb + rank;
    cb += n;
    return true;
  }
};

// Where its hits go: the hit of input entry j goes to output segment j / P, P being the input queue's entries per
// block-task -- the hit queue then lists the hits in the order of the rays (to 1/PT_SEGS of the queue), which for view
// rays is the order of the pixel tiles.  What shade makes of it keeps that order, so the light-sample and sun / sky rays
// that a wave, a workgroup and -- with the queue parts of nextChunk -- an XCD have in flight start close to each other
// and share the nodes and triangles they fetch.  (With one output region per workgroup, as the kernels for scenes in
// LDS have it, a region is a sample of the whole launch, and the rays in flight in the next stage come from all over
// the frame.)  A segment takes at most P hits, fewer than it holds; lanes of a wave that append to the same segment
// share one atomic add on its count (a wave's rays come from one or two chunks, so from one or two segments).
// View rays whose hits are put in order before they are shaded (Wavefront::regionPix != 0, k_hit_order): segment =
// the pixel's place in the batch / regionPix, so that a segment holds the hits of ALL samples of its 1/PT_SEGS of the
// batch's pixels -- the denser the hits of a segment lie, the more the rays that start from them have in common.
// The counts must be zero at launch (launchExtend clears them).
#ifndef PT_EXTEND_BY_SEGMENT
#define PT_EXTEND_BY_SEGMENT 1
#endif
__device__ __forceinline__ uint32_t appendToSegment(bool isHit, uint32_t seg, uint32_t* counts) {
  uint32_t idx = 0;
  bool pending = isHit;
  const uint32_t lane = threadIdx.x & 63u;
  for (;;) {
    const unsigned long long todo = __ballot(pending);
    if (!todo) break;
    const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)seg, (int)leader);
    const bool mine = pending && seg == s0;
    const unsigned long long m = __ballot(mine);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&counts[s0], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
    if (mine) { idx = base + rank; pending = false; }
  }
  return idx;
}

template <bool PRIMARY>
__global__ void PT_TRAVERSAL_BOUNDS k_extend_refill(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc, Wavefront wf) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  __syncthreads();
  const QMeta q = wf.rayIn;
  const uint32_t total = q.prefix[PT_SEGS], perTask = max(q.prefix[PT_SEGS + 1], 1u);
  const uint32_t segCapIn = (PRIMARY ? wf.capSlots : wf.capQueue) / PT_SEGS, segCapOut = wf.capHit / PT_SEGS;
#if PT_EXTEND_BY_SEGMENT
  const uint32_t outLimit = 0;
#else
  const uint32_t segFirst = (uint32_t)(((uint64_t)blockIdx.x * PT_SEGS) / gridDim.x);
  const uint32_t segLast = (uint32_t)(((uint64_t)(blockIdx.x + 1) * PT_SEGS) / gridDim.x);
  const uint32_t outBase = segFirst * segCapOut, outLimit = (segLast - segFirst) * segCapOut;
#endif
  if (!PRIMARY && blockIdx.x == 0 && threadIdx.x == 0) statAdd(wf.stats, STAT_RAYS_INDIRECT, (unsigned long long)total);
  SegCursor cur; cur.seg = 0; cur.lo = 1; cur.hi = 0;
  bool hasRay = false; uint32_t slot = 0, outSeg = 0; float carry = 0.0f;
  LaneTrav t; idleTrav(t);
  uint32_t stackRest[PT_STACK_REST];
  const auto stack = globalSceneStack(stackRest);
  WaveSupply sup;
  for (;;) {
    bool need = !hasRay;
    while (sup.open && __any(need)) {
      uint32_t j;
      if (!sup.take(&pool, wf.taskCounters, q, total, outLimit, need, j)) break;
      if (j != PT_NONE) {
        uint32_t e = physIndexHint(q, j, sup.seg, segCapIn, cur);
        if (PRIMARY) {
          float4 sd = wf.slotQ[e];
          slot = f2u(sd.w); startTrav(sc, t, cfg.cam.pos, xyz(sd));
        } else {
          float4 ro = wf.rayO[e], rd = wf.rayD[e];
          slot = f2u(ro.w); carry = rd.w; startTrav(sc, t, xyz(ro), xyz(rd));
        }
        hasRay = true; need = false;
        outSeg = min(j / perTask, PT_SEGS - 1u);
        if (PRIMARY && wf.regionPix) {          // by the pixel's place in the batch, whatever the sample: see above
          const uint32_t smp = fdiv(pm.divNpix, slot), lp = slot - smp * pm.npix;
          outSeg = fdiv(wf.divRegion, lp);
        }
      }
    }
    if (!__any(hasRay)) break;
    advanceTrav<false, PT_LEAF_WAIT_EXTEND>(sc, t, stack, cfg.eps, sup.open ? (uint32_t)PT_REFILL_BELOW : 1u);
    const bool fin = hasRay && !t.busy();
    const bool isHit = fin && t.found;
#if PT_EXTEND_BY_SEGMENT
    const uint32_t seg = outSeg;
    const uint32_t idx = appendToSegment(isHit, seg, wf.hitQ.count);
    const uint32_t e = seg * segCapOut + idx;
#else
    const uint32_t idx = waveAppend(isHit, &pool.out);
    const uint32_t e = outBase + idx;
#endif
    if (isHit) {
      wf.hitA[e] = f4(t.o + t.d * t.h.t, u2f(t.h.tri));
      wf.hitB[e] = make_float4(t.h.u, t.h.v, u2f(slot), PRIMARY ? t.h.t : carry);
    } else if (fin && PRIMARY) {
      wf.Lacc[slot] = f4(sampleSky(t.d), 0.0f);
    }
    if (fin) hasRay = false;
  }
#if !PT_EXTEND_BY_SEGMENT
  __syncthreads();
  uint32_t n = pool.out;
  for (uint32_t s = segFirst + threadIdx.x; s < segLast; s += blockDim.x) {
    uint32_t before = (s - segFirst) * segCapOut;
    wf.hitQ.count[s] = n > before ? min(n - before, segCapOut) : 0u;
  }
#endif
}

// (k_connect_refill and k_occluded_refill below are the same scheme for the shadow rays.  A first attempt at them,
// made while the queues were still in launch-wide sample order, gained nothing; with the hit queue in ray order and
// the queue handed out by XCD they take 16 % and 22 % off their launches on the 10 M-triangle soup.)

// Raygen and the extend of the view rays in one traversal launch: a wave takes runs of 64 path slots (a pixel tile at
// one sample number), decides each slot as k_raygen does (constant sky: nothing; not covered: sampleSky; covered: the
// view ray) and traces the covered ones right away -- no view-ray queue (16 B written and read per traced ray), no
// raygen launch, no scan.  Runs whose pixels are all culled cost a mask word per lane.  Slots are handed out like queue
// entries (a global counter gives a workgroup PT_RUN_PIECE slots at a time, its waves take 64 at a time), hits are
// appended like extend's.  The counter must be zero at launch: the scan of the hit queue, which follows this kernel
// in every batch, clears it for the next batch.
constexpr uint32_t PT_RUN_PIECE = PT_SUPER;      // PT_HIT_SLACK is sized for pieces of PT_SUPER (kernels.h)
__device__ __forceinline__ bool nextSlots(WorkPool* p, uint32_t* counter, uint32_t total, uint32_t outLimit, uint32_t& begin) {
  uint32_t b = 0xFFFFFFFFu;
  if ((threadIdx.x & 63) == 0) {
    volatile WorkPool* vp = p;
    for (;;) {
      unsigned long long st = vp->state;
      uint32_t next = (uint32_t)st, end = (uint32_t)(st >> 32);
      if (next < end) {
        if (atomicCAS(&p->state, st, st + 64) == st) { b = next; break; }
        continue;
      }
      if (vp->done) break;
      if (atomicCAS(&p->lock, 0u, 1u) == 0u) {
        st = vp->state;
        if ((uint32_t)st >= (uint32_t)(st >> 32) && !vp->done) {
          // every slot can become a hit: the workgroup stops taking slots when a piece more could overflow its region
          bool full = vp->out + blockDim.x + PT_RUN_PIECE > outLimit;
          uint32_t g = full ? 0xFFFFFFFFu : atomicAdd(counter, PT_RUN_PIECE);
          if (g >= total) vp->done = 1;
          else { __threadfence_block(); atomicExch(&p->state, ((unsigned long long)min(g + PT_RUN_PIECE, total) << 32) | g); }
        }
        __threadfence_block();
        atomicExch(&p->lock, 0u);
      } else {
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  begin = __builtin_amdgcn_readfirstlane(b);
  return begin != 0xFFFFFFFFu;
}

template <bool LDSS>
__global__ void PT_TRAVERSAL_BOUNDS k_raygen_extend(SceneBox box, ViewMasks masks, DevScene sc, DevCfg cfg, PixMap pm,
                                                    const float4* __restrict__ acc, uint32_t nslots, Wavefront wf) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  stageScene<LDSS>(sc);
  __syncthreads();
  const uint32_t segCapOut = wf.capHit / PT_SEGS;
  const uint32_t segFirst = (uint32_t)(((uint64_t)blockIdx.x * PT_SEGS) / gridDim.x);
  const uint32_t segLast = (uint32_t)(((uint64_t)(blockIdx.x + 1) * PT_SEGS) / gridDim.x);
  const uint32_t outBase = segFirst * segCapOut, outLimit = (segLast - segFirst) * segCapOut;
  uint32_t culled = 0, nConst = 0, cN = 0, cT = 0, begin;
  while (nextSlots(&pool, wf.taskCounters + (PT_TASK_COUNTERS - 1) * PT_TASK_PARTS, nslots, outLimit, begin)) {
    if (masks.activeCols && pm.columns) {            // a run of a column of constant sky: counted, nothing else (see k_sample_traced)
      const uint32_t run = begin >> 6, col = run - fdiv(pm.divColumns, run) * pm.columns;
      if (!((masks.activeCols[col >> 5] >> (col & 31u)) & 1u)) { culled++; nConst++; continue; }
    }
    const uint32_t i = begin + (threadIdx.x & 63u);
    const bool valid = i < nslots;
    uint32_t s, gp = slotPixel(cfg, pm, valid ? i : 0u, s);
    const bool constant = valid && masks.constantSky && maskBit(masks.constantSky, gp);
    const bool live = valid && !constant;
    if (constant) { culled++; nConst++; }
    bool isHit = false; V3 d = v3(0, 0, 0); Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0;
    if (live) {
      d = primaryRayFrom(cfg, gp, s, acc[gp].w, true).d;
      bool enter;
      if (masks.covered) enter = maskBit(masks.covered, gp);
      else {
        RayInv ri = makeRayInv(cfg.cam.pos, d);
        float tn;
        enter = slab(ri, box.mn[0], box.mn[1], box.mn[2], box.mx[0], box.mx[1], box.mx[2], PT_FLT_MAX, tn);
      }
      if (enter) isHit = traverse<false, LDSS, false>(sc, cfg.cam.pos, d, cfg.eps, h, cN, cT);
      else culled++;
      if (!isHit) wf.Lacc[i] = f4(sampleSky(d), 0.0f);                                   // rendering.d:480
    }
    uint32_t idx = waveAppend(isHit, &pool.out);
    if (isHit) {
      uint32_t e = outBase + idx;
      wf.hitA[e] = f4(cfg.cam.pos + d * h.t, u2f(h.tri));
      wf.hitB[e] = make_float4(h.u, h.v, u2f(i), h.t);
    }
  }
  for (int off = 32; off > 0; off >>= 1) { culled += __shfl_down(culled, off); nConst += __shfl_down(nConst, off); }
  if ((threadIdx.x & 63) == 0 && culled) statAdd(wf.stats, STAT_RAYS_CULLED, (unsigned long long)culled);
  if ((threadIdx.x & 63) == 0 && nConst) statAdd(wf.stats, STAT_CONSTANT_SKY, (unsigned long long)nConst);
  __syncthreads();
  uint32_t n = pool.out;
  for (uint32_t sg = segFirst + threadIdx.x; sg < segLast; sg += blockDim.x) {
    uint32_t before = (sg - segFirst) * segCapOut;
    wf.hitQ.count[sg] = n > before ? min(n - before, segCapOut) : 0u;
  }
}

// View rays generated in registers, no cull, no queue: the parity probe (getViewRay without jitter + Scene.trace).
template <bool LDSS, bool COUNT>
__global__ void PT_TRAVERSAL_BOUNDS k_extend_primary(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc,
                                                         uint32_t nslots, int jitter, float4* __restrict__ hit,
                                                         unsigned long long* stats) {
  stageScene<LDSS>(sc);
  uint32_t cN = 0, cT = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
    PrimaryRay pr = primaryRay(cfg, pm, acc, i, jitter != 0);
    Hit h;
    traverse<false, LDSS, COUNT, false>(sc, cfg.cam.pos, pr.d, cfg.eps, h, cN, cT);
    hit[i] = make_float4(h.t, h.u, h.v, u2f(h.tri));
  }
  flushCounts<COUNT>(stats, STAT_NODES_CLOSEST, cN, cT);
}

// The tables shading reads per hit.  LDST: the workgroup's copy in LDS (kernels.h: shadeTableBytes) -- an iteration
// of shade is a chain of dependent reads (hit -> normals -> material -> light group -> light triangle), and at four
// waves per SIMD the chain's latency, not bandwidth, is what a launch on a small scene waits for.
template <bool LDST> struct ShadeTables {
  const DevScene& sc;
  uint32_t base;              // first float4 of the copy in the workgroup's dynamic LDS
  __device__ __forceinline__ uint32_t oMat() const { return base + 3 * sc.numOrigTris; }
  __device__ __forceinline__ uint32_t oLight() const { return oMat() + sc.numMats; }
  __device__ __forceinline__ uint32_t oGroup() const { return oLight() + 3 * sc.numLightTris; }
  __device__ __forceinline__ void stage() const {
    if constexpr (LDST) {
      for (uint32_t i = threadIdx.x; i < 3 * sc.numOrigTris; i += blockDim.x) g_lds[base + i] = sc.nrm[i];
      for (uint32_t i = threadIdx.x; i < sc.numMats; i += blockDim.x) g_lds[oMat() + i] = sc.mats[i];
      float* lt = (float*)(g_lds + oLight());
      for (uint32_t i = threadIdx.x; i < 12 * sc.numLightTris; i += blockDim.x) lt[i] = sc.lights[i];
      uint32_t* gr = (uint32_t*)(g_lds + oGroup());
      for (uint32_t i = threadIdx.x; i < sc.numLightGroups; i += blockDim.x) {
        gr[i] = sc.lightFirst[i]; gr[sc.numLightGroups + i] = sc.lightCount[i];
      }
    }
  }
  __device__ __forceinline__ float4 nrm(uint32_t i) const { if constexpr (LDST) return g_lds[base + i]; else return sc.nrm[i]; }
  __device__ __forceinline__ float4 mat(uint32_t i) const { if constexpr (LDST) return g_lds[oMat() + i]; else return sc.mats[i]; }
  __device__ __forceinline__ float light(uint32_t t, uint32_t k) const {
    if constexpr (LDST) return ((const float*)(g_lds + oLight()))[t * 12 + k]; else return sc.lights[(size_t)t * 12 + k];
  }
  __device__ __forceinline__ uint32_t lightFirst(uint32_t li) const {
    if constexpr (LDST) return ((const uint32_t*)(g_lds + oGroup()))[li]; else return sc.lightFirst[li];
  }
  __device__ __forceinline__ uint32_t lightCount(uint32_t li) const {
    if constexpr (LDST) return ((const uint32_t*)(g_lds + oGroup()))[sc.numLightGroups + li]; else return sc.lightCount[li];
  }
};

// ------------------------------------------------------------------------------------------------ connect
// Light-sample rays: Scene.trace + the body of `if(distanceY > lightDistance - FloatEpsilon)`, rendering.d:414-425.
// At most one entry per radiance record and launch, so the read-modify-write of Lacc needs no atomics and the
// per-record order of additions (light 0, light 1, ..., sun, sky) is the launch order.  The record was set to
// (0, 0, 0, tag) by the shading of the vertex (`vec3 L;`, rendering.d:399).
template <bool LDSS, bool COUNT>
__global__ void PT_TRAVERSAL_BOUNDS k_connect(DevScene sc, float eps, QMeta q, uint32_t segCap, const float4* __restrict__ shO,
                                                  const float4* __restrict__ shD, const float4* __restrict__ shW,
                                                  uint32_t* taskCounter, float area, float4* __restrict__ Lacc,
                                                  unsigned long long* stats) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  stageScene<LDSS>(sc);
  __syncthreads();
  uint32_t cN = 0, cT = 0;
  const uint32_t total = q.prefix[PT_SEGS];
  if (blockIdx.x == 0 && threadIdx.x == 0) statAdd(stats, STAT_RAYS_LIGHT, (unsigned long long)total);
  SegCursor cur; cur.seg = 0; cur.lo = 1; cur.hi = 0;      // empty: the first entry searches
  Chunk ck;
  while (nextChunk(&pool, taskCounter, q, total, 0, ck)) {
    uint32_t j = ck.begin + (threadIdx.x & 63u);
    if (j < ck.end) {
      uint32_t e = physIndexHint(q, j, ck.seg, segCap, cur);
      float4 ro = shO[e], rd = shD[e];
      V3 phi = xyz(rd);
      Hit h;
      if (traverse<false, LDSS, COUNT>(sc, xyz(ro), phi, eps, h, cN, cT)) {
        float lightDistance = rd.w;
        if (h.t > lightDistance - eps) {
          float4 w = shW[e];
          float4 q0 = sc.nrm[3 * h.tri], q1 = sc.nrm[3 * h.tri + 1], q2 = sc.nrm[3 * h.tri + 2];
          V3 normalY = interpNormal(xyz(q0), xyz(q1), xyz(q2), h.u, h.v);
          float4 m = sc.mats[f2u(q0.w)];
          V3 Le = m.x * v3(m.y, m.z, m.w);
          float G = w.w * dot(normalY, -phi) / (lightDistance * lightDistance);
          if (!(G > 0.0f)) G = 0.0f;
          V3 c = xyz(w) * Le * G * area;
          uint32_t slot = f2u(ro.w);
          float4 L = Lacc[slot];
          L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z;
          Lacc[slot] = L;
        }
      }
    }
  }
  flushCounts<COUNT>(stats, STAT_NODES_CLOSEST, cN, cT);
}

// Sun and sky rays of one vertex: Scene.hitsNothing + `L += ...`, rendering.d:440-443 and 455-457.  An entry is
// (origin, lane slot), (sun direction, n.sunDir), (sky direction, n.dir) and X; a zero direction is not traced.
// direct: the vertex is a view-ray hit whose L nothing else adds to before this kernel; X = (color, material tag) and
// the kernel stores L = 0 + ... with the tag (see k_accumulate) without reading it.  Otherwise X = (BRDF * color, -)
// and L is read, added to and written back when a ray got through.
template <bool LDSS, bool COUNT>
__global__ void PT_TRAVERSAL_BOUNDS k_occluded(DevScene sc, float eps, V3 sunRadiance, V3 skyRadiance, int direct, QMeta q,
                                                   uint32_t segCap, const float4* __restrict__ shO, const float4* __restrict__ shD,
                                                   const float4* __restrict__ shW, const float4* __restrict__ shX,
                                                   uint32_t* taskCounter, float4* __restrict__ Lacc, unsigned long long* stats) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  stageScene<LDSS>(sc);
  __syncthreads();
  uint32_t cN = 0, cT = 0, nRays = 0;
  const uint32_t total = q.prefix[PT_SEGS];
  if (blockIdx.x == 0 && threadIdx.x == 0) statAdd(stats, STAT_ENV_ENTRIES, (unsigned long long)total);
  SegCursor cur; cur.seg = 0; cur.lo = 1; cur.hi = 0;      // empty: the first entry searches
  Chunk ck;
  while (nextChunk(&pool, taskCounter, q, total, 0, ck)) {
    uint32_t j = ck.begin + (threadIdx.x & 63u);
#ifdef PT_PROBE_OCC
    if (j < ck.end && (threadIdx.x & 63u) < PT_PROBE_OCC) {
#else
    if (j < ck.end) {
#endif
      uint32_t e = physIndexHint(q, j, ck.seg, segCap, cur);
      float4 ro = shO[e], a = shD[e], b = shW[e];
      bool eSun = a.x != 0.0f || a.y != 0.0f || a.z != 0.0f, eSky = b.x != 0.0f || b.y != 0.0f || b.z != 0.0f;
      nRays += (eSun ? 1u : 0u) + (eSky ? 1u : 0u);
      // One traversal after the other, each with the lanes that have that ray.  (Both rays of a lane in one loop, the
      // sky ray starting as soon as the lane's sun ray is decided, was measured 5 % slower: sun rays are nearly
      // parallel and walk the tree together, and mixing them with sky rays breaks that up.)
      bool vSun = false, vSky = false;
      Hit h;
      if (eSun) vSun = !traverse<true, LDSS, COUNT, !COUNT>(sc, xyz(ro), xyz(a), eps, h, cN, cT);   // one octant: see traverse
      if (eSky) vSky = !traverse<true, LDSS, COUNT>(sc, xyz(ro), xyz(b), eps, h, cN, cT);
      if (direct || vSun || vSky) {
        float4 x = shX[e];
        uint32_t slot = f2u(ro.w);
        V3 Tw; float4 L;
        if (direct) { V3 color = xyz(x); Tw = v3(1, 1, 1) * (PT_BRDF * color); L = make_float4(0.0f, 0.0f, 0.0f, x.w); }
        else { Tw = xyz(x); L = Lacc[slot]; }
        if (vSun) { V3 c = Tw * sunRadiance * a.w; L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z; }
        if (vSky) { V3 c = Tw * skyRadiance * b.w; L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z; }
        Lacc[slot] = L;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) nRays += __shfl_down(nRays, off);
  if ((threadIdx.x & 63) == 0 && nRays) statAdd(stats, STAT_RAYS_OCCL, (unsigned long long)nRays);
  flushCounts<COUNT>(stats, STAT_NODES_ANY, cN, cT);
}

// k_connect for scenes traversed from global memory, with lanes that take the next ray when theirs is decided.
#ifndef PT_REFILL_CONNECT
#define PT_REFILL_CONNECT 48
#endif
__global__ void PT_TRAVERSAL_BOUNDS k_connect_refill(DevScene sc, float eps, QMeta q, uint32_t segCap, const float4* __restrict__ shO,
                                                     const float4* __restrict__ shD, const float4* __restrict__ shW,
                                                     uint32_t* taskCounter, float area, float4* __restrict__ Lacc,
                                                     unsigned long long* stats) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  __syncthreads();
  const uint32_t total = q.prefix[PT_SEGS];
  if (blockIdx.x == 0 && threadIdx.x == 0) statAdd(stats, STAT_RAYS_LIGHT, (unsigned long long)total);
  SegCursor cur; cur.seg = 0; cur.lo = 1; cur.hi = 0;
  uint32_t e = 0, slot = 0; bool hasRay = false; float lightDistance = 0.0f;
  LaneTrav t; idleTrav(t);
  uint32_t stackRest[PT_STACK_REST];
  const auto stack = globalSceneStack(stackRest);
  WaveSupply sup;
  for (;;) {
    if (hasRay && !t.busy()) {
      // the body of `if(distanceY > lightDistance - FloatEpsilon)`, rendering.d:416-425
      if (t.found && t.h.t > lightDistance - eps) {
        const float4 w = shW[e];
        const float4 q0 = sc.nrm[3 * t.h.tri], q1 = sc.nrm[3 * t.h.tri + 1], q2 = sc.nrm[3 * t.h.tri + 2];
        const V3 normalY = interpNormal(xyz(q0), xyz(q1), xyz(q2), t.h.u, t.h.v);
        const float4 m = sc.mats[f2u(q0.w)];
        const V3 Le = m.x * v3(m.y, m.z, m.w);
        float G = w.w * dot(normalY, -t.d) / (lightDistance * lightDistance);
        if (!(G > 0.0f)) G = 0.0f;
        const V3 c = xyz(w) * Le * G * area;
        float4 L = Lacc[slot];
        L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z;
        Lacc[slot] = L;
      }
      hasRay = false;
    }
    bool need = !hasRay;
    while (sup.open && __any(need)) {
      uint32_t j;
      if (!sup.take(&pool, taskCounter, q, total, 0, need, j)) break;
      if (j != PT_NONE) {
        e = physIndexHint(q, j, sup.seg, segCap, cur);
        const float4 ro = shO[e], rd = shD[e];
        slot = f2u(ro.w); lightDistance = rd.w;
        startTrav(sc, t, xyz(ro), xyz(rd));
        hasRay = true; need = false;
      }
    }
    if (!__any(hasRay)) break;
    advanceTrav<false, PT_LEAF_WAIT_CONNECT>(sc, t, stack, eps, sup.open ? (uint32_t)PT_REFILL_CONNECT : 1u, lightDistance - eps);
  }
}

// k_occluded for scenes traversed from global memory, with lanes that take the next ray when theirs is decided (the
// scheme of k_extend_refill): a lane works through its entry's sun ray, then its sky ray, then the next entry of the
// wave's chunk; whenever fewer than PT_REFILL_ANY lanes are traversing, the decided rays are written out and the free
// lanes supplied.  An entry's L is stored once, when its last ray is decided, in the order sun, sky.
#ifndef PT_REFILL_ANY
#define PT_REFILL_ANY 40
#endif
__global__ void PT_TRAVERSAL_BOUNDS k_occluded_refill(DevScene sc, float eps, V3 sunRadiance, V3 skyRadiance, int direct, QMeta q,
                                                      uint32_t segCap, const float4* __restrict__ shO, const float4* __restrict__ shD,
                                                      const float4* __restrict__ shW, const float4* __restrict__ shX,
                                                      uint32_t* taskCounter, float4* __restrict__ Lacc, unsigned long long* stats) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  __syncthreads();
  const uint32_t total = q.prefix[PT_SEGS];
  if (blockIdx.x == 0 && threadIdx.x == 0) statAdd(stats, STAT_ENV_ENTRIES, (unsigned long long)total);
  SegCursor cur; cur.seg = 0; cur.lo = 1; cur.hi = 0;
  uint32_t nRays = 0;
  // the lane's entry: where it lies, what is still to be traced, what got through, the cosines its radiance is weighted with
  uint32_t e = 0, slot = 0; bool haveEntry = false, skyTodo = false, onSky = false, vSun = false, vSky = false, tracing = false;
  float sSun = 0.0f, sSky = 0.0f;
  LaneTrav t; idleTrav(t);
  uint32_t stackRest[PT_STACK_REST];
  const auto stack = globalSceneStack(stackRest);
  WaveSupply sup;
  for (;;) {
    // lanes whose ray is decided: note the answer, go on with the entry's sky ray or close the entry
    if (tracing && !t.busy()) {
      if (onSky) vSky = !t.found; else vSun = !t.found;
      tracing = false;
      if (!onSky && skyTodo) {
        const float4 b = shW[e];
        onSky = true; skyTodo = false; tracing = true;
        startTrav(sc, t, t.o, xyz(b));
      }
    }
    auto closeEntry = [&]() {
      if (direct || vSun || vSky) {
        const float4 x = shX[e];
        V3 Tw; float4 L;
        if (direct) { V3 color = xyz(x); Tw = v3(1, 1, 1) * (PT_BRDF * color); L = make_float4(0.0f, 0.0f, 0.0f, x.w); }
        else { Tw = xyz(x); L = Lacc[slot]; }
        if (vSun) { V3 c = Tw * sunRadiance * sSun; L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z; }
        if (vSky) { V3 c = Tw * skyRadiance * sSky; L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z; }
        Lacc[slot] = L;
      }
      haveEntry = false;
    };
    if (haveEntry && !tracing) closeEntry();
    bool need = !haveEntry;
    while (sup.open && __any(need)) {
      uint32_t j;
      if (!sup.take(&pool, taskCounter, q, total, 0, need, j)) break;
      if (j != PT_NONE) {
        e = physIndexHint(q, j, sup.seg, segCap, cur);
        const float4 ro = shO[e], a = shD[e], b = shW[e];
        const bool eSun = a.x != 0.0f || a.y != 0.0f || a.z != 0.0f, eSky = b.x != 0.0f || b.y != 0.0f || b.z != 0.0f;
        nRays += (eSun ? 1u : 0u) + (eSky ? 1u : 0u);
        slot = f2u(ro.w); sSun = a.w; sSky = b.w; vSun = false; vSky = false; haveEntry = true;
        if (eSun) { onSky = false; skyTodo = eSky; tracing = true; startTrav(sc, t, xyz(ro), xyz(a)); }
        else if (eSky) { onSky = true; skyTodo = false; tracing = true; startTrav(sc, t, xyz(ro), xyz(b)); }
        else closeEntry();                                     // (shade emits no such entry; kept for safety)
        need = !haveEntry;
      }
    }
    if (!__any(tracing)) break;
    advanceTrav<true, PT_LEAF_WAIT_ANY>(sc, t, stack, eps, sup.open ? (uint32_t)PT_REFILL_ANY : 1u);
  }
  for (int off = 32; off > 0; off >>= 1) nRays += __shfl_down(nRays, off);
  if ((threadIdx.x & 63) == 0 && nRays) statAdd(stats, STAT_RAYS_OCCL, (unsigned long long)nRays);
}

// ------------------------------------------------------------------------------------------------ hit order
// Scenes traversed from global memory: the hits of a hit-queue segment, ordered by the cell of the hit position before
// they are shaded.  A segment's hits come in the order of the rays, for view rays the order of the pixel tiles: a wave
// of 64 consecutive hits lies behind 64 neighbouring pixels -- at 64 different depths where the geometry is finer than
// a pixel, so the light-sample, sun and sky rays that start there share little of what they fetch.  The cell grid is
// the 16-bit grid of the quantised nodes (DevScene::qBase / qScale) coarsened to at most 2^PT_ORDER_BITS cells per
// axis, and further until the box of the segment's hits is at most 65536 cells; the key of a hit is the number of its
// cell in that box, the axis along which the box has the fewest cells running fastest (behind a patch of pixels the
// box is long and thin: consecutive keys are consecutive slabs of it).  Hits of one cell keep their order.
// The sort is a radix sort of the 16-bit keys in two stable counting passes: every wave counts a contiguous piece of
// the segment per digit value, one scan over values x waves, every wave places its piece in order.
// order[s * segCap + r] = index within segment s of the hit that is shaded r-th; what shade emits keeps that order
// (its appends go by wave ballot), so it carries over to all shadow queues and the ray queue of the stage.  Which hit
// is shaded when does not change any result: every entry belongs to a radiance record of its own.
#ifndef PT_ORDER_BITS
#define PT_ORDER_BITS 10
#endif
constexpr uint32_t PT_ORDER_THREADS = 1024, PT_ORDER_WAVES = PT_ORDER_THREADS / 64, PT_ORDER_DIGIT = 256;
struct CellFrame { uint32_t sh, mn[3], m1, m2; int a0, a1, a2; };      // key = c[a0] + m1 * c[a1] + m2 * c[a2]
__device__ __forceinline__ void gridOf(const DevScene& sc, const float4& p, uint32_t g[3]) {
  const float v[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) g[k] = (uint32_t)fminf(fmaxf((v[k] - sc.qBase[k]) / sc.qScale[k], 0.0f), 65535.0f);
}
__device__ __forceinline__ uint32_t cellKey(const CellFrame& f, const uint32_t g[3]) {
  uint32_t c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = (g[k] >> f.sh) - f.mn[k];
  const uint32_t c0 = f.a0 == 0 ? c[0] : f.a0 == 1 ? c[1] : c[2];
  const uint32_t c1 = f.a1 == 0 ? c[0] : f.a1 == 1 ? c[1] : c[2];
  const uint32_t c2 = f.a2 == 0 ? c[0] : f.a2 == 1 ? c[1] : c[2];
  return c0 + f.m1 * c1 + f.m2 * c2;
}
// Lanes of the wave with the same 8-bit digit as this one: how many come before it (rank) and how many there are (n);
// leader: the first of them.  One ballot per bit of the digit, whatever the number of different digits in the wave.
// Converged code; lanes with valid == false take no part.
__device__ __forceinline__ void rankByDigit(bool valid, uint32_t d, uint32_t& rank, uint32_t& n, bool& leader) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (uint32_t bit = 0; bit < 8; ++bit) {
    const bool one = (d >> bit) & 1u;
    const unsigned long long b = __ballot(one);
    peers &= one ? b : ~b;
  }
  rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
  n = (uint32_t)__popcll(peers); leader = valid && rank == 0;
}
// One stable counting pass of a workgroup over n (key, index) pairs by the digit (key >> shift) & 255.  inIdx == null:
// the index of pair r is r.  outKey == null: only the indices are written.  sCount: PT_ORDER_WAVES * PT_ORDER_DIGIT
// words, sTotal: PT_ORDER_DIGIT words.  Ends with a barrier.
__device__ __forceinline__ void countingPass(const uint32_t* __restrict__ inKey, const uint32_t* __restrict__ inIdx, uint32_t shift,
                                             uint32_t* __restrict__ outKey, uint32_t* __restrict__ outIdx, uint32_t n,
                                             uint32_t* sCount, uint32_t* sTotal) {
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  for (uint32_t i = tid; i < PT_ORDER_WAVES * PT_ORDER_DIGIT; i += PT_ORDER_THREADS) sCount[i] = 0;
  __syncthreads();
  const uint32_t piece = ((n + PT_ORDER_WAVES - 1u) / PT_ORDER_WAVES + 63u) & ~63u;      // every wave a contiguous piece, in steps of 64
  const uint32_t lo = min(wave * piece, n), hi = min(lo + piece, n);
  uint32_t* const mine = sCount + wave * PT_ORDER_DIGIT;
  for (uint32_t b = lo; b < hi; b += 64u) {
    const uint32_t r = b + lane; const bool valid = r < hi;
    const uint32_t d = valid ? (inKey[r] >> shift) & (PT_ORDER_DIGIT - 1u) : 0u;
    uint32_t rank, cnt; bool leader;
    rankByDigit(valid, d, rank, cnt, leader);
    if (leader) mine[d] += cnt;
  }
  __syncthreads();
  // exclusive scan, value-major and wave-minor
  if (tid < PT_ORDER_DIGIT) {
    uint32_t t = 0;
    for (uint32_t w = 0; w < PT_ORDER_WAVES; ++w) t += sCount[w * PT_ORDER_DIGIT + tid];
    sTotal[tid] = t;
  }
  __syncthreads();
  if (wave == 0) {
    constexpr uint32_t PER = PT_ORDER_DIGIT / 64;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { v[k] = sTotal[lane * PER + k]; sum += v[k]; }
    uint32_t x = sum;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, off); if (lane >= (uint32_t)off) x += y; }
    uint32_t run = x - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { sTotal[lane * PER + k] = run; run += v[k]; }
  }
  __syncthreads();
  if (tid < PT_ORDER_DIGIT) {
    uint32_t run = sTotal[tid];
    for (uint32_t w = 0; w < PT_ORDER_WAVES; ++w) { const uint32_t t = sCount[w * PT_ORDER_DIGIT + tid]; sCount[w * PT_ORDER_DIGIT + tid] = run; run += t; }
  }
  __syncthreads();
  for (uint32_t b = lo; b < hi; b += 64u) {
    const uint32_t r = b + lane; const bool valid = r < hi;
    const uint32_t key = valid ? inKey[r] : 0u, d = (key >> shift) & (PT_ORDER_DIGIT - 1u);
    uint32_t rank, cnt; bool leader;
    rankByDigit(valid, d, rank, cnt, leader);
    if (valid) {
      const uint32_t pos = mine[d] + rank;
      outIdx[pos] = inIdx ? inIdx[r] : r;
      if (outKey) outKey[pos] = key;
    }
    if (leader) mine[d] += cnt;             // (after the reads of this step: a wave's LDS accesses keep their order)
  }
  __syncthreads();
}
// work: 3 words per entry of the hit queue (keys, keys and indices after the first pass)
__global__ void __launch_bounds__(PT_ORDER_THREADS) k_hit_order(DevScene sc, QMeta hitQ, uint32_t segCap, uint32_t capHit,
                                                                const float4* __restrict__ hitA, uint32_t* __restrict__ order,
                                                                uint32_t* __restrict__ work) {
  __shared__ uint32_t sCount[PT_ORDER_WAVES * PT_ORDER_DIGIT];
  __shared__ uint32_t sTotal[PT_ORDER_DIGIT];
  __shared__ uint32_t sMin[3], sMax[3];
  const uint32_t seg = blockIdx.x, n = hitQ.prefix[seg + 1] - hitQ.prefix[seg];
  if (n == 0) return;
  const size_t base = (size_t)seg * segCap;
  const float4* in = hitA + base;
  uint32_t* const key0 = work + base; uint32_t* const key1 = work + capHit + base; uint32_t* const idx1 = work + 2 * (size_t)capHit + base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (tid < 3) { sMin[tid] = 0xFFFFu; sMax[tid] = 0u; }
  __syncthreads();
  // the box of the segment's hits on the grid
  uint32_t mn[3] = {0xFFFFu, 0xFFFFu, 0xFFFFu}, mx[3] = {0u, 0u, 0u};
  for (uint32_t i = tid; i < n; i += PT_ORDER_THREADS) {
    uint32_t g[3]; gridOf(sc, in[i], g);
#pragma unroll
    for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], g[k]); mx[k] = max(mx[k], g[k]); }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int off = 32; off > 0; off >>= 1) { mn[k] = min(mn[k], (uint32_t)__shfl_xor((int)mn[k], off)); mx[k] = max(mx[k], (uint32_t)__shfl_xor((int)mx[k], off)); }
    if (lane == 0) { atomicMin(&sMin[k], mn[k]); atomicMax(&sMax[k], mx[k]); }
  }
  __syncthreads();
  CellFrame f; f.sh = 16u - PT_ORDER_BITS;
  uint32_t ext[3];
  for (;; f.sh++) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { f.mn[k] = sMin[k] >> f.sh; ext[k] = (sMax[k] >> f.sh) - f.mn[k] + 1u; }
    if ((uint64_t)ext[0] * ext[1] * ext[2] <= 65536u) break;        // (one cell at the latest when sh = 16)
  }
  // axes by the number of cells, fewest first
  f.a0 = 0; f.a1 = 1; f.a2 = 2;
  if (ext[f.a1] < ext[f.a0]) { const int t = f.a0; f.a0 = f.a1; f.a1 = t; }
  if (ext[f.a2] < ext[f.a1]) { const int t = f.a1; f.a1 = f.a2; f.a2 = t; }
  if (ext[f.a1] < ext[f.a0]) { const int t = f.a0; f.a0 = f.a1; f.a1 = t; }
  const uint32_t e0 = f.a0 == 0 ? ext[0] : f.a0 == 1 ? ext[1] : ext[2], e1 = f.a1 == 0 ? ext[0] : f.a1 == 1 ? ext[1] : ext[2];
  f.m1 = e0; f.m2 = e0 * e1;
  for (uint32_t i = tid; i < n; i += PT_ORDER_THREADS) { uint32_t g[3]; gridOf(sc, in[i], g); key0[i] = cellKey(f, g); }
  __syncthreads();
  countingPass(key0, nullptr, 0u, key1, idx1, n, sCount, sTotal);
  countingPass(key1, idx1, 8u, nullptr, order + base, n, sCount, sTotal);
}

// ------------------------------------------------------------------------------------------------ shade
struct ShadeCtx {
  const DevScene& sc; const DevCfg& cfg; const Wavefront& wf; uint32_t* sCnt; uint32_t T, segCapRay, segCapSh;
  // FUSE (rays traced where they are made, see k_shade): segment size of the hit queue the continuation rays' hits go
  // to, and what the lane traced (indirect rays, light-sample rays, sun/sky rays, vertices with a sun or sky ray)
  uint32_t segCapHit = 0; uint32_t nInd = 0, nLight = 0, nOccl = 0, nEnv = 0;
  float4* outA = nullptr; float4* outB = nullptr;      // the hit records the continuation rays' hits go to
  uint32_t segBits = 0;                                // ... dealt in turn to the 1 << segBits segments from T on
};

// One computeLindirect head (rendering.d:378-385): Russian roulette, direction, ray; appended to ray queue qOut.
// FIRST: the chain starts here, at a view-ray hit.  Its depth (1) and generator state (the chain's stream after these
// three draws) follow from the lane slot, so they are not stored: the shading of the ray's hit works them out again
// (most such rays hit nothing and nobody would read what was stored).  Otherwise the generator goes to the lane slot's
// state; the depth of a vertex is its bounce round + 1.  No throughput is carried along a chain: every vertex keeps
// its own computeLdirect sum and k_accumulate folds the chain from its end, as the recursion does (rendering.d:391).
template <bool FIRST, bool FUSE>
__device__ __forceinline__ bool bounce(ShadeCtx& cx, bool active, Rng& g, uint32_t depth, V3 pos, V3 normal, uint32_t laneSlot) {
  bool alive = false; V3 dir = v3(0, 0, 0);
  if (active) {
    float a = g.uniform(0.0f, 1.0f);
    if (!(a > cx.cfg.rrProb || depth > cx.cfg.maxDepth)) {
      float psi = g.uniform(0.0f, 2.0f * PT_PI_F);
      float phi = g.uniform(0.0f, PT_PI_2_F);
#ifdef PT_PROBE_D
      dir = normal + v3(phi, psi, 0.0f) * 1e-3f;
#else
      dir = angleToDirection(phi, psi, normal);
#endif
      alive = true;
    }
  }
  if constexpr (FUSE) {
    // the ray is traced here (Scene.trace, rendering.d:389) and what is appended is its hit, in the form extend writes
    bool isHit = false; Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0;
    const V3 o = pos + normal * 0.1f;
    if (alive) {
      uint32_t cN = 0, cT = 0;
#ifndef PT_PROBE_A
      isHit = traverse<false, true, false, PT_FUSE_OCTANTS>(cx.sc, o, dir, cx.cfg.eps, h, cN, cT);
#endif
      cx.nInd++;
    }
    uint32_t idx = waveAppend(isHit, &cx.sCnt[0]);
    if (isHit) {
      uint32_t e = (cx.T + (idx & ((1u << cx.segBits) - 1u))) * cx.segCapHit + (idx >> cx.segBits);
      cx.outA[e] = f4(o + dir * h.t, u2f(h.tri));
      cx.outB[e] = make_float4(h.u, h.v, u2f(laneSlot), 0.0f);
      if (!FIRST) cx.wf.rng[laneSlot] = make_uint2(g.s0, g.s1);
    }
    return isHit;
  } else {
    uint32_t idx = waveAppend(alive, &cx.sCnt[0]);
    if (alive) {
      uint32_t e = cx.T * cx.segCapRay + idx;
      cx.wf.rayO[e] = f4(pos + normal * 0.1f, u2f(laneSlot));
      cx.wf.rayD[e] = f4(dir, 0.0f);
      if (!FIRST) cx.wf.rng[laneSlot] = make_uint2(g.s0, g.s1);
    }
    return alive;
  }
}

__device__ __forceinline__ void emitShadow(ShadeCtx& cx, int kind, bool emit, V3 org, V3 dir, float dw, V3 w, float ww, uint32_t laneSlot) {
  uint32_t idx = waveAppend(emit, &cx.sCnt[1 + kind]);
  if (emit) {
    size_t e = (size_t)kind * cx.wf.capQueue + cx.T * cx.segCapSh + idx;
    cx.wf.shO[e] = f4(org, u2f(laneSlot));
    cx.wf.shD[e] = f4(dir, dw);
    cx.wf.shW[e] = f4(w, ww);
  }
}

// pickRandomLightPoint (rendering.d:164-193) for light group li and the direction to the sample (rendering.d:404-412);
// returns whether a ray is to be traced towards it.
template <bool LDST>
__device__ __forceinline__ bool sampleLight(const ShadeTables<LDST>& tb, Rng& g, uint32_t li, V3 x, V3 normalX, float eps, V3& phi,
                                            float& lightDistance, float& ndot) {
  uint32_t first = tb.lightFirst(li), cnt = tb.lightCount(li);
  float a = g.uniform(0.0f, 1.0f);
  uint32_t chosen = cnt - 1;
  for (uint32_t j = 0; j < cnt; j++) {
    float prob = tb.light(first + j, 10);
    if (prob > a) { chosen = j; break; }
    a -= prob;
  }
  uint32_t lt = first + chosen;
  float u, v;
  do { u = g.uniform(0.0f, 1.0f); v = g.uniform(0.0f, 1.0f); } while (!(u + v <= 1.0f));
  V3 y = v3(tb.light(lt, 0), tb.light(lt, 1), tb.light(lt, 2)) + v3(tb.light(lt, 3), tb.light(lt, 4), tb.light(lt, 5)) * v +
         v3(tb.light(lt, 6), tb.light(lt, 7), tb.light(lt, 8)) * u;
  lightDistance = length(y - x);
  phi = normalized(y - x);
  ndot = dot(normalX, phi);
  return ndot > eps;
}

// The sun direction (rendering.d:432-439) and the sky direction (rendering.d:448-454) sampled at a vertex, each with
// whether its ray is traced and the cosine its radiance is weighted with.
struct EnvSample { bool eSun, eSky; V3 dSun, dSky; float sSun, sSky; };
__device__ __forceinline__ EnvSample sampleEnv(const DevCfg& cfg, Rng& g, V3 normalX) {
  EnvSample s;
  float psi = g.uniform(0.0f, 2.0f * PT_PI_F);
  float phi = g.uniform(PT_PI_2_F - PT_PI_2_F * 0.01f, PT_PI_2_F);
  s.dSun = angleToDirection(phi, psi, cfg.sunDir);
  s.eSun = dot(normalX, s.dSun) > cfg.eps;
  s.sSun = dot(normalX, cfg.sunDir);
  psi = g.uniform(0.0f, 2.0f * PT_PI_F);
  phi = g.uniform(0.0f, PT_PI_2_F);
  s.dSky = angleToLocalDirection(phi, psi);
  s.sSky = dot(normalX, s.dSky);
  s.eSky = s.sSky > cfg.eps;
  return s;
}

// computeLdirect up to (not including) the trace calls, rendering.d:396-462: emits one shadow ray per light group and
// one entry for the sun and sky rays together (envX: what the kernel that traces them needs besides, see k_occluded).
// Returns whether the lane emitted that entry.
template <bool LDST>
__device__ __forceinline__ bool directLight(ShadeCtx& cx, const ShadeTables<LDST>& tb, bool active, Rng& g, V3 x, V3 normalX, V3 Tw, float4 envX,
                                            uint32_t laneSlot) {
  const DevScene& sc = cx.sc; const DevCfg& cfg = cx.cfg;
  float eps = cfg.eps;
  int kind = 0;
  if (cfg.hasLights) {
    for (uint32_t li = 0; li < sc.numLightGroups; li++, kind++) {
      bool emit = false; V3 phi = v3(0, 0, 0); float lightDistance = 0.0f, ndot = 0.0f;
      if (active) emit = sampleLight(tb, g, li, x, normalX, eps, phi, lightDistance, ndot);
      emitShadow(cx, kind, emit, x + normalX * eps, phi, lightDistance, Tw, ndot, laneSlot);
    }
  }
  bool emitted = false;
  if (cfg.hasSky) {
    // sun and sky leave from the same point: one entry carries both directions, each with the cosine its radiance is
    // weighted with; a direction that is not traced is stored as 0
    EnvSample s; s.eSun = s.eSky = false; s.dSun = s.dSky = v3(0, 0, 0); s.sSun = s.sSky = 0.0f;
    if (active) s = sampleEnv(cfg, g, normalX);
    emitted = s.eSun || s.eSky;
    uint32_t idx = waveAppend(emitted, &cx.sCnt[1 + kind]);
    if (emitted) {
      size_t ex = cx.T * cx.segCapSh + idx, e = (size_t)kind * cx.wf.capQueue + ex;
      if (!s.eSun) s.dSun = v3(0, 0, 0);
      if (!s.eSky) s.dSky = v3(0, 0, 0);
      cx.wf.shO[e] = f4(x + normalX * 0.1f, u2f(laneSlot));
      cx.wf.shD[e] = f4(s.dSun, s.sSun);
      cx.wf.shW[e] = f4(s.dSky, s.sSky);
      cx.wf.shX[ex] = envX;
    }
  }
  return emitted;
}

// computeLdirect whole (rendering.d:396-462) for the shading variant that traces its own rays: every light group's
// sample ray (the body of k_connect), then the sun and the sky ray (the body of k_occluded), each contribution added
// to L in that order -- the order in which the queued kernels are launched.  Returns the `vec3 L` of the vertex,
// gathered from 0; the caller stores it with the record's tag word.
template <bool LDST>
__device__ __forceinline__ V3 directLightTraced(ShadeCtx& cx, const ShadeTables<LDST>& tb, bool active, Rng& g, V3 x, V3 normalX, V3 Tw) {
  const DevScene& sc = cx.sc; const DevCfg& cfg = cx.cfg;
  const float eps = cfg.eps;
  V3 L = v3(0.0f, 0.0f, 0.0f);
  auto add = [&](V3 c) { L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z; };
  if (active) {
    Hit h; uint32_t cN = 0, cT = 0;
    if (cfg.hasLights) {
      for (uint32_t li = 0; li < sc.numLightGroups; li++) {
        V3 phi; float lightDistance, ndot;
        if (sampleLight(tb, g, li, x, normalX, eps, phi, lightDistance, ndot)) {
          cx.nLight++;
          // Scene.trace + `if(distanceY > lightDistance - FloatEpsilon)`, rendering.d:414-425
          if (traverse<false, true, false, PT_FUSE_OCTANTS, false, true>(sc, x + normalX * eps, phi, eps, h, cN, cT, lightDistance - eps) && h.t > lightDistance - eps) {
            float4 q0 = tb.nrm(3 * h.tri), q1 = tb.nrm(3 * h.tri + 1), q2 = tb.nrm(3 * h.tri + 2);
            V3 normalY = interpNormal(xyz(q0), xyz(q1), xyz(q2), h.u, h.v);
            float4 m = tb.mat(f2u(q0.w));
            V3 Le = m.x * v3(m.y, m.z, m.w);
            float G = ndot * dot(normalY, -phi) / (lightDistance * lightDistance);
            if (!(G > 0.0f)) G = 0.0f;
            add(Tw * Le * G * sc.lightArea[li]);
          }
        }
      }
    }
    if (cfg.hasSky) {
      const EnvSample s = sampleEnv(cfg, g, normalX);
      if (s.eSun || s.eSky) {
        // Scene.hitsNothing + `L += ...`, rendering.d:440-443 and 455-457
        const V3 org = x + normalX * 0.1f;
        bool vSun = false, vSky = false;
#ifndef PT_PROBE_C
        if (s.eSun) vSun = !traverse<true, true, false, PT_FUSE_OCTANTS>(sc, org, s.dSun, eps, h, cN, cT);
        if (s.eSky) vSky = !traverse<true, true, false>(sc, org, s.dSky, eps, h, cN, cT);
#endif
        cx.nOccl += (s.eSun ? 1u : 0u) + (s.eSky ? 1u : 0u); cx.nEnv++;
        if (vSun) add(Tw * cfg.sunRadiance * s.sSun);
        if (vSky) add(Tw * cfg.skyRadiance * s.sSky);
      }
    }
  }
  return L;
}

// PRIMARY: the part of computeL after the view-ray trace has hit (rendering.d:472-477); the generator of
// (pixel, sample) is re-seeded and stepped past the two jitter draws of getViewRay.
// !PRIMARY: the part of computeLindirect after its trace has hit (rendering.d:390-391) = computeLrefl at the hit.
// Block-task T = blockIdx.x reads flat entries [T*P, (T+1)*P) of the hit queue and fills segment T of the
// output queues.
// FUSE (scenes small enough for every shade workgroup to hold nodes and triangles in LDS beside its tables): the rays a
// vertex spawns are traced right here instead of being queued -- the continuation ray of a chain (what is appended
// is its hit, to the next stage's hit queue: no ray queue, no extend launch for indirect rays), the light-sample rays
// and the sun and sky rays (no queue entries, no k_connect / k_occluded launches; L is gathered in registers in the
// order those launches would have added to it).  Same device functions, same operations per ray; what disappears is
// 64 B of queue traffic per indirect ray, 128 B per light-sample ray and 80 B per sun/sky entry -- on scenes of a few
// dozen triangles that traffic is what the launches wait for.  PRIMARY: a chain whose first ray hit
// something is noted in the top bits of the view-ray record's tag word (wf.chainMask), so that accumulate reads only
// the chain records that can hold anything.
// What shading reads of the triangle that was hit: the interpolated normal (scene.d:711-723), the material's colour
// and the tag (material index + 1) that marks a view-ray record as a hit.
template <bool LDST>
__device__ __forceinline__ void vertexTables(const ShadeTables<LDST>& tb, uint32_t tri, float u, float v, V3& normal, V3& color, uint32_t& matTag) {
  float4 q0 = tb.nrm(3 * tri), q1 = tb.nrm(3 * tri + 1), q2 = tb.nrm(3 * tri + 2);
  normal = interpNormal(xyz(q0), xyz(q1), xyz(q2), u, v);
  float4 m = tb.mat(f2u(q0.w));
  color = v3(m.y, m.z, m.w);
  matTag = f2u(q0.w) + 1u;
}

// One vertex: computeLdirect, and the heads of the indirect chains (PRIMARY, rendering.d:474-476) or the chain's
// continuation (rendering.d:376-394).  Called from converged code (the appends rank lanes with a wave ballot);
// `active` says whether the lane has a vertex.  g: the vertex's generator (view-ray hit: stream 0 of its sample past
// the two jitter draws; chain hit: the chain's stream).
// recSlot: the vertex's radiance record.  A view-ray hit: its path slot, (L, tag word).  A chain vertex of bounce round
// r: lane slot + r * chainStride, (L, (material index + 1) | continues << 31) -- computeLdirect at that vertex alone,
// and whether the chain has a vertex in round r + 1.  Where the rays are traced in place the record is written once,
// complete; where they are queued, shading writes (0, 0, 0, material index + 1), the kernels that trace the shadow
// rays add to it, and the record of round r + 1 is marked absent (w = 0) until the shading of that round finds the
// continuation ray's hit in its queue -- every record is written in the batch that reads it.
template <bool PRIMARY, bool LDST, bool FUSE>
__device__ __forceinline__ void shadeVertex(ShadeCtx& cx, const ShadeTables<LDST>& tb, bool active, V3 hitPos, V3 normal, V3 color,
                                            uint32_t matTag, uint32_t depth, Rng& g, uint32_t gp, uint32_t sampleIdx,
                                            uint32_t laneSlot, uint32_t recSlot, float4* recOut = nullptr) {
  const DevCfg& cfg = cx.cfg; const Wavefront& wf = cx.wf;
  uint32_t tagWord = matTag;
  // every chain has a generator stream of its own, so the place of the loop in this kernel is free: the traced
  // variant runs it ahead of computeLdirect, because the record that carries the tag word (with the chains that hit)
  // is written there
  auto chainHeads = [&]() {
    uint32_t hitMask = 0;
#ifdef PT_PROBE_B
    for (uint32_t c = 0; c < 0; c++) {
#else
    for (uint32_t c = 0; c < cfg.numIndirect; c++) {
#endif
      uint32_t lane = (c + 1) * wf.capSlots + laneSlot;
      Rng gc; gc.s0 = 1; gc.s1 = 0;
      if (active) gc.seed(cfg.seed, gp, sampleIdx, c + 1);
      if (bounce<true, FUSE>(cx, active, gc, 0, hitPos, normal, lane)) hitMask |= 1u << c;
      // queued: whether the ray hits is not known yet -- the chain's first record is absent until round 0 shades the hit
      if (!FUSE && active) ((float*)&wf.Lacc[lane])[3] = 0.0f;
    }
    return hitMask;
  };
  const V3 Tw = v3(1, 1, 1) * (PT_BRDF * color);
  if constexpr (FUSE) {
    if (PRIMARY && cfg.mode == 1) { uint32_t hitMask = chainHeads(); tagWord |= hitMask << 24; }
    const V3 L = directLightTraced(cx, tb, active, g, hitPos, normal, Tw);
    bool more = false;
    if (!PRIMARY && cfg.mode == 1) more = bounce<false, FUSE>(cx, active, g, depth, hitPos, normal, laneSlot);
    if (active) {
      const float4 rec = f4(L, u2f(PRIMARY ? tagWord : (matTag | (more ? 0x80000000u : 0u))));
      if (recOut) *recOut = rec; else wf.Lacc[recSlot] = rec;       // (recOut: the caller adds the sample up itself)
    }
  } else {
    const bool direct = envDirect(cfg, cx.sc, PRIMARY);
    bool env = directLight(cx, tb, active, g, hitPos, normal, Tw, direct ? f4(color, u2f(tagWord)) : f4(Tw, 0.0f), recSlot);
    // `vec3 L;` of computeLdirect (rendering.d:399) + the tag; when the sun / sky entry carries it, the kernel that
    // traces them stores L
    if (active && !(direct && env)) wf.Lacc[recSlot] = make_float4(0.0f, 0.0f, 0.0f, u2f(tagWord));
    if (PRIMARY && cfg.mode == 1) chainHeads();
    if (!PRIMARY && cfg.mode == 1) {
      if (active && depth <= cfg.maxDepth) ((float*)&wf.Lacc[recSlot + wf.chainStride])[3] = 0.0f;     // the next round's, until it is shaded
      bounce<false, FUSE>(cx, active, g, depth, hitPos, normal, laneSlot);
    }
  }
}

// Ray counts of the traced variants: wave sums, one atomic per wave and counter.
__device__ __forceinline__ void flushTraced(ShadeCtx& cx, bool whole = false) {
  uint32_t nInd = cx.nInd, nOccl = cx.nOccl, nEnv = cx.nEnv, nLight = cx.nLight;
  for (int off = 32; off > 0; off >>= 1) {
    nInd += __shfl_down(nInd, off); nOccl += __shfl_down(nOccl, off); nEnv += __shfl_down(nEnv, off); nLight += __shfl_down(nLight, off);
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* st = cx.wf.stats;
    if (nLight) { statAdd(st, STAT_RAYS_LIGHT, (unsigned long long)nLight); statAdd(st, STAT_FUSED_LIGHT, (unsigned long long)nLight); }
    if (nInd) { statAdd(st, STAT_RAYS_INDIRECT, (unsigned long long)nInd); statAdd(st, STAT_FUSED_INDIRECT, (unsigned long long)nInd); }
    if (nOccl) { statAdd(st, STAT_RAYS_OCCL, (unsigned long long)nOccl); statAdd(st, STAT_FUSED_OCCL, (unsigned long long)nOccl); }
    if (nEnv) statAdd(st, STAT_ENV_ENTRIES, (unsigned long long)nEnv);
    if (whole) {
      if (nLight) statAdd(st, STAT_WHOLE_LIGHT, (unsigned long long)nLight);
      if (nInd) statAdd(st, STAT_WHOLE_INDIRECT, (unsigned long long)nInd);
      if (nOccl) statAdd(st, STAT_WHOLE_OCCL, (unsigned long long)nOccl);
    }
  }
}

template <bool PRIMARY, bool LDST, bool FUSE>
__device__ __forceinline__ void shadeBody(const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* __restrict__ acc,
                                          const Wavefront& wf, int numKinds) {
  static_assert(!FUSE || LDST, "the fused variant keeps scene and tables in LDS");
  __shared__ uint32_t sCnt[1 + PT_MAX_KINDS];
  if (threadIdx.x <= PT_MAX_KINDS) sCnt[threadIdx.x] = 0;
  if constexpr (FUSE) stageScene<true>(sc);
  const ShadeTables<LDST> tb{sc, FUSE ? 2 * sc.numNodes + 3 * sc.numTris : 0};
  tb.stage();
  __syncthreads();
  uint32_t T = blockIdx.x;
  uint32_t segCap = wf.capQueue / PT_SEGS;
  ShadeCtx cx{sc, cfg, wf, sCnt, T, segCap, segCap};
  cx.segCapHit = wf.capHit / PT_SEGS; cx.outA = wf.hitOutA; cx.outB = wf.hitOutB;
  // OWN_SEGMENT (the traced shading of chain hits): block-task T shades what segment T of the queue holds -- the
  // queue was filled by another launch of these kernels, whose workgroups' shares are even enough, and needs no scan --
  // and goes through all bounce rounds: what the workgroup appended in one round (the hits of its vertices'
  // continuation rays, in segment T of the other pair of record arrays) is what it shades in the next.  No launch, no
  // scan and no exchange between rounds; nobody else reads or writes segment T of either pair.
  constexpr bool OWN_SEGMENT = FUSE && !PRIMARY;
  const QMeta& hitQ = wf.hitQ;
  Task tk; SegCursor seg;
  if constexpr (OWN_SEGMENT) {
    tk.begin = 0; tk.end = hitQ.count[T]; tk.seg = T; seg.seg = T; seg.lo = 0; seg.hi = tk.end;
    if (threadIdx.x == 0 && tk.end) statAdd(wf.stats, STAT_HITS, (unsigned long long)tk.end);
  } else {
    tk = taskOf(hitQ, T); seg = segCursor(hitQ, tk.seg);
    if (T == 0 && threadIdx.x == 0) {
      statAdd(wf.stats, STAT_HITS, (unsigned long long)hitQ.prefix[PT_SEGS]);
      if (PRIMARY) statAdd(wf.stats, STAT_HITS_PRIMARY, (unsigned long long)hitQ.prefix[PT_SEGS]);
    }
  }
  const FastDiv divCap = wf.divCap;
  uint32_t round = wf.round;
  const float4* inA = wf.hitA; const float4* inB = wf.hitB;
  for (;;) {
  for (uint32_t base = tk.begin; base < tk.end; base += blockDim.x) {
    uint32_t j = base + threadIdx.x;
    bool active = j < tk.end;
    V3 hitPos = v3(0, 0, 0), normal = v3(0, 0, 1), color = v3(0, 0, 0);
    uint32_t laneSlot = 0, gp = 0, sampleIdx = 0;
    const uint32_t depth = PRIMARY ? 0u : round + 1u;
    uint32_t matTag = 0;                     // material index + 1: marks the slot as a view-ray hit (k_accumulate)
    Rng g; g.s0 = 1; g.s1 = 0;
    if (active) {
      uint32_t e = physIndex(hitQ, j, seg, wf.capHit / PT_SEGS);
      if (!FUSE && wf.hitOrder) e = seg.seg * (wf.capHit / PT_SEGS) + wf.hitOrder[e];       // see k_hit_order
      float4 ha = inA[e], hb = inB[e];
      hitPos = xyz(ha); laneSlot = f2u(hb.z);
      uint32_t tri = f2u(ha.w);
      if (PRIMARY) {
        uint32_t s = fdiv(pm.divNpix, laneSlot), lp = laneSlot - s * pm.npix;
        gp = mapPixel(pm, cfg.W, lp);
        sampleIdx = cfg.sampleBase + ((uint32_t)acc[gp].w + s) * cfg.sampleStride;
        g.seed(cfg.seed, gp, sampleIdx, 0);
        g.next(); g.next();                                            // the x and y jitter of getViewRay
      } else if (round == 0) {
        // hit of a chain's first ray: what bounce<true> did not store
        uint32_t chain = fdiv(divCap, laneSlot), slot = laneSlot - chain * wf.capSlots;         // chain = 1 + index
        uint32_t s = fdiv(pm.divNpix, slot), lp = slot - s * pm.npix;
        gp = mapPixel(pm, cfg.W, lp);
        sampleIdx = cfg.sampleBase + ((uint32_t)acc[gp].w + s) * cfg.sampleStride;
        g.seed(cfg.seed, gp, sampleIdx, chain);
        g.next(); g.next(); g.next();                                  // Russian roulette, psi, phi of the chain's head
      } else {
        uint2 r = wf.rng[laneSlot]; g.s0 = r.x; g.s1 = r.y;
      }
      vertexTables(tb, tri, hb.x, hb.y, normal, color, matTag);
    }
    shadeVertex<PRIMARY, LDST, FUSE>(cx, tb, active, hitPos, normal, color, matTag, depth, g, gp, sampleIdx, laneSlot,
                                     PRIMARY ? laneSlot : laneSlot + round * wf.chainStride);
  }
  if constexpr (!OWN_SEGMENT) break;
  else {
    __syncthreads();                                   // the round's appends are counted, its records written
    const uint32_t n = sCnt[0];
    if (n == 0 || round >= cfg.maxDepth) break;        // (a vertex deeper than maxDepth spawns nothing, rendering.d:380)
    __syncthreads();
    if (threadIdx.x == 0) { sCnt[0] = 0; statAdd(wf.stats, STAT_HITS, (unsigned long long)n); }
    __syncthreads();
    round++;
    float4* const nextA = cx.outA; float4* const nextB = cx.outB;
    cx.outA = const_cast<float4*>(inA); cx.outB = const_cast<float4*>(inB); inA = nextA; inB = nextB;
    tk.begin = 0; tk.end = n; seg.seg = T; seg.lo = 0; seg.hi = n;
  }
  }
  if constexpr (FUSE) flushTraced(cx);
  __syncthreads();
  if (threadIdx.x == 0) wf.rayOut.count[T] = OWN_SEGMENT ? 0u : sCnt[0];
  if (threadIdx.x >= 1 && threadIdx.x <= (uint32_t)numKinds) wf.sh[threadIdx.x - 1].count[T] = sCnt[threadIdx.x];
}

template <bool PRIMARY, bool LDST>
__global__ void __launch_bounds__(PT_SHADE_THREADS) k_shade(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc, Wavefront wf,
                                               int numKinds) {
  shadeBody<PRIMARY, LDST, false>(sc, cfg, pm, acc, wf, numKinds);
}
template <bool PRIMARY>
__global__ void __launch_bounds__(PT_FUSE_THREADS, PT_FUSE_WAVES) k_shade_traced(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc,
                                                                  Wavefront wf, int numKinds) {
  shadeBody<PRIMARY, true, true>(sc, cfg, pm, acc, wf, numKinds);
}

// A whole sample in one kernel, for the scenes the traced shading handles: raygen (getViewRay, the view-ray masks),
// Scene.trace of the view ray and the shading of its hit, slot by slot -- no view-ray queue, no hit queue for view
// rays, no generator re-seeded in shade (the lane still holds the sample's stream past its two jitter draws).  What
// leaves the kernel is the sample's radiance record and the hits of the chains' first rays.
// Its launch fills the chip in one round: PT_SAMPLE_GROUPS workgroups of PT_SAMPLE_THREADS threads are two workgroups
// of 12 waves per CU, six waves per SIMD (1024 workgroups of 8 waves needed a second round a third full).
#ifndef PT_SAMPLE_THREADS
#define PT_SAMPLE_THREADS 768
#endif
#ifndef PT_SAMPLE_GROUPS
#define PT_SAMPLE_GROUPS 512
#endif
#ifndef PT_POOL_CHUNK
#define PT_POOL_CHUNK 4          // runs a wave takes from the pool at a time
#endif
#ifndef PT_POOL_ALL_FROM
#define PT_POOL_ALL_FROM 512     // a launch with this many runs per workgroup pools all of them, a smaller one a quarter
#endif
#ifndef PT_SAMPLE_WAVES
#define PT_SAMPLE_WAVES PT_FUSE_WAVES
#endif
static_assert(PT_SEGS % PT_SAMPLE_GROUPS == 0 && ((PT_SEGS / PT_SAMPLE_GROUPS) & (PT_SEGS / PT_SAMPLE_GROUPS - 1)) == 0,
              "a workgroup appends to a power of two of whole segments");
__global__ void __launch_bounds__(PT_SAMPLE_THREADS, PT_SAMPLE_WAVES) k_sample_traced(SceneBox box, ViewMasks masks, DevScene sc, DevCfg cfg, PixMap pm,
                                                                   const float4* __restrict__ acc, uint32_t nslots, Wavefront wf) {
  __shared__ uint32_t sCnt[1 + PT_MAX_KINDS];
  if (threadIdx.x <= PT_MAX_KINDS) sCnt[threadIdx.x] = 0;
  stageScene<true>(sc);
  const ShadeTables<true> tb{sc, 2 * sc.numNodes + 3 * sc.numTris};
  tb.stage();
  __syncthreads();
  // Runs of 64 slots (one wave: a tile of pixels at one sample number) are dealt out to the workgroups in turn, run c
  // to workgroup c % gridDim.x, and a workgroup's runs to its waves as they become free: the work of a run ranges from
  // nothing (sky) to a vertex with all its rays, and contiguous shares would leave the workgroups that cover background
  // idle.  A workgroup appends to its PT_SEGS / gridDim.x adjacent segments as one region; it gets at most
  // capSlots / (64 * gridDim.x) runs, so what it appends fits.  Waves do not wait for each other.
  const uint32_t G = gridDim.x, segsPer = PT_SEGS / G, runs = (nslots + 63u) >> 6;
  const uint32_t T = blockIdx.x * segsPer;                 // first segment of this workgroup's region
  const uint32_t segCap = wf.capQueue / PT_SEGS;
  ShadeCtx cx{sc, cfg, wf, sCnt, T, segCap, segCap};
  cx.segCapHit = wf.capHit / PT_SEGS; cx.outA = wf.hitOutA; cx.outB = wf.hitOutB;
  cx.segBits = 31u - (uint32_t)__builtin_clz(segsPer);       // the consumers take a segment each: the hits go to all of them in turn
  uint32_t culled = 0, nConst = 0, hits = 0, mine = 0;
  // A workgroup's own runs go to whichever of its waves is free next.  The other runs are a pool that all workgroups
  // take from, PT_POOL_CHUNK at a time, when their own are done (wf.runPool; PT_POOL_COUNTERS counters that deal out
  // every PT_POOL_COUNTERS-th run of the pool each, a workgroup moves on to the next counter when one is used up):
  // fixed shares differ by several per cent in the vertices they hold, and a launch ends with its slowest workgroup.
  // A large launch pools every run; a small one (a wave gets to do a handful of runs) keeps three quarters as fixed
  // shares, taken from the counter in LDS, because a piece of four is a large part of such a wave's work.
  // A piece is taken only while the region has room for it and for every run the workgroup's waves are working on,
  // each counted with all the hits it can append (perRun).  The region holds more than a full share of such runs plus
  // a piece, so a workgroup declines only after it has done a share, and some workgroup can always take what is left.
  __shared__ uint32_t sNextRun, sBusy, sUsedUp;
  if (threadIdx.x == 0) { sNextRun = 0; sBusy = 0; sUsedUp = 0; }
  __syncthreads();
  const uint32_t perRun = (cfg.mode == 1 ? cfg.numIndirect : 0u) * 64u, cap = segsPer * cx.segCapHit, share = (runs + G - 1) / G;
  const bool pooled = wf.runPool && (perRun == 0 || cap / perRun >= share + PT_POOL_CHUNK);
  const uint32_t own = !pooled ? share : share >= PT_POOL_ALL_FROM ? 0u : share - share / 4, firstPooled = G * own;
  uint32_t pending = 0, pendingRun = 0;                       // the rest of the piece this wave took from the pool
  for (;;) {
    uint32_t run = PT_NONE;
    if (pending) { run = pendingRun; pendingRun += PT_POOL_COUNTERS; pending--; }
    else {
      uint32_t got = 0;
      if ((threadIdx.x & 63u) == 0) {
        const uint32_t k = atomicAdd(&sNextRun, 1u);
        if (k < own) {
          run = blockIdx.x + G * k;
          if (run < runs) { got = 1; atomicAdd(&sBusy, 1u); }
        } else if (pooled && firstPooled < runs) {
          const uint32_t busy = atomicAdd(&sBusy, (uint32_t)PT_POOL_CHUNK) + PT_POOL_CHUNK;
          if (atomicAdd(&sCnt[0], 0u) + busy * perRun <= cap) {
            for (;;) {
              const uint32_t usedUp = atomicAdd(&sUsedUp, 0u);
              if (usedUp >= PT_POOL_COUNTERS) break;
              const uint32_t c = (blockIdx.x + usedUp) % PT_POOL_COUNTERS;
              // (single runs instead of pieces towards the end of the pool, so that the launch does not end with waves on
              // their last four: measured, 2841 -> 2859 us)
              const uint32_t r = firstPooled + c + PT_POOL_COUNTERS * atomicAdd(&wf.runPool[c * PT_POOL_STRIDE], (uint32_t)PT_POOL_CHUNK);
              if (r < runs) { run = r; got = min((uint32_t)PT_POOL_CHUNK, (runs - r + PT_POOL_COUNTERS - 1) / PT_POOL_COUNTERS); break; }
              atomicMax(&sUsedUp, usedUp + 1u);
            }
          }
          atomicSub(&sBusy, PT_POOL_CHUNK - got);
        }
      }
      got = (uint32_t)__builtin_amdgcn_readfirstlane(got);
      if (!got) break;
      run = (uint32_t)__builtin_amdgcn_readfirstlane(run);
      pending = got - 1; pendingRun = run + PT_POOL_COUNTERS;
    }
    // a run of a column of constant sky: its 64 samples are counted and that is all (two runs in three of the chest
    // frame; the test reads one word through the scalar cache, where the run would cost 50 vector instructions and
    // three loads per lane to find the same out)
    if (masks.activeCols && pm.columns) {
      const uint32_t col = run - fdiv(pm.divColumns, run) * pm.columns;
      if (!((masks.activeCols[col >> 5] >> (col & 31u)) & 1u)) {
        mine++; culled++; nConst++;
        if ((threadIdx.x & 63u) == 0) atomicSub(&sBusy, 1u);
        continue;
      }
    }
    const uint32_t i = (run << 6) + (threadIdx.x & 63u);
    const bool valid = i < nslots;
    if (valid) mine++;
    uint32_t s, gp = slotPixel(cfg, pm, valid ? i : 0u, s);
    // the pixel's mask words and its sample count are asked for together: one memory round trip in front of a run, not
    // one for the constant-sky bit and a second for the rest
    const uint32_t skyWord = masks.constantSky ? masks.constantSky[gp >> 5] : 0u;
    const uint32_t coveredWord = masks.covered ? masks.covered[gp >> 5] : 0u;
    const float accW = acc[gp].w;
    const bool constant = valid && ((skyWord >> (gp & 31u)) & 1u);
    const bool live = valid && !constant;
    bool active = false;
    V3 hitPos = v3(0, 0, 0), normal = v3(0, 0, 1), color = v3(0, 0, 0);
    uint32_t matTag = 0, sampleIdx = 0;
    Rng g; g.s0 = 1; g.s1 = 0;
    if (constant) { culled++; nConst++; }
    if (live) {
      PrimaryRay pr = primaryRayFrom(cfg, gp, s, accW, true);
      bool enter;
      if (masks.covered) enter = (coveredWord >> (gp & 31u)) & 1u;
      else {
        RayInv ri = makeRayInv(cfg.cam.pos, pr.d);
        float tn;
        enter = slab(ri, box.mn[0], box.mn[1], box.mn[2], box.mx[0], box.mx[1], box.mx[2], PT_FLT_MAX, tn);
      }
      Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0;
      if (enter) { uint32_t cN = 0, cT = 0; active = traverse<false, true, false, PT_FUSE_OCTANTS>(sc, cfg.cam.pos, pr.d, cfg.eps, h, cN, cT); }
      else culled++;
      if (active) {
        hitPos = cfg.cam.pos + pr.d * h.t;
        vertexTables(tb, h.tri, h.u, h.v, normal, color, matTag);
        g = pr.g; sampleIdx = pr.sampleIdx; hits++;
      } else {
        wf.Lacc[i] = f4(sampleSky(pr.d), 0.0f);                                     // rendering.d:480
      }
    }
    // (a run none of whose view rays hit -- a tile of sky -- has nothing to shade: two runs in three of the chest frame)
    if (__any(active)) shadeVertex<true, true, true>(cx, tb, active, hitPos, normal, color, matTag, 0u, g, gp, sampleIdx, i, i);
    if ((threadIdx.x & 63u) == 0) atomicSub(&sBusy, 1u);              // after the run's appends
  }
  flushTraced(cx, true);
  for (int off = 32; off > 0; off >>= 1) {
    culled += __shfl_down(culled, off); nConst += __shfl_down(nConst, off); hits += __shfl_down(hits, off); mine += __shfl_down(mine, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (culled) statAdd(wf.stats, STAT_RAYS_CULLED, (unsigned long long)culled);
    if (nConst) statAdd(wf.stats, STAT_CONSTANT_SKY, (unsigned long long)nConst);
    if (hits) {
      statAdd(wf.stats, STAT_HITS, (unsigned long long)hits); statAdd(wf.stats, STAT_HITS_PRIMARY, (unsigned long long)hits);
    }
    if (mine > culled) statAdd(wf.stats, STAT_FUSED_PRIMARY, (unsigned long long)(mine - culled));     // view rays traced here
  }
  __syncthreads();
  if (threadIdx.x == 0 && sCnt[0]) statAdd(wf.stats, STAT_WHOLE_CHAIN_HITS, (unsigned long long)sCnt[0]);
  if (threadIdx.x < segsPer) {
    const uint32_t n = sCnt[0];
    wf.rayOut.count[T + threadIdx.x] = (n + segsPer - 1u - threadIdx.x) >> cx.segBits;      // hit idx went to segment idx % segsPer
  }
}

// Mode 0 on the same scenes, frames large enough to fill the chip with pixel tiles: a whole PIXEL per lane.  A wave
// takes a column of 64 pixels (a tile) and walks their samples in order -- raygen, view ray, shading with every ray
// traced in place, exactly the code of k_sample_traced -- adding each sample to the pixels' sums in registers as
// k_accumulate would (emissive * color + L for a hit, sampleSky for a miss, the sky constant for a constant-sky
// pixel; rendering.d:477-482), in sample order, so the sums are the same bits.  No radiance record is written or
// read back (32 B of HBM traffic per sample), no accumulate launch, and no state per sample: the batch is the whole
// window whatever its size.  Columns are dealt out one at a time (below) -- a column with all its samples is a
// twentieth of a wave's share of a 4K frame, and pieces of four columns cost 20 % (cornell 4K 64 spp: 32.2 ms against
// 26.8; with the sums in LDS 26.5; the recorded path 28.7).  The pool counters (wf.runPool) are zero at launch; the
// host clears them behind the launch.
#ifndef PT_PIXEL_POOL_FROM
#define PT_PIXEL_POOL_FROM 32      // columns per workgroup from which a launch pools them
#endif
__global__ void __launch_bounds__(PT_SAMPLE_THREADS, PT_SAMPLE_WAVES) k_pixel_traced(SceneBox box, ViewMasks masks, DevScene sc, DevCfg cfg, PixMap pm,
                                                                  float4* __restrict__ acc, Wavefront wf) {
  __shared__ uint32_t sCnt[1 + PT_MAX_KINDS];
  if (threadIdx.x <= PT_MAX_KINDS) sCnt[threadIdx.x] = 0;
  stageScene<true>(sc);
  const ShadeTables<true> tb{sc, 2 * sc.numNodes + 3 * sc.numTris};
  tb.stage();
  __syncthreads();
  ShadeCtx cx{sc, cfg, wf, sCnt, 0u, 1u, 1u};
  // the pixels' sums wait in LDS while a sample is worked out (four registers less across the sample's code)
  __shared__ float4 sSum[PT_SAMPLE_THREADS];
  const uint32_t columns = (pm.npix + 63u) >> 6, lane = threadIdx.x & 63u;
  uint32_t culled = 0, nConst = 0, hits = 0, mine = 0;
  // Who renders which column.  A launch with few columns per workgroup deals column c to workgroup c % gridDim.x, and a
  // workgroup's columns to whichever of its waves is free (a counter in LDS): no global atomic -- in a 320 x 240 pass
  // the waves' takes from the pool, and the eight probes with which every one of them finds it empty, cost as much as
  // the pass itself (0.133 ms against 0.076).  A larger launch pools all its columns: the waves of all workgroups take
  // them one at a time from the PT_POOL_COUNTERS counters (each deals out every PT_POOL_COUNTERS-th column), and a
  // workgroup remembers the counters it has found used up.
  __shared__ uint32_t sNextCol, sUsedUp;
  if (threadIdx.x == 0) { sNextCol = 0; sUsedUp = 0; }
  __syncthreads();
  const uint32_t G = gridDim.x, share = (columns + G - 1) / G;
  const bool pooled = wf.runPool && share >= PT_PIXEL_POOL_FROM;
  for (;;) {
    uint32_t col = PT_NONE;
    if (lane == 0) {
      if (!pooled) {
        const uint32_t k = atomicAdd(&sNextCol, 1u);
        if (k < share && blockIdx.x + G * k < columns) col = blockIdx.x + G * k;
      } else {
        for (;;) {
          const uint32_t usedUp = atomicAdd(&sUsedUp, 0u);
          if (usedUp >= PT_POOL_COUNTERS) break;
          const uint32_t c = (blockIdx.x + usedUp) % PT_POOL_COUNTERS;
          const uint32_t r = c + PT_POOL_COUNTERS * atomicAdd(&wf.runPool[c * PT_POOL_STRIDE], 1u);
          if (r < columns) { col = r; break; }
          atomicMax(&sUsedUp, usedUp + 1u);
        }
      }
    }
    col = (uint32_t)__builtin_amdgcn_readfirstlane(col);
    if (col == PT_NONE) break;
    const uint32_t lp = (col << 6) + lane;
    const bool valid = lp < pm.npix;
    const uint32_t gp = mapPixel(pm, cfg.W, valid ? lp : 0u);
    const bool constant = valid && masks.constantSky && maskBit(masks.constantSky, gp);
    const bool enterMask = valid && masks.covered && maskBit(masks.covered, gp);
    float4 a = acc[gp];
    const float accW = a.w;                                   // Pixel.n before this call: numbers the samples
    if (valid) mine += pm.k;
    if (constant) { culled += pm.k; nConst += pm.k; }
    const bool live = valid && !constant;
    if (!__any(live)) {
      // every sample of these pixels is the sky constant (rendering.d:338)
      if (constant) {
        const V3 c = v3(0.682f, 0.977f, 1.0f);
        for (uint32_t s = 0; s < pm.k; s++) { a.x += c.x; a.y += c.y; a.z += c.z; a.w += 1.0f; }
        acc[gp] = a;
      }
      continue;
    }
    sSum[threadIdx.x] = a;
    for (uint32_t s = 0; s < pm.k; s++) {
      bool active = false;
      V3 hitPos = v3(0, 0, 0), normal = v3(0, 0, 1), color = v3(0, 0, 0);
      uint32_t matTag = 0, sampleIdx = 0;
      Rng g; g.s0 = 1; g.s1 = 0;
      float4 rec = make_float4(0.682f, 0.977f, 1.0f, 0.0f);       // what accumulate adds for a constant-sky pixel
      if (live) {
        PrimaryRay pr = primaryRayFrom(cfg, gp, s, accW, true);
        bool enter;
        if (masks.covered) enter = enterMask;
        else {
          RayInv ri = makeRayInv(cfg.cam.pos, pr.d);
          float tn;
          enter = slab(ri, box.mn[0], box.mn[1], box.mn[2], box.mx[0], box.mx[1], box.mx[2], PT_FLT_MAX, tn);
        }
        Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0;
        if (enter) { uint32_t cN = 0, cT = 0; active = traverse<false, true, false, PT_FUSE_OCTANTS>(sc, cfg.cam.pos, pr.d, cfg.eps, h, cN, cT); }
        else culled++;
        if (active) {
          hitPos = cfg.cam.pos + pr.d * h.t;
          vertexTables(tb, h.tri, h.u, h.v, normal, color, matTag);
          g = pr.g; sampleIdx = pr.sampleIdx; hits++;
        } else {
          rec = f4(sampleSky(pr.d), 0.0f);                                              // rendering.d:480
        }
      }
      if (__any(active)) shadeVertex<true, true, true>(cx, tb, active, hitPos, normal, color, matTag, 0u, g, gp, sampleIdx, 0u, 0u, &rec);
      if (valid) {
        // k_accumulate's addSample
        V3 add = xyz(rec);
        const uint32_t tag = f2u(rec.w);
        if (tag) { const float4 m = tb.mat(tag - 1u); add = m.x * v3(m.y, m.z, m.w) + add; }
        float4 b = sSum[threadIdx.x];
        b.x += add.x; b.y += add.y; b.z += add.z; b.w += 1.0f;
        sSum[threadIdx.x] = b;
      }
    }
    if (valid) acc[gp] = sSum[threadIdx.x];
  }
  flushTraced(cx, true);
  for (int off = 32; off > 0; off >>= 1) {
    culled += __shfl_down(culled, off); nConst += __shfl_down(nConst, off); hits += __shfl_down(hits, off); mine += __shfl_down(mine, off);
  }
  if (lane == 0) {
    if (culled) statAdd(wf.stats, STAT_RAYS_CULLED, (unsigned long long)culled);
    if (nConst) statAdd(wf.stats, STAT_CONSTANT_SKY, (unsigned long long)nConst);
    if (hits) {
      statAdd(wf.stats, STAT_HITS, (unsigned long long)hits); statAdd(wf.stats, STAT_HITS_PRIMARY, (unsigned long long)hits);
    }
    if (mine > culled) statAdd(wf.stats, STAT_FUSED_PRIMARY, (unsigned long long)(mine - culled));     // view rays traced here
  }
}

// View-ray hits of a scene whose vertices trace nothing but a sun and a sky ray (mode 0, sky, no light group), when
// the scene is too large for the shading kernel to hold: the shading of the hit (the body of k_shade<PRIMARY>:
// generator, normal, material, the two directions) moves into the traversal kernel that traces those rays, which
// reads the hit queue instead of a queue of sun/sky entries.  No shade launch, no 64-byte entry written and read
// back per vertex; L = 0 + sun + sky is stored once, with the tag.  Same operations per vertex as k_shade followed by
// k_occluded (direct).
template <bool LDSS>
__global__ void PT_TRAVERSAL_BOUNDS k_shade_occluded(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc, Wavefront wf) {
  __shared__ __align__(16) WorkPool pool;
  initPool(&pool);
  stageScene<LDSS>(sc);
  __syncthreads();
  const ShadeTables<false> tb{sc, 0};
  const QMeta q = wf.hitQ;
  const uint32_t total = q.prefix[PT_SEGS], segCap = wf.capHit / PT_SEGS;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    statAdd(wf.stats, STAT_HITS, (unsigned long long)total); statAdd(wf.stats, STAT_HITS_PRIMARY, (unsigned long long)total);
  }
  uint32_t nRays = 0, nEnv = 0, cN = 0, cT = 0;
  SegCursor cur; cur.seg = 0; cur.lo = 1; cur.hi = 0;
  Chunk ck;
  while (nextChunk(&pool, wf.taskCounters + 1 * PT_TASK_PARTS, q, total, 0, ck)) {
    uint32_t j = ck.begin + (threadIdx.x & 63u);
    if (j < ck.end) {
      uint32_t e = physIndexHint(q, j, ck.seg, segCap, cur);
      float4 ha = wf.hitA[e], hb = wf.hitB[e];
      const V3 hitPos = xyz(ha); const uint32_t laneSlot = f2u(hb.z);
      uint32_t s = fdiv(pm.divNpix, laneSlot), lp = laneSlot - s * pm.npix;
      uint32_t gp = mapPixel(pm, cfg.W, lp);
      uint32_t sampleIdx = cfg.sampleBase + ((uint32_t)acc[gp].w + s) * cfg.sampleStride;
      Rng g; g.seed(cfg.seed, gp, sampleIdx, 0);
      g.next(); g.next();                                              // the x and y jitter of getViewRay
      V3 normal, color; uint32_t matTag;
      vertexTables(tb, f2u(ha.w), hb.x, hb.y, normal, color, matTag);
      const V3 Tw = v3(1, 1, 1) * (PT_BRDF * color);
      const EnvSample es = sampleEnv(cfg, g, normal);
      float4 L = make_float4(0.0f, 0.0f, 0.0f, u2f(matTag));
      if (es.eSun || es.eSky) {
        const V3 org = hitPos + normal * 0.1f;
        bool vSun = false, vSky = false;
        Hit h;
        if (es.eSun) vSun = !traverse<true, LDSS, false, true>(sc, org, es.dSun, cfg.eps, h, cN, cT);
        if (es.eSky) vSky = !traverse<true, LDSS, false>(sc, org, es.dSky, cfg.eps, h, cN, cT);
        nRays += (es.eSun ? 1u : 0u) + (es.eSky ? 1u : 0u); nEnv++;
        if (vSun) { V3 c = Tw * cfg.sunRadiance * es.sSun; L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z; }
        if (vSky) { V3 c = Tw * cfg.skyRadiance * es.sSky; L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z; }
      }
      wf.Lacc[laneSlot] = L;
    }
  }
  for (int off = 32; off > 0; off >>= 1) { nRays += __shfl_down(nRays, off); nEnv += __shfl_down(nEnv, off); }
  if ((threadIdx.x & 63) == 0) {
    if (nRays) statAdd(wf.stats, STAT_RAYS_OCCL, (unsigned long long)nRays);
    if (nEnv) statAdd(wf.stats, STAT_ENV_ENTRIES, (unsigned long long)nEnv);
  }
}

// ------------------------------------------------------------------------------------------------ accumulate
// pixel.sum += L; pixel.n += 1 for each of the batch's k samples of a pixel, in sample order
// (rendering.d:477-482; the (sum / N) of the indirect chains as in the block at rendering.d:474-477).
// A slot's radiance record is (L, tag): for a view-ray hit L is computeLdirect's sum, gathered from 0 in the
// reference's order (light groups, sun, sky), and tag = material index + 1, so that emissive * color is added in
// front of it as the source does (`emissive * color + computeLdirect(...)`); for a miss L = sampleSky and tag = 0.
__global__ void __launch_bounds__(256) k_accumulate(ViewMasks masks, DevCfg cfg, PixMap pm, float4* __restrict__ acc, Wavefront wf,
                                                    const float4* __restrict__ mats) {
  uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < PT_POOL_COUNTERS && wf.runPool) wf.runPool[threadIdx.x * PT_POOL_STRIDE] = 0;   // the batch is over
  if (lp >= pm.npix) return;
  uint32_t gp = mapPixel(pm, cfg.W, lp);
  float4 a = acc[gp];
  if (masks.constantSky && maskBit(masks.constantSky, gp)) {
    // every sample of this pixel is sampleSky away from its sun disc (rendering.d:338); raygen wrote nothing
    const V3 c = v3(0.682f, 0.977f, 1.0f);
    for (uint32_t s = 0; s < pm.k; s++) { a.x += c.x; a.y += c.y; a.z += c.z; a.w += 1.0f; }
    acc[gp] = a;
    return;
  }
  // the records of four samples are requested before the first is looked at: the additions below stay in sample
  // order, but a lane's loads no longer wait for each other (the launch reads 16 B per sample and is bound by HBM)
  auto addSample = [&](float4 L, uint32_t slot) {
    V3 add = xyz(L);
    const uint32_t tagWord = f2u(L.w);
    if (tagWord) {
      // wf.chainMask: the low 24 bits are the tag, the bits above say which chains' first rays hit anything -- the
      // record of a chain that did not was never added to, and adding its zero would not change the sum
      const uint32_t tag = wf.chainMask ? tagWord & 0xFFFFFFu : tagWord, chains = wf.chainMask ? tagWord >> 24 : 0xFFFFFFFFu;
      float4 m = mats[tag - 1u];
      add = m.x * v3(m.y, m.z, m.w) + add;
      if (cfg.mode == 1) {
        // computeLindirect of every chain, folded from the chain's end as the recursion returns (rendering.d:391, 372):
        // Lrefl(v) = Ldirect(v) + color(v) * Lrefl(next vertex), or + 0 where the chain ends; the chain is worth
        // color(view-ray hit) * Lrefl(its first vertex).  Vertex r of chain c keeps (Ldirect, material + 1 | continues << 31)
        // in record slot + (c + 1) * capSlots + r * chainStride.
        const V3 color0 = v3(m.y, m.z, m.w);
        V3 sum = v3(0, 0, 0);
        for (uint32_t c = 0; c < cfg.numIndirect; c++) {
          const uint32_t rec0 = (c + 1u) * wf.capSlots + slot;
          // how many vertices the chain has: told by the chain mask and the `continues` bits where shading traced its own
          // rays, else by the records' tag words (shading marked the record after the last vertex absent)
          uint32_t n = 0;
          if (wf.chainMask) {
            bool more = (chains >> (c & 31u)) & 1u;
            while (more && n < wf.recRounds) { more = f2u(wf.Lacc[rec0 + n * wf.chainStride].w) >> 31; n++; }
          } else {
            while (n < wf.recRounds && (f2u(wf.Lacc[rec0 + n * wf.chainStride].w) & 0x7FFFFFFFu) != 0u) n++;
          }
          if (n == 0) continue;                                              // `sum += 0`
          V3 R = v3(0, 0, 0);
          for (uint32_t r = n; r-- > 0;) {
            const float4 Lr = wf.Lacc[rec0 + r * wf.chainStride];
            const float4 mr = mats[(f2u(Lr.w) & 0x7FFFFFFFu) - 1u];
            R = xyz(Lr) + (r + 1 == n ? v3(0, 0, 0) : v3(mr.y, mr.z, mr.w) * R);
          }
          sum = sum + color0 * R;
        }
        add = add + sum / (float)cfg.numIndirect;
      }
    }
    a.x += add.x; a.y += add.y; a.z += add.z; a.w += 1.0f;
  };
  uint32_t s = 0, slot = lp;
  for (; s + 4 <= pm.k; s += 4, slot += 4 * pm.npix) {
    const float4 L0 = wf.Lacc[slot], L1 = wf.Lacc[slot + pm.npix], L2 = wf.Lacc[slot + 2 * pm.npix], L3 = wf.Lacc[slot + 3 * pm.npix];
    addSample(L0, slot); addSample(L1, slot + pm.npix); addSample(L2, slot + 2 * pm.npix); addSample(L3, slot + 3 * pm.npix);
  }
  for (; s < pm.k; s++, slot += pm.npix) addSample(wf.Lacc[slot], slot);
  acc[gp] = a;
}

// ------------------------------------------------------------------------------------------------ operators
__global__ void k_view_rays(DevCfg cfg, uint32_t pixelOffset, uint32_t count, float* __restrict__ rays6) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t p = pixelOffset + i, row = fdiv(cfg.divW, p), col = p - row * cfg.W;
  float x, y; viewRayXY(col, row, cfg.W, cfg.H, nullptr, x, y);
  V3 d = screenRayDir(cfg.cam, x, y);
  float* r = rays6 + 6 * (size_t)i;
  r[0] = cfg.cam.pos.x; r[1] = cfg.cam.pos.y; r[2] = cfg.cam.pos.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
}

template <bool LDSS>
__global__ void PT_TRAVERSAL_BOUNDS k_trace_rays(DevScene sc, float eps, const float* __restrict__ rays6, uint32_t count,
                                                     int32_t* triId, float* t, float* u, float* v, float* normal3) {
  stageScene<LDSS>(sc);
  uint32_t cN = 0, cT = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float* r = rays6 + 6 * (size_t)i;
    Hit h;
    bool ok = traverse<false, LDSS, false, false>(sc, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), eps, h, cN, cT);
    if (triId) triId[i] = ok ? (int32_t)h.tri : -1;
    if (t) t[i] = h.t;
    if (u) u[i] = h.u;
    if (v) v[i] = h.v;
    if (normal3) {
      V3 nn = v3(0, 0, 0);
      if (ok) { float4 q0 = sc.nrm[3 * h.tri], q1 = sc.nrm[3 * h.tri + 1], q2 = sc.nrm[3 * h.tri + 2];
                nn = interpNormal(xyz(q0), xyz(q1), xyz(q2), h.u, h.v); }
      normal3[3 * (size_t)i] = nn.x; normal3[3 * (size_t)i + 1] = nn.y; normal3[3 * (size_t)i + 2] = nn.z;
    }
  }
}

// Scene.trace, texcoord variant (scene.d:745-796): closest hit on either side of the surface unless the caller
// ignores back faces; reports TraceResult (scene.d:55-60: 0 FrontFaceHit, 1 BackFaceHit, 2 NoHit), the distance
// and the texture coordinates interpolated with scene.d:68-105.  tex6 holds three (s, t) pairs per triangle in
// input order.
template <bool LDSS, bool BOTH>
__global__ void PT_TRAVERSAL_BOUNDS k_trace_tex(DevScene sc, float eps, const float* __restrict__ tex6,
                                                    const float* __restrict__ rays6, uint32_t count, int32_t* result,
                                                    int32_t* triId, float* t, float* tc2) {
  stageScene<LDSS>(sc);
  uint32_t cN = 0, cT = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float* r = rays6 + 6 * (size_t)i;
    Hit h;
    bool ok = traverse<false, LDSS, false, false, BOTH>(sc, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), eps, h, cN, cT);
    result[i] = ok ? (h.front ? 0 : 1) : 2;
    if (triId) triId[i] = ok ? (int32_t)h.tri : -1;
    if (t) t[i] = h.t;
    if (tc2) {
      float s = 0.0f, tt = 0.0f;
      if (ok) {
        const float* q = tex6 + 6 * (size_t)h.tri;
        s = interpolateComponent(h.u, h.v, q[0], q[2], q[4]);
        tt = interpolateComponent(h.u, h.v, q[1], q[3], q[5]);
      }
      tc2[2 * (size_t)i] = s; tc2[2 * (size_t)i + 1] = tt;
    }
  }
}

template <bool LDSS, bool COUNT>
__global__ void PT_TRAVERSAL_BOUNDS k_hits_nothing(DevScene sc, float eps, const float* __restrict__ rays6, uint32_t count,
                                                       uint8_t* out, unsigned long long* stats) {
  stageScene<LDSS>(sc);
  uint32_t cN = 0, cT = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float* r = rays6 + 6 * (size_t)i;
    Hit h;
    out[i] = traverse<true, LDSS, COUNT, false>(sc, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), eps, h, cN, cT) ? 0 : 1;
  }
  flushCounts<COUNT>(stats, STAT_NODES_ANY, cN, cT);
}

// drawScreen's scalar path, main.d:82-95
__global__ void k_tonemap(const float4* __restrict__ acc, uint32_t pixelOffset, uint32_t count, uint8_t* __restrict__ rgb) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float4 a = acc[pixelOffset + i];
  float c[3] = {a.x / a.w, a.y / a.w, a.z / a.w};                     // pixel.color = pixel.sum / pixel.n
  for (int k = 0; k < 3; k++) rgb[3 * (size_t)i + k] = tonemapChannel(a.w > 0.0f ? c[k] : 0.0f);
}

// Pixel[] records (rendering.d:24-30: color, n, sum; 7 floats) from the (sum, n) accumulator, one thread per float so
// that the 28-byte records are written with dense 4-byte stores.  color = sum / n (rendering.d:483), 0 while n == 0.
__global__ void k_pack_pixels(const float4* __restrict__ acc, uint32_t pixelOffset, uint32_t count, float* __restrict__ out7) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count * 7u) return;
  uint32_t p = j / 7u, c = j - p * 7u;
  float4 a = acc[pixelOffset + p];
  uint32_t k = c < 3 ? c : c - 4;                                  // channel (unused for c == 3)
  float s = k == 0 ? a.x : (k == 1 ? a.y : a.z);
  float v;
  if (c < 3) v = a.w > 0.0f ? s / a.w : 0.0f;
  else if (c == 3) v = a.w;
  else v = s;
  out7[j] = v;
}
// The inverse for the project file (main.d:227-234): (sum, n) from Pixel[] records.
__global__ void k_unpack_pixels(const float* __restrict__ in7, uint32_t pixelOffset, uint32_t count, float4* __restrict__ acc) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= count) return;
  const float* r = in7 + 7 * (size_t)p;
  acc[pixelOffset + p] = make_float4(r[4], r[5], r[6], r[3]);
}

// angleToDirection (rendering.d:316-330) as the shading kernels evaluate it; a zero normal selects
// angleToLocalDirection (rendering.d:309-314).  in5 = (phi, psi, normal) per row.
__global__ void k_angle_directions(const float* __restrict__ in5, uint32_t n, float* __restrict__ out3) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = in5 + 5 * (size_t)i;
  V3 nn = v3(p[2], p[3], p[4]);
  V3 d = (nn.x == 0.0f && nn.y == 0.0f && nn.z == 0.0f) ? angleToLocalDirection(p[0], p[1]) : angleToDirection(p[0], p[1], nn);
  out3[3 * (size_t)i] = d.x; out3[3 * (size_t)i + 1] = d.y; out3[3 * (size_t)i + 2] = d.z;
}

__global__ void k_rng_stream(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, uint32_t* out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    Rng g; g.seed(seed, pixel, sample, stream);
    for (uint32_t i = 0; i < n; i++) out[i] = g.next();
  }
}

// ------------------------------------------------------------------------------------------------ launch geometry
struct Geo { dim3 grid, block; size_t lds; };

// Traversal kernels are persistent: enough workgroups to fill the chip.  With the scene in LDS the workgroup
// size grows with the scene so that one copy serves many waves.
Geo traversalGeo(const LaunchCfg& lc, const DevScene& sc, uint64_t maxItems) {
  Geo g; g.lds = sc.ldsBytes;
  uint32_t bs, perCU;
  if (sc.ldsBytes == 0) {
    bs = PT_GLOBAL_THREADS; perCU = 4;
#ifdef PT_DIAG_ENV
    if (const char* e = getenv("PT_DIAG_PERCU")) perCU = (uint32_t)atoi(e);      // diagnostic builds: occupancy of the global-memory kernels
#endif
  }
  else { perCU = std::max(1u, ldsWorkgroupsPerCU(sc.ldsBytes)); bs = perCU == 8 ? 256 : perCU == 4 ? 512 : 1024; }
  uint64_t want = (maxItems + bs - 1) / bs;
  uint64_t cap = (uint64_t)lc.numCUs * perCU;
  if (lc.maxGroups) cap = std::min<uint64_t>(cap, lc.maxGroups);
  g.block = dim3(bs); g.grid = dim3((unsigned)std::max<uint64_t>(1, std::min(want, cap)));
  return g;
}

// Dynamic LDS above 64 KB has to be allowed per kernel and device, once, before the first launch.
void allowLds(const LaunchCfg& lc, const void* kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return;
  static std::mutex mu; static std::set<std::pair<int, const void*>> done;
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert({lc.device, kernel}).second)
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);   // + static LDS <= 160 KB
}

}  // namespace

#define PT_DISPATCH_LC(KERNEL, ...)                                                              \
  do {                                                                                           \
    bool l_ = sc.ldsBytes != 0, c_ = lc.countVisits;                                             \
    if (l_ && c_) { allowLds(lc, (const void*)KERNEL<true, true>, g.lds); hipLaunchKernelGGL((KERNEL<true, true>), g.grid, g.block, g.lds, lc.stream, __VA_ARGS__); } \
    else if (l_) { allowLds(lc, (const void*)KERNEL<true, false>, g.lds); hipLaunchKernelGGL((KERNEL<true, false>), g.grid, g.block, g.lds, lc.stream, __VA_ARGS__); } \
    else if (c_) hipLaunchKernelGGL((KERNEL<false, true>), g.grid, g.block, 0, lc.stream, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<false, false>), g.grid, g.block, 0, lc.stream, __VA_ARGS__);          \
  } while (0)

void launchRaygen(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevCfg& cfg, const PixMap& pm,
                  const float4* acc, uint32_t nslots, const Wavefront& wf) {
  hipLaunchKernelGGL(k_raygen, dim3(PT_SEGS), dim3(PT_RAYGEN_THREADS), 0, lc.stream, box, masks, cfg, pm, acc, nslots, wf);
}

void launchScan(const LaunchCfg& lc, const QMeta* metas, int n, uint32_t* taskCounters) {
  hipLaunchKernelGGL(k_scan, dim3(n), dim3(PT_SEGS), 0, lc.stream, metas, taskCounters);
}

void launchExtendPrimary(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                         uint32_t nslots, int jitter, float4* hit, unsigned long long* stats) {
  Geo g = traversalGeo(lc, sc, nslots);
  PT_DISPATCH_LC(k_extend_primary, sc, cfg, pm, acc, nslots, jitter, hit, stats);
}

template <bool PRIMARY>
static void dispatchExtend(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                           const Wavefront& wf) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  g.grid.x = std::min<unsigned>(g.grid.x, PT_SEGS);      // every workgroup owns at least one output segment
  bool l_ = sc.ldsBytes != 0, c_ = lc.countVisits;
  if (l_ && c_) { allowLds(lc, (const void*)k_extend<PRIMARY, true, true>, g.lds); hipLaunchKernelGGL((k_extend<PRIMARY, true, true>), g.grid, g.block, g.lds, lc.stream, sc, cfg, pm, acc, wf); }
  else if (l_) { allowLds(lc, (const void*)k_extend<PRIMARY, true, false>, g.lds); hipLaunchKernelGGL((k_extend<PRIMARY, true, false>), g.grid, g.block, g.lds, lc.stream, sc, cfg, pm, acc, wf); }
  else if (c_) hipLaunchKernelGGL((k_extend<PRIMARY, false, true>), g.grid, g.block, 0, lc.stream, sc, cfg, pm, acc, wf);
  else if (PT_REFILL_BELOW) {
    if (PT_EXTEND_BY_SEGMENT) (void)hipMemsetAsync(wf.hitQ.count, 0, PT_SEGS * sizeof(uint32_t), lc.stream);   // it adds to them
    hipLaunchKernelGGL((k_extend_refill<PRIMARY>), g.grid, g.block, 0, lc.stream, sc, cfg, pm, acc, wf);
  }
  else hipLaunchKernelGGL((k_extend<PRIMARY, false, false>), g.grid, g.block, 0, lc.stream, sc, cfg, pm, acc, wf);
}

void launchExtend(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                  const Wavefront& wf, bool primary) {
  if (primary) dispatchExtend<true>(lc, sc, cfg, pm, acc, wf); else dispatchExtend<false>(lc, sc, cfg, pm, acc, wf);
}

void launchConnect(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind, float area) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  size_t e = (size_t)kind * wf.capQueue;
  if (!sc.ldsBytes && !lc.countVisits && PT_REFILL_CONNECT) {
    hipLaunchKernelGGL(k_connect_refill, g.grid, g.block, 0, lc.stream, sc, cfg.eps, wf.sh[kind], wf.capQueue / PT_SEGS,
                       (const float4*)(wf.shO + e), (const float4*)(wf.shD + e), (const float4*)(wf.shW + e),
                       wf.taskCounters + (1 + kind) * PT_TASK_PARTS, area, wf.Lacc, wf.stats);
    return;
  }
  PT_DISPATCH_LC(k_connect, sc, cfg.eps, wf.sh[kind], wf.capQueue / PT_SEGS, (const float4*)(wf.shO + e), (const float4*)(wf.shD + e),
                 (const float4*)(wf.shW + e), wf.taskCounters + (1 + kind) * PT_TASK_PARTS, area, wf.Lacc, wf.stats);
}

void launchOccluded(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind, bool primary) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  size_t e = (size_t)kind * wf.capQueue;
  if (!sc.ldsBytes && !lc.countVisits && PT_REFILL_ANY) {
    hipLaunchKernelGGL(k_occluded_refill, g.grid, g.block, 0, lc.stream, sc, cfg.eps, cfg.sunRadiance, cfg.skyRadiance,
                       envDirect(cfg, sc, primary) ? 1 : 0, wf.sh[kind], wf.capQueue / PT_SEGS, (const float4*)(wf.shO + e),
                       (const float4*)(wf.shD + e), (const float4*)(wf.shW + e), (const float4*)wf.shX,
                       wf.taskCounters + (1 + kind) * PT_TASK_PARTS, wf.Lacc, wf.stats);
    return;
  }
  PT_DISPATCH_LC(k_occluded, sc, cfg.eps, cfg.sunRadiance, cfg.skyRadiance, envDirect(cfg, sc, primary) ? 1 : 0, wf.sh[kind],
                 wf.capQueue / PT_SEGS, (const float4*)(wf.shO + e), (const float4*)(wf.shD + e), (const float4*)(wf.shW + e),
                 (const float4*)wf.shX, wf.taskCounters + (1 + kind) * PT_TASK_PARTS, wf.Lacc, wf.stats);
}

void launchHitOrder(const LaunchCfg& lc, const DevScene& sc, const Wavefront& wf) {
  hipLaunchKernelGGL(k_hit_order, dim3(PT_SEGS), dim3(PT_ORDER_THREADS), 0, lc.stream, sc, wf.hitQ, wf.capHit / PT_SEGS, wf.capHit,
                     (const float4*)wf.hitA, wf.hitOrder, wf.hitOrder + wf.capHit);
}

void launchShade(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                 const Wavefront& wf, bool primary) {
  int numKinds = (cfg.hasLights ? (int)sc.numLightGroups : 0) + (cfg.hasSky ? 1 : 0);
  const uint32_t lds = shadeTableBytes(sc);
  const dim3 grid(PT_SEGS), block(PT_SHADE_THREADS);
  if (wf.fuse) {
    const uint32_t all = lds + (uint32_t)sc.ldsBytes;
    const dim3 fblock(PT_FUSE_THREADS);
    if (primary) hipLaunchKernelGGL((k_shade_traced<true>), grid, fblock, all, lc.stream, sc, cfg, pm, acc, wf, numKinds);
    else hipLaunchKernelGGL((k_shade_traced<false>), grid, fblock, all, lc.stream, sc, cfg, pm, acc, wf, numKinds);
  }
  else if (primary && lds) hipLaunchKernelGGL((k_shade<true, true>), grid, block, lds, lc.stream, sc, cfg, pm, acc, wf, numKinds);
  else if (primary) hipLaunchKernelGGL((k_shade<true, false>), grid, block, 0, lc.stream, sc, cfg, pm, acc, wf, numKinds);
  else if (lds) hipLaunchKernelGGL((k_shade<false, true>), grid, block, lds, lc.stream, sc, cfg, pm, acc, wf, numKinds);
  else hipLaunchKernelGGL((k_shade<false, false>), grid, block, 0, lc.stream, sc, cfg, pm, acc, wf, numKinds);
}

void launchRaygenExtend(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevScene& sc, const DevCfg& cfg,
                        const PixMap& pm, const float4* acc, uint32_t nslots, const Wavefront& wf) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  g.grid.x = std::min<unsigned>(g.grid.x, PT_SEGS);      // every workgroup owns at least one output segment
  if (sc.ldsBytes) {
    allowLds(lc, (const void*)k_raygen_extend<true>, g.lds);
    hipLaunchKernelGGL(k_raygen_extend<true>, g.grid, g.block, g.lds, lc.stream, box, masks, sc, cfg, pm, acc, nslots, wf);
  } else {
    hipLaunchKernelGGL(k_raygen_extend<false>, g.grid, g.block, 0, lc.stream, box, masks, sc, cfg, pm, acc, nslots, wf);
  }
}

void launchShadeOccluded(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc, const Wavefront& wf) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  // scenes in LDS only (renderBatch: a scene in global memory runs k_shade and k_occluded_refill)
  allowLds(lc, (const void*)k_shade_occluded<true>, g.lds);
  hipLaunchKernelGGL(k_shade_occluded<true>, g.grid, g.block, g.lds, lc.stream, sc, cfg, pm, acc, wf);
}

void launchSampleTraced(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevScene& sc, const DevCfg& cfg,
                        const PixMap& pm, const float4* acc, uint32_t nslots, const Wavefront& wf) {
  const uint32_t all = shadeTableBytes(sc) + (uint32_t)sc.ldsBytes;
  hipLaunchKernelGGL(k_sample_traced, dim3(PT_SAMPLE_GROUPS), dim3(PT_SAMPLE_THREADS), all, lc.stream, box, masks, sc, cfg, pm, acc, nslots, wf);
}

void launchPixelTraced(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevScene& sc, const DevCfg& cfg,
                       const PixMap& pm, float4* acc, const Wavefront& wf) {
  const uint32_t all = shadeTableBytes(sc) + (uint32_t)sc.ldsBytes;
  hipLaunchKernelGGL(k_pixel_traced, dim3(PT_SAMPLE_GROUPS), dim3(PT_SAMPLE_THREADS), all, lc.stream, box, masks, sc, cfg, pm, acc, wf);
  if (wf.runPool) (void)hipMemsetAsync(wf.runPool, 0, PT_POOL_WORDS * sizeof(uint32_t), lc.stream);      // zero between launches
}

// One wave per column: does any of its pixels lie outside the constant-sky mask?
__global__ void __launch_bounds__(256) k_active_columns(ViewMasks masks, DevCfg cfg, PixMap pm, uint32_t* __restrict__ bits) {
  const uint32_t col = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (col >= pm.columns) return;
  const uint32_t gp = mapPixel(pm, cfg.W, (col << 6) + lane);
  const bool live = !maskBit(masks.constantSky, gp);
  if (__any(live) && lane == 0) atomicOr(&bits[col >> 5], 1u << (col & 31u));
}
void launchActiveColumns(const LaunchCfg& lc, const ViewMasks& masks, const DevCfg& cfg, const PixMap& pm, uint32_t* bits) {
  (void)hipMemsetAsync(bits, 0, ((size_t)pm.columns + 31) / 32 * 4, lc.stream);
  hipLaunchKernelGGL(k_active_columns, dim3((pm.columns * 64u + 255u) / 256u), dim3(256), 0, lc.stream, masks, cfg, pm, bits);
}

void launchAccumulate(const LaunchCfg& lc, const ViewMasks& masks, const DevCfg& cfg, const PixMap& pm, float4* acc,
                      const Wavefront& wf, const DevScene& sc) {
  hipLaunchKernelGGL(k_accumulate, dim3((pm.npix + 255) / 256), dim3(256), 0, lc.stream, masks, cfg, pm, acc, wf, sc.mats);
}

void launchViewRays(const LaunchCfg& lc, const DevCfg& cfg, uint32_t pixelOffset, uint32_t count, float* rays6) {
  hipLaunchKernelGGL(k_view_rays, dim3((count + 255) / 256), dim3(256), 0, lc.stream, cfg, pixelOffset, count, rays6);
}

void launchTraceRays(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count,
                     int32_t* triId, float* t, float* u, float* v, float* normal3) {
  Geo g = traversalGeo(lc, sc, count);
  if (sc.ldsBytes) { allowLds(lc, (const void*)k_trace_rays<true>, g.lds); hipLaunchKernelGGL(k_trace_rays<true>, g.grid, g.block, g.lds, lc.stream, sc, cfg.eps, rays6, count, triId, t, u, v, normal3); }
  else hipLaunchKernelGGL(k_trace_rays<false>, g.grid, g.block, 0, lc.stream, sc, cfg.eps, rays6, count, triId, t, u, v, normal3);
}

template <bool BOTH>
static void launchTraceTexT(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* tex6, const float* rays6,
                            uint32_t count, int32_t* result, int32_t* triId, float* t, float* tc2) {
  Geo g = traversalGeo(lc, sc, count);
  if (sc.ldsBytes) {
    allowLds(lc, (const void*)k_trace_tex<true, BOTH>, g.lds);
    hipLaunchKernelGGL((k_trace_tex<true, BOTH>), g.grid, g.block, g.lds, lc.stream, sc, cfg.eps, tex6, rays6, count, result, triId, t, tc2);
  } else {
    hipLaunchKernelGGL((k_trace_tex<false, BOTH>), g.grid, g.block, 0, lc.stream, sc, cfg.eps, tex6, rays6, count, result, triId, t, tc2);
  }
}

void launchTraceTex(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* tex6, const float* rays6,
                    uint32_t count, bool ignoreBackfaces, int32_t* result, int32_t* triId, float* t, float* tc2) {
  if (ignoreBackfaces) launchTraceTexT<false>(lc, sc, cfg, tex6, rays6, count, result, triId, t, tc2);
  else launchTraceTexT<true>(lc, sc, cfg, tex6, rays6, count, result, triId, t, tc2);
}

void launchHitsNothing(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count, uint8_t* out,
                       unsigned long long* stats) {
  Geo g = traversalGeo(lc, sc, count);
  PT_DISPATCH_LC(k_hits_nothing, sc, cfg.eps, rays6, count, out, stats);
}

void launchTonemap(const LaunchCfg& lc, const float4* acc, uint32_t pixelOffset, uint32_t count, uint8_t* rgb) {
  hipLaunchKernelGGL(k_tonemap, dim3((count + 255) / 256), dim3(256), 0, lc.stream, acc, pixelOffset, count, rgb);
}

void launchPackPixels(const LaunchCfg& lc, const float4* acc, uint32_t pixelOffset, uint32_t count, float* out7) {
  uint64_t n = (uint64_t)count * 7;
  hipLaunchKernelGGL(k_pack_pixels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lc.stream, acc, pixelOffset, count, out7);
}
void launchUnpackPixels(const LaunchCfg& lc, const float* in7, uint32_t pixelOffset, uint32_t count, float4* acc) {
  hipLaunchKernelGGL(k_unpack_pixels, dim3((count + 255) / 256), dim3(256), 0, lc.stream, in7, pixelOffset, count, acc);
}

// The rows of a rank's bands (bands first, first + stride, ... of bandRows scanlines) as one dense buffer, for the
// exchange between GPUs: pack copies them out of a full-frame accumulator, unpack copies them into one.
__global__ void k_band_rows(float4* __restrict__ frame, float4* __restrict__ packed, uint32_t W, uint32_t H, uint32_t bandRows,
                            uint32_t first, uint32_t stride, uint32_t packedRows, int unpack) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= packedRows * W) return;
  uint32_t pr = i / W, col = i - pr * W;
  uint32_t j = pr / bandRows, r = pr - j * bandRows;                   // j-th band of the rank, row r inside it
  uint32_t row = (first + j * stride) * bandRows + r;
  if (row >= H) return;
  size_t f = (size_t)row * W + col;
  if (unpack) frame[f] = packed[i]; else packed[i] = frame[f];
}
void launchBandRows(const LaunchCfg& lc, float4* frame, float4* packed, uint32_t W, uint32_t H, uint32_t bandRows, uint32_t first,
                    uint32_t stride, uint32_t packedRows, bool unpack) {
  uint64_t n = (uint64_t)packedRows * W;
  if (!n) return;
  hipLaunchKernelGGL(k_band_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lc.stream, frame, packed, W, H, bandRows, first,
                     stride, packedRows, unpack ? 1 : 0);
}

void launchAngleDirections(const LaunchCfg& lc, const float* in5, uint32_t n, float* out3) {
  hipLaunchKernelGGL(k_angle_directions, dim3((n + 255) / 256), dim3(256), 0, lc.stream, in5, n, out3);
}

__global__ void k_init_metas(QMeta* desc, uint32_t* words, uint32_t n) {
  uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  uint32_t* p = words + (size_t)m * PT_QMETA_WORDS;
  QMeta q; q.count = p; q.prefix = p + PT_SEGS; q.start = p + 2 * PT_SEGS + 2;
  desc[m] = q;
}
void launchInitMetas(const LaunchCfg& lc, QMeta* desc, uint32_t* words, uint32_t n) {
  hipLaunchKernelGGL(k_init_metas, dim3((n + 255) / 256), dim3(256), 0, lc.stream, desc, words, n);
}

void launchRngStream(const LaunchCfg& lc, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, uint32_t* out) {
  hipLaunchKernelGGL(k_rng_stream, dim3(1), dim3(64), 0, lc.stream, seed, pixel, sample, stream, n, out);
}

}  // namespace pt
