// Wavefront path tracer kernels for MI355X (gfx950, wave64).
//
// One sample of the reference's computeL (rendering.d:464-484) is split into stages that each run over a
// whole batch of paths; the stages talk through structure-of-arrays queues in HBM:
//
//   extend (primary)   getViewRay + Scene.trace            rendering.d:467-471, scene.d:681-743
//   shade              computeL / computeLdirect / computeLindirect bodies between the trace calls
//                                                           rendering.d:376-462, 472-481
//   connect            Scene.trace of a light-sample ray + the G term   rendering.d:414-425
//   occluded           Scene.hitsNothing of the sun / sky rays          rendering.d:440-443, 455-457
//   extend (queue)     Scene.trace of the indirect rays                 rendering.d:389
//   accumulate         sum += L; n += 1                                  rendering.d:477-483
//
// Live paths are compacted when they are appended to the next queue: a wave64 ballot + mbcnt gives each
// lane its rank, wave totals are combined through LDS and ONE atomic per workgroup reserves the range
// (a single 4-byte word sustains < 10^8 atomics/s, far less than one per wave at these ray rates).
//
// Traversal: stack-based, near child first, over the BVH2 of bvh.h.  When the scene fits, nodes and
// triangles are staged in LDS once per (persistent) workgroup; otherwise they are read from HBM/L2.
// The ray-box test is conservative (boxes are padded at build time, far distance inflated by 2 ulp) and
// the ray-triangle test is evaluated in one fixed order with explicit fma (pt_math.h), so the closest
// hit (t, u, v, triangle id) does not depend on the tree.
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <algorithm>

namespace pt {

extern __shared__ float4 g_lds[];   // [0, 2*numNodes) nodes, then 3*numTris triangle rows

namespace {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ V3 xyz(float4 a) { return v3(a.x, a.y, a.z); }
__device__ __forceinline__ float4 f4(V3 a, float w) { return make_float4(a.x, a.y, a.z, w); }

__device__ __forceinline__ uint32_t mapPixel(const PixMap& pm, uint32_t lp) {
  uint32_t q = pm.lpBase + lp;
  if (pm.bandPix == 0) return pm.offset + q;
  uint32_t j = q / pm.bandPix, r = q - j * pm.bandPix;
  return (pm.bandFirst + j * pm.bandStride) * pm.bandPix + r;
}

template <bool LDSS> __device__ __forceinline__ float4 ldNode(const DevScene& sc, uint32_t i) {
  if constexpr (LDSS) return g_lds[i]; else return sc.nodes[i];
}
template <bool LDSS> __device__ __forceinline__ float4 ldTri(const DevScene& sc, uint32_t i) {
  if constexpr (LDSS) return g_lds[2 * sc.numNodes + i]; else return sc.tris[i];
}

template <bool LDSS> __device__ __forceinline__ void stageScene(const DevScene& sc) {
  if constexpr (LDSS) {
    uint32_t nn = 2 * sc.numNodes, nt = 3 * sc.numTris;
    for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) g_lds[i] = sc.nodes[i];
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) g_lds[nn + i] = sc.tris[i];
    __syncthreads();
  }
}

struct Hit { float t, u, v; uint32_t tri; };

// Closest front-face hit (ANY = false: Scene.trace, scene.d:691-740) or first hit of either face
// (ANY = true: Scene.hitsNothing, scene.d:808-835).  Returns true if something was accepted.
template <bool ANY, bool LDSS, bool COUNT>
__device__ __forceinline__ bool traverse(const DevScene& sc, V3 o, V3 d, float eps, Hit& h, uint32_t& cN, uint32_t& cT) {
  float dx = fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x;
  float dy = fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y;
  float dz = fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z;
  float ix = __builtin_amdgcn_rcpf(dx), iy = __builtin_amdgcn_rcpf(dy), iz = __builtin_amdgcn_rcpf(dz);
  float ox = -o.x * ix, oy = -o.y * iy, oz = -o.z * iz;

  h.t = PT_FLT_MAX; h.u = 0.0f; h.v = 0.0f; h.tri = 0xFFFFFFFFu;
  bool found = false;
  uint32_t stack[64];
  int sp = 0;
  uint32_t ref = sc.rootRef;
  bool live = true;
  while (live) {
    while ((ref >> BVH_COUNT_SHIFT) == 0) {
      uint32_t b = ref * 4;
      float4 a0 = ldNode<LDSS>(sc, b), a1 = ldNode<LDSS>(sc, b + 1);
      float4 b0 = ldNode<LDSS>(sc, b + 2), b1 = ldNode<LDSS>(sc, b + 3);
      if (COUNT) cN += 2;
      float p0x = fmaf(a0.x, ix, ox), p1x = fmaf(a1.x, ix, ox);
      float p0y = fmaf(a0.y, iy, oy), p1y = fmaf(a1.y, iy, oy);
      float p0z = fmaf(a0.z, iz, oz), p1z = fmaf(a1.z, iz, oz);
      float tn0 = fmaxf(fmaxf(fminf(p0x, p1x), fminf(p0y, p1y)), fmaxf(fminf(p0z, p1z), 0.0f));
      float tf0 = fminf(fminf(fmaxf(p0x, p1x), fmaxf(p0y, p1y)), fmaxf(p0z, p1z)) * 1.0000005f;
      float q0x = fmaf(b0.x, ix, ox), q1x = fmaf(b1.x, ix, ox);
      float q0y = fmaf(b0.y, iy, oy), q1y = fmaf(b1.y, iy, oy);
      float q0z = fmaf(b0.z, iz, oz), q1z = fmaf(b1.z, iz, oz);
      float tn1 = fmaxf(fmaxf(fminf(q0x, q1x), fminf(q0y, q1y)), fmaxf(fminf(q0z, q1z), 0.0f));
      float tf1 = fminf(fminf(fmaxf(q0x, q1x), fmaxf(q0y, q1y)), fmaxf(q0z, q1z)) * 1.0000005f;
      bool h0 = tn0 <= fminf(tf0, h.t), h1 = tn1 <= fminf(tf1, h.t);
      uint32_t r0 = f2u(a0.w), r1 = f2u(b0.w);
      if (h0 && h1) {
        bool sw = tn1 < tn0;
        stack[sp++] = sw ? r0 : r1;
        ref = sw ? r1 : r0;
      } else if (h0 || h1) {
        ref = h0 ? r0 : r1;
      } else {
        if (sp == 0) { live = false; break; }
        ref = stack[--sp];
      }
    }
    if (!live) break;
    uint32_t cnt = ref >> BVH_COUNT_SHIFT, first = ref & BVH_INDEX_MASK;
    for (uint32_t i = 0; i < cnt; i++) {
      uint32_t r = (first + i) * 3;
      float4 t0 = ldTri<LDSS>(sc, r), tA = ldTri<LDSS>(sc, r + 1), tB = ldTri<LDSS>(sc, r + 2);
      if (COUNT) cT += 1;
      V3 A = xyz(tA), B = xyz(tB);
      V3 p = cross(d, B);
      float det = dot(A, p);
      if (!(fabsf(det) < eps)) {
        float inv = 1.0f / det;
        V3 s = o - xyz(t0);
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
              if ((t < h.t || (t == h.t && orig < h.tri)) && t >= eps && det < 0.0f) {
                h.t = t; h.u = u; h.v = v; h.tri = orig; found = true;
              }
            }
          }
        }
      }
    }
    if (sp == 0) break;
    ref = stack[--sp];
  }
  return found;
}

template <bool COUNT> __device__ __forceinline__ void flushCounts(unsigned long long* stats, int nodeSlot, uint32_t cN, uint32_t cT) {
  if constexpr (COUNT) {
    for (int off = 32; off > 0; off >>= 1) { cN += __shfl_down(cN, off); cT += __shfl_down(cT, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[nodeSlot], (unsigned long long)cN); atomicAdd(&stats[nodeSlot + 1], (unsigned long long)cT); }
  }
}

// Workgroup-aggregated queue append.  Every thread of the workgroup must call it the same number of times.
// sm: 2 * 20 words of LDS scratch, `phase` alternates 0/1 between consecutive calls.
__device__ __forceinline__ uint32_t blockAppend(bool pred, uint32_t* counter, uint32_t* sm, int& phase) {
  uint32_t* s = sm + phase * 20; phase ^= 1;
  unsigned long long mask = __ballot(pred);
  uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  uint32_t wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) s[wave] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < nw; w++) { uint32_t c = s[w]; s[w] = total; total += c; }
    s[18] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  return s[18] + s[wave] + rank;
}

// ------------------------------------------------------------------------------------------------ extend
// Primary rays: the view ray is generated in registers (no ray queue) and the hit record is the only HBM write.
template <bool LDSS, bool COUNT>
__global__ void __launch_bounds__(1024) k_extend_primary(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc,
                                                         uint32_t nslots, int jitter, float4* __restrict__ hit,
                                                         unsigned long long* stats) {
  stageScene<LDSS>(sc);
  uint32_t cN = 0, cT = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
    uint32_t s = i / pm.npix, lp = i - s * pm.npix;
    uint32_t gp = mapPixel(pm, lp);
    float x, y;
    if (jitter) {
      Rng g; g.seed(cfg.seed, gp, cfg.sampleBase + (uint32_t)acc[gp].w + s, 0);
      viewRayXY(gp, cfg.W, cfg.H, &g, x, y);
    } else viewRayXY(gp, cfg.W, cfg.H, nullptr, x, y);
    V3 d = screenRayDir(cfg.cam, x, y);
    Hit h;
    traverse<false, LDSS, COUNT>(sc, cfg.cam.pos, d, cfg.eps, h, cN, cT);
    hit[i] = make_float4(h.t, h.u, h.v, u2f(h.tri));
  }
  flushCounts<COUNT>(stats, STAT_NODES_CLOSEST, cN, cT);
}

// Queued rays (indirect bounces): closest hit for every entry of ray queue q.
template <bool LDSS, bool COUNT>
__global__ void __launch_bounds__(1024) k_extend_queue(DevScene sc, float eps, const float4* __restrict__ rayO,
                                                       const float4* __restrict__ rayD, const uint32_t* __restrict__ count,
                                                       float4* __restrict__ hit, unsigned long long* stats, int statSlot) {
  stageScene<LDSS>(sc);
  uint32_t n = *count, cN = 0, cT = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[statSlot], (unsigned long long)n);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 ro = rayO[i], rd = rayD[i];
    Hit h;
    traverse<false, LDSS, COUNT>(sc, xyz(ro), xyz(rd), eps, h, cN, cT);
    hit[i] = make_float4(h.t, h.u, h.v, u2f(h.tri));
  }
  flushCounts<COUNT>(stats, STAT_NODES_CLOSEST, cN, cT);
}

// ------------------------------------------------------------------------------------------------ connect
// Light-sample rays: Scene.trace + the body of `if(distanceY > lightDistance - FloatEpsilon)`, rendering.d:414-425.
// At most one entry per lane slot and launch, so the read-modify-write of Lacc needs no atomics and the
// per-lane order of additions (light 0, light 1, ..., sun, sky) is the launch order.
template <bool LDSS, bool COUNT>
__global__ void __launch_bounds__(1024) k_connect(DevScene sc, float eps, const float4* __restrict__ shO,
                                                  const float4* __restrict__ shD, const float4* __restrict__ shW,
                                                  const uint32_t* __restrict__ count, float area, float4* __restrict__ Lacc,
                                                  unsigned long long* stats) {
  stageScene<LDSS>(sc);
  uint32_t n = *count, cN = 0, cT = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[STAT_RAYS_LIGHT], (unsigned long long)n);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 ro = shO[i], rd = shD[i];
    V3 phi = xyz(rd);
    Hit h;
    if (traverse<false, LDSS, COUNT>(sc, xyz(ro), phi, eps, h, cN, cT)) {
      float lightDistance = rd.w;
      if (h.t > lightDistance - eps) {
        float4 w = shW[i];
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
  flushCounts<COUNT>(stats, STAT_NODES_CLOSEST, cN, cT);
}

// Sun / sky rays: Scene.hitsNothing + `L += ...`, rendering.d:440-443 and 455-457.
template <bool LDSS, bool COUNT>
__global__ void __launch_bounds__(1024) k_occluded(DevScene sc, float eps, const float4* __restrict__ shO,
                                                   const float4* __restrict__ shD, const float4* __restrict__ shW,
                                                   const uint32_t* __restrict__ count, float4* __restrict__ Lacc,
                                                   unsigned long long* stats) {
  stageScene<LDSS>(sc);
  uint32_t n = *count, cN = 0, cT = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[STAT_RAYS_OCCL], (unsigned long long)n);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 ro = shO[i], rd = shD[i];
    Hit h;
    if (!traverse<true, LDSS, COUNT>(sc, xyz(ro), xyz(rd), eps, h, cN, cT)) {
      float4 w = shW[i];
      uint32_t slot = f2u(ro.w);
      float4 L = Lacc[slot];
      L.x = L.x + w.x; L.y = L.y + w.y; L.z = L.z + w.z;
      Lacc[slot] = L;
    }
  }
  flushCounts<COUNT>(stats, STAT_NODES_ANY, cN, cT);
}

// ------------------------------------------------------------------------------------------------ shade
struct ShadeCtx {
  const DevScene& sc; const DevCfg& cfg; const Wavefront& wf; uint32_t* sm; int phase;
};

// One computeLindirect head (rendering.d:378-385): Russian roulette, direction, ray; appended to ray queue qOut.
__device__ __forceinline__ void bounce(ShadeCtx& cx, bool active, Rng& g, uint32_t depth, V3 T, V3 color, V3 pos, V3 normal,
                                       uint32_t laneSlot, int qOut) {
  bool alive = false; V3 dir = v3(0, 0, 0);
  if (active) {
    float a = g.uniform(0.0f, 1.0f);
    if (!(a > cx.cfg.rrProb || depth > cx.cfg.maxDepth)) {
      float psi = g.uniform(0.0f, 2.0f * PT_PI_F);
      float phi = g.uniform(0.0f, PT_PI_2_F);
      dir = angleToDirection(phi, psi, normal);
      alive = true;
    }
  }
  uint32_t idx = blockAppend(alive, &cx.wf.cntOut[0], cx.sm, cx.phase);
  if (alive) {
    cx.wf.rayO[qOut][idx] = f4(pos + normal * 0.1f, u2f(laneSlot));
    cx.wf.rayD[qOut][idx] = f4(dir, 0.0f);
    // BRDF * PI * color == color: the factor the caller multiplies computeLrefl with (rendering.d:391)
    cx.wf.thr[laneSlot] = f4(T * color, u2f(depth + 1));
    cx.wf.rng[laneSlot] = make_uint2(g.s0, g.s1);
  }
}

// computeLdirect up to (not including) the trace calls, rendering.d:396-462: emits one shadow ray per kind.
__device__ __forceinline__ void directLight(ShadeCtx& cx, bool active, Rng& g, V3 x, V3 normalX, V3 Tw, uint32_t laneSlot) {
  const DevScene& sc = cx.sc; const DevCfg& cfg = cx.cfg; const Wavefront& wf = cx.wf;
  float eps = cfg.eps;
  int kind = 0;
  if (cfg.hasLights) {
    for (uint32_t li = 0; li < sc.numLightGroups; li++, kind++) {
      bool emit = false; V3 phi = v3(0, 0, 0); float lightDistance = 0.0f, ndot = 0.0f;
      if (active) {
        // pickRandomLightPoint, rendering.d:164-193
        uint32_t first = sc.lightFirst[li], cnt = sc.lightCount[li];
        float a = g.uniform(0.0f, 1.0f);
        uint32_t chosen = cnt - 1;
        for (uint32_t j = 0; j < cnt; j++) {
          float prob = sc.lights[(first + j) * 12 + 10];
          if (prob > a) { chosen = j; break; }
          a -= prob;
        }
        const float* lt = sc.lights + (size_t)(first + chosen) * 12;
        float u, v;
        do { u = g.uniform(0.0f, 1.0f); v = g.uniform(0.0f, 1.0f); } while (!(u + v <= 1.0f));
        V3 y = v3(lt[0], lt[1], lt[2]) + v3(lt[3], lt[4], lt[5]) * v + v3(lt[6], lt[7], lt[8]) * u;
        lightDistance = length(y - x);
        phi = normalized(y - x);
        ndot = dot(normalX, phi);
        emit = ndot > eps;
      }
      uint32_t idx = blockAppend(emit, &wf.cntOut[1 + kind], cx.sm, cx.phase);
      if (emit) {
        size_t e = (size_t)kind * wf.capQueue + idx;
        wf.shO[e] = f4(x + normalX * eps, u2f(laneSlot));
        wf.shD[e] = f4(phi, lightDistance);
        wf.shW[e] = f4(Tw, ndot);
      }
    }
  }
  if (cfg.hasSky) {
    {  // sun, rendering.d:432-445
      bool emit = false; V3 dir = v3(0, 0, 0), c = v3(0, 0, 0);
      if (active) {
        float psi = g.uniform(0.0f, 2.0f * PT_PI_F);
        float phi = g.uniform(PT_PI_2_F - PT_PI_2_F * 0.01f, PT_PI_2_F);
        dir = angleToDirection(phi, psi, cfg.sunDir);
        emit = dot(normalX, dir) > eps;
        c = Tw * cfg.sunRadiance * dot(normalX, cfg.sunDir);
      }
      uint32_t idx = blockAppend(emit, &wf.cntOut[1 + kind], cx.sm, cx.phase);
      if (emit) {
        size_t e = (size_t)kind * wf.capQueue + idx;
        wf.shO[e] = f4(x + normalX * 0.1f, u2f(laneSlot));
        wf.shD[e] = f4(dir, 0.0f);
        wf.shW[e] = f4(c, 0.0f);
      }
      kind++;
    }
    {  // sky, rendering.d:448-459
      bool emit = false; V3 dir = v3(0, 0, 0), c = v3(0, 0, 0);
      if (active) {
        float psi = g.uniform(0.0f, 2.0f * PT_PI_F);
        float phi = g.uniform(0.0f, PT_PI_2_F);
        dir = angleToLocalDirection(phi, psi);
        float nd = dot(normalX, dir);
        emit = nd > eps;
        c = Tw * cfg.skyRadiance * nd;
      }
      uint32_t idx = blockAppend(emit, &wf.cntOut[1 + kind], cx.sm, cx.phase);
      if (emit) {
        size_t e = (size_t)kind * wf.capQueue + idx;
        wf.shO[e] = f4(x + normalX * 0.1f, u2f(laneSlot));
        wf.shD[e] = f4(dir, 0.0f);
        wf.shW[e] = f4(c, 0.0f);
      }
      kind++;
    }
  }
}

// PRIMARY: the part of computeL after the view-ray trace (rendering.d:471-481).  Entry i is path slot i; the
// view ray is regenerated from (pixel, sample) instead of being read back.
// !PRIMARY: the part of computeLindirect after its trace (rendering.d:389-392) = computeLrefl at the hit.
template <bool PRIMARY>
__global__ void __launch_bounds__(512) k_shade(DevScene sc, DevCfg cfg, PixMap pm, const float4* __restrict__ acc, Wavefront wf,
                                               uint32_t nIn, int qIn, int qOut) {
  __shared__ uint32_t sm[40];
  ShadeCtx cx{sc, cfg, wf, sm, 0};
  uint32_t n = PRIMARY ? nIn : wf.cntIn[0];
  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    uint32_t i = base + threadIdx.x;
    bool valid = i < n, isHit = false;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 0), T = v3(1, 1, 1);
    uint32_t laneSlot = 0, depth = 0, gp = 0, sampleIdx = 0;
    Rng g; g.s0 = 1; g.s1 = 0;
    float4 hr = make_float4(0, 0, 0, 0);
    if (valid) {
      hr = wf.hit[i];
      isHit = f2u(hr.w) != 0xFFFFFFFFu;
      if (PRIMARY) {
        uint32_t s = i / pm.npix, lp = i - s * pm.npix;
        gp = mapPixel(pm, lp);
        sampleIdx = cfg.sampleBase + (uint32_t)acc[gp].w + s;
        g.seed(cfg.seed, gp, sampleIdx, 0);
        float x, y; viewRayXY(gp, cfg.W, cfg.H, &g, x, y);
        o = cfg.cam.pos; d = screenRayDir(cfg.cam, x, y);
        laneSlot = i;
      } else {
        float4 ro = wf.rayO[qIn][i], rd = wf.rayD[qIn][i];
        o = xyz(ro); d = xyz(rd); laneSlot = f2u(ro.w);
        if (isHit) {
          float4 t4 = wf.thr[laneSlot]; T = xyz(t4); depth = f2u(t4.w);
          uint2 r = wf.rng[laneSlot]; g.s0 = r.x; g.s1 = r.y;
        }
      }
    }
    bool active = valid && isHit;
    V3 hitPos = v3(0, 0, 0), normal = v3(0, 0, 1), color = v3(0, 0, 0);
    if (active) {
      uint32_t tri = f2u(hr.w);
      float4 q0 = sc.nrm[3 * tri], q1 = sc.nrm[3 * tri + 1], q2 = sc.nrm[3 * tri + 2];
      normal = interpNormal(xyz(q0), xyz(q1), xyz(q2), hr.y, hr.z);
      hitPos = o + d * hr.x;                                           // Ray.get(rayPos)
      float4 m = sc.mats[f2u(q0.w)];
      color = v3(m.y, m.z, m.w);
      if (PRIMARY) {
        V3 e = m.x * color;                                            // emissive * color, rendering.d:477
        wf.Lacc[laneSlot] = f4(e, 1.0f);
      }
    } else if (valid && PRIMARY) {
      wf.Lacc[laneSlot] = f4(sampleSky(d), 0.0f);                      // rendering.d:480
    }
    V3 Tw = T * (PT_BRDF * color);
    directLight(cx, active, g, hitPos, normal, Tw, laneSlot);
    if (cfg.mode == 1) {
      if (PRIMARY) {
        for (uint32_t c = 0; c < cfg.numIndirect; c++) {               // the loop at rendering.d:474-476
          uint32_t lane = (c + 1) * wf.capSlots + i;
          Rng gc; gc.s0 = 1; gc.s1 = 0;
          if (active) { gc.seed(cfg.seed, gp, sampleIdx, c + 1); wf.Lacc[lane] = make_float4(0, 0, 0, 0); }
          bounce(cx, active, gc, 0, v3(1, 1, 1), color, hitPos, normal, lane, qOut);
        }
      } else {
        bounce(cx, active, g, depth, T, color, hitPos, normal, laneSlot, qOut);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ accumulate
// pixel.sum += L; pixel.n += 1 for each of the batch's k samples of a pixel, in sample order
// (rendering.d:477-482; the (sum / N) of the indirect chains as in the block at rendering.d:474-477).
__global__ void __launch_bounds__(256) k_accumulate(DevCfg cfg, PixMap pm, float4* __restrict__ acc, Wavefront wf) {
  uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
  if (lp >= pm.npix) return;
  uint32_t gp = mapPixel(pm, lp);
  float4 a = acc[gp];
  for (uint32_t s = 0; s < pm.k; s++) {
    uint32_t slot = s * pm.npix + lp;
    float4 L = wf.Lacc[slot];
    V3 add = xyz(L);
    if (cfg.mode == 1 && L.w != 0.0f) {
      V3 sum = v3(0, 0, 0);
      for (uint32_t c = 0; c < cfg.numIndirect; c++) sum = sum + xyz(wf.Lacc[(size_t)(c + 1) * wf.capSlots + slot]);
      add = add + sum / (float)cfg.numIndirect;
    }
    a.x += add.x; a.y += add.y; a.z += add.z; a.w += 1.0f;
  }
  acc[gp] = a;
}

// ------------------------------------------------------------------------------------------------ operators
__global__ void k_view_rays(DevCfg cfg, uint32_t pixelOffset, uint32_t count, float* __restrict__ rays6) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float x, y; viewRayXY(pixelOffset + i, cfg.W, cfg.H, nullptr, x, y);
  V3 d = screenRayDir(cfg.cam, x, y);
  float* r = rays6 + 6 * (size_t)i;
  r[0] = cfg.cam.pos.x; r[1] = cfg.cam.pos.y; r[2] = cfg.cam.pos.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
}

template <bool LDSS>
__global__ void __launch_bounds__(1024) k_trace_rays(DevScene sc, float eps, const float* __restrict__ rays6, uint32_t count,
                                                     int32_t* triId, float* t, float* u, float* v, float* normal3) {
  stageScene<LDSS>(sc);
  uint32_t cN = 0, cT = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float* r = rays6 + 6 * (size_t)i;
    Hit h;
    bool ok = traverse<false, LDSS, false>(sc, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), eps, h, cN, cT);
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

template <bool LDSS>
__global__ void __launch_bounds__(1024) k_hits_nothing(DevScene sc, float eps, const float* __restrict__ rays6, uint32_t count,
                                                       uint8_t* out) {
  stageScene<LDSS>(sc);
  uint32_t cN = 0, cT = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float* r = rays6 + 6 * (size_t)i;
    Hit h;
    out[i] = traverse<true, LDSS, false>(sc, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), eps, h, cN, cT) ? 0 : 1;
  }
}

// drawScreen's scalar path, main.d:82-95
__global__ void k_tonemap(const float4* __restrict__ acc, uint32_t pixelOffset, uint32_t count, uint8_t* __restrict__ rgb) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float4 a = acc[pixelOffset + i];
  const float A = 0.055f, aPlusOne = 1 + 0.055f, power = 1 / 2.4f;
  float c[3] = {a.x / a.w, a.y / a.w, a.z / a.w};
  for (int k = 0; k < 3; k++) {
    float x = a.w > 0.0f ? c[k] : 0.0f;
    x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
    x = (x <= 0.0031308f) ? x * 12.92f : aPlusOne * powf(x, power) - A;
    rgb[3 * (size_t)i + k] = (uint8_t)(x * 255.0f);
  }
}

__global__ void k_rng_stream(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, uint32_t* out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    Rng g; g.seed(seed, pixel, sample, stream);
    for (uint32_t i = 0; i < n; i++) out[i] = g.next();
  }
}

// ------------------------------------------------------------------------------------------------ launch geometry
struct Geo { dim3 grid, block; size_t lds; };

// Traversal kernels are persistent: enough workgroups to fill the chip, grid-stride over the queue.
// With the scene in LDS the workgroup size grows with the scene so that one copy serves many waves.
Geo traversalGeo(const LaunchCfg& lc, const DevScene& sc, uint64_t maxItems) {
  Geo g; g.lds = sc.ldsBytes;
  uint32_t bs, perCU;
  if (sc.ldsBytes == 0) { bs = 256; perCU = 8; }
  else if (sc.ldsBytes <= 18 * 1024) { bs = 256; perCU = 8; }
  else if (sc.ldsBytes <= 38 * 1024) { bs = 512; perCU = 4; }
  else if (sc.ldsBytes <= 78 * 1024) { bs = 1024; perCU = 2; }
  else { bs = 1024; perCU = 1; }
  uint64_t want = (maxItems + bs - 1) / bs;
  uint64_t cap = (uint64_t)lc.numCUs * perCU;
  g.block = dim3(bs); g.grid = dim3((unsigned)std::max<uint64_t>(1, std::min(want, cap)));
  return g;
}

// Dynamic LDS above 64 KB has to be allowed per kernel (and per device) before the launch.
template <typename K> void allowLds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

#define PT_DISPATCH_LC(KERNEL, ...)                                                              \
  do {                                                                                           \
    bool l_ = sc.ldsBytes != 0, c_ = lc.countVisits;                                             \
    if (l_ && c_) { allowLds(KERNEL<true, true>, g.lds); hipLaunchKernelGGL((KERNEL<true, true>), g.grid, g.block, g.lds, lc.stream, __VA_ARGS__); } \
    else if (l_) { allowLds(KERNEL<true, false>, g.lds); hipLaunchKernelGGL((KERNEL<true, false>), g.grid, g.block, g.lds, lc.stream, __VA_ARGS__); } \
    else if (c_) hipLaunchKernelGGL((KERNEL<false, true>), g.grid, g.block, 0, lc.stream, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<false, false>), g.grid, g.block, 0, lc.stream, __VA_ARGS__);          \
  } while (0)

void launchExtendPrimary(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                         uint32_t nslots, int jitter, float4* hit, unsigned long long* stats) {
  Geo g = traversalGeo(lc, sc, nslots);
  PT_DISPATCH_LC(k_extend_primary, sc, cfg, pm, acc, nslots, jitter, hit, stats);
}

void launchExtendQueue(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int q, int statSlot) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  PT_DISPATCH_LC(k_extend_queue, sc, cfg.eps, (const float4*)wf.rayO[q], (const float4*)wf.rayD[q],
                 (const uint32_t*)&wf.cntIn[0], wf.hit, wf.stats, statSlot);
}

void launchConnect(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind, float area) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  size_t e = (size_t)kind * wf.capQueue;
  PT_DISPATCH_LC(k_connect, sc, cfg.eps, (const float4*)(wf.shO + e), (const float4*)(wf.shD + e), (const float4*)(wf.shW + e),
                 (const uint32_t*)&wf.cntOut[1 + kind], area, wf.Lacc, wf.stats);
}

void launchOccluded(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind) {
  Geo g = traversalGeo(lc, sc, wf.capQueue);
  size_t e = (size_t)kind * wf.capQueue;
  PT_DISPATCH_LC(k_occluded, sc, cfg.eps, (const float4*)(wf.shO + e), (const float4*)(wf.shD + e), (const float4*)(wf.shW + e),
                 (const uint32_t*)&wf.cntOut[1 + kind], wf.Lacc, wf.stats);
}

void launchShade(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                 const Wavefront& wf, bool primary, uint32_t nslots, int qIn, int qOut) {
  uint32_t bs = 512;
  uint64_t items = primary ? nslots : wf.capQueue;
  uint64_t want = (items + bs - 1) / bs, cap = (uint64_t)lc.numCUs * 4;
  dim3 grid((unsigned)std::max<uint64_t>(1, std::min(want, cap)));
  if (primary) hipLaunchKernelGGL(k_shade<true>, grid, dim3(bs), 0, lc.stream, sc, cfg, pm, acc, wf, nslots, qIn, qOut);
  else hipLaunchKernelGGL(k_shade<false>, grid, dim3(bs), 0, lc.stream, sc, cfg, pm, acc, wf, nslots, qIn, qOut);
}

void launchAccumulate(const LaunchCfg& lc, const DevCfg& cfg, const PixMap& pm, float4* acc, const Wavefront& wf) {
  hipLaunchKernelGGL(k_accumulate, dim3((pm.npix + 255) / 256), dim3(256), 0, lc.stream, cfg, pm, acc, wf);
}

void launchViewRays(const LaunchCfg& lc, const DevCfg& cfg, uint32_t pixelOffset, uint32_t count, float* rays6) {
  hipLaunchKernelGGL(k_view_rays, dim3((count + 255) / 256), dim3(256), 0, lc.stream, cfg, pixelOffset, count, rays6);
}

void launchTraceRays(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count,
                     int32_t* triId, float* t, float* u, float* v, float* normal3) {
  Geo g = traversalGeo(lc, sc, count);
  if (sc.ldsBytes) { allowLds(k_trace_rays<true>, g.lds); hipLaunchKernelGGL(k_trace_rays<true>, g.grid, g.block, g.lds, lc.stream, sc, cfg.eps, rays6, count, triId, t, u, v, normal3); }
  else hipLaunchKernelGGL(k_trace_rays<false>, g.grid, g.block, 0, lc.stream, sc, cfg.eps, rays6, count, triId, t, u, v, normal3);
}

void launchHitsNothing(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count, uint8_t* out) {
  Geo g = traversalGeo(lc, sc, count);
  if (sc.ldsBytes) { allowLds(k_hits_nothing<true>, g.lds); hipLaunchKernelGGL(k_hits_nothing<true>, g.grid, g.block, g.lds, lc.stream, sc, cfg.eps, rays6, count, out); }
  else hipLaunchKernelGGL(k_hits_nothing<false>, g.grid, g.block, 0, lc.stream, sc, cfg.eps, rays6, count, out);
}

void launchTonemap(const LaunchCfg& lc, const float4* acc, uint32_t pixelOffset, uint32_t count, uint8_t* rgb) {
  hipLaunchKernelGGL(k_tonemap, dim3((count + 255) / 256), dim3(256), 0, lc.stream, acc, pixelOffset, count, rgb);
}

void launchRngStream(const LaunchCfg& lc, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, uint32_t* out) {
  hipLaunchKernelGGL(k_rng_stream, dim3(1), dim3(64), 0, lc.stream, seed, pixel, sample, stream, n, out);
}

}  // namespace pt
