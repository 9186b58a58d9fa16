// Device-side data layout and launch entry points of the wavefront path tracer (gfx950).
#pragma once
#include "pt_math.h"
#include "bvh.h"

namespace pt {

// Everything the traversal and shading kernels read.  All arrays live in HBM; when ldsBytes != 0 the
// traversal kernels copy nodes + tris into LDS once per workgroup and traverse from there.
struct DevScene {
  const float4* nodes;        // 2 x float4 per node: (min.xyz, ref) (max.xyz, 0); sibling pairs adjacent
  const float4* tris;         // 3 x float4 per leaf-order triangle: (v0, origId) (v2-v0, 0) (v1-v0, 0)
  const float4* nrm;          // 3 x float4 per ORIGINAL triangle: (n0, matIdx) (n1, 0) (n2, 0)
  const float4* mats;         // (emissive, r, g, b)                                   rendering.d:33-37
  const float* lights;        // 12 floats per light triangle: v0 e1 e2 area probability pad   rendering.d:40-46
  const uint32_t* lightFirst; // per light group
  const uint32_t* lightCount;
  const float* lightArea;     // g_totalLightSourceArea
  uint32_t numNodes, numTris, numOrigTris, rootRef, numLightGroups;
  uint32_t ldsBytes;          // 0 => traverse from global memory
};

struct DevCfg {               // PtConfig + camera as the kernels need them
  uint32_t W, H, mode, hasSky, hasLights;
  V3 sunDir, sunRadiance, skyRadiance;
  float eps, rrProb;
  uint32_t maxDepth, numIndirect, seed, sampleBase;
  Camera cam;
};

// Which pixels a batch covers.  Local pixel lp (0..npix) of the batch is pixel
//   offset + lpBase + lp                                   (bandPix == 0: contiguous window)
//   (bandFirst + j*bandStride)*bandPix + r,  j = q / bandPix, r = q % bandPix, q = lpBase + lp   (band set)
// and path slot = sample * npix + lp.
struct PixMap { uint32_t offset, bandPix, bandFirst, bandStride, lpBase, npix, k; };

constexpr int PT_MAX_KINDS = 10;   // shadow-ray kinds per vertex: light groups + sun + sky
constexpr int PT_COUNTER_SET = 16;  // words per counter set (1 + PT_MAX_KINDS, padded)

struct Wavefront {
  float4* hit;                // per ray-queue entry: (t, u, v, triId)
  float4* rayO[2];            // ray queues (ping-pong): (origin.xyz, lane slot)
  float4* rayD[2];            //                          (dir.xyz, 0)
  float4* Lacc;               // per lane slot: radiance gathered by this lane (xyz), primary-hit flag (w)
  float4* thr;                // per lane slot: throughput of the chain (xyz), depth (w)
  uint2* rng;                 // per lane slot: generator state
  float4* shO;                // shadow queues, kind-major: (origin.xyz, lane slot)
  float4* shD;                //                            (dir.xyz, light distance)
  float4* shW;                //                            (weight.xyz, n.omega)
  uint32_t* cntIn;            // counter set written by the previous round: [0] = ray queue length
  uint32_t* cntOut;           // counter set of this round: [0] ray queue being filled, [1 + kind] shadow queues
  unsigned long long* stats;  // see STAT_*
  uint32_t capSlots;          // primary slots per batch
  uint32_t capQueue;          // entries per queue (capSlots * max(1, numIndirect) in mode 1)
  uint32_t lanes;             // 1 + numIndirect in mode 1, else 1
};

enum { STAT_RAYS_LIGHT = 0, STAT_RAYS_INDIRECT, STAT_RAYS_OCCL, STAT_NODES_CLOSEST, STAT_TRIS_CLOSEST,
       STAT_NODES_ANY, STAT_TRIS_ANY, STAT_COUNT };

struct LaunchCfg { int numCUs; bool countVisits; hipStream_t stream; };

// ---- wavefront stages (all asynchronous on lc.stream) ----
void launchExtendPrimary(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                         uint32_t nslots, int jitter, float4* hit, unsigned long long* stats);
void launchExtendQueue(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int q, int statSlot);
void launchShade(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                 const Wavefront& wf, bool primary, uint32_t nslots, int qIn, int qOut);
void launchConnect(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind, float area);
void launchOccluded(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind);
void launchAccumulate(const LaunchCfg& lc, const DevCfg& cfg, const PixMap& pm, float4* acc, const Wavefront& wf);

// ---- stand-alone operators ----
void launchViewRays(const LaunchCfg& lc, const DevCfg& cfg, uint32_t pixelOffset, uint32_t count, float* rays6);
void launchTraceRays(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count,
                     int32_t* triId, float* t, float* u, float* v, float* normal3);
void launchHitsNothing(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count,
                       uint8_t* out);
void launchTonemap(const LaunchCfg& lc, const float4* acc, uint32_t pixelOffset, uint32_t count, uint8_t* rgb);
void launchRngStream(const LaunchCfg& lc, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n,
                     uint32_t* out);

}  // namespace pt
