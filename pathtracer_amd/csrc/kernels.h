// Device-side data layout and launch entry points of the wavefront path tracer (gfx950).
#pragma once
#include "pt_math.h"
#include "bvh.h"

namespace pt {

// Division by a run-time constant without the ~30-instruction software divide (Granlund-Montgomery):
// q = (t + ((n - t) >> s1)) >> s2 with t = umulhi(m, n); exact for every 32-bit n.
struct FastDiv { uint32_t m, s1, s2, d; };
inline FastDiv makeFastDiv(uint32_t d) {
  FastDiv f; f.d = d ? d : 1;
  uint32_t l = 0; while ((1ull << l) < f.d) l++;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - f.d)) / f.d + 1);
  f.s1 = l < 1 ? l : 1; f.s2 = l > 1 ? l - 1 : 0;
  return f;
}
PT_HD uint32_t fdiv(const FastDiv& f, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t t = __umulhi(f.m, n);
#else
  uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
#endif
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// Everything the traversal and shading kernels read.  All arrays live in HBM; when ldsBytes != 0 the
// traversal kernels copy nodes + tris into LDS once per workgroup and traverse from there.
struct DevScene {
  const float4* nodes;        // 2 x float4 per node: (min.xyz, ref) (max.xyz, 0); sibling pairs adjacent.  Only
                              // present when ldsBytes != 0.
  const uint4* qnodes;        // the same tree for traversal from global memory (ldsBytes == 0), 16 B per node: box
                              // planes on a 16-bit grid over the root box, x: min | max << 16, y, z likewise, w: ref.
                              // plane = qBase + q * qScale, rounded outwards: the boxes only grow
  const float4* tris;         // 3 x float4 per leaf-order triangle: (v0, origId) (v2-v0, 0) (v1-v0, 0)
  const float4* nrm;          // 3 x float4 per ORIGINAL triangle: (n0, matIdx) (n1, 0) (n2, 0)
  const float4* mats;         // (emissive, r, g, b)                                   rendering.d:33-37
  const float* lights;        // 12 floats per light triangle: v0 e1 e2 area probability pad   rendering.d:40-46
  const uint32_t* lightFirst; // per light group
  const uint32_t* lightCount;
  const float* lightArea;     // g_totalLightSourceArea
  uint32_t numNodes, numTris, numOrigTris, rootRef, numLightGroups, numMats, numLightTris;
  uint32_t ldsBytes;          // 0 => traverse from global memory
  float qBase[3], qScale[3];
};

// The root box of the BVH.  A view ray that misses it cannot hit anything; raygen answers it without a traversal.
// (Testing a deeper cut of the tree in raygen was measured: 15 more boxes cull 14 % more rays but cost more in
// raygen than those rays cost in extend.)
struct SceneBox { float mn[3], mx[3]; };

// Per-pixel bit masks of the view rays (scene_host.h: buildCoverageMask), both null when no mask is in use.
//   covered      some triangle's screen rectangle reaches the pixel: its view rays are traced
//   constantSky  not covered and every sample of the pixel misses sampleSky's sun disc: each sample's radiance is
//                the same constant, raygen skips the slot and accumulate adds the constant without reading it
//   activeCols   per batch geometry (PixMap), or null: bit c says that column c of the batch -- local pixels 64 c ..
//                64 c + 63, the pixels of a run of 64 path slots -- has a pixel that is NOT constant sky; a run of a
//                column without one is skipped before anything is worked out per lane (launchActiveColumns)
struct ViewMasks { const uint32_t* covered; const uint32_t* constantSky; const uint32_t* activeCols; };

struct DevCfg {               // PtConfig + camera as the kernels need them
  uint32_t W, H, mode, hasSky, hasLights;
  V3 sunDir, sunRadiance, skyRadiance;
  float eps, rrProb;
  uint32_t maxDepth, numIndirect, seed, sampleBase, sampleStride;
  Camera cam;
  FastDiv divW;
};

// Which pixels a batch covers and in which order.  Path slot i = sample * npix + lp.  Local pixel lp is first
// mapped to a position q in the batch's pixel list -- linearly (tileW == 0), or so that each run of 64 slots is a
// tileW x (64/tileW) block of pixels (coherent rays per wave; needs whole rows) -- and q to a frame pixel:
//   offset + lpBase + q                                    (bandPix == 0: contiguous window)
//   (bandFirst + j*bandStride)*bandPix + r, j = (lpBase+q) / bandPix, r = (lpBase+q) % bandPix   (band set)
// Tiles are numbered strip by strip: vertical strips of stripW tiles, row-major inside a strip (stripW == 0: row-major
// over the whole width).  Consecutive runs of slots -- what the waves of a launch, and the workgroups that share an
// L2, have in flight at a time -- then cover a compact block of pixels instead of a band across the frame, so the
// rays in flight meet in a smaller part of a scene that is traversed from global memory.  The last strip takes what
// is left of the width (lastW tiles; nFull strips are full).
struct PixMap {
  uint32_t offset, bandPix, bandFirst, bandStride, lpBase, npix, k;
  uint32_t tileW, tilesPerRow;      // tile shape and tiles per row of the frame
  uint32_t stripW, stripTiles, nFull, lastW;      // stripTiles = stripW * rows of tiles in the batch
  uint32_t columns;                               // npix / 64 when that is whole (run r is column r % columns), else 0
  FastDiv divNpix, divBand, divTilesPerRow, divStripTiles, divStripW, divLastW, divColumns;
};

PT_HD uint32_t mapPixel(const PixMap& pm, uint32_t W, uint32_t lp) {
  uint32_t q = lp;
  if (pm.tileW) {
    uint32_t tile = lp >> 6, in = lp & 63u;
    uint32_t tr, tc;
    if (pm.stripW) {
      uint32_t strip = fdiv(pm.divStripTiles, tile);
      if (strip >= pm.nFull) { uint32_t t = tile - pm.nFull * pm.stripTiles; tr = fdiv(pm.divLastW, t); tc = pm.nFull * pm.stripW + (t - tr * pm.lastW); }
      else { uint32_t t = tile - strip * pm.stripTiles; tr = fdiv(pm.divStripW, t); tc = strip * pm.stripW + (t - tr * pm.stripW); }
    } else { tr = fdiv(pm.divTilesPerRow, tile); tc = tile - tr * pm.tilesPerRow; }
    uint32_t tx, ty;
    if (pm.tileW == 8) { tx = in & 7u; ty = in >> 3; } else { tx = in & 15u; ty = in >> 4; }
    uint32_t th = 64u / pm.tileW;
    q = (tr * th + ty) * W + tc * pm.tileW + tx;
  }
  q += pm.lpBase;
  if (pm.bandPix == 0) return pm.offset + q;
  uint32_t j = fdiv(pm.divBand, q), r = q - j * pm.bandPix;
  return (pm.bandFirst + j * pm.bandStride) * pm.bandPix + r;
}

constexpr int PT_MAX_KINDS = 10;    // shadow queues per vertex: one per light group + one for the sun and sky rays

// Queues are SEGMENTED: every stage splits its input into PT_SEGS block-tasks of P consecutive entries, and
// block-task T appends what it emits to segment T of each output queue (entries T*segCap .. T*segCap+count[T]).
// No global atomic and no device-wide ordering is involved in an append: lanes rank themselves with a wave64
// ballot + mbcnt and waves of the workgroup share an LDS counter.  A one-workgroup scan per queue then turns the
// PT_SEGS counts into prefix sums, so that the next stage sees a dense queue: flat entry j lives in the segment s
// with prefix[s] <= j < prefix[s+1], at s*segCap + (j - prefix[s]).
constexpr uint32_t PT_SEGS = 1024;
struct QMeta {
  uint32_t* count;            // [PT_SEGS]      entries per segment, written by the producing stage
  uint32_t* prefix;           // [PT_SEGS + 2]  exclusive prefix sums; [PT_SEGS] = total, [PT_SEGS + 1] = P of the consumers
  uint32_t* start;            // [PT_SEGS]      segment that holds flat entry T*P: where consumer block-task T starts
};
constexpr uint32_t PT_QMETA_WORDS = 3 * PT_SEGS + 2;
constexpr int PT_TASK_COUNTERS = 16;   // one dynamic work counter per traversal launch of a stage ...
// ... which is PT_TASK_PARTS words: a launch's queue is cut into that many contiguous parts with a counter each, and a
// workgroup starts with the part of its index modulo PT_TASK_PARTS -- workgroups are dealt to the 8 XCDs in turn, so the
// workgroups that share an XCD (and its L2) work their way through one contiguous part of the queue together instead of
// every L2 seeing rays from all over the queue; a workgroup whose part is used up goes on with the next one.
constexpr int PT_TASK_PARTS = 8, PT_TASK_WORDS = PT_TASK_COUNTERS * PT_TASK_PARTS;
// Counters of the runs that the whole-sample kernel's workgroups take from a common pool (k_sample_traced), each on a
// cache line of its own; they follow the task counters and are zero between batches (k_accumulate clears them).
constexpr int PT_POOL_COUNTERS = 8, PT_POOL_STRIDE = 64, PT_POOL_WORDS = PT_POOL_COUNTERS * PT_POOL_STRIDE;

// Traversal kernels consume a queue dynamically: a workgroup takes PT_SUPER consecutive flat entries from the
// launch's global counter into an LDS pool, and its waves take 64 at a time from the pool, independently of
// each other (no barrier between chunks: a wave with long traversals does not hold the others back).  Extend,
// the one traversal kernel with an output queue, gives each workgroup a run of adjacent segments to append to;
// PT_HIT_SLACK extra entries per segment guarantee that the workgroups that still have room can take all input:
// a workgroup stops taking input only when its region could not hold one more refill plus the chunks in flight
// (out + blockDim + PT_SUPER > region), i.e. when out > region - PT_HIT_SLACK >= its share of capQueue.  If every
// workgroup had stopped, the outputs would sum to more than capQueue, but there are never more outputs than input
// entries and the input queue holds at most capQueue.
constexpr uint32_t PT_SUPER = 2048;
constexpr uint32_t PT_HIT_SLACK = PT_SUPER + 1024;

struct Wavefront {
  float4* slotQ;              // view rays that may hit something: (dir.xyz, path slot); the origin is the camera
  float4* rayO;               // indirect-ray queue: (origin.xyz, lane slot)
  float4* rayD;               //                     (dir.xyz, 0)
  float4* hitA;               // hit queue (filled by extend, hits only): (hit position.xyz, triangle id)
  float4* hitB;               //                                          (u, v, lane slot, t)
  uint32_t* hitOrder;         // 4 words per entry of the hit queue, or null: [0, capHit) the order in which shade takes a
                              // segment's hits, the rest is what the sort that makes it works in (launchHitOrder)
  uint32_t regionPix;         // != 0: the hits of the view rays of local pixels [s * regionPix, (s + 1) * regionPix) go to segment s
  FastDiv divRegion;          //       of the hit queue, whatever their sample (k_extend_refill); 0: segments by the order of the rays
  float4* Lacc;               // radiance records.  [0, capSlots): per path slot, the sample's computeLdirect sum or sampleSky
                              // (xyz); w: bits of (material index + 1) when the view ray hit (above bit 24: which
                              // chains' first rays hit, see chainMask), 0 when it missed.  Then, per vertex of a chain,
                              // record lane slot + round * chainStride: computeLdirect at that vertex (xyz), w: bits of
                              // (material index + 1) | (the chain goes on) << 31, or 0 where the chain has no vertex
  uint2* rng;                 // per lane slot: generator state
  float4* shO;                // shadow queues, kind-major: (origin.xyz, lane slot)
  float4* shD;                //                            (dir.xyz, light distance)
  float4* shW;                //                            (weight.xyz, n.omega)
  float4* shX;                // the sun/sky queue (the last kind) holds (origin, lane slot), (sun dir, n.sunDir),
                              // (sky dir, n.dir) in shO/shD/shW and a fourth vector here: see k_occluded
  QMeta rayIn;                // slot / ray queue this stage's extend reads (filled by the previous stage)
  QMeta hitQ;                 // hit queue of this stage: extend fills it, shade reads it
  QMeta rayOut;               // ray queue this stage's shade fills
  QMeta sh[PT_MAX_KINDS];     // shadow queues this stage's shade fills
  uint32_t* taskCounters;     // [PT_TASK_WORDS], zeroed by the scan that precedes the launches using them; counter c of
                              // a stage = words [c * PT_TASK_PARTS, (c + 1) * PT_TASK_PARTS)
  uint32_t* runPool;          // [PT_POOL_WORDS] or null (every workgroup keeps to its own share of the runs)
  unsigned long long* stats;  // see STAT_*
  uint32_t capSlots;          // primary slots per batch (multiple of 64 * PT_SEGS)
  uint32_t capQueue;          // entries per ray / shadow queue (capSlots * max(1, numIndirect) in mode 1)
  uint32_t capHit;            // entries of the hit queue: capQueue + PT_SEGS * PT_HIT_SLACK
  uint32_t lanes;             // 1 + numIndirect in mode 1, else 1
  uint32_t chainStride;       // numIndirect * capSlots: from a chain vertex's record to that of the next bounce round
  uint32_t recRounds;         // bounce rounds that have records: maxDepth + 1
  FastDiv divCap;             // division by capSlots
  uint32_t round;             // bounce round of the stage being launched (0 = hits of the rays the view-ray hits spawned)
  // Shading that traces the rays it spawns (k_shade<.., FUSE>, fuse != 0).  The hits of a stage's continuation rays go
  // to (hitOutA, hitOutB), the next stage's hit queue; the host
  // alternates the two record arrays (hitA/hitB and rayO/rayD, both rayCap >= capHit entries then) between stages.
  float4* hitOutA; float4* hitOutB;
  uint32_t rayCap;            // entries of rayO / rayD
  uint32_t fuse;
  uint32_t chainMask;         // 1 (fuse in mode 1): the tag word of a view-ray record is (chains whose first ray hit)
                              // << 24 | tag, and a chain vertex's record says whether the chain goes on
};

// Scene and tables of the fused shading variant share a shade workgroup's LDS; above this the general path runs.
constexpr uint32_t PT_FUSE_LDS_BYTES = 24 * 1024;

enum { STAT_RAYS_LIGHT = 0, STAT_RAYS_INDIRECT, STAT_RAYS_OCCL, STAT_NODES_CLOSEST, STAT_TRIS_CLOSEST,
       STAT_NODES_ANY, STAT_TRIS_ANY, STAT_RAYS_CULLED, STAT_HITS, STAT_ENV_ENTRIES, STAT_FUSED_INDIRECT, STAT_FUSED_OCCL, STAT_HITS_PRIMARY, STAT_FUSED_LIGHT, STAT_FUSED_PRIMARY,
       STAT_WHOLE_LIGHT, STAT_WHOLE_INDIRECT, STAT_WHOLE_OCCL,      // of the FUSED ones: traced by the whole-sample kernel
       STAT_WHOLE_CHAIN_HITS,                                        // hits it appended for the shading of chain hits
       STAT_CONSTANT_SKY,                                            // of RAYS_CULLED: samples of constant-sky pixels
       STAT_COUNT };
constexpr int PT_STAT_REPLICAS = 64, PT_STAT_STRIDE = 32;   // copies of the counters, 256 B apart (kernels.hip: statAdd)
static_assert(STAT_COUNT <= PT_STAT_STRIDE, "a copy of the counters is two 128-byte lines of its own");

// Workgroups of a traversal kernel per CU when the scene copy in LDS takes `bytes` (0: it does not fit at all).
inline uint32_t ldsWorkgroupsPerCU(uint64_t bytes) {
  return bytes <= 18 * 1024 ? 8u : bytes <= 38 * 1024 ? 4u : bytes <= 78 * 1024 ? 2u : bytes <= 156 * 1024 ? 1u : 0u;
}

// Shading tables (vertex normals, materials, light triangles, light groups) small enough for every shade workgroup
// to keep a copy in LDS: bytes of that copy, or 0 when they are read from global memory.
inline uint32_t shadeTableBytes(const DevScene& sc) {
  uint64_t b = ((uint64_t)3 * sc.numOrigTris + sc.numMats + (uint64_t)3 * sc.numLightTris) * 16 + (uint64_t)2 * sc.numLightGroups * 4;
  return b <= 24576 ? (uint32_t)b : 0u;
}

// Whether the sun/sky kernel stores L of a vertex outright instead of adding to it: view-ray hits of a scene in which
// no light-group kernel adds to L before it.
PT_HD bool envDirect(const DevCfg& cfg, const DevScene& sc, bool primary) {
  return primary && cfg.hasSky && !(cfg.hasLights && sc.numLightGroups);
}

// maxGroups != 0 bounds the workgroups of a traversal launch (extend's hit queue needs PT_HIT_SLACK spare entries
// per workgroup: a caller with a small hit queue launches fewer).
struct LaunchCfg { int device; int numCUs; bool countVisits; hipStream_t stream; uint32_t maxGroups = 0; };

// ---- wavefront stages (all asynchronous on lc.stream) ----
// masks.covered == null: cull by the scene box
void launchRaygen(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevCfg& cfg, const PixMap& pm,
                  const float4* acc, uint32_t nslots, const Wavefront& wf);
// prefix sums + consumer start segments for n queues, and zeroing of the stage's dynamic task counters
void launchScan(const LaunchCfg& lc, const QMeta* metas, int n, uint32_t* taskCounters);
// primary: entries of wf.rayIn index wf.slotQ (view direction + path slot); else wf.rayO / wf.rayD
void launchExtend(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                  const Wavefront& wf, bool primary);
// scenes in global memory: fills wf.hitOrder from the hit queue (after its scan, before the shade that reads it)
void launchHitOrder(const LaunchCfg& lc, const DevScene& sc, const Wavefront& wf);
void launchShade(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                 const Wavefront& wf, bool primary);
void launchConnect(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind, float area);
// primary: the queue was filled by the shading of view-ray hits
void launchOccluded(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const Wavefront& wf, int kind, bool primary);
// raygen + extend of the view rays in one traversal launch: fills wf.hitQ (counts) and the radiance records of the
// misses.  wf.taskCounters[PT_TASK_COUNTERS - 1] must be zero (it is after any launchScan).
void launchRaygenExtend(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevScene& sc, const DevCfg& cfg,
                        const PixMap& pm, const float4* acc, uint32_t nslots, const Wavefront& wf);
// shading of the view-ray hits + their sun and sky rays in one traversal launch (mode 0, sky, no light groups): reads
// wf.hitQ, stores the radiance records
void launchShadeOccluded(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc, const Wavefront& wf);
// raygen + trace of the view rays + shading of their hits in one launch (wf.fuse != 0): fills the radiance records and
// wf.rayOut = the hit queue of the chains' first rays
void launchSampleTraced(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevScene& sc, const DevCfg& cfg,
                        const PixMap& pm, const float4* acc, uint32_t nslots, const Wavefront& wf);
// mode 0 with wf.fuse: the pm.k samples of every pixel of pm, one pixel per lane, added to acc in sample order -- a
// batch's whole-sample launch and its accumulate in one, with no radiance records in between; any pm.k
void launchPixelTraced(const LaunchCfg& lc, const SceneBox& box, const ViewMasks& masks, const DevScene& sc, const DevCfg& cfg,
                       const PixMap& pm, float4* acc, const Wavefront& wf);
// Whether a window of npix pixels takes that path rather than whole-sample launches + accumulate: its columns of 64
// pixels are the units of work, and a wave walks a column's samples one after the other, so there have to be enough
// columns for the waves of the chip -- or one sample each, where a column is no more work than a run.
inline bool pixelSumsPay(uint64_t npix, uint32_t spp) { return spp == 1 || npix >= 64ull * 12288ull; }
// bits[c >> 5] bit c & 31 = column c of pm has a pixel outside masks.constantSky (pm.columns columns; bits are cleared first)
void launchActiveColumns(const LaunchCfg& lc, const ViewMasks& masks, const DevCfg& cfg, const PixMap& pm, uint32_t* bits);
void launchAccumulate(const LaunchCfg& lc, const ViewMasks& masks, const DevCfg& cfg, const PixMap& pm, float4* acc,
                      const Wavefront& wf, const DevScene& sc);

// ---- stand-alone operators ----
void launchExtendPrimary(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const PixMap& pm, const float4* acc,
                         uint32_t nslots, int jitter, float4* hit, unsigned long long* stats);
void launchViewRays(const LaunchCfg& lc, const DevCfg& cfg, uint32_t pixelOffset, uint32_t count, float* rays6);
void launchTraceRays(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count,
                     int32_t* triId, float* t, float* u, float* v, float* normal3);
void launchTraceTex(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* tex6, const float* rays6,
                    uint32_t count, bool ignoreBackfaces, int32_t* result, int32_t* triId, float* t, float* tc2);
void launchHitsNothing(const LaunchCfg& lc, const DevScene& sc, const DevCfg& cfg, const float* rays6, uint32_t count,
                       uint8_t* out, unsigned long long* stats);
void launchTonemap(const LaunchCfg& lc, const float4* acc, uint32_t pixelOffset, uint32_t count, uint8_t* rgb);
void launchPackPixels(const LaunchCfg& lc, const float4* acc, uint32_t pixelOffset, uint32_t count, float* out7);
void launchUnpackPixels(const LaunchCfg& lc, const float* in7, uint32_t pixelOffset, uint32_t count, float4* acc);
void launchBandRows(const LaunchCfg& lc, float4* frame, float4* packed, uint32_t W, uint32_t H, uint32_t bandRows, uint32_t first,
                    uint32_t stride, uint32_t packedRows, bool unpack);
void launchAngleDirections(const LaunchCfg& lc, const float* in5, uint32_t n, float* out3);
// Writes n queue descriptors over the counter words at `words` (PT_QMETA_WORDS each), as the host lays them out.
void launchInitMetas(const LaunchCfg& lc, QMeta* desc, uint32_t* words, uint32_t n);
void launchRngStream(const LaunchCfg& lc, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n,
                     uint32_t* out);

}  // namespace pt
