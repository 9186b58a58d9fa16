// C ABI of libptgpu (include/ptgpu.h): handle, scene upload, wavefront scheduling, read-back.
#include "../../include/ptgpu.h"
#include "kernels.h"
#include "bvh_gpu.h"
#include "mask_gpu.h"
#include "scene_host.h"
#include "thmodel.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace pt;

namespace {

thread_local std::string g_err;

struct ApiError { int code; std::string msg; };
[[noreturn]] void fail(int code, const std::string& m) { throw ApiError{code, m}; }

#define HIPCHECK(expr)                                                                                        \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) fail(PT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                \
  } while (0)

struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  void alloc(size_t n) {
    if (n <= bytes && p) return;
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) { p = nullptr; fail(PT_E_NOMEM, "hipMalloc(" + std::to_string(n) + " bytes): " + hipGetErrorString(e)); }
    bytes = n;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  ~DevBuf() { release(); }
  template <typename T> T* as() const { return (T*)p; }
};

// Wavefront state of ptgpu_render_noalloc: queues, per-slot records and queue counters of ONE batch, in device
// memory that exists from the moment the code object is loaded, so that the call itself allocates nothing.  Every
// device the library runs on holds its own copy until the process ends.  `scratch` names the array for
// hipGetSymbolAddress only; the address host code works with is the per-device pointer taken from it at set-up.
constexpr size_t PT_NOALLOC_BYTES = (size_t)PT_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[PT_NOALLOC_BYTES];

constexpr int PT_MAX_DEVICES = 64;
void* g_scratchPtr[PT_MAX_DEVICES] = {};
std::mutex g_scratchMu;
// All calls on one GPU share that GPU's array.  They are put in order instead of racing: a call enqueues its work
// under the device's mutex, first making its stream wait for the event the previous call on that device recorded
// behind its last kernel, and records the event again behind its own.  Neither the wait nor the record allocates.
hipEvent_t g_scratchDone[PT_MAX_DEVICES] = {};
std::mutex g_scratchCall[PT_MAX_DEVICES];

// Address of `scratch` on the current device (which must be `device`); looked up once per device.
void* scratchOnDevice(int device) {
  if (device < 0 || device >= PT_MAX_DEVICES) fail(PT_E_INVALID, "device ordinal beyond " + std::to_string(PT_MAX_DEVICES - 1));
  std::lock_guard<std::mutex> lk(g_scratchMu);
  if (!g_scratchPtr[device]) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) fail(PT_E_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    hipEvent_t ev = nullptr;
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) fail(PT_E_DEVICE, std::string("hipEventCreateWithFlags: ") + hipGetErrorString(e));
    g_scratchDone[device] = ev;
    g_scratchPtr[device] = p;
  }
  return g_scratchPtr[device];
}

// Page-locked host memory that lives as long as the handle: what the frame crosses PCIe through when the caller's
// own buffer is pageable (a pageable copy runs at a tenth of the link's rate).
struct PinnedBuf {
  void* p = nullptr; size_t bytes = 0;
  void alloc(size_t n) {
    if (n <= bytes && p) return;
    release();
    hipError_t e = hipHostMalloc(&p, n ? n : 16, hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; fail(PT_E_NOMEM, "hipHostMalloc(" + std::to_string(n) + " bytes): " + hipGetErrorString(e)); }
    bytes = n ? n : 16;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
  ~PinnedBuf() { release(); }
};

// Whether the device can copy straight into / out of p (page-locked memory of hipHostMalloc / ptgpu_host_alloc).
bool isPinnedHost(const void* p) {
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

struct EventPair { hipEvent_t a, b; int kind; };
constexpr uint64_t PT_GRAPH_MAX_SLOTS = 1ull << 22;       // batches up to this many samples are replayed from a graph
constexpr int KIND_RENDER = PT_K_COUNT;       // span of one ptgpu_render call
constexpr size_t MAX_EVENT_PAIRS = 16384;

}  // namespace

struct PtHandle {
  PtConfig cfg{};
  int numCUs = 256;
  hipStream_t stream = nullptr; bool ownStream = false;
  bool haveScene = false, haveCamera = false;
  Camera cam{};
  FlatScene flat; LightTable lights; Bvh bvh;
  std::string bvhNote;          // why the GPU builder was not used although asked for (empty otherwise)
  DevBuf dNodes, dTris, dNrm, dMats, dLights, dLightFirst, dLightCount, dLightArea;
  DevScene dsc{};
  DevBuf accOwn; float4* acc = nullptr; uint64_t accBytes = 0;
  // wavefront buffers
  DevBuf bSlotQ, bHitA, bHitB, bHitOrder, bRayO, bRayD, bLacc, bRng, bShO, bShD, bShW, bShX, bCounters, bStats;
  SceneBox box{};
  // coverage mask of the view rays (scene_host.h: buildCoverageMask); rebuilt when the scene or the camera changes
  DevBuf dMask, dSkyMask; bool maskDirty = true, maskUsable = false;
  DevBuf dVerts, dMaskRows, dMaskBig, dMaskFlags; double sceneScale = 0.0;   // for the GPU build of the masks (mask_gpu.h)
  DevBuf dTex; bool texDirty = true;   // TriangleData.tex in input order, uploaded at the first texcoord trace
  ViewMasks masks() const { return maskUsable ? ViewMasks{dMask.as<uint32_t>(), dSkyMask.as<uint32_t>(), nullptr} : ViewMasks{nullptr, nullptr, nullptr}; }
  // which columns of the current batch geometry have anything but constant sky (ViewMasks::activeCols): built on the
  // stream when the geometry or the masks change
  DevBuf dActiveCols; PixMap colKey{}; uint64_t colMaskGen = 0, maskGen = 0; bool colValid = false;
  DevBuf bMetaDesc; std::vector<QMeta> metas;     // queue descriptors: device copy and host copy, stage-major
  Wavefront wf{}; uint32_t wfKinds = 0, wfRounds = 0; bool wfLean = false;
  // ptgpu_render_noalloc: address of the static scratch array on this handle's device, and the host copy of the
  // queue descriptors laid over it (sized at creation for the largest maxDepth, so that the call allocates nothing)
  unsigned char* scratchDev = nullptr; std::vector<QMeta> metasNoAlloc;
  // frame transfer: device-side records and bytes, page-locked staging, a second stream for copies that overlap the
  // next pass, one event per tone-map buffer (two buffers alternate)
  DevBuf dPixelRec, dRgb[2]; PinnedBuf pinPixels, pinRgb[2];
  hipStream_t copyStream = nullptr; hipEvent_t evTone[2] = {nullptr, nullptr}, evCopy[2] = {nullptr, nullptr};
  int toneNext = 0, tonePending = -1; uint8_t* toneDst = nullptr; size_t toneBytes = 0; bool toneDirect = false;
  // small batches (the progressive loop at the reference's frame size: a dozen launches of a few microseconds each)
  // are replayed from a HIP graph: the first call of a shape runs normally, the second is captured, later ones replay
  std::vector<uint8_t> graphSeen, graphKey; hipGraphExec_t graphExec = nullptr; uint64_t sceneGeneration = 0;
  bool noEvents = false;        // set while ptgpu_render_noalloc runs: no event is created or recorded
  // statistics
  PtStats st{};
  std::vector<EventPair> pending; std::vector<hipEvent_t> freeEvents;
  bool eventOverflow = false;
};

namespace {

// device < 0 makes a host-only handle: scene preparation (reader, flattening, lights, BVH) works, anything
// that needs the GPU fails.
void useDevice(PtHandle* h) {
  if (h->cfg.device < 0) fail(PT_E_DEVICE, "this handle was created without a device (device < 0): host-side scene preparation only");
  HIPCHECK(hipSetDevice(h->cfg.device));
}

void normalizeConfig(PtConfig& c) {
  if (c.floatEpsilon == 0.0f) c.floatEpsilon = 1e-5f;
  if (c.rrProb == 0.0f) c.rrProb = 0.7f;
  if (c.maxDepth == 0) c.maxDepth = 5;
  if (c.numIndirect == 0) c.numIndirect = 5;
  if (c.sampleStride == 0) c.sampleStride = 1;
  if (c.maxLeafTris > BVH_MAX_LEAF) c.maxLeafTris = BVH_MAX_LEAF;
}

void validateConfig(const PtConfig& c) {
  if (c.width == 0 || c.height == 0) fail(PT_E_INVALID, "width and height must be non-zero");
  if ((uint64_t)c.width * c.height > 0x7FFFFFFFull) fail(PT_E_INVALID, "frame has more than 2^31 pixels");
  if (c.mode > 1) fail(PT_E_INVALID, "mode must be 0 (direct only) or 1 (+ indirect chains)");
  if (c.numIndirect > 64) fail(PT_E_INVALID, "numIndirect > 64");
  if (c.maxDepth > 64) fail(PT_E_INVALID, "maxDepth > 64");
}

DevCfg makeDevCfg(const PtHandle* h) {
  const PtConfig& c = h->cfg; DevCfg d;
  d.W = c.width; d.H = c.height; d.mode = c.mode; d.hasSky = c.hasSky; d.hasLights = c.hasLights;
  d.sunDir = v3(c.sunDir[0], c.sunDir[1], c.sunDir[2]);
  d.sunRadiance = v3(c.sunRadiance[0], c.sunRadiance[1], c.sunRadiance[2]);
  d.skyRadiance = v3(c.skyRadiance[0], c.skyRadiance[1], c.skyRadiance[2]);
  d.eps = c.floatEpsilon; d.rrProb = c.rrProb; d.maxDepth = c.maxDepth; d.numIndirect = c.numIndirect;
  d.seed = c.seed; d.sampleBase = c.sampleBase; d.sampleStride = c.sampleStride; d.cam = h->cam;
  d.divW = makeFastDiv(c.width);
  return d;
}

LaunchCfg makeLaunchCfg(const PtHandle* h) {
  return LaunchCfg{h->cfg.device, h->numCUs, (h->cfg.flags & PT_FLAG_COUNT_VISITS) != 0, h->stream};
}

hipEvent_t getEvent(PtHandle* h) {
  if (!h->freeEvents.empty()) { hipEvent_t e = h->freeEvents.back(); h->freeEvents.pop_back(); return e; }
  hipEvent_t e; HIPCHECK(hipEventCreate(&e)); return e;
}

// Brackets a launch with events when asked to; the pairs are resolved in ptgpu_get_stats.
struct Timed {
  PtHandle* h; int kind; bool on; hipEvent_t a{}, b{};
  Timed(PtHandle* h_, int kind_, bool force = false) : h(h_), kind(kind_) {
    const bool wanted = !h->noEvents && (force || (h->cfg.flags & PT_FLAG_TIME_KERNELS));
    on = wanted && h->pending.size() < MAX_EVENT_PAIRS;
    if (!on && wanted) h->eventOverflow = true;
    if (on) { a = getEvent(h); b = getEvent(h); HIPCHECK(hipEventRecord(a, h->stream)); }
    if (kind < PT_K_COUNT) h->st.kernelLaunches[kind]++;
  }
  void done() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fail(PT_E_DEVICE, "kernel launch failed (kind " + std::to_string(kind) + "): " + hipGetErrorString(e));
    if (on) { HIPCHECK(hipEventRecord(b, h->stream)); h->pending.push_back(EventPair{a, b, kind}); on = false; }
  }
};

void resolveEvents(PtHandle* h) {
  for (EventPair& p : h->pending) {
    float ms = 0.0f;
    HIPCHECK(hipEventSynchronize(p.b));
    HIPCHECK(hipEventElapsedTime(&ms, p.a, p.b));
    if (p.kind == KIND_RENDER) h->st.renderMs += ms; else h->st.kernelMs[p.kind] += ms;
    h->freeEvents.push_back(p.a); h->freeEvents.push_back(p.b);
  }
  h->pending.clear();
}

void uploadScene(PtHandle* h) {
  const FlatScene& fs = h->flat;
  uint32_t T = (uint32_t)fs.numTriangles();
  if (T >= (1u << BVH_COUNT_SHIFT)) fail(PT_E_INVALID, "too many triangles (limit 2^28 - 1)");
  for (uint32_t i = 0; i < T; i++)
    if (fs.matIdx[i] >= fs.mats.size()) fail(PT_E_INVALID, "material index out of range at triangle " + std::to_string(i));
  findLights(fs, h->lights);
  if (h->lights.numGroups() + 1 > (size_t)PT_MAX_KINDS)
    fail(PT_E_INVALID, "more than " + std::to_string(PT_MAX_KINDS - 1) + " emissive materials");
  // The binned-SAH build on one host core takes ~1 s per million triangles; from 1 M triangles up the tree is built
  // on the GPU instead (Morton-order LBVH: more traversal steps per ray, but the soup builds in a fraction of a second).
  const uint32_t f = h->cfg.flags;
  const bool autoLeaf = h->cfg.maxLeafTris == 0;          // 0: the library chooses (4, more when that fits the scene in LDS)
  const uint32_t leafCap = autoLeaf ? 4u : h->cfg.maxLeafTris;
  bool onGpu = false;
  if (h->cfg.device >= 0 && !(f & PT_FLAG_HOST_BVH) && ((f & PT_FLAG_GPU_BVH) || T >= 1000000u)) {
    std::string why;
    // leaves of at most 2 triangles: the collapse of the radix tree is blind to surface areas, and with 4 the soup's
    // rays test 5x as many triangles as in the SAH tree (measured, 10 M triangles: build 0.28 s vs 9.1 s on the host,
    // 698 vs 735 Mrays/s; leaves of 4: 675, of 1: 672)
    onGpu = buildBvhGpu(fs.v.data(), T, std::min(leafCap, 2u), h->bvh, why);
    h->bvhNote = onGpu ? "" : why;
  }
  if (!onGpu) {
    buildBvh(fs.v.data(), T, leafCap, h->bvh);
    // Tree shape for LDS residency: the traversal kernels keep a copy of nodes + triangles per workgroup.  Up to 78 KB
    // a CU holds 32 waves of them (8 x 256, 4 x 512 or 2 x 1024 threads), up to 156 KB one workgroup of 16 waves,
    // beyond that the scene is traversed from global memory.  A scene just above one of these two steps is rebuilt
    // with fuller leaves (fewer nodes) when that gets it under the step: the waves outweigh the extra triangle tests
    // (1300 big triangles: 130.9 KB -> 76.4 KB, 18.9 -> 15.0 ms per 64 spp at 1080p; 2400: 237 KB, traversed from
    // global memory, -> 147.5 KB in LDS, 35.0 -> 31.7 ms).  With an explicit maxLeafTris only the cost changes.
    auto footprint = [](const Bvh& b) { return ((uint64_t)2 * b.numNodes + (uint64_t)3 * b.triOrder.size()) * 16; };
    if (!(f & PT_FLAG_NO_LDS_SCENE) && (uint64_t)T * 48 <= 156 * 1024) {
      auto residency = [](uint64_t bytes) { return bytes <= 78 * 1024 ? 2u : bytes <= 156 * 1024 ? 1u : 0u; };
      uint32_t best = residency(footprint(h->bvh));
      const uint32_t leaves[3] = {leafCap, autoLeaf ? std::max(leafCap, 6u) : leafCap, autoLeaf ? std::max(leafCap, 8u) : leafCap};
      for (int k = 0; k < 3 && best < 2; k++) {
        if (k > 0 && leaves[k] == leaves[k - 1]) continue;
        if (residency((uint64_t)T * 48) <= best) break;                    // the triangles alone do not fit the next step
        Bvh alt; buildBvh(fs.v.data(), T, leaves[k], alt, 0.25f);
        uint32_t cls = residency(footprint(alt));
        if (cls > best) { alt.buildSeconds += h->bvh.buildSeconds; h->bvh = std::move(alt); best = cls; }
      }
    }
  }
  h->st.bvhOnGpu = onGpu ? 1u : 0u; h->st.bvhBuildSeconds = h->bvh.buildSeconds;
  {
    // a scene that will be traversed from global memory gets its node pairs laid out by cache line
    const size_t lds = ((size_t)2 * h->bvh.numNodes + (size_t)3 * h->bvh.triOrder.size()) * 16;
    if (((f & PT_FLAG_NO_LDS_SCENE) || lds > 156 * 1024) && !(f & PT_FLAG_NO_TREELETS)) {
      const auto t0 = std::chrono::steady_clock::now();
      treeletLayout(h->bvh);
      h->st.bvhBuildSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  const Bvh& bvh = h->bvh;

  // triangles in leaf order: (v0, origId) (v2 - v0) (v1 - v0)
  uint32_t NT = (uint32_t)bvh.triOrder.size();
  std::vector<float> tris((size_t)NT * 12, 0.0f);
  for (uint32_t i = 0; i < NT; i++) {
    uint32_t o = bvh.triOrder[i];
    float* r = &tris[(size_t)i * 12];
    if (o == 0xFFFFFFFFu) { std::memcpy(&r[3], &o, 4); continue; }
    const float* p = &fs.v[(size_t)o * 9];
    r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; std::memcpy(&r[3], &o, 4);
    r[4] = p[6] - p[0]; r[5] = p[7] - p[1]; r[6] = p[8] - p[2];
    r[8] = p[3] - p[0]; r[9] = p[4] - p[1]; r[10] = p[5] - p[2];
  }
  std::vector<float> nrm((size_t)std::max(1u, T) * 12, 0.0f);
  for (uint32_t i = 0; i < T; i++) {
    float* r = &nrm[(size_t)i * 12]; const float* q = &fs.n[(size_t)i * 9];
    for (int k = 0; k < 3; k++) { r[4 * k] = q[3 * k]; r[4 * k + 1] = q[3 * k + 1]; r[4 * k + 2] = q[3 * k + 2]; }
    uint32_t m = fs.matIdx[i]; std::memcpy(&r[3], &m, 4);
  }
  std::vector<float> mats(std::max<size_t>(1, fs.mats.size()) * 4, 0.0f);
  for (size_t i = 0; i < fs.mats.size(); i++) {
    mats[4 * i] = fs.mats[i].emissive; mats[4 * i + 1] = fs.mats[i].color[0];
    mats[4 * i + 2] = fs.mats[i].color[1]; mats[4 * i + 3] = fs.mats[i].color[2];
  }
  const LightTable& lt = h->lights;
  std::vector<float> lights(std::max<size_t>(1, lt.tris.size()) * 12, 0.0f);
  for (size_t i = 0; i < lt.tris.size(); i++) std::memcpy(&lights[i * 12], &lt.tris[i], 44);
  std::vector<uint32_t> lf(lt.first), lc(lt.count); std::vector<float> la(lt.totalArea);
  if (lf.empty()) { lf.push_back(0); lc.push_back(0); la.push_back(0.0f); }

  h->st.numTriangles = T; h->st.numNodes = bvh.numNodes; h->st.bvhDepth = bvh.depth;
  h->st.numLightGroups = (uint32_t)lt.numGroups();
  h->maskDirty = true; h->texDirty = true;
  size_t ldsNeed = ((size_t)2 * bvh.numNodes + (size_t)3 * NT) * 16;
  DevScene& d = h->dsc;
  d.ldsBytes = (!(h->cfg.flags & PT_FLAG_NO_LDS_SCENE) && ldsNeed <= 156 * 1024) ? (uint32_t)ldsNeed : 0;
  h->st.sceneInLds = d.ldsBytes ? 1 : 0;
  if (h->cfg.device < 0) { h->haveScene = true; return; }
  auto up = [&](DevBuf& b, const void* src, size_t bytes) {
    b.alloc(bytes);
    HIPCHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  };
  HIPCHECK(hipStreamSynchronize(h->stream));
  if (d.ldsBytes) up(h->dNodes, bvh.nodes.data(), bvh.nodes.size() * 4);
  else {                   // traversal from global memory reads the 16-byte quantised nodes
    std::vector<uint32_t> q((size_t)std::max(1u, bvh.numNodes) * 4, 0u);
    quantizeNodes(bvh.nodes.data(), bvh.numNodes, q.data(), d.qBase, d.qScale);
    up(h->dNodes, q.data(), q.size() * 4);
  }
  up(h->dTris, tris.data(), tris.size() * 4);
  if (fs.v.empty()) h->dVerts.alloc(16);
  else up(h->dVerts, fs.v.data(), fs.v.size() * 4);                          // input order and bits: what the masks project
  h->sceneScale = sceneScale(fs);
  up(h->dNrm, nrm.data(), nrm.size() * 4);
  up(h->dMats, mats.data(), mats.size() * 4);
  up(h->dLights, lights.data(), lights.size() * 4);
  up(h->dLightFirst, lf.data(), lf.size() * 4);
  up(h->dLightCount, lc.data(), lc.size() * 4);
  up(h->dLightArea, la.data(), la.size() * 4);

  d.nodes = d.ldsBytes ? h->dNodes.as<float4>() : nullptr; d.qnodes = d.ldsBytes ? nullptr : h->dNodes.as<uint4>();
  d.tris = h->dTris.as<float4>(); d.nrm = h->dNrm.as<float4>();
  d.mats = h->dMats.as<float4>(); d.lights = h->dLights.as<float>();
  d.lightFirst = h->dLightFirst.as<uint32_t>(); d.lightCount = h->dLightCount.as<uint32_t>();
  d.lightArea = h->dLightArea.as<float>();
  d.numNodes = bvh.numNodes; d.numTris = NT; d.numOrigTris = T; d.rootRef = bvh.rootRef;
  d.numLightGroups = (uint32_t)lt.numGroups(); d.numMats = (uint32_t)fs.mats.size(); d.numLightTris = (uint32_t)lt.tris.size();
  for (int k = 0; k < 3; k++) { h->box.mn[k] = bvh.nodes[k]; h->box.mx[k] = bvh.nodes[4 + k]; }   // node 0 = root
  h->haveScene = true; h->sceneGeneration++;
  h->st.sceneBytes = (uint64_t)bvh.numNodes * (d.ldsBytes ? 32 : 16) + (uint64_t)NT * 48;
  h->st.numTriangles = T; h->st.numNodes = bvh.numNodes; h->st.bvhDepth = bvh.depth;
  h->st.numLightGroups = d.numLightGroups; h->st.sceneInLds = d.ldsBytes ? 1 : 0;
}

uint32_t numKinds(const PtHandle* h) { return h->dsc.numLightGroups + 1; }

// Whether the batches of this handle run the shading that traces the rays it spawns (kernels.h: Wavefront::fuse), as far
// as scene and configuration decide it: scene and shading tables small enough to share a workgroup's LDS, a tree the
// kernels' stacks hold, and some ray to spawn at all.
bool sceneFusable(const PtHandle* h) {
  const PtConfig& c = h->cfg;
  if (c.flags & (PT_FLAG_NO_FUSE | PT_FLAG_COUNT_VISITS)) return false;     // visits are counted by the traversal kernels proper
  const uint32_t tables = shadeTableBytes(h->dsc);
  if (!h->dsc.ldsBytes || !tables || h->dsc.ldsBytes + tables > PT_FUSE_LDS_BYTES) return false;
  const bool chains = c.mode == 1 && c.numIndirect > 0;
  // which chains' first rays hit is kept in the top eight bits of the view-ray record's tag word
  if (chains && (c.numIndirect > 8 || h->dsc.numMats >= 0xFFFFFFu)) return false;
  return chains || c.hasSky || (c.hasLights && h->dsc.numLightGroups);
}

void ensureWavefront(PtHandle* h) {
  const PtConfig& c = h->cfg;
  // Lean: every batch is one whole-sample launch plus traced shading, which use neither the view-ray queue nor the
  // shadow queues (more than a third of the state in mode 1), and the default batch is twice as large for it: a
  // batch costs about 0.1 ms of launches, gaps and tails whatever its size (chest1 1080p mode 1, ms per 1024 spp: 64 Mi
  // slots 64.0, 128 Mi 60.9, 256 Mi 59.0).
  const bool lean = sceneFusable(h) && !(c.flags & PT_FLAG_QUEUED_VIEW_RAYS);
  // Large batches: every launch ends with a tail in which the last waves finish alone, and a batch is ~10 launches.
  // Measured on the 1080p teapot frame (ms per 256 spp): 4 M slots 70, 16 M 31.6, 32 M 28.0, 64 M 25.1, 128 M 24.5,
  // 256 M 23.7, 512 M 24.0.
  // HBM has room: 128 B per slot in mode 0 without light groups (256 M slots = 34 GB), ~0.9 KB in mode 1 with
  // 5 indirect chains (64 M slots = 60 GB; chest1 1080p mode 1, ms per 64 spp: 32 M 15.1, 64 M 13.6).
  uint64_t want = c.batchSlots ? c.batchSlots : (c.mode == 1 ? (lean ? 1u << 27 : 1u << 26) : (1u << 28));
  uint64_t frame = (uint64_t)c.width * c.height;
  // no point in more slots than 256 samples of the whole frame; a multiple of 64 * PT_SEGS so that every block-task's
  // share of a full batch is whole waves and fits its output segment
  const uint64_t gran = 64ull * PT_SEGS;
  want = std::min<uint64_t>(want, frame * 256);
  uint32_t NI = c.mode == 1 ? c.numIndirect : 0;
  uint32_t lanes = 1 + NI;
  // radiance records: one per path slot and one per chain and bounce round (kernels.h: Wavefront::Lacc)
  const uint64_t recLanes = 1 + (uint64_t)NI * (c.maxDepth + 1);
  if (!c.batchSlots) {       // the default yields to the memory that is actually free (the caller may hold most of it)
    // decided once per shape of the wavefront state, not per render call
    if (h->wf.capSlots && h->wf.lanes == lanes && h->wfKinds == numKinds(h) && h->wfRounds == (c.mode == 1 ? c.maxDepth + 4 : 2) &&
        h->wfLean == lean) return;
    size_t freeB = 0, totalB = 0;
    HIPCHECK(hipMemGetInfo(&freeB, &totalB));
    uint64_t perSlot = 16ull * recLanes + (NI ? 8ull * lanes : 0) + (uint64_t)std::max(1u, NI) * (32 + (NI ? 32 : 0));
    if (!lean) perSlot += 16 + (uint64_t)std::max(1u, NI) * (48ull * numKinds(h) + 16 + 16);      // (the last 16: hitOrder)
    uint64_t held = h->wf.capSlots ? (uint64_t)h->wf.capSlots * perSlot : 0;       // what this handle would give back
    while (want > gran && want * perSlot > (freeB + held) / 2) want /= 2;
  }
  while (want > gran && want * recLanes > 0xFFFFFFFFull) want /= 2;          // records are indexed with 32 bits
  uint32_t capSlots = (uint32_t)std::min<uint64_t>(((want + gran - 1) / gran) * gran, 0x40000000ull);
  uint64_t capQueue = (uint64_t)capSlots * std::max(1u, NI);
  if (capQueue > 0x7FFFFFFFull || (uint64_t)capSlots * recLanes > 0xFFFFFFFFull) fail(PT_E_INVALID, "batchSlots * numIndirect * (maxDepth + 1) too large");
  uint32_t kinds = numKinds(h), rounds = c.mode == 1 ? c.maxDepth + 4 : 2;
  Wavefront& w = h->wf;
  if (w.capSlots == capSlots && w.capQueue == (uint32_t)capQueue && w.lanes == lanes && h->wfKinds == kinds && h->wfRounds == rounds &&
      h->wfLean == lean) return;
  HIPCHECK(hipStreamSynchronize(h->stream));
  h->bSlotQ.alloc(lean ? 16 : (uint64_t)capSlots * 16);
  uint64_t capHit = capQueue + (uint64_t)PT_SEGS * PT_HIT_SLACK;
  h->bHitA.alloc(capHit * 16); h->bHitB.alloc(capHit * 16);
  h->bHitOrder.alloc(lean ? 16 : capHit * 16);             // the order and three words per entry for the sort that makes it
  // indirect rays only; as large as the hit arrays, so that the fused shading can use the two pairs alternately as the
  // hit queues of successive bounce rounds (kernels.h: Wavefront::hitOutA)
  h->bRayO.alloc(NI ? capHit * 16 : 16); h->bRayD.alloc(NI ? capHit * 16 : 16);
  h->bLacc.alloc(recLanes * capSlots * 16);
  h->bRng.alloc(NI ? (uint64_t)lanes * capSlots * 8 : 16);
  h->bShO.alloc(lean ? 16 : (uint64_t)kinds * capQueue * 16);
  h->bShD.alloc(lean ? 16 : (uint64_t)kinds * capQueue * 16);
  h->bShW.alloc(lean ? 16 : (uint64_t)kinds * capQueue * 16);
  h->bShX.alloc(lean ? 16 : capQueue * 16);
  // queue bookkeeping: stage s owns 2 + PT_MAX_KINDS queues: [0] the hit queue its extend fills, [1] the ray queue
  // and [2 + kind] the shadow queues its shade fills
  const uint32_t perStage = 2 + PT_MAX_KINDS;
  h->bCounters.alloc((size_t)rounds * perStage * PT_QMETA_WORDS * 4 + (PT_TASK_WORDS + PT_POOL_WORDS) * 4);
  h->bMetaDesc.alloc((size_t)rounds * perStage * sizeof(QMeta));
  h->metas.resize((size_t)rounds * perStage);
  uint32_t* words = h->bCounters.as<uint32_t>();
  for (size_t m = 0; m < h->metas.size(); m++) {
    uint32_t* p = words + m * PT_QMETA_WORDS;
    h->metas[m] = QMeta{p, p + PT_SEGS, p + 2 * PT_SEGS + 2};
  }
  HIPCHECK(hipMemset(h->bCounters.p, 0, h->bCounters.bytes));
  HIPCHECK(hipMemcpy(h->bMetaDesc.p, h->metas.data(), h->metas.size() * sizeof(QMeta), hipMemcpyHostToDevice));
  w.taskCounters = words + h->metas.size() * PT_QMETA_WORDS;
  w.runPool = (h->cfg.flags & PT_FLAG_STATIC_RUNS) ? nullptr : w.taskCounters + PT_TASK_WORDS;
  w.slotQ = h->bSlotQ.as<float4>(); w.hitA = h->bHitA.as<float4>(); w.hitB = h->bHitB.as<float4>();
  w.hitOrder = lean ? nullptr : h->bHitOrder.as<uint32_t>();
  w.rayO = h->bRayO.as<float4>(); w.rayD = h->bRayD.as<float4>();
  w.Lacc = h->bLacc.as<float4>(); w.rng = h->bRng.as<uint2>();
  w.shO = h->bShO.as<float4>(); w.shD = h->bShD.as<float4>(); w.shW = h->bShW.as<float4>(); w.shX = h->bShX.as<float4>();
  w.stats = h->bStats.as<unsigned long long>();
  w.capSlots = capSlots; w.capQueue = (uint32_t)capQueue; w.capHit = (uint32_t)capHit; w.lanes = lanes;
  w.chainStride = NI * capSlots; w.recRounds = c.maxDepth + 1;
  w.rayCap = NI ? (uint32_t)capHit : 0u; w.hitOutA = w.hitOutB = nullptr; w.fuse = 0; w.chainMask = 0;
  h->wfKinds = kinds; h->wfRounds = rounds; h->wfLean = lean;
}

// Shadow rays of one vertex generation: light groups by Scene.trace, sun and sky (one queue) by Scene.hitsNothing.
void launchShadowKinds(PtHandle* h, const LaunchCfg& lc, const DevCfg& dc, const Wavefront& wf, bool primary) {
  int kind = 0;
  if (dc.hasLights)
    for (uint32_t li = 0; li < h->dsc.numLightGroups; li++, kind++) {
      Timed t(h, PT_K_CONNECT); launchConnect(lc, h->dsc, dc, wf, kind, h->lights.totalArea[li]); t.done();
    }
  if (dc.hasSky)
    { Timed t(h, PT_K_OCCLUDED); launchOccluded(lc, h->dsc, dc, wf, kind, primary); t.done(); }
}

// One wavefront batch: npix pixels x k samples.  Counter set s belongs to stage s: 0 = raygen, 1 = shading of the
// view-ray hits, 2 + r = shading of the indirect hits of round r; a stage reads the ray queue its predecessor filled.
// Where a batch keeps its state: the handle's own buffers (ensureWavefront) or the static scratch array.
struct WfState { Wavefront wf; const QMeta* hostMetas; const QMeta* devMetas; ViewMasks masks; };

// Whether this batch runs the shading variant that traces the rays it spawns (kernels.h: Wavefront::fuse): scene and
// shading tables small enough to share a shade workgroup's LDS and, for the chains, ray arrays as large as the hit arrays.
uint32_t fuseBits(const PtHandle* h, const DevCfg& dc, const LaunchCfg& lc, const Wavefront& wf) {
  if (!sceneFusable(h) || lc.countVisits) return 0;
  const bool chains = dc.mode == 1 && dc.numIndirect > 0;
  if (chains && wf.rayCap < wf.capHit) return 0;
  return 1u;
}

// Whether a window of npix pixels x spp samples is rendered a pixel per lane, the samples added up in registers
// (k_pixel_traced): mode 0 on a scene whose shading traces its own rays, enough pixels to fill the chip.
bool pixelSums(const PtHandle* h, const DevCfg& dc, const LaunchCfg& lc, uint64_t npix, uint32_t spp) {
  if (dc.mode != 0 || lc.countVisits || !sceneFusable(h)) return false;
  if (h->cfg.flags & (PT_FLAG_QUEUED_VIEW_RAYS | PT_FLAG_NO_PIXEL_SUMS)) return false;
  return pixelSumsPay(npix, spp);
}

const uint32_t* activeColumns(PtHandle* h, const DevCfg& dc, const LaunchCfg& lc, const PixMap& pm, const ViewMasks& masks);

void renderBatch(PtHandle* h, const DevCfg& dc, const LaunchCfg& lc, PixMap pm, const WfState& ws) {
  Wavefront wf = ws.wf;
  wf.divCap = makeFastDiv(wf.capSlots); wf.round = 0;
  wf.fuse = fuseBits(h, dc, lc, wf);
  wf.chainMask = wf.fuse && dc.mode == 1 ? 1u : 0u;          // (sceneFusable has made sure that the mask fits)
  // The hits of a scene in global memory are shaded in the order of their cells (kernels.hip: k_hit_order) when the
  // geometry is about as fine as the pixels: then neighbouring pixels see hits at unrelated depths and the order of the
  // pixels is no order in space.  Under coarser geometry the hits of neighbouring pixels are neighbours already, and
  // the sort and the gathering of the hits only cost (the teapot forced into global memory, 1080p: 40.0 -> 50.7 ms per
  // 256 spp; the 10 M-triangle soup at 3840x2160: 65.5 -> 59.7 ms per 4 spp).
  const bool fine = (uint64_t)h->dsc.numOrigTris * 2 >= (uint64_t)h->cfg.width * h->cfg.height;
  if (h->dsc.ldsBytes || (h->cfg.flags & PT_FLAG_NO_HIT_ORDER) || !(fine || (h->cfg.flags & PT_FLAG_HIT_ORDER))) wf.hitOrder = nullptr;
  wf.regionPix = 0; wf.divRegion = makeFastDiv(1);
  if (wf.hitOrder && !lc.countVisits) {
    // the view rays and the hits of a segment's pixels at all k samples must fit the segment
    const uint32_t R = (pm.npix + PT_SEGS - 1) / PT_SEGS;
    if ((uint64_t)R * pm.k <= wf.capSlots / PT_SEGS) { wf.regionPix = R; wf.divRegion = makeFastDiv(R); }      // (capHit > capSlots)
  }
  const uint32_t perStage = 2 + PT_MAX_KINDS;
  const int kinds = (int)((dc.hasLights ? h->dsc.numLightGroups : 0) + (dc.hasSky ? 1 : 0));
  const QMeta* desc = ws.devMetas;
  // stage s: 0 = raygen, 1 = view-ray hits, 2 + r = indirect hits of round r.  Per stage: extend reads the previous
  // stage's ray queue and fills the hit queue [0]; shade reads it and fills the ray queue [1] and shadow queues
  // [2 + kind]; a scan after each producer makes its queues readable.
  auto bind = [&](uint32_t stage) {
    wf.rayIn = ws.hostMetas[(size_t)(stage - 1) * perStage + 1];
    wf.hitQ = ws.hostMetas[(size_t)stage * perStage];
    wf.rayOut = ws.hostMetas[(size_t)stage * perStage + 1];
    for (int k = 0; k < PT_MAX_KINDS; k++) wf.sh[k] = ws.hostMetas[(size_t)stage * perStage + 2 + k];
  };
  auto scan = [&](size_t first, int n) {
    Timed t(h, PT_K_OTHER); launchScan(lc, desc + first, n, wf.taskCounters); t.done();
  };
  auto stage = [&](uint32_t s, bool primary) {
    bind(s);
    wf.round = primary ? 0u : s - 2u;
    { Timed t(h, PT_K_EXTEND); launchExtend(lc, h->dsc, dc, pm, h->acc, wf, primary); t.done(); }
    scan((size_t)s * perStage, 1);
    if (wf.hitOrder) { Timed t(h, PT_K_OTHER); launchHitOrder(lc, h->dsc, wf); t.done(); }
    { Timed t(h, PT_K_SHADE); launchShade(lc, h->dsc, dc, pm, h->acc, wf, primary); t.done(); }
    scan((size_t)s * perStage + 1, 1 + kinds);
    launchShadowKinds(h, lc, dc, wf, primary);
  };
  // Fused shading: the shade of stage s reads the hit queue of stage s and appends the hits of the rays it traced to the
  // hit queue of stage s + 1, whose records lie in the other pair of arrays (odd stages: hitA/hitB, even: rayO/rayD).
  float4* const pairA[2] = {ws.wf.rayO, ws.wf.hitA};
  float4* const pairB[2] = {ws.wf.rayD, ws.wf.hitB};
  auto fusedStage = [&](uint32_t s, bool primary) {
    bind(s);
    wf.round = primary ? 0u : s - 2u;
    wf.hitA = pairA[s & 1u]; wf.hitB = pairB[s & 1u];
    wf.hitOutA = pairA[(s + 1u) & 1u]; wf.hitOutB = pairB[(s + 1u) & 1u];
    // continuation rays exist in mode 1 only (mode 0 has two stages of queue descriptors and nothing to append)
    if (dc.mode == 1) wf.rayOut = ws.hostMetas[(size_t)(s + 1) * perStage];              // the hit queue of the next stage
    if (primary) {
      Timed t(h, PT_K_EXTEND); launchExtend(lc, h->dsc, dc, pm, h->acc, wf, true); t.done();
      scan((size_t)s * perStage, 1);
    }
    { Timed t(h, primary ? PT_K_SHADE : PT_K_SHADE_TRACED); launchShade(lc, h->dsc, dc, pm, h->acc, wf, primary); t.done(); }
    // (no scan of what it appended: the traced shading of chain hits takes its queue segment by segment)
  };
  h->st.raysPrimary += (uint64_t)pm.npix * pm.k; h->st.samples += (uint64_t)pm.npix * pm.k;
  const bool wholeSample = wf.fuse && !(h->cfg.flags & PT_FLAG_QUEUED_VIEW_RAYS);
  if (wholeSample && pixelSums(h, dc, lc, pm.npix, pm.k)) {
    // mode 0: a pixel per lane, its samples added up in registers -- no radiance records, no accumulate launch
    { Timed t(h, PT_K_PIXEL_TRACED); launchPixelTraced(lc, h->box, ws.masks, h->dsc, dc, pm, h->acc, wf); t.done(); }
    HIPCHECK(hipGetLastError());
    return;
  }
  uint32_t nslots = pm.npix * pm.k;
  if (wholeSample) {
    // raygen, the view rays and the shading of their hits in one launch; it appends what stage 1's shade would
    bind(1);
    wf.hitOutA = pairA[0]; wf.hitOutB = pairB[0];
    if (dc.mode == 1) wf.rayOut = ws.hostMetas[(size_t)2 * perStage];
    ViewMasks vm = ws.masks;
    if (ws.hostMetas == h->metas.data()) vm.activeCols = activeColumns(h, dc, lc, pm, ws.masks);      // (not for a caller's state: it may allocate)
    { Timed t(h, PT_K_SAMPLE_TRACED); launchSampleTraced(lc, h->box, vm, h->dsc, dc, pm, h->acc, nslots, wf); t.done(); }
    if (dc.mode == 1) fusedStage(2, false);                // one launch goes through all bounce rounds
    { Timed t(h, PT_K_ACCUM); launchAccumulate(lc, ws.masks, dc, pm, h->acc, wf, h->dsc); t.done(); }
    HIPCHECK(hipGetLastError());
    return;
  }
  // the view rays generated and traced by one traversal launch (not with the visit counters, which count in extend)
  // (scenes in global memory keep the queue: their extend hands freed lanes the next ray, which needs one)
  const bool rayExt = !wf.fuse && h->dsc.ldsBytes && !lc.countVisits && !(h->cfg.flags & PT_FLAG_NO_FUSE);
  if (rayExt) {
    bind(1);
    ViewMasks vm = ws.masks;
    if (ws.hostMetas == h->metas.data()) vm.activeCols = activeColumns(h, dc, lc, pm, ws.masks);
    { Timed t(h, PT_K_RAYGEN_EXTEND); launchRaygenExtend(lc, h->box, vm, h->dsc, dc, pm, h->acc, nslots, wf); t.done(); }
    scan((size_t)perStage, 1);
    if (dc.mode == 0 && envDirect(dc, h->dsc, true)) {
      { Timed t(h, PT_K_SHADE_OCCLUDED); launchShadeOccluded(lc, h->dsc, dc, pm, h->acc, wf); t.done(); }
    } else {
      { Timed t(h, PT_K_SHADE); launchShade(lc, h->dsc, dc, pm, h->acc, wf, true); t.done(); }
      scan((size_t)perStage + 1, 1 + kinds);
      launchShadowKinds(h, lc, dc, wf, true);
      if (dc.mode == 1)
        for (uint32_t round = 0; round <= dc.maxDepth; round++) stage(round + 2, false);
    }
    { Timed t(h, PT_K_ACCUM); launchAccumulate(lc, ws.masks, dc, pm, h->acc, wf, h->dsc); t.done(); }
    HIPCHECK(hipGetLastError());
    return;
  }
  wf.rayOut = ws.hostMetas[1];
  { Timed t(h, PT_K_RAYGEN); launchRaygen(lc, h->box, ws.masks, dc, pm, h->acc, nslots, wf); t.done(); }
  scan(1, 1);
  if (wf.fuse) {
    fusedStage(1, true);
    if (dc.mode == 1) fusedStage(2, false);
  // (A scene in global memory keeps shade and the sun / sky kernel apart even when a vertex spawns nothing else: that
  // kernel's lanes take the next ray when theirs is decided, which the kernel that shades its hits does not do --
  // 10 M-triangle soup without its light, 3840x2160, 4 spp: 17.0 + 1.0 ms against 24.7 ms.)
  } else {
    stage(1, true);
    if (dc.mode == 1)      // depth 0..maxDepth can still spawn a ray (the check is depth > maxDepth, rendering.d:380)
      for (uint32_t round = 0; round <= dc.maxDepth; round++) stage(round + 2, false);
  }
  { Timed t(h, PT_K_ACCUM); launchAccumulate(lc, ws.masks, dc, pm, h->acc, wf, h->dsc); t.done(); }
  HIPCHECK(hipGetLastError());
}

#ifndef PT_STRIP_TILES
#define PT_STRIP_TILES 32u      // strip width in tiles; 0: tiles row-major over the whole width (A/B builds)
#endif
// Slots are laid out so that a wave's 64 paths are a compact block of pixels (8x8, or 16x4 when only multiples of
// 4 rows are available) whenever the chunk is made of whole rows; otherwise in scanline order.
void chooseTiling(PixMap& pm, uint32_t W, bool wholeRows, bool strips) {
  pm.tileW = 0; pm.tilesPerRow = 1;
  if (wholeRows && pm.npix % W == 0 && pm.lpBase % W == 0) {
    uint32_t rows = pm.npix / W;
    if (rows % 8 == 0 && W % 8 == 0) pm.tileW = 8;
    else if (rows % 4 == 0 && W % 16 == 0) pm.tileW = 16;
    if (pm.tileW) pm.tilesPerRow = W / pm.tileW;
  }
  pm.divTilesPerRow = makeFastDiv(pm.tilesPerRow);
  // tiles numbered by vertical strips (kernels.h: PixMap) when the batch is wider than one and the scene is traversed
  // from global memory (with the scene in LDS the order only moves work between workgroups: chest1 +4 %, teapot +5 %)
  pm.stripW = 0; pm.stripTiles = 1; pm.nFull = 0; pm.lastW = 1;
  const uint32_t stripTiles = PT_STRIP_TILES;      // (soup 4K 4 spp, ms: row-major 67.5; strips of 8 / 16 / 32 / 64 / 120 tiles 66.0 / 66.3 / 65.7 / 65.6 / 66.0)
  if (strips && pm.tileW && stripTiles && pm.tilesPerRow > stripTiles) {
    const uint32_t tileRows = pm.npix / W / (64u / pm.tileW);
    pm.stripW = stripTiles; pm.stripTiles = pm.stripW * tileRows;
    pm.nFull = pm.tilesPerRow / pm.stripW; pm.lastW = std::max(1u, pm.tilesPerRow - pm.nFull * pm.stripW);
  }
  pm.divStripTiles = makeFastDiv(pm.stripTiles); pm.divStripW = makeFastDiv(std::max(1u, pm.stripW)); pm.divLastW = makeFastDiv(pm.lastW);
  pm.columns = pm.npix % 64u == 0 ? pm.npix / 64u : 0u; pm.divColumns = makeFastDiv(std::max(1u, pm.columns));
}

// The bitmap of the batch's columns that hold anything but constant sky, for the whole-sample kernel (kernels.h:
// ViewMasks::activeCols): built on the handle's stream when the batch geometry or the masks have changed since the
// last batch; a frame rendered in batches of the same pixels builds it once.
const uint32_t* activeColumns(PtHandle* h, const DevCfg& dc, const LaunchCfg& lc, const PixMap& pm, const ViewMasks& masks) {
  if (!masks.constantSky || !pm.columns) return nullptr;
  PixMap key = pm; key.k = 0;
  if (!h->colValid || h->colMaskGen != h->maskGen || std::memcmp(&key, &h->colKey, sizeof(PixMap)) != 0) {
    h->dActiveCols.alloc(((size_t)pm.columns + 31) / 32 * 4);
    launchActiveColumns(lc, masks, dc, pm, h->dActiveCols.as<uint32_t>());
    h->colKey = key; h->colMaskGen = h->maskGen; h->colValid = true;
  }
  return h->dActiveCols.as<uint32_t>();
}

// The view-ray masks for the current scene and camera, built on the GPU (mask_gpu.h) on the handle's stream; the call
// waits for that stream once, to learn whether the masks are usable at all (no triangle behind the camera plane).
void buildMasks(PtHandle* h) {
  const auto tMask = std::chrono::steady_clock::now();
  h->maskUsable = false;
  if (!(h->cfg.flags & PT_FLAG_NO_COVERAGE_MASK)) {
    const uint32_t W = h->cfg.width, H = h->cfg.height, T = (uint32_t)h->flat.numTriangles();
    const size_t words = ((size_t)W * H + 31) / 32;
    h->dMask.alloc(words * 4); h->dSkyMask.alloc(words * 4);
    h->dMaskRows.alloc(maskRowWords(W, H) * 4); h->dMaskBig.alloc(std::max<size_t>(T, 1) * 4); h->dMaskFlags.alloc(8);
    bool usable = false;
    HIPCHECK(buildCoverageMaskGpu(h->stream, h->dVerts.as<float>(), T, h->cam, W, H, h->sceneScale, h->dMaskRows.as<uint32_t>(),
                                  h->dMaskBig.as<uint32_t>(), h->dMaskFlags.as<uint32_t>(), h->dMask.as<uint32_t>(),
                                  h->dSkyMask.as<uint32_t>(), &usable));
    h->maskUsable = usable;
  }
  h->maskDirty = false; h->maskGen++;
  h->st.maskBuildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tMask).count();
}

// external: the batch state to use instead of the handle's own (the caller has laid it out); nothing is allocated,
// built on the host or timed with events on that path.
void renderPixels(PtHandle* h, PixMap base, uint64_t totalPix, uint32_t spp, const WfState* external = nullptr,
                  uint32_t maxGroups = 0) {
  if (!h->haveScene) fail(PT_E_STATE, "no scene: call ptgpu_upload_scene or ptgpu_load_thmodel first");
  if (!h->haveCamera) fail(PT_E_STATE, "no camera: call ptgpu_set_camera or ptgpu_camera_look_at first");
  if (totalPix == 0 || spp == 0) return;
  if (!external) ensureWavefront(h);
  if (!external && h->maskDirty) buildMasks(h);
  DevCfg dc = makeDevCfg(h); LaunchCfg lc = makeLaunchCfg(h);
  lc.maxGroups = maxGroups;
  WfState own{h->wf, h->metas.data(), h->bMetaDesc.as<QMeta>(), h->masks()};
  const WfState& ws = external ? *external : own;
  Timed span(h, KIND_RENDER, true);
  uint32_t W = h->cfg.width;
  bool wholeRows = (base.bandPix != 0 || base.offset % W == 0) && totalPix % W == 0;
  uint64_t cap = ws.wf.capSlots;
  // (a window whose samples are added up in registers keeps no state per sample: one batch whatever its size)
  if (pixelSums(h, dc, lc, totalPix, spp)) cap = std::max<uint64_t>(cap, totalPix * spp);
  uint64_t chunk = std::min<uint64_t>(cap, totalPix);
  if (wholeRows && chunk < totalPix && chunk >= 8ull * W) chunk -= chunk % (8ull * W);     // chunks of whole 8-row strips
  base.divBand = makeFastDiv(base.bandPix);
  // ---- one small batch: replay it from a graph (launch-bound otherwise: ~13 launches for ~100 us of kernels) ----
  const uint64_t nslotsAll = totalPix * spp;
  const bool oneBatch = chunk == totalPix && nslotsAll <= cap;
  if (!external && oneBatch && nslotsAll <= PT_GRAPH_MAX_SLOTS &&
      !(h->cfg.flags & (PT_FLAG_TIME_KERNELS | PT_FLAG_COUNT_VISITS | PT_FLAG_NO_GRAPH))) {
    PixMap pm = base; pm.lpBase = 0; pm.npix = (uint32_t)totalPix; pm.k = spp; pm.divNpix = makeFastDiv(pm.npix);
    chooseTiling(pm, W, wholeRows, h->dsc.ldsBytes == 0);
    std::vector<uint8_t> key;
    auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; key.insert(key.end(), b, b + n); };
    // everything the launches of a batch take as arguments (the structs are assembled field by field from zeroed
    // memory or copied whole, so equal contents mean equal bytes)
    put(&pm, sizeof(pm)); put(&dc, sizeof(dc)); put(&ws.wf, sizeof(ws.wf)); put(&ws.masks, sizeof(ws.masks));
    put(&h->acc, sizeof(h->acc)); put(&h->dsc, sizeof(h->dsc)); put(&h->box, sizeof(h->box)); put(&h->stream, sizeof(h->stream));
    put(&h->sceneGeneration, sizeof(h->sceneGeneration)); put(&h->cfg.flags, sizeof(h->cfg.flags));
    auto countBatch = [&]() { h->st.raysPrimary += nslotsAll; h->st.samples += nslotsAll; };
    if (h->graphExec && key == h->graphKey) {
      HIPCHECK(hipGraphLaunch(h->graphExec, h->stream));
      countBatch(); span.done(); return;
    }
    // Captured only on the stream the handle made itself: a caller's stream may be the legacy null stream, which cannot
    // be captured, or be under a capture of the caller's own.  When the capture cannot begin, the batch runs as usual.
    bool capturing = false;
    if (key == h->graphSeen && h->ownStream) {     // second call of this shape: capture it
      if (h->graphExec) { (void)hipGraphExecDestroy(h->graphExec); h->graphExec = nullptr; h->graphKey.clear(); }
      if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
      else { (void)hipGetLastError(); h->graphSeen.clear(); }
    }
    if (capturing) {
      hipGraph_t graph = nullptr;
      try {
        h->noEvents = true;
        renderBatch(h, dc, lc, pm, ws);
        h->noEvents = false;
      } catch (...) {
        h->noEvents = false; h->graphSeen.clear();
        (void)hipStreamEndCapture(h->stream, &graph); if (graph) (void)hipGraphDestroy(graph);
        throw;
      }
      hipError_t ce = hipStreamEndCapture(h->stream, &graph);
      if (ce != hipSuccess) { h->graphSeen.clear(); fail(PT_E_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce)); }
      hipError_t ge = hipGraphInstantiate(&h->graphExec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ge != hipSuccess) { h->graphExec = nullptr; h->graphSeen.clear(); fail(PT_E_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ge)); }
      h->graphKey = key;
      HIPCHECK(hipGraphLaunch(h->graphExec, h->stream));
      span.done(); return;                     // (renderBatch has counted the batch while it was captured)
    }
    h->graphSeen = key;                        // first call: run it as usual (this also sets kernel attributes)
  }
  for (uint64_t lp0 = 0; lp0 < totalPix; lp0 += chunk) {
    uint32_t npix = (uint32_t)std::min<uint64_t>(chunk, totalPix - lp0);
    uint32_t kmax = std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>(cap / npix, 0xFFFFFFFFull));
    // samples added up in registers: a launch of 64 samples per pixel ends with waves finishing their last column alone,
    // and longer launches only make that tail longer (cornell 4K, ms per sample: 16 spp 0.448, 32 0.419, 64 0.412, 256 0.414)
    if (pixelSums(h, dc, lc, totalPix, spp)) kmax = std::min(kmax, 64u);
    for (uint32_t s0 = 0; s0 < spp; s0 += kmax) {
      PixMap pm = base; pm.lpBase = (uint32_t)lp0; pm.npix = npix; pm.k = std::min(kmax, spp - s0);
      pm.divNpix = makeFastDiv(npix);
      chooseTiling(pm, W, wholeRows, h->dsc.ldsBytes == 0);
      renderBatch(h, dc, lc, pm, ws);
    }
  }
  span.done();
}

void checkWindow(const PtHandle* h, uint32_t off, uint32_t count) {
  uint64_t frame = (uint64_t)h->cfg.width * h->cfg.height;
  if ((uint64_t)off + count > frame) fail(PT_E_INVALID, "pixel window [" + std::to_string(off) + ", +" + std::to_string(count) + ") exceeds the frame");
}

template <typename F> int guarded(F&& f) {
  try { f(); return PT_OK; }
  catch (const ApiError& e) { g_err = e.msg; return e.code; }
  catch (const IoError& e) { g_err = e.what(); return PT_E_IO; }
  catch (const std::bad_alloc&) { g_err = "out of host memory"; return PT_E_NOMEM; }
  catch (const std::exception& e) { g_err = e.what(); return PT_E_INVALID; }
}

}  // namespace

extern "C" {

const char* ptgpu_last_error(void) { return g_err.c_str(); }

int ptgpu_create(const PtConfig* cfg, PtHandle** out) {
  if (out) *out = nullptr;
  PtHandle* h = nullptr;
  int rc = guarded([&]() {
    if (!cfg || !out) fail(PT_E_INVALID, "null argument");
    PtConfig c = *cfg; normalizeConfig(c); validateConfig(c);
    if (c.device < 0) { h = new PtHandle(); h->cfg = c; h->accBytes = (uint64_t)c.width * c.height * 16; return; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) fail(PT_E_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (c.device >= ndev) fail(PT_E_INVALID, "device ordinal " + std::to_string(c.device) + " out of range");
    h = new PtHandle(); h->cfg = c;
    useDevice(h);
    hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, c.device));
    h->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->ownStream = true;
    h->accBytes = (uint64_t)c.width * c.height * 16;
    h->accOwn.alloc(h->accBytes); h->acc = h->accOwn.as<float4>();
    HIPCHECK(hipMemset(h->acc, 0, h->accBytes));
    h->bStats.alloc(PT_STAT_REPLICAS * PT_STAT_STRIDE * 8); HIPCHECK(hipMemset(h->bStats.p, 0, h->bStats.bytes));
    HIPCHECK(hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      HIPCHECK(hipEventCreateWithFlags(&h->evTone[k], hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&h->evCopy[k], hipEventDisableTiming));
    }
    h->scratchDev = (unsigned char*)scratchOnDevice(c.device);
    h->metasNoAlloc.resize((size_t)(64 + 4) * (2 + PT_MAX_KINDS));        // maxDepth <= 64 (validateConfig)
  });
  if (rc != PT_OK) {                        // give back the streams and events made so far (ptgpu_destroy keeps g_err's message apart)
    if (h) { const std::string why = g_err; (void)ptgpu_destroy(h); g_err = why; }
    return rc;
  }
  *out = h; return PT_OK;
}

int ptgpu_destroy(PtHandle* h) {
  if (!h) return PT_OK;
  return guarded([&]() {
    if (h->cfg.device < 0) { delete h; return; }
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (EventPair& p : h->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t e : h->freeEvents) (void)hipEventDestroy(e);
    if (h->graphExec) (void)hipGraphExecDestroy(h->graphExec);
    if (h->copyStream) { (void)hipStreamSynchronize(h->copyStream); (void)hipStreamDestroy(h->copyStream); }
    for (int k = 0; k < 2; k++) { if (h->evTone[k]) (void)hipEventDestroy(h->evTone[k]); if (h->evCopy[k]) (void)hipEventDestroy(h->evCopy[k]); }
    if (h->ownStream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
  });
}

int ptgpu_set_config(PtHandle* h, const PtConfig* cfg) {
  return guarded([&]() {
    if (!h || !cfg) fail(PT_E_INVALID, "null argument");
    PtConfig c = *cfg; normalizeConfig(c); validateConfig(c);
    if (c.device != h->cfg.device || c.width != h->cfg.width || c.height != h->cfg.height || c.batchSlots != h->cfg.batchSlots)
      fail(PT_E_INVALID, "device, width, height and batchSlots are fixed at ptgpu_create");
    bool rebuild = c.maxLeafTris != h->cfg.maxLeafTris ||
                   ((c.flags ^ h->cfg.flags) & (PT_FLAG_NO_LDS_SCENE | PT_FLAG_GPU_BVH | PT_FLAG_HOST_BVH | PT_FLAG_NO_TREELETS));
    if ((c.flags ^ h->cfg.flags) & PT_FLAG_NO_COVERAGE_MASK) h->maskDirty = true;
    h->cfg = c;
    if (rebuild && h->haveScene) { if (h->cfg.device >= 0) useDevice(h); uploadScene(h); }
  });
}

int ptgpu_upload_scene(PtHandle* h, const float* v, const float* n, const uint32_t* matIdx, const PtMaterial* mats,
                       uint32_t T, uint32_t M) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (T && (!v || !n || !matIdx)) fail(PT_E_INVALID, "null triangle array");
    if (M == 0 || !mats) fail(PT_E_INVALID, "at least one material is required");
    if (h->cfg.device >= 0) useDevice(h);
    FlatScene fs;
    fs.v.assign(v, v + (size_t)T * 9); fs.n.assign(n, n + (size_t)T * 9);
    fs.matIdx.assign(matIdx, matIdx + T); fs.mats.assign(mats, mats + M); fs.matNames.assign(M, "");
    h->flat = std::move(fs);
    uploadScene(h);
  });
}

int ptgpu_load_thmodel(PtHandle* h, const char* path, int materialTable, const float* modelMatrix) {
  return guarded([&]() {
    if (!h || !path) fail(PT_E_INVALID, "null argument");
    if (materialTable != 0 && materialTable != 1) fail(PT_E_INVALID, "materialTable must be 0 or 1");
    if (h->cfg.device >= 0) useDevice(h);
    ModelData model; LoadFile(path, model);
    FlatScene fs; flattenModel(model, materialTable, modelMatrix, fs);
    h->flat = std::move(fs);
    uploadScene(h);
  });
}

int ptgpu_get_scene(PtHandle* h, float* v, float* n, uint32_t* matIdx, PtMaterial* mats, uint32_t* numMaterials) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    const FlatScene& fs = h->flat;
    if (v) std::memcpy(v, fs.v.data(), fs.v.size() * 4);
    if (n) std::memcpy(n, fs.n.data(), fs.n.size() * 4);
    if (matIdx) std::memcpy(matIdx, fs.matIdx.data(), fs.matIdx.size() * 4);
    if (mats) std::memcpy(mats, fs.mats.data(), fs.mats.size() * sizeof(PtMaterial));
    if (numMaterials) *numMaterials = (uint32_t)fs.mats.size();
  });
}

int ptgpu_get_light_group(PtHandle* h, uint32_t g, float* rows11, uint32_t cap, uint32_t* count, float* totalArea) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    if (g >= h->lights.numGroups()) fail(PT_E_INVALID, "light group out of range");
    uint32_t c = h->lights.count[g];
    if (rows11) for (uint32_t i = 0; i < c && i < cap; i++) std::memcpy(rows11 + 11 * (size_t)i, &h->lights.tris[h->lights.first[g] + i], 44);
    if (count) *count = c;
    if (totalArea) *totalArea = h->lights.totalArea[g];
  });
}

int ptgpu_get_bvh(PtHandle* h, float* nodes8, uint32_t* triOrder) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    if (nodes8 && h->dsc.ldsBytes == 0) {       // the boxes the kernels traverse: on the grid
      std::vector<uint32_t> q((size_t)std::max(1u, h->bvh.numNodes) * 4, 0u);
      float base[3], scale[3];
      quantizeNodes(h->bvh.nodes.data(), h->bvh.numNodes, q.data(), base, scale);
      dequantizeNodes(q.data(), h->bvh.numNodes, base, scale, nodes8);
    } else if (nodes8) std::memcpy(nodes8, h->bvh.nodes.data(), h->bvh.nodes.size() * 4);
    if (triOrder) std::memcpy(triOrder, h->bvh.triOrder.data(), h->bvh.triOrder.size() * 4);
  });
}

int ptgpu_set_camera(PtHandle* h, const float pos[3], const float dir[3], const float right[3], const float up[3],
                     float tanFov, float aspect) {
  return guarded([&]() {
    if (!h || !pos || !dir || !right || !up) fail(PT_E_INVALID, "null argument");
    h->cam.pos = v3(pos[0], pos[1], pos[2]); h->cam.dir = v3(dir[0], dir[1], dir[2]);
    h->cam.right = v3(right[0], right[1], right[2]); h->cam.up = v3(up[0], up[1], up[2]);
    h->cam.tanFov = tanFov; h->cam.aspect = aspect; h->haveCamera = true; h->maskDirty = true;
  });
}

int ptgpu_camera_look_at(PtHandle* h, float fovDegrees, float aspect, const float from[3], const float to[3],
                         const float up[3], float* out14) {
  return guarded([&]() {
    if (!from || !to || !up) fail(PT_E_INVALID, "null argument");
    Camera c = cameraLookAt(fovDegrees, aspect, v3(from[0], from[1], from[2]), v3(to[0], to[1], to[2]), v3(up[0], up[1], up[2]));
    if (h) { h->cam = c; h->haveCamera = true; h->maskDirty = true; }
    if (out14) {
      float r[14] = {c.pos.x, c.pos.y, c.pos.z, c.dir.x, c.dir.y, c.dir.z, c.right.x, c.right.y, c.right.z,
                     c.up.x, c.up.y, c.up.z, c.tanFov, c.aspect};
      std::memcpy(out14, r, sizeof(r));
    }
  });
}

int ptgpu_render(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, uint32_t spp) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    checkWindow(h, pixelOffset, pixelCount);
    useDevice(h);
    PixMap pm{}; pm.offset = pixelOffset;
    renderPixels(h, pm, pixelCount, spp);
  });
}

// Largest batch of ptgpu_render_noalloc for the handle's current scene and configuration, and where its pieces lie
// in the scratch array (offsets in bytes, each a multiple of 256).
namespace {
struct NoAllocLayout {
  size_t slotQ, hitA, hitB, rayO, rayD, Lacc, rng, shO, shD, shW, shX, words, desc, total;
  uint32_t capSlots, capQueue, capHit, lanes, recLanes, kinds, rounds, rayCap;
  bool lean;
};
constexpr uint32_t PT_NOALLOC_GROUPS = 128;      // traversal workgroups per launch: bounds the hit queue's slack

NoAllocLayout layoutNoAlloc(const PtHandle* h, uint32_t capSlots) {
  const PtConfig& c = h->cfg;
  NoAllocLayout L{};
  const uint32_t NI = c.mode == 1 ? c.numIndirect : 0;
  L.capSlots = capSlots; L.lanes = 1 + NI; L.recLanes = 1 + NI * (c.maxDepth + 1); L.kinds = numKinds(h); L.rounds = c.mode == 1 ? c.maxDepth + 4 : 2;
  // as ensureWavefront: a scene whose batches are whole-sample launches keeps no view-ray queue and no shadow queues,
  // and its ray arrays are as large as the hit arrays (the two pairs alternate between bounce rounds)
  L.lean = sceneFusable(h) && !(c.flags & PT_FLAG_QUEUED_VIEW_RAYS);
  const uint64_t capQueue = (uint64_t)capSlots * std::max(1u, NI);
  uint64_t capHit = capQueue + (uint64_t)PT_NOALLOC_GROUPS * PT_HIT_SLACK;
  capHit = (capHit + PT_SEGS - 1) / PT_SEGS * PT_SEGS;
  if (capHit > 0x7FFFFFFFull) { L.total = ~(size_t)0; return L; }
  L.capQueue = (uint32_t)capQueue; L.capHit = (uint32_t)capHit;
  size_t at = 0;
  auto piece = [&](uint64_t bytes) { size_t o = at; at += (size_t)((bytes + 255) / 256 * 256); return o; };
  L.slotQ = piece(L.lean ? 16 : (uint64_t)capSlots * 16);
  L.hitA = piece(capHit * 16); L.hitB = piece(capHit * 16);
  const uint64_t rayEntries = !NI ? 1 : L.lean ? capHit : capQueue;
  L.rayCap = NI && L.lean ? (uint32_t)capHit : 0u;
  L.rayO = piece(rayEntries * 16); L.rayD = piece(rayEntries * 16);
  L.Lacc = piece((uint64_t)L.recLanes * capSlots * 16);
  L.rng = piece(NI ? (uint64_t)L.lanes * capSlots * 8 : 16);
  const uint64_t sh = L.lean ? 16 : (uint64_t)L.kinds * capQueue * 16;
  L.shO = piece(sh); L.shD = piece(sh); L.shW = piece(sh); L.shX = piece(L.lean ? 16 : capQueue * 16);
  const uint64_t nMetas = (uint64_t)L.rounds * (2 + PT_MAX_KINDS);
  L.words = piece(nMetas * PT_QMETA_WORDS * 4 + (PT_TASK_WORDS + PT_POOL_WORDS) * 4);
  L.desc = piece(nMetas * sizeof(QMeta));
  L.total = at;
  return L;
}

// The largest batch (a multiple of 64 * PT_SEGS slots) whose layout fits the array; the layout grows with the batch.
uint64_t maxNoAllocSlots(const PtHandle* h) {
  const uint64_t gran = 64ull * PT_SEGS;
  uint64_t lo = 0, hi = 0x40000000ull / gran + 1;          // fits(lo * gran) (0: nothing), !fits(hi * gran)
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) / 2;
    if (layoutNoAlloc(h, (uint32_t)(mid * gran)).total <= PT_NOALLOC_BYTES) lo = mid; else hi = mid;
  }
  return lo * gran;
}
}  // namespace

int ptgpu_render_noalloc(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, uint32_t spp) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    checkWindow(h, pixelOffset, pixelCount);
    useDevice(h);
    if (!h->haveScene) fail(PT_E_STATE, "no scene: call ptgpu_upload_scene or ptgpu_load_thmodel first");
    if (!h->haveCamera) fail(PT_E_STATE, "no camera: call ptgpu_set_camera or ptgpu_camera_look_at first");
    if (pixelCount == 0 || spp == 0) return;
    const uint64_t gran = 64ull * PT_SEGS;
    const uint64_t nslots = (uint64_t)pixelCount * spp;
    const uint64_t maxSlots = maxNoAllocSlots(h);          // what the scratch array holds for this scene and configuration
    if (nslots > maxSlots)
      fail(PT_E_INVALID, "ptgpu_render_noalloc: pixelCount * spp = " + std::to_string(nslots) + " exceeds the " +
                             std::to_string(maxSlots) + " samples per call that the " + std::to_string(PT_NOALLOC_BYTES >> 20) +
                             " MiB static scratch buffer holds for this scene and configuration (mode, numIndirect, light groups); "
                             "split the window or use ptgpu_render");
    const NoAllocLayout L = layoutNoAlloc(h, (uint32_t)((nslots + gran - 1) / gran * gran));
    unsigned char* base = h->scratchDev;
    WfState ws{};
    Wavefront& w = ws.wf;
    w.slotQ = (float4*)(base + L.slotQ); w.hitA = (float4*)(base + L.hitA); w.hitB = (float4*)(base + L.hitB);
    w.rayO = (float4*)(base + L.rayO); w.rayD = (float4*)(base + L.rayD); w.Lacc = (float4*)(base + L.Lacc);
    w.rng = (uint2*)(base + L.rng);
    w.shO = (float4*)(base + L.shO); w.shD = (float4*)(base + L.shD); w.shW = (float4*)(base + L.shW); w.shX = (float4*)(base + L.shX);
    w.stats = h->bStats.as<unsigned long long>();
    w.capSlots = L.capSlots; w.capQueue = L.capQueue; w.capHit = L.capHit; w.lanes = L.lanes; w.rayCap = L.rayCap;
    w.chainStride = (L.lanes - 1) * L.capSlots; w.recRounds = h->cfg.maxDepth + 1;
    uint32_t* words = (uint32_t*)(base + L.words);
    const size_t nMetas = (size_t)L.rounds * (2 + PT_MAX_KINDS);
    for (size_t m = 0; m < nMetas; m++) {
      uint32_t* p = words + m * PT_QMETA_WORDS;
      h->metasNoAlloc[m] = QMeta{p, p + PT_SEGS, p + 2 * PT_SEGS + 2};
    }
    w.taskCounters = words + nMetas * PT_QMETA_WORDS;
    w.runPool = (h->cfg.flags & PT_FLAG_STATIC_RUNS) ? nullptr : w.taskCounters + PT_TASK_WORDS;
    ws.hostMetas = h->metasNoAlloc.data(); ws.devMetas = (const QMeta*)(base + L.desc);
    // masks built for the current scene and camera are used; none is built here (that would allocate): the view rays
    // are then culled by the scene's bounding box, which gives the same image
    ws.masks = (!h->maskDirty && h->maskUsable) ? h->masks() : ViewMasks{nullptr, nullptr, nullptr};
    // one call at a time per GPU works on the array: this one starts behind the last kernel of the previous one (any
    // handle, any stream of this device) and the next one behind this one's.  Under a stream capture the order is the
    // capturing caller's to give.
    std::lock_guard<std::mutex> turn(g_scratchCall[h->cfg.device]);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool ordered = cap == hipStreamCaptureStatusNone;
    if (ordered) HIPCHECK(hipStreamWaitEvent(h->stream, g_scratchDone[h->cfg.device], 0));
    // the array keeps the last call's contents (of this or any other handle on the device): counters start at zero
    HIPCHECK(hipMemsetAsync(words, 0, nMetas * PT_QMETA_WORDS * 4 + (PT_TASK_WORDS + PT_POOL_WORDS) * 4, h->stream));
    // (every radiance record a batch reads was written by that batch: nothing of them needs clearing)
    LaunchCfg lc = makeLaunchCfg(h);
    launchInitMetas(lc, (QMeta*)(base + L.desc), words, (uint32_t)nMetas);
    HIPCHECK(hipGetLastError());
    PixMap pm{}; pm.offset = pixelOffset;
    h->noEvents = true;
    try { renderPixels(h, pm, pixelCount, spp, &ws, PT_NOALLOC_GROUPS); } catch (...) { h->noEvents = false; throw; }
    h->noEvents = false;
    if (ordered) HIPCHECK(hipEventRecord(g_scratchDone[h->cfg.device], h->stream));
  });
}

int ptgpu_render_bands(PtHandle* h, uint32_t bandRows, uint32_t bandFirst, uint32_t bandStride, uint32_t spp) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (bandRows == 0 || bandStride == 0 || bandFirst >= bandStride) fail(PT_E_INVALID, "need bandRows > 0 and bandFirst < bandStride");
    useDevice(h);
    uint32_t W = h->cfg.width, H = h->cfg.height;
    uint32_t numBands = (H + bandRows - 1) / bandRows;
    uint64_t rows = 0;
    for (uint32_t b = bandFirst; b < numBands; b += bandStride) rows += std::min(bandRows, H - b * bandRows);
    PixMap pm{}; pm.bandPix = bandRows * W; pm.bandFirst = bandFirst; pm.bandStride = bandStride;
    renderPixels(h, pm, rows * W, spp);
  });
}

int ptgpu_synchronize(PtHandle* h) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    useDevice(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
  });
}

int ptgpu_read_pixels(PtHandle* h, uint32_t off, uint32_t count, PtPixel* out) {
  return guarded([&]() {
    if (!h || (!out && count)) fail(PT_E_INVALID, "null argument");
    checkWindow(h, off, count); useDevice(h);
    if (!count) return;
    // the 28-byte records are made on the device (pixel.color = pixel.sum / pixel.n, an IEEE division there as here)
    // and cross PCIe once: straight into `out` when the caller page-locked it, else through the handle's staging
    const size_t bytes = (size_t)count * sizeof(PtPixel);
    h->dPixelRec.alloc(bytes);
    launchPackPixels(makeLaunchCfg(h), h->acc, off, count, h->dPixelRec.as<float>());
    HIPCHECK(hipGetLastError());
    const bool direct = isPinnedHost(out);
    if (!direct) h->pinPixels.alloc(bytes);
    HIPCHECK(hipMemcpyAsync(direct ? (void*)out : h->pinPixels.p, h->dPixelRec.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (!direct) std::memcpy(out, h->pinPixels.p, bytes);
  });
}

int ptgpu_write_pixels(PtHandle* h, uint32_t off, uint32_t count, const PtPixel* in) {
  return guarded([&]() {
    if (!h || (!in && count)) fail(PT_E_INVALID, "null argument");
    checkWindow(h, off, count); useDevice(h);
    if (!count) return;
    const size_t bytes = (size_t)count * sizeof(PtPixel);
    h->dPixelRec.alloc(bytes);
    const bool direct = isPinnedHost(in);
    if (!direct) { HIPCHECK(hipStreamSynchronize(h->stream)); h->pinPixels.alloc(bytes); std::memcpy(h->pinPixels.p, in, bytes); }
    HIPCHECK(hipMemcpyAsync(h->dPixelRec.p, direct ? (const void*)in : h->pinPixels.p, bytes, hipMemcpyHostToDevice, h->stream));
    launchUnpackPixels(makeLaunchCfg(h), h->dPixelRec.as<float>(), off, count, h->acc);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(h->stream));       // `in` and the staging buffer may be reused when this returns
  });
}

int ptgpu_clear_pixels(PtHandle* h) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    useDevice(h);
    HIPCHECK(hipMemsetAsync(h->acc, 0, h->accBytes, h->stream));
  });
}

namespace {
// Tone map of a window into one of the handle's two device buffers on the render stream, then its copy to the host on
// the copy stream, so that kernels queued on the render stream afterwards (the next pass) run beside the copy.
void tonemapBegin(PtHandle* h, uint32_t off, uint32_t count, uint8_t* rgb) {
  if (h->tonePending >= 0) fail(PT_E_STATE, "a ptgpu_tonemap_begin is outstanding: call ptgpu_tonemap_end first");
  const int k = h->toneNext;
  const size_t bytes = (size_t)count * 3;
  h->dRgb[k].alloc(bytes);
  h->toneDirect = isPinnedHost(rgb);
  if (!h->toneDirect) h->pinRgb[k].alloc(bytes);
  launchTonemap(makeLaunchCfg(h), h->acc, off, count, h->dRgb[k].as<uint8_t>());
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipEventRecord(h->evTone[k], h->stream));
  HIPCHECK(hipStreamWaitEvent(h->copyStream, h->evTone[k], 0));
  HIPCHECK(hipMemcpyAsync(h->toneDirect ? (void*)rgb : h->pinRgb[k].p, h->dRgb[k].p, bytes, hipMemcpyDeviceToHost, h->copyStream));
  HIPCHECK(hipEventRecord(h->evCopy[k], h->copyStream));
  h->tonePending = k; h->toneDst = rgb; h->toneBytes = bytes; h->toneNext = k ^ 1;
}
void tonemapEnd(PtHandle* h) {
  if (h->tonePending < 0) fail(PT_E_STATE, "no ptgpu_tonemap_begin is outstanding");
  const int k = h->tonePending;
  h->tonePending = -1;
  HIPCHECK(hipEventSynchronize(h->evCopy[k]));
  if (!h->toneDirect) std::memcpy(h->toneDst, h->pinRgb[k].p, h->toneBytes);
}
}  // namespace

int ptgpu_tonemap(PtHandle* h, uint32_t off, uint32_t count, uint8_t* rgb) {
  return guarded([&]() {
    if (!h || (!rgb && count)) fail(PT_E_INVALID, "null argument");
    checkWindow(h, off, count); useDevice(h);
    if (!count) return;
    tonemapBegin(h, off, count, rgb);
    tonemapEnd(h);
  });
}

int ptgpu_tonemap_begin(PtHandle* h, uint32_t off, uint32_t count, uint8_t* rgb) {
  return guarded([&]() {
    if (!h || !rgb) fail(PT_E_INVALID, "null argument");
    checkWindow(h, off, count); useDevice(h);
    if (!count) fail(PT_E_INVALID, "empty window");
    tonemapBegin(h, off, count, rgb);
  });
}

int ptgpu_tonemap_end(PtHandle* h) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    useDevice(h);
    tonemapEnd(h);
  });
}

int ptgpu_host_alloc(PtHandle* h, uint64_t bytes, void** out) {
  return guarded([&]() {
    if (!h || !out || !bytes) fail(PT_E_INVALID, "null argument");
    *out = nullptr;
    useDevice(h);
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); fail(PT_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    *out = p;
  });
}

int ptgpu_host_free(PtHandle* h, void* p) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (!p) return;
    useDevice(h);
    HIPCHECK(hipStreamSynchronize(h->stream)); HIPCHECK(hipStreamSynchronize(h->copyStream));
    HIPCHECK(hipHostFree(p));
  });
}

int ptgpu_scene_trace(PtHandle* h, const float* rays6, uint32_t count, int32_t* triId, float* t, float* u, float* v, float* normal3) {
  return guarded([&]() {
    if (!h || (!rays6 && count)) fail(PT_E_INVALID, "null argument");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    useDevice(h);
    if (!count) return;
    DevBuf dr, dtri, dt, du, dv, dn;
    dr.alloc((size_t)count * 24); dtri.alloc((size_t)count * 4); dt.alloc((size_t)count * 4); du.alloc((size_t)count * 4);
    dv.alloc((size_t)count * 4); dn.alloc((size_t)count * 12);
    HIPCHECK(hipMemcpyAsync(dr.p, rays6, (size_t)count * 24, hipMemcpyHostToDevice, h->stream));
    launchTraceRays(makeLaunchCfg(h), h->dsc, makeDevCfg(h), dr.as<float>(), count, dtri.as<int32_t>(), dt.as<float>(),
                    du.as<float>(), dv.as<float>(), dn.as<float>());
    HIPCHECK(hipGetLastError());
    if (triId) HIPCHECK(hipMemcpyAsync(triId, dtri.p, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    if (t) HIPCHECK(hipMemcpyAsync(t, dt.p, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    if (u) HIPCHECK(hipMemcpyAsync(u, du.p, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    if (v) HIPCHECK(hipMemcpyAsync(v, dv.p, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    if (normal3) HIPCHECK(hipMemcpyAsync(normal3, dn.p, (size_t)count * 12, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  });
}

int ptgpu_set_texcoords(PtHandle* h, const float* tex6) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    size_t T = h->flat.numTriangles();
    if (!tex6 && T) fail(PT_E_INVALID, "null argument");
    h->flat.tex.assign(tex6, tex6 + T * 6);
    h->texDirty = true;
  });
}

int ptgpu_get_texcoords(PtHandle* h, float* tex6) {
  return guarded([&]() {
    if (!h || !tex6) fail(PT_E_INVALID, "null argument");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    size_t T = h->flat.numTriangles();
    if (h->flat.tex.size() == T * 6) std::memcpy(tex6, h->flat.tex.data(), T * 24);
    else std::memset(tex6, 0, T * 24);
  });
}

int ptgpu_scene_trace_tex(PtHandle* h, const float* rays6, uint32_t count, int ignoreBackfaces, int32_t* result, int32_t* triId,
                          float* t, float* tc2) {
  return guarded([&]() {
    if (!h || ((!rays6 || !result) && count)) fail(PT_E_INVALID, "null argument");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    useDevice(h);
    if (!count) return;
    size_t T = h->flat.numTriangles();
    if (h->texDirty) {
      h->dTex.alloc(std::max<size_t>(T, 1) * 24);
      if (h->flat.tex.size() == T * 6 && T) HIPCHECK(hipMemcpy(h->dTex.p, h->flat.tex.data(), T * 24, hipMemcpyHostToDevice));
      else HIPCHECK(hipMemset(h->dTex.p, 0, std::max<size_t>(T, 1) * 24));
      h->texDirty = false;
    }
    DevBuf dr, dres, dtri, dt, dtc;
    dr.alloc((size_t)count * 24); dres.alloc((size_t)count * 4); dtri.alloc((size_t)count * 4); dt.alloc((size_t)count * 4);
    dtc.alloc((size_t)count * 8);
    HIPCHECK(hipMemcpyAsync(dr.p, rays6, (size_t)count * 24, hipMemcpyHostToDevice, h->stream));
    launchTraceTex(makeLaunchCfg(h), h->dsc, makeDevCfg(h), h->dTex.as<float>(), dr.as<float>(), count, ignoreBackfaces != 0,
                   dres.as<int32_t>(), dtri.as<int32_t>(), dt.as<float>(), dtc.as<float>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(result, dres.p, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    if (triId) HIPCHECK(hipMemcpyAsync(triId, dtri.p, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    if (t) HIPCHECK(hipMemcpyAsync(t, dt.p, (size_t)count * 4, hipMemcpyDeviceToHost, h->stream));
    if (tc2) HIPCHECK(hipMemcpyAsync(tc2, dtc.p, (size_t)count * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  });
}

int ptgpu_scene_hits_nothing(PtHandle* h, const float* rays6, uint32_t count, uint8_t* out) {
  return guarded([&]() {
    if (!h || ((!rays6 || !out) && count)) fail(PT_E_INVALID, "null argument");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    useDevice(h);
    if (!count) return;
    DevBuf dr, dout; dr.alloc((size_t)count * 24); dout.alloc(count);
    HIPCHECK(hipMemcpyAsync(dr.p, rays6, (size_t)count * 24, hipMemcpyHostToDevice, h->stream));
    launchHitsNothing(makeLaunchCfg(h), h->dsc, makeDevCfg(h), dr.as<float>(), count, dout.as<uint8_t>(), h->bStats.as<unsigned long long>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(out, dout.p, count, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  });
}

int ptgpu_trace_primary(PtHandle* h, uint32_t off, uint32_t count, int32_t* triId, float* t, float* u, float* v) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    if (!h->haveCamera) fail(PT_E_STATE, "no camera");
    checkWindow(h, off, count); useDevice(h);
    if (!count) return;
    DevBuf dh; dh.alloc((size_t)count * 16);
    PixMap pm{}; pm.offset = off; pm.npix = count; pm.k = 1;
    pm.divNpix = makeFastDiv(count); pm.divBand = makeFastDiv(0); pm.tilesPerRow = 1; pm.divTilesPerRow = makeFastDiv(1);
    launchExtendPrimary(makeLaunchCfg(h), h->dsc, makeDevCfg(h), pm, h->acc, count, 0, dh.as<float4>(), h->bStats.as<unsigned long long>());
    HIPCHECK(hipGetLastError());
    std::vector<float> tmp((size_t)count * 4);
    HIPCHECK(hipMemcpyAsync(tmp.data(), dh.p, (size_t)count * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    for (uint32_t i = 0; i < count; i++) {
      const float* r = &tmp[(size_t)i * 4];
      if (t) t[i] = r[0];
      if (u) u[i] = r[1];
      if (v) v[i] = r[2];
      if (triId) { int32_t id; std::memcpy(&id, &r[3], 4); triId[i] = id; }
    }
  });
}

int ptgpu_view_rays(PtHandle* h, uint32_t off, uint32_t count, float* rays6) {
  return guarded([&]() {
    if (!h || (!rays6 && count)) fail(PT_E_INVALID, "null argument");
    if (!h->haveCamera) fail(PT_E_STATE, "no camera");
    checkWindow(h, off, count); useDevice(h);
    if (!count) return;
    DevBuf d; d.alloc((size_t)count * 24);
    launchViewRays(makeLaunchCfg(h), makeDevCfg(h), off, count, d.as<float>());
    HIPCHECK(hipMemcpyAsync(rays6, d.p, (size_t)count * 24, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  });
}

int ptgpu_get_coverage_mask(PtHandle* h, uint32_t* covered, uint32_t* constantSky, uint32_t* usable) {
  return guarded([&]() {
    if (!h || !usable) fail(PT_E_INVALID, "null argument");
    if (!h->haveScene) fail(PT_E_STATE, "no scene");
    if (!h->haveCamera) fail(PT_E_STATE, "no camera");
    if (h->cfg.device >= 0 && !(h->cfg.flags & PT_FLAG_NO_COVERAGE_MASK)) {     // the masks rendering uses: built on the GPU
      useDevice(h);
      if (h->maskDirty) buildMasks(h);
      *usable = h->maskUsable ? 1u : 0u;
      const size_t bytes = (((size_t)h->cfg.width * h->cfg.height + 31) / 32) * 4;
      HIPCHECK(hipStreamSynchronize(h->stream));
      if (h->maskUsable && covered) HIPCHECK(hipMemcpy(covered, h->dMask.p, bytes, hipMemcpyDeviceToHost));
      if (h->maskUsable && constantSky) HIPCHECK(hipMemcpy(constantSky, h->dSkyMask.p, bytes, hipMemcpyDeviceToHost));
      return;
    }
    std::vector<uint32_t> m, s;                       // host-only handle (or masks switched off): the host builder
    bool ok = buildCoverageMask(h->flat, h->cam, h->cfg.width, h->cfg.height, m, s);
    *usable = ok ? 1u : 0u;
    if (ok && covered) std::memcpy(covered, m.data(), m.size() * 4);
    if (ok && constantSky) std::memcpy(constantSky, s.data(), s.size() * 4);
  });
}

int ptgpu_rng_stream(PtHandle* h, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, uint32_t* out) {
  return guarded([&]() {
    if (!h || (!out && n)) fail(PT_E_INVALID, "null argument");
    useDevice(h);
    if (!n) return;
    DevBuf d; d.alloc((size_t)n * 4);
    launchRngStream(makeLaunchCfg(h), h->cfg.seed, pixel, sample, stream, n, d.as<uint32_t>());
    HIPCHECK(hipMemcpyAsync(out, d.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  });
}

int ptgpu_angle_directions(PtHandle* h, const float* in5, uint32_t n, float* out3) {
  return guarded([&]() {
    if (!h || ((!in5 || !out3) && n)) fail(PT_E_INVALID, "null argument");
    useDevice(h);
    if (!n) return;
    DevBuf di, dout; di.alloc((size_t)n * 20); dout.alloc((size_t)n * 12);
    HIPCHECK(hipMemcpyAsync(di.p, in5, (size_t)n * 20, hipMemcpyHostToDevice, h->stream));
    launchAngleDirections(makeLaunchCfg(h), di.as<float>(), n, dout.as<float>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(out3, dout.p, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  });
}

int ptgpu_get_stats(PtHandle* h, PtStats* out) {
  return guarded([&]() {
    if (!h || !out) fail(PT_E_INVALID, "null argument");
    if (h->cfg.device < 0) { *out = h->st; return; }
    useDevice(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    resolveEvents(h);
    unsigned long long all[PT_STAT_REPLICAS * PT_STAT_STRIDE], s[STAT_COUNT] = {};
    HIPCHECK(hipMemcpy(all, h->bStats.p, sizeof(all), hipMemcpyDeviceToHost));
    for (int r = 0; r < PT_STAT_REPLICAS; r++)
      for (int k = 0; k < STAT_COUNT; k++) s[k] += all[r * PT_STAT_STRIDE + k];
    h->st.raysLight = s[STAT_RAYS_LIGHT]; h->st.raysIndirect = s[STAT_RAYS_INDIRECT]; h->st.raysOcclusion = s[STAT_RAYS_OCCL];
    h->st.nodesClosest = s[STAT_NODES_CLOSEST]; h->st.trisClosest = s[STAT_TRIS_CLOSEST];
    h->st.nodesAny = s[STAT_NODES_ANY]; h->st.trisAny = s[STAT_TRIS_ANY]; h->st.raysCulled = s[STAT_RAYS_CULLED];
    h->st.hits = s[STAT_HITS]; h->st.envEntries = s[STAT_ENV_ENTRIES];
    h->st.hitsPrimary = s[STAT_HITS_PRIMARY]; h->st.raysPrimaryInShade = s[STAT_FUSED_PRIMARY]; h->st.raysLightInShade = s[STAT_FUSED_LIGHT]; h->st.raysIndirectInShade = s[STAT_FUSED_INDIRECT]; h->st.raysOcclusionInShade = s[STAT_FUSED_OCCL];
    h->st.raysLightInSample = s[STAT_WHOLE_LIGHT]; h->st.raysIndirectInSample = s[STAT_WHOLE_INDIRECT]; h->st.raysOcclusionInSample = s[STAT_WHOLE_OCCL];
    h->st.chainHitsFromSample = s[STAT_WHOLE_CHAIN_HITS]; h->st.samplesConstantSky = s[STAT_CONSTANT_SKY];
    *out = h->st;
  });
}

int ptgpu_reset_stats(PtHandle* h) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    useDevice(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    resolveEvents(h);
    HIPCHECK(hipMemset(h->bStats.p, 0, h->bStats.bytes));
    PtStats keep = h->st; h->st = PtStats();
    h->st.numTriangles = keep.numTriangles; h->st.numNodes = keep.numNodes; h->st.bvhDepth = keep.bvhDepth;
    h->st.numLightGroups = keep.numLightGroups; h->st.sceneInLds = keep.sceneInLds;
    h->st.bvhOnGpu = keep.bvhOnGpu; h->st.bvhBuildSeconds = keep.bvhBuildSeconds; h->st.maskBuildMs = keep.maskBuildMs;
    h->st.sceneBytes = keep.sceneBytes;
    h->eventOverflow = false;
  });
}

int ptgpu_bind_accumulator(PtHandle* h, void* devicePtr, uint64_t bytes) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    useDevice(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (!devicePtr) { h->accOwn.alloc(h->accBytes); h->acc = h->accOwn.as<float4>(); return; }
    if (bytes < h->accBytes) fail(PT_E_INVALID, "accumulator needs width*height*16 bytes");
    if (((uintptr_t)devicePtr & 15) != 0) fail(PT_E_INVALID, "accumulator must be 16-byte aligned");
    h->acc = (float4*)devicePtr;
  });
}

int ptgpu_get_accumulator(PtHandle* h, void** devicePtr, uint64_t* bytes) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    if (devicePtr) *devicePtr = h->acc;
    if (bytes) *bytes = h->accBytes;
  });
}

int ptgpu_set_stream(PtHandle* h, void* hipStream) {
  return guarded([&]() {
    if (!h) fail(PT_E_INVALID, "null handle");
    useDevice(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (h->ownStream) { (void)hipStreamDestroy(h->stream); h->ownStream = false; }
    h->stream = (hipStream_t)hipStream;
  });
}

}  // extern "C"

// ======================================================================================================================
// Several GPUs behind one object: the reference's task decomposition (one ComputeOutputTask per 4 scanlines pulled by
// worker threads, a barrier per pass, main.d:184-215, 273-288) with a GPU in the place of a worker thread.  The scene
// is replicated; GPU i renders bands i, i + G, ... (or, for the sample split, every pixel with its own residue class
// of sample indices); one exchange over RCCL puts the frame together on one GPU.
// ======================================================================================================================
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is opened at run time, not linked
#include <thread>

namespace {

// librccl, opened on first use.  In a process that already holds an RCCL (torch's), the loader hands back that one.
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
  bool ok() const { return lib != nullptr; }
};
Rccl& rccl() {
  static Rccl r; static std::once_flag once;
  std::call_once(once, []() {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { const char* e = dlerror(); r.why = e ? e : "librccl.so not found"; return; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) { r.why = std::string("librccl lacks ") + n; } return p; };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll"); r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))sym("ncclReduce"); r.Send = (decltype(r.Send))sym("ncclSend"); r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart"); r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.why.empty()) { dlclose(r.lib); r.lib = nullptr; }
  });
  return r;
}
#define NCCLCHECK(expr)                                                                                          \
  do {                                                                                                           \
    ncclResult_t r_ = (expr);                                                                                    \
    if (r_ != ncclSuccess) fail(PT_E_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(r_));              \
  } while (0)

// Rows of the bands first, first + stride, ... of bandRows scanlines in a frame of `height` rows, in order.
void bandRowsOf(uint32_t height, uint32_t bandRows, uint32_t first, uint32_t stride, std::vector<uint32_t>& rows) {
  rows.clear();
  uint32_t numBands = (height + bandRows - 1) / bandRows;
  for (uint32_t b = first; b < numBands; b += stride)
    for (uint32_t r = b * bandRows; r < std::min(height, (b + 1) * bandRows); r++) rows.push_back(r);
}
uint32_t bandsOfRank(uint32_t height, uint32_t bandRows, uint32_t first, uint32_t stride) {
  uint32_t numBands = (height + bandRows - 1) / bandRows;
  return first < numBands ? (numBands - first + stride - 1) / stride : 0;
}

}  // namespace

struct PtGroup {
  std::vector<PtHandle*> ranks;       // one renderer per GPU
  PtHandle* frame = nullptr;          // a handle without a scene on the root GPU: its accumulator is the combined frame
  uint32_t root = 0;
  std::vector<ncclComm_t> comms;      // empty when the exchange runs without RCCL (one GPU and no library)
  std::vector<DevBuf> packed;         // per rank: its band rows, dense (on the rank's GPU)
  std::vector<DevBuf> recv;           // per rank: the same rows as received on the root GPU
  // what the last ptgpu_group_render shared out, so that combine knows what to collect
  int split = -1; uint32_t bandRows = 4;
  bool hostOnly = false;
};

namespace {

// Runs f(rank) on one host thread per GPU (the calling thread takes rank 0) and rethrows the first failure.
template <typename F> void onEveryRank(PtGroup* g, F&& f) {
  const size_t n = g->ranks.size();
  std::vector<int> codes(n, PT_OK); std::vector<std::string> msgs(n);
  auto run = [&](size_t r) {
    try { f((uint32_t)r); }
    catch (const ApiError& e) { codes[r] = e.code; msgs[r] = e.msg; }
    catch (const IoError& e) { codes[r] = PT_E_IO; msgs[r] = e.what(); }
    catch (const std::exception& e) { codes[r] = PT_E_INVALID; msgs[r] = e.what(); }
  };
  std::vector<std::thread> th;
  for (size_t r = 1; r < n; r++) th.emplace_back(run, r);
  run(0);
  for (auto& t : th) t.join();
  for (size_t r = 0; r < n; r++) if (codes[r] != PT_OK) fail(codes[r], "GPU " + std::to_string(r) + ": " + msgs[r]);
}
void must(int rc) { if (rc != PT_OK) fail(rc, ptgpu_last_error()); }

}  // namespace

extern "C" {

int ptgpu_band_rows(uint32_t height, uint32_t bandRows, uint32_t bandFirst, uint32_t bandStride, uint32_t* rows, uint32_t cap,
                    uint32_t* count) {
  return guarded([&]() {
    if (!count) fail(PT_E_INVALID, "null argument");
    if (bandRows == 0 || bandStride == 0 || bandFirst >= bandStride) fail(PT_E_INVALID, "need bandRows > 0 and bandFirst < bandStride");
    std::vector<uint32_t> r; bandRowsOf(height, bandRows, bandFirst, bandStride, r);
    *count = (uint32_t)r.size();
    if (rows) for (uint32_t i = 0; i < r.size() && i < cap; i++) rows[i] = r[i];
  });
}

int ptgpu_group_create(const PtConfig* cfg, const int32_t* devices, uint32_t numDevices, uint32_t root, PtGroup** out) {
  if (out) *out = nullptr;
  PtGroup* g = nullptr;
  int rc = guarded([&]() {
    if (!cfg || !devices || !out || numDevices == 0) fail(PT_E_INVALID, "null argument or no device");
    if (root >= numDevices) fail(PT_E_INVALID, "root must be one of the devices");
    g = new PtGroup(); g->root = root;
    g->packed.resize(numDevices); g->recv.resize(numDevices);
    g->hostOnly = devices[0] < 0;
    for (uint32_t i = 0; i < numDevices; i++) {
      if ((devices[i] < 0) != g->hostOnly) fail(PT_E_INVALID, "host-only (-1) and real devices cannot be mixed");
      for (uint32_t j = 0; j < i; j++) if (!g->hostOnly && devices[j] == devices[i]) fail(PT_E_INVALID, "device listed twice");
      PtConfig c = *cfg; c.device = devices[i];
      PtHandle* h = nullptr; must(ptgpu_create(&c, &h));
      g->ranks.push_back(h);
    }
    { PtConfig c = *cfg; c.device = devices[root]; must(ptgpu_create(&c, &g->frame)); }
    if (!g->hostOnly) {
      Rccl& r = rccl();
      if (r.ok()) {
        g->comms.resize(numDevices);
        std::vector<int> devs(devices, devices + numDevices);
        NCCLCHECK(r.CommInitAll(g->comms.data(), (int)numDevices, devs.data()));
      } else if (numDevices > 1) {
        fail(PT_E_DEVICE, "more than one GPU needs RCCL, which could not be opened: " + r.why);
      }
    }
  });
  if (rc != PT_OK) { if (g) ptgpu_group_destroy(g); return rc; }
  *out = g; return PT_OK;
}

int ptgpu_group_destroy(PtGroup* g) {
  if (!g) return PT_OK;
  return guarded([&]() {
    for (PtHandle* h : g->ranks) if (h && h->cfg.device >= 0) { (void)hipSetDevice(h->cfg.device); (void)hipStreamSynchronize(h->stream); }
    for (ncclComm_t c : g->comms) if (c) (void)rccl().CommDestroy(c);
    for (size_t r = 0; r < g->ranks.size(); r++) {
      if (g->ranks[r] && g->ranks[r]->cfg.device >= 0) (void)hipSetDevice(g->ranks[r]->cfg.device);
      g->packed[r].release();
    }
    if (g->frame && g->frame->cfg.device >= 0) (void)hipSetDevice(g->frame->cfg.device);
    for (DevBuf& b : g->recv) b.release();
    for (PtHandle* h : g->ranks) (void)ptgpu_destroy(h);
    (void)ptgpu_destroy(g->frame);
    delete g;
  });
}

int ptgpu_group_info(PtGroup* g, uint32_t* numDevices, uint32_t* root, uint32_t* usesRccl) {
  return guarded([&]() {
    if (!g) fail(PT_E_INVALID, "null group");
    if (numDevices) *numDevices = (uint32_t)g->ranks.size();
    if (root) *root = g->root;
    if (usesRccl) *usesRccl = g->comms.empty() ? 0u : 1u;
  });
}

int ptgpu_group_handle(PtGroup* g, uint32_t rank, PtHandle** out) {
  return guarded([&]() {
    if (!g || !out) fail(PT_E_INVALID, "null argument");
    if (rank >= g->ranks.size()) fail(PT_E_INVALID, "rank out of range");
    *out = g->ranks[rank];
  });
}

int ptgpu_group_frame(PtGroup* g, PtHandle** out) {
  return guarded([&]() {
    if (!g || !out) fail(PT_E_INVALID, "null argument");
    *out = g->frame;
  });
}

int ptgpu_group_load_thmodel(PtGroup* g, const char* path, int materialTable, const float* modelMatrix) {
  return guarded([&]() {
    if (!g || !path) fail(PT_E_INVALID, "null argument");
    onEveryRank(g, [&](uint32_t r) { must(ptgpu_load_thmodel(g->ranks[r], path, materialTable, modelMatrix)); });
  });
}

int ptgpu_group_upload_scene(PtGroup* g, const float* v, const float* n, const uint32_t* matIdx, const PtMaterial* mats,
                             uint32_t numTriangles, uint32_t numMaterials) {
  return guarded([&]() {
    if (!g) fail(PT_E_INVALID, "null group");
    onEveryRank(g, [&](uint32_t r) { must(ptgpu_upload_scene(g->ranks[r], v, n, matIdx, mats, numTriangles, numMaterials)); });
  });
}

int ptgpu_group_set_camera(PtGroup* g, const float pos[3], const float dir[3], const float right[3], const float up[3], float tanFov,
                           float aspect) {
  return guarded([&]() {
    if (!g) fail(PT_E_INVALID, "null group");
    for (PtHandle* h : g->ranks) must(ptgpu_set_camera(h, pos, dir, right, up, tanFov, aspect));
  });
}

int ptgpu_group_camera_look_at(PtGroup* g, float fovDegrees, float aspect, const float from[3], const float to[3], const float up[3]) {
  return guarded([&]() {
    if (!g) fail(PT_E_INVALID, "null group");
    for (PtHandle* h : g->ranks) must(ptgpu_camera_look_at(h, fovDegrees, aspect, from, to, up, nullptr));
  });
}

int ptgpu_group_render(PtGroup* g, int split, uint32_t bandRows, uint32_t spp) {
  return guarded([&]() {
    if (!g) fail(PT_E_INVALID, "null group");
    if (split != PT_SPLIT_BANDS && split != PT_SPLIT_SAMPLES) fail(PT_E_INVALID, "split must be PT_SPLIT_BANDS or PT_SPLIT_SAMPLES");
    if (split == PT_SPLIT_BANDS && bandRows == 0) fail(PT_E_INVALID, "bandRows must be > 0");
    if (g->split >= 0 && g->split != split) fail(PT_E_STATE, "the split of a group is fixed by its first ptgpu_group_render (clear with ptgpu_group_clear)");
    if (g->split == PT_SPLIT_BANDS && g->bandRows != bandRows) fail(PT_E_STATE, "bandRows changed between passes");
    const uint32_t G = (uint32_t)g->ranks.size();
    onEveryRank(g, [&](uint32_t r) {
      PtHandle* h = g->ranks[r];
      if (split == PT_SPLIT_BANDS) must(ptgpu_render_bands(h, bandRows, r, G, spp));
      else {
        PtConfig c = h->cfg; c.sampleBase = r; c.sampleStride = G;
        must(ptgpu_set_config(h, &c));
        must(ptgpu_render(h, 0, h->cfg.width * h->cfg.height, spp));
      }
    });
    g->split = split; g->bandRows = bandRows;
  });
}

int ptgpu_group_clear(PtGroup* g) {
  return guarded([&]() {
    if (!g) fail(PT_E_INVALID, "null group");
    for (PtHandle* h : g->ranks) must(ptgpu_clear_pixels(h));
    must(ptgpu_clear_pixels(g->frame));
    g->split = -1;
  });
}

int ptgpu_group_combine(PtGroup* g) {
  return guarded([&]() {
    if (!g) fail(PT_E_INVALID, "null group");
    if (g->hostOnly) fail(PT_E_DEVICE, "this group was created without devices");
    if (g->split < 0) fail(PT_E_STATE, "nothing rendered yet");
    const uint32_t G = (uint32_t)g->ranks.size(), root = g->root;
    PtHandle* F = g->frame;
    const uint32_t W = F->cfg.width, H = F->cfg.height;
    const size_t frameFloats = (size_t)W * H * 4;
    const bool useRccl = !g->comms.empty();
    Rccl& R = rccl();
    if (g->split == PT_SPLIT_SAMPLES) {
      // every rank holds (sum, n) of all pixels for its own samples: one reduce(sum) onto the root's frame buffer
      if (useRccl) {
        NCCLCHECK(R.GroupStart());
        for (uint32_t r = 0; r < G; r++) {
          PtHandle* h = g->ranks[r];
          HIPCHECK(hipSetDevice(h->cfg.device));
          NCCLCHECK(R.Reduce(h->acc, r == root ? (void*)F->acc : nullptr, frameFloats, ncclFloat, ncclSum, (int)root, g->comms[r], h->stream));
        }
        NCCLCHECK(R.GroupEnd());
      } else {
        HIPCHECK(hipSetDevice(F->cfg.device));
        HIPCHECK(hipMemcpyAsync(F->acc, g->ranks[0]->acc, frameFloats * 4, hipMemcpyDeviceToDevice, g->ranks[0]->stream));
      }
    } else {
      // disjoint rows: every rank packs its bands into a dense buffer, the others send theirs to the root (1/G of the
      // frame each instead of a reduce over G whole frames), the root scatters all of them into the frame buffer
      std::vector<uint32_t> packedRows(G);
      for (uint32_t r = 0; r < G; r++) {
        PtHandle* h = g->ranks[r];
        packedRows[r] = bandsOfRank(H, g->bandRows, r, G) * g->bandRows;
        HIPCHECK(hipSetDevice(h->cfg.device));
        g->packed[r].alloc((size_t)packedRows[r] * W * 16);
        LaunchCfg lc = makeLaunchCfg(h);
        launchBandRows(lc, h->acc, g->packed[r].as<float4>(), W, H, g->bandRows, r, G, packedRows[r], false);
        HIPCHECK(hipGetLastError());
      }
      HIPCHECK(hipSetDevice(F->cfg.device));
      for (uint32_t r = 0; r < G; r++) if (r != root) g->recv[r].alloc((size_t)packedRows[r] * W * 16);
      PtHandle* hr = g->ranks[root];
      if (G > 1) {
        NCCLCHECK(R.GroupStart());
        for (uint32_t r = 0; r < G; r++) {
          if (r == root || !packedRows[r]) continue;
          const size_t cnt = (size_t)packedRows[r] * W * 4;
          HIPCHECK(hipSetDevice(g->ranks[r]->cfg.device));
          NCCLCHECK(R.Send(g->packed[r].p, cnt, ncclFloat, (int)root, g->comms[r], g->ranks[r]->stream));
          HIPCHECK(hipSetDevice(hr->cfg.device));
          NCCLCHECK(R.Recv(g->recv[r].p, cnt, ncclFloat, (int)r, g->comms[root], hr->stream));
        }
        NCCLCHECK(R.GroupEnd());
      }
      HIPCHECK(hipSetDevice(hr->cfg.device));
      LaunchCfg lc = makeLaunchCfg(hr);                    // on the root rank's stream, behind its receives
      for (uint32_t r = 0; r < G; r++)
        launchBandRows(lc, F->acc, r == root ? g->packed[r].as<float4>() : g->recv[r].as<float4>(), W, H, g->bandRows, r, G, packedRows[r], true);
      HIPCHECK(hipGetLastError());
    }
    for (PtHandle* h : g->ranks) { HIPCHECK(hipSetDevice(h->cfg.device)); HIPCHECK(hipStreamSynchronize(h->stream)); }
    HIPCHECK(hipSetDevice(F->cfg.device));
  });
}

}  // extern "C"
