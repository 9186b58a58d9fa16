/*
 * ptgpu -- MI355X (gfx950) implementation of PathTracer's per-pixel Monte Carlo integrator and
 * ray-scene intersection, behind a C ABI.
 *
 * The reference program has no plugin/FFI interface for this path; the boundary is the set of D call
 * sites that the host program keeps (all citations are into the reference checkout):
 *
 *   scene ready      end of loadScene()                       rendering.d:70-162
 *   render call      computeOutputColor(offset, pixels, gen)   rendering.d:494-500, main.d:120-123
 *   frame read-back  drawScreen(screen, pixels)                main.d:31-99, main.d:275
 *   checkpoint       RawFile("sponza.project") of Pixel[]      main.d:227-234, 333-340
 *
 * Conventions: every function returns 0 on success or a negative PT_E_* code; ptgpu_last_error()
 * returns a message for the calling thread.  No exception crosses this boundary (the reference throws
 * RCException, modelloader.d:134, scene.d:1007).  All structs are plain data, fp32/uint32 only.
 * Functions on ONE handle are not re-entrant; different handles (= different GPUs) may be driven from
 * different threads.  Host memory passed in is copied before return and never freed by the library.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_E_INVALID (-1)   /* bad argument / out-of-range pixel window            */
#define PT_E_STATE (-2)     /* call order: scene or camera not set                  */
#define PT_E_DEVICE (-3)    /* HIP runtime error (message has the hipError string)  */
#define PT_E_IO (-4)        /* file missing / not a thModel / malformed chunk       */
#define PT_E_NOMEM (-5)

typedef struct PtHandle PtHandle;

/* The reference's source-level switches as one POD (rendering.d:19-21 resolution, :54-55 g_hasSky /
 * g_hasLights, :72-74 sun + sky, :379-380 Russian roulette 0.7 / depth 5, :466 N = 5, FloatEpsilon). */
typedef struct PtConfig {
  int32_t device;            /* HIP device ordinal                                                    */
  uint32_t width, height;    /* g_width, g_height                                                     */
  uint32_t mode;             /* 0 = HEAD behaviour: direct lighting only (rendering.d:477)            */
                             /* 1 = + numIndirect computeLindirect chains (the block at :474-477)     */
  uint32_t hasSky;           /* g_hasSky                                                              */
  uint32_t hasLights;        /* g_hasLights                                                           */
  float sunDir[3];           /* g_sunDir      (normalised by the caller)                              */
  float sunRadiance[3];      /* g_sunRadiance                                                         */
  float skyRadiance[3];      /* g_skyRadiance                                                         */
  float floatEpsilon;        /* thBase FloatEpsilon (not in the reference checkout); 1e-5 if 0        */
  float rrProb;              /* `absorb` of computeLindirect, 0.7 if 0                                */
  uint32_t maxDepth;         /* depth cap of computeLindirect, 5 if 0                                 */
  uint32_t numIndirect;      /* N of computeL, 5 if 0                                                 */
  uint32_t seed;             /* RNG key; streams are keyed (seed, pixel, sample, stream)              */
  uint32_t sampleBase;       /* sample k of a pixel (k = Pixel.n so far + position in the call) draws the streams   */
  uint32_t sampleStride;     /* of index sampleBase + k * sampleStride (stride 1 if 0).  G GPUs that each render   */
                             /* every pixel use base = rank, stride = G: their index sets stay disjoint over any   */
                             /* number of passes and across checkpoint / resume                                     */
  uint32_t batchSlots;       /* path slots per wavefront batch, 0 = default                           */
  uint32_t maxLeafTris;      /* BVH leaf size bound (1..15); 0 = the library chooses: 4, or up to 8     */
                             /* when fuller leaves let one more workgroup per CU keep the scene in LDS */
  uint32_t flags;            /* PT_FLAG_*                                                             */
} PtConfig;

#define PT_FLAG_COUNT_VISITS 1u   /* count BVH node boxes tested / triangles tested per ray kind (slow) */
#define PT_FLAG_NO_LDS_SCENE 2u   /* keep nodes/triangles in global memory even if they fit in LDS     */
#define PT_FLAG_TIME_KERNELS 4u   /* bracket every kernel launch with events (ptgpu_get_stats)         */
#define PT_FLAG_NO_COVERAGE_MASK 8u /* cull view rays by the scene's bounding box only, not by the per-pixel
                                       screen-space coverage mask of the triangles                       */
#define PT_FLAG_GPU_BVH 16u       /* build the BVH on the GPU (Morton-order LBVH) whatever the scene size;
                                     default: on the GPU from 1 M triangles up, binned SAH on the host below */
#define PT_FLAG_HOST_BVH 32u      /* always build the BVH on the host                                     */
#define PT_FLAG_NO_GRAPH 64u      /* never replay a small batch from a HIP graph (default: a ptgpu_render / _bands call that is
                                     one batch of at most 4 Mi samples is captured the second time it is issued with the
                                     same arguments, scene, camera and configuration, and replayed from then on -- on the
                                     stream the handle made itself; after ptgpu_set_stream every call launches its kernels) */
#define PT_FLAG_NO_FUSE 128u      /* one launch per stage of the wavefront (raygen, extend, shade, connect, occluded), queues
                                     in between.  Default: stages are merged where that saves queue traffic and
                                     launches with the same rays and results -- scenes whose nodes, triangles and
                                     shading tables fit in 24 KB of LDS: shading traces the rays it spawns (light-sample
                                     rays, sun / sky rays, continuation rays of the chains), and raygen + view ray + its
                                     shading are one kernel; larger scenes in LDS: the view rays are generated by the
                                     kernel that traces them; mode 0 with sky and no light group: the kernel that traces
                                     the sun / sky rays shades the view-ray hits itself                             */
#define PT_FLAG_QUEUED_VIEW_RAYS 256u /* with traced shading: keep raygen, extend and shade as separate launches for the
                                     view rays (default: one kernel generates a sample's view ray, traces it and shades
                                     its hit)                                                                        */
#define PT_FLAG_STATIC_RUNS 512u  /* the one-kernel-per-sample path: every workgroup keeps to its own share of the samples
                                     (default: the 64-sample runs of a large launch, and the last quarter of those
                                     of a small one, are a pool that workgroups take from as they finish)        */
#define PT_FLAG_NO_TREELETS 1024u   /* scenes traversed from global memory: keep the builder's order of the node pairs
                                     (default: pairs are laid out so that a 128-byte line holds a pair with
                                     descendants of it)                                                              */
#define PT_FLAG_NO_PIXEL_SUMS 2048u /* mode 0 on the one-kernel-per-sample path: keep a radiance record per sample and add the
                                     records up in a second launch, as mode 1 does (default, for windows of at least
                                     786432 pixels or one sample per pixel: a lane renders the samples of one pixel one
                                     after the other and adds them up in registers; same sums, bit for bit)            */
#define PT_FLAG_NO_HIT_ORDER 4096u  /* scenes traversed from global memory: shade the hits in the order of their rays (default
                                     for a scene with at least half as many triangles as the frame has pixels: within
                                     each 1/1024 of the hit queue, in the order of the grid cells the hit positions lie
                                     in, so that the shadow rays in flight together start close to each other; same
                                     image, bit for bit)                                                                  */
#define PT_FLAG_HIT_ORDER 8192u     /* that order for any scene traversed from global memory, however coarse its triangles  */

typedef struct PtMaterial { float emissive; float color[3]; } PtMaterial;        /* rendering.d:33-37 */
typedef struct PtPixel { float color[3]; float n; float sum[3]; } PtPixel;        /* rendering.d:24-30 */

/* Kernel kinds reported by ptgpu_get_stats: one per kernel of the render path.  OTHER = the queue scans. */
enum { PT_K_EXTEND = 0, PT_K_CONNECT = 1, PT_K_OCCLUDED = 2, PT_K_SHADE = 3, PT_K_ACCUM = 4, PT_K_OTHER = 5,
       PT_K_RAYGEN = 6,          /* view rays generated into a queue                                                 */
       PT_K_RAYGEN_EXTEND = 7,   /* view rays generated by the kernel that traces them (scenes in LDS)              */
       PT_K_SHADE_OCCLUDED = 8,  /* the sun / sky traversal kernel that shades the view-ray hits it reads            */
       PT_K_SHADE_TRACED = 9,    /* shading of chain hits that traces the rays it spawns, all bounce rounds          */
       PT_K_SAMPLE_TRACED = 10,  /* a whole sample per lane: raygen, view ray, shading and the rays it spawns        */
       PT_K_PIXEL_TRACED = 11,   /* a whole pixel per lane (mode 0): its samples one after the other, summed in registers */
       PT_K_COUNT = 12 };

typedef struct PtStats {
  uint64_t raysPrimary;      /* Scene.trace calls from computeL             (rendering.d:471)            */
  uint64_t raysLight;        /* Scene.trace calls from computeLdirect        (rendering.d:414)            */
  uint64_t raysIndirect;     /* Scene.trace calls from computeLindirect      (rendering.d:389)            */
  uint64_t raysOcclusion;    /* Scene.hitsNothing calls                      (rendering.d:440,455)        */
  uint64_t samples;          /* computeL invocations                                                     */
  uint64_t nodesClosest, trisClosest;   /* with PT_FLAG_COUNT_VISITS: box tests / triangle tests, closest-hit rays */
  uint64_t nodesAny, trisAny;           /* same for any-hit rays                                          */
  uint64_t raysCulled;       /* view rays answered without a traversal: pixel outside every triangle's screen
                                rectangle, or ray outside the scene's bounding box (included in raysPrimary)  */
  uint64_t hits;             /* Scene.trace calls of computeL / computeLindirect that hit: vertices shaded     */
  uint64_t envEntries;       /* shaded vertices that trace a sun ray, a sky ray or both (one queue entry each) */
  uint64_t hitsPrimary;      /* of hits: view rays that hit                                                    */
  uint64_t raysPrimaryInShade;   /* of raysPrimary / raysLight / raysIndirect / raysOcclusion: rays the shading kernel */
  uint64_t raysLightInShade;     /* (or the whole-sample kernel) traced itself                                          */
  uint64_t raysIndirectInShade;  /* instead of queueing them for the traversal kernels (small scenes, see              */
  uint64_t raysOcclusionInShade; /* PT_FLAG_NO_FUSE)                                                                   */
  uint64_t raysLightInSample;    /* of the three above: rays traced by the whole-sample kernel (PT_K_SAMPLE_TRACED); the rest  */
  uint64_t raysIndirectInSample; /* were traced by the shading of chain hits (PT_K_SHADE_TRACED, or PT_K_SHADE with             */
  uint64_t raysOcclusionInSample;/* PT_FLAG_QUEUED_VIEW_RAYS)                                                                   */
  uint64_t chainHitsFromSample;  /* hits of the chains' first rays that the whole-sample kernel appended for PT_K_SHADE_TRACED  */
  uint64_t samplesConstantSky;   /* of raysCulled: samples of pixels whose every sample is the sky constant (nothing is read    */
                                 /* or written for them until accumulate adds the constant)                                     */
  uint64_t sceneBytes;           /* nodes + triangles as the traversal kernels read them (LDS copy or global memory)            */
  double kernelMs[PT_K_COUNT];          /* with PT_FLAG_TIME_KERNELS: summed event time per kernel kind   */
  uint64_t kernelLaunches[PT_K_COUNT];
  double renderMs;           /* event time first kernel -> last kernel of all ptgpu_render calls          */
  uint32_t numTriangles, numNodes, bvhDepth, numLightGroups;
  uint32_t sceneInLds;       /* 1 if nodes + triangles are staged in LDS by the traversal kernels         */
  uint32_t bvhOnGpu;         /* 1 if the BVH of the current scene was built on the GPU                    */
  double bvhBuildSeconds;    /* wall time of that build (host or GPU, transfers included)                 */
  double maskBuildMs;        /* host time of the last build + upload of the view-ray masks (done inside the first
                                ptgpu_render after the scene or the camera changed)                         */
} PtStats;

const char* ptgpu_last_error(void);

/* New!Camera + New!Scene happen on a handle.  Replaces the globals g_camera/g_scene (rendering.d:17-18). */
int ptgpu_create(const PtConfig* cfg, PtHandle** out);
int ptgpu_destroy(PtHandle* h);
/* Switch any PtConfig field except device/width/height/batchSlots after creation. */
int ptgpu_set_config(PtHandle* h, const PtConfig* cfg);

/* Scene from flattened triangles: what Scene's accessors expose after construction (scene.d:925-946):
 * v = T x (v0,v1,v2) world-space positions, n = T x 3 vertex normals, matIdx = T material indices,
 * mats = M materials.  Builds the acceleration structure, the light groups (one per emissive material,
 * rendering.d:116-161) and uploads everything. */
int ptgpu_upload_scene(PtHandle* h, const float* v, const float* n, const uint32_t* matIdx, const PtMaterial* mats,
                       uint32_t numTriangles, uint32_t numMaterials);

/* New!Scene(path, matFunc, modelMatrix) for a .thModel file (scene.d:193-321, modelloader.d:125-704).
 * materialTable: 0 = fillMaterial (rendering.d:247-275), 1 = fillMaterialCitymap (rendering.d:195-244).
 * modelMatrix: 16 floats, row-major, row-vector convention; NULL = inverse(root node transform), which
 * puts the scene in raw mesh space. */
int ptgpu_load_thmodel(PtHandle* h, const char* path, int materialTable, const float* modelMatrix);

/* Read the flattened scene back (host copies): pass NULL to skip an array.  Sizes from ptgpu_get_stats. */
int ptgpu_get_scene(PtHandle* h, float* v, float* n, uint32_t* matIdx, PtMaterial* mats, uint32_t* numMaterials);
/* Light table as built by loadScene: for group g, up to cap rows of 11 floats (v0,e1,e2,area,probability). */
int ptgpu_get_light_group(PtHandle* h, uint32_t g, float* rows11, uint32_t cap, uint32_t* count, float* totalArea);
/* Flattened BVH: nodes = numNodes x 8 floats (min.xyz, ref, max.xyz, unused), refs as uint32 bit patterns;
 * triOrder = numTriangles original triangle ids in leaf order.  When the scene is traversed from global memory
 * (PtStats.sceneInLds == 0) the boxes are the ones the kernels use: rounded outwards to a 16-bit grid over the root box. */
int ptgpu_get_bvh(PtHandle* h, float* nodes8, uint32_t* triOrder);

/* Exactly Camera's private state (camera.d:11-12) ... */
int ptgpu_set_camera(PtHandle* h, const float pos[3], const float dir[3], const float right[3], const float up[3],
                     float tanFov, float aspect);
/* ... or Camera(fov, aspect) + setTransform(from, to, up) (camera.d:16-29). out14 (optional) receives
 * pos dir right up tanFov aspect. */
int ptgpu_camera_look_at(PtHandle* h, float fovDegrees, float aspect, const float from[3], const float to[3],
                         const float up[3], float* out14);

/* computeOutputColor(pixelOffset, pixels[0..pixelCount], gen) repeated spp times (rendering.d:494-500):
 * every pixel of the window gets spp more samples; sum += L, n += 1, color = sum / n.  Asynchronous on
 * the handle's stream, with two exceptions: the first call after the scene or the camera changed builds the view-ray
 * masks on the GPU and waits for that build once (PtStats.maskBuildMs: 1-4 ms), and the first call of a new batch shape
 * allocates the batch state.  ptgpu_render_noalloc does neither allocation nor, once the masks exist, any waiting. */
int ptgpu_render(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, uint32_t spp);
/* The reference's task decomposition (main.d:184-192) as one call: renders the scanline bands
 * b = bandFirst, bandFirst + bandStride, ... (band = bandRows rows) -- the per-GPU share of a frame. */
int ptgpu_render_bands(PtHandle* h, uint32_t bandRows, uint32_t bandFirst, uint32_t bandStride, uint32_t spp);
int ptgpu_synchronize(PtHandle* h);

/* ptgpu_render for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same result as ptgpu_render for the same
 * window and spp, bit for bit, but the wavefront state (ray, hit and shadow queues, per-sample radiance, queue
 * counters) lives in a static device array of PT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds for
 * the life of the process, and the call makes no hipMalloc / hipFree, creates no event and builds nothing on the
 * host.  The whole window is ONE batch: pixelCount * spp must fit the array for the handle's scene and configuration
 * (a scene whose samples are one launch each, i.e. up to 24 KB of nodes, triangles and shading tables, keeps no ray
 * queues: 1769472 samples in mode 0, 65536 in mode 1 with five chains; other scenes, mode 0: 655360 samples without
 * light groups, 458752 with one; mode 1 with five chains: 65536); a larger request fails with PT_E_INVALID and a message
 * that states the limit.  The view-ray masks are used when an earlier ptgpu_render built them for the current
 * camera, otherwise the rays are culled by the scene's bounding box (same image).  All calls on one GPU share the
 * array.  The library puts them in order: a call's kernels start behind the last kernel of the previous call on that
 * GPU, whatever handle, stream or host thread issued it (a stream wait on an event, no host wait), so calls that would
 * overlap run one after the other instead of racing.  Under a stream capture no such wait is recorded: the order is
 * the capturing caller's to give. */
#define PT_NOALLOC_SCRATCH_BYTES (96u << 20)
int ptgpu_render_noalloc(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, uint32_t spp);

/* Pixel[] access for drawScreen (main.d:275) and the project file (main.d:227-234, 333-340). */
int ptgpu_read_pixels(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, PtPixel* out);
int ptgpu_write_pixels(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, const PtPixel* in);
int ptgpu_clear_pixels(PtHandle* h);
/* drawScreen's scalar path on the device: saturate -> sRGB OETF -> (ubyte)(x*255), 3 bytes per pixel (main.d:82-95). */
int ptgpu_tonemap(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, uint8_t* rgb);
/* The same in two halves, for the progressive loop (main.d:273-287: draw pass k, then render pass k + 1): begin queues
 * the tone map behind the passes rendered so far and its copy to the host on a second stream, and returns; calls made
 * before end (ptgpu_render of the next pass) run beside the copy; end waits until rgb holds the bytes.  One begin may
 * be outstanding per handle.  rgb must stay valid until end. */
int ptgpu_tonemap_begin(PtHandle* h, uint32_t pixelOffset, uint32_t pixelCount, uint8_t* rgb);
int ptgpu_tonemap_end(PtHandle* h);
/* Page-locked host memory for the caller's Pixel[] (NewArray!Pixel, main.d:175) and frame bytes: ptgpu_read_pixels,
 * ptgpu_write_pixels and ptgpu_tonemap* copy straight into / out of such memory; any other buffer goes through a
 * page-locked staging buffer of the handle and one more memcpy.  (Page-locking memory the caller already owns,
 * hipHostRegister, is refused by the host driver this was developed on; hence allocation.) */
int ptgpu_host_alloc(PtHandle* h, uint64_t bytes, void** out);
int ptgpu_host_free(PtHandle* h, void* p);

/* Scene.trace (scene.d:681-743) for caller-supplied rays, 6 floats each (pos, dir): closest front-face hit.
 * Outputs may be NULL.  triId = -1 and t = FLT_MAX on a miss. */
int ptgpu_scene_trace(PtHandle* h, const float* rays6, uint32_t count, int32_t* triId, float* t, float* u, float* v,
                      float* normal3);
/* Scene.trace, texcoord variant (scene.d:745-796): closest hit of caller-supplied rays on either side of the
 * surface, or front faces only when ignoreBackfaces != 0 (IgnoreBackfaces, scene.d:62-66).  result[i] is TraceResult
 * (scene.d:55-60): 0 FrontFaceHit, 1 BackFaceHit, 2 NoHit.  tc2 = 2 floats per ray, TriangleData.tex interpolated by
 * interpolate() (scene.d:68-105); (0, 0) on a miss.  triId / t as for ptgpu_scene_trace.  triId, t, tc2 may be NULL. */
int ptgpu_scene_trace_tex(PtHandle* h, const float* rays6, uint32_t count, int ignoreBackfaces, int32_t* result,
                          int32_t* triId, float* t, float* tc2);
/* TriangleData.tex (scene.d:305-307) of the current scene: numTriangles x 3 x (s, t) in triangle input order.
 * ptgpu_load_thmodel fills it from the file (texcoord set 1 as the reference indexes it; the last set present when
 * the mesh has fewer; zeros when it has none); ptgpu_upload_scene leaves it zero.  set replaces it. */
int ptgpu_set_texcoords(PtHandle* h, const float* tex6);
int ptgpu_get_texcoords(PtHandle* h, float* tex6);
/* Scene.hitsNothing (scene.d:801-838): out[i] = 1 if ray i hits no geometry. */
int ptgpu_scene_hits_nothing(PtHandle* h, const float* rays6, uint32_t count, uint8_t* out);
/* Parity probe: getViewRay without jitter (rendering.d:278-285) + Scene.trace for a pixel window. */
int ptgpu_trace_primary(PtHandle* h, uint32_t pixelOffset, uint32_t count, int32_t* triId, float* t, float* u, float* v);
/* The view rays themselves, 6 floats each. */
int ptgpu_view_rays(PtHandle* h, uint32_t pixelOffset, uint32_t count, float* rays6);
/* First n words of RNG stream (seed, pixel, sample, stream) as the kernels draw them. */
int ptgpu_rng_stream(PtHandle* h, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, uint32_t* out);
/* Parity probe: angleToDirection(phi, psi, normal) (rendering.d:316-330) as the shading kernels evaluate it, for n rows
 * of 5 floats (phi, psi, normal.xyz); a row with a zero normal gets angleToLocalDirection(phi, psi) (rendering.d:309-314).
 * out3 = n x 3.  cosf / sinf are one fixed sequence of binary32 operations (pt_math.h: sincosFixed). */
int ptgpu_angle_directions(PtHandle* h, const float* in5, uint32_t n, float* out3);
/* Parity probe (no reference counterpart): the masks the view rays of getViewRay (rendering.d:278-285) are culled
 * with for the current scene and camera, ceil(width*height/32) words each, either may be NULL.  Bit p of covered is
 * set when pixel p may see a triangle; bit p of constantSky when it cannot and in addition none of its sample
 * positions sees sampleSky's sun disc (rendering.d:332-339), so that every sample of the pixel is the sky constant.
 * *usable = 0 when no mask can be built for this camera (a triangle reaches behind it); rendering then culls by the
 * scene's bounding box.  Works on host-only handles. */
int ptgpu_get_coverage_mask(PtHandle* h, uint32_t* covered, uint32_t* constantSky, uint32_t* usable);

int ptgpu_get_stats(PtHandle* h, PtStats* out);
int ptgpu_reset_stats(PtHandle* h);

/* Plumbing for callers that own device memory and streams (one process per GPU with torch.distributed):
 * the accumulator is width*height float4 (sum.r, sum.g, sum.b, n) in device memory. */
int ptgpu_bind_accumulator(PtHandle* h, void* devicePtr, uint64_t bytes);
int ptgpu_get_accumulator(PtHandle* h, void** devicePtr, uint64_t* bytes);
int ptgpu_set_stream(PtHandle* h, void* hipStream);

/* ---- several GPUs of one node behind one object -------------------------------------------------------------------
 * The reference spreads a frame over worker threads as one task per 4 scanlines with a barrier per pass
 * (main.d:184-215, 273-288).  A group does the same over GPUs: the scene is replicated, GPU i of G renders the bands
 * i, i + G, ... (PT_SPLIT_BANDS: the frame is bit-identical for any G), or every pixel with the sample indices
 * i, i + G, ... (PT_SPLIT_SAMPLES: G x spp samples per pixel), each on its own host thread; ptgpu_group_combine puts the
 * frame together on the root GPU with one exchange over RCCL (bands: every GPU sends its rows, 1/G of the frame, to the
 * root; samples: one reduce(sum) of the (sum, n) accumulators).  The library is opened with dlopen when the first group
 * is created; one GPU works without it. */
typedef struct PtGroup PtGroup;
#define PT_SPLIT_BANDS 0
#define PT_SPLIT_SAMPLES 1
/* cfg->device is ignored; devices = HIP ordinals (all -1: a host-only group for scene preparation); root indexes devices. */
int ptgpu_group_create(const PtConfig* cfg, const int32_t* devices, uint32_t numDevices, uint32_t root, PtGroup** out);
int ptgpu_group_destroy(PtGroup* g);
int ptgpu_group_info(PtGroup* g, uint32_t* numDevices, uint32_t* root, uint32_t* usesRccl);
/* The renderer of one GPU (any ptgpu_* call works on it) and the handle whose Pixel[] is the combined frame: read it
 * with ptgpu_read_pixels / ptgpu_tonemap* after ptgpu_group_combine. */
int ptgpu_group_handle(PtGroup* g, uint32_t rank, PtHandle** out);
int ptgpu_group_frame(PtGroup* g, PtHandle** out);
/* ptgpu_load_thmodel / ptgpu_upload_scene / ptgpu_set_camera / ptgpu_camera_look_at on every GPU. */
int ptgpu_group_load_thmodel(PtGroup* g, const char* path, int materialTable, const float* modelMatrix);
int ptgpu_group_upload_scene(PtGroup* g, const float* v, const float* n, const uint32_t* matIdx, const PtMaterial* mats,
                             uint32_t numTriangles, uint32_t numMaterials);
int ptgpu_group_set_camera(PtGroup* g, const float pos[3], const float dir[3], const float right[3], const float up[3],
                           float tanFov, float aspect);
int ptgpu_group_camera_look_at(PtGroup* g, float fovDegrees, float aspect, const float from[3], const float to[3],
                               const float up[3]);
/* One pass of spp samples per pixel, shared out as above (asynchronous on every GPU).  Each GPU keeps its own Pixel[]
 * between passes; the split is fixed from the first pass until ptgpu_group_clear. */
int ptgpu_group_render(PtGroup* g, int split, uint32_t bandRows, uint32_t spp);
int ptgpu_group_combine(PtGroup* g);
int ptgpu_group_clear(PtGroup* g);
/* The scanlines of the bands bandFirst, bandFirst + bandStride, ... (main.d:184-192 with bandRows = 4), in order:
 * up to cap of them in rows (may be NULL), their number in *count.  Pure host arithmetic. */
int ptgpu_band_rows(uint32_t height, uint32_t bandRows, uint32_t bandFirst, uint32_t bandStride, uint32_t* rows, uint32_t cap,
                    uint32_t* count);

#ifdef __cplusplus
}
#endif
#endif
