// View-ray masks built on the GPU (mask_gpu.hip).
#pragma once
#include "pt_math.h"
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pt {

// Words of the row-padded scratch image buildCoverageMaskGpu rasterises into.
size_t maskRowWords(uint32_t W, uint32_t H);

// dVerts9: T x 9 floats (v0, v1, v2) in device memory; scale: largest finite |coordinate| of the scene.  dRows:
// maskRowWords(W, H) words, dBigList: T words, dFlags: 2 words (scratch); dBits / dSky: ceil(W*H/32) words each, the
// masks of scene_host.h: buildCoverageMask.  Queues everything on `stream`, then waits for it to learn whether a
// triangle reaches behind the camera (*usable = false: the masks must not be used).
hipError_t buildCoverageMaskGpu(hipStream_t stream, const float* dVerts9, uint32_t T, const Camera& cam, uint32_t W, uint32_t H,
                                double scale, uint32_t* dRows, uint32_t* dBigList, uint32_t* dFlags, uint32_t* dBits, uint32_t* dSky,
                                bool* usable);

}  // namespace pt
