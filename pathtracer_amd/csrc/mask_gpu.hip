// View-ray masks (scene_host.h: buildCoverageMask) built on the GPU: the projected triangles are rasterised into a
// bit image with atomic ORs, one thread per triangle (one workgroup for a triangle that spans many rows), then packed
// to one bit per pixel together with the constant-sky test.  Same arithmetic as the host builder (mask_math.h), so the
// same bits; what it buys is time -- the host loop takes a second for ten million triangles and blocks the first
// render after every camera change.
#include "mask_gpu.h"
#include "mask_math.h"
#include <hip/hip_runtime.h>

namespace pt {
namespace {

constexpr uint32_t MASK_BIG_ROWS = 16;      // triangles spanning more rows go to the one-workgroup-per-triangle kernel

__device__ __forceinline__ void orSpan(uint32_t* row, uint32_t a, uint32_t b) {
  uint32_t wa = a >> 5, wb = b >> 5;
  uint32_t first = 0xFFFFFFFFu << (a & 31), last = 0xFFFFFFFFu >> (31 - (b & 31));
  if (wa == wb) { atomicOr(&row[wa], first & last); return; }
  atomicOr(&row[wa], first);
  for (uint32_t w = wa + 1; w < wb; w++) if (row[w] != 0xFFFFFFFFu) atomicOr(&row[w], 0xFFFFFFFFu);
  atomicOr(&row[wb], last);
}

// flags[0]: set when a triangle reaches behind the camera plane (no mask can be used); flags[1]: entries of bigList
__global__ void __launch_bounds__(256) k_mask_triangles(MaskCam mc, const float* __restrict__ v9, uint32_t T, uint32_t* rows,
                                                        uint32_t wordsPerRow, uint32_t* flags, uint32_t* bigList) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  MaskTri t;
  int st = maskTriSetup(mc, v9 + 9 * (size_t)i, t);
  if (st == MASK_TRI_BEHIND) { flags[0] = 1u; return; }
  if (st != MASK_TRI_OK) return;
  if (t.rb - t.ra >= MASK_BIG_ROWS) { bigList[atomicAdd(&flags[1], 1u)] = i; return; }
  for (uint32_t r = t.ra; r <= t.rb; r++) {
    uint32_t a, b;
    if (maskRowSpan(t, r, a, b)) orSpan(rows + (size_t)r * wordsPerRow, a, b);
  }
}

__global__ void __launch_bounds__(256) k_mask_big_triangles(MaskCam mc, const float* __restrict__ v9, uint32_t* rows,
                                                            uint32_t wordsPerRow, const uint32_t* flags, const uint32_t* bigList) {
  const uint32_t n = flags[1];
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
    MaskTri t;
    if (maskTriSetup(mc, v9 + 9 * (size_t)bigList[k], t) != MASK_TRI_OK) continue;
    for (uint32_t r = t.ra + threadIdx.x; r <= t.rb; r += blockDim.x) {
      uint32_t a, b;
      if (maskRowSpan(t, r, a, b)) orSpan(rows + (size_t)r * wordsPerRow, a, b);
    }
  }
}

// One thread per word of the per-pixel masks (pixel index = row * W + col).
__global__ void __launch_bounds__(256) k_mask_pack(MaskCam mc, const uint32_t* __restrict__ rows, uint32_t wordsPerRow,
                                                   uint32_t* __restrict__ bits, uint32_t* __restrict__ sky, uint32_t words) {
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= words) return;
  const uint64_t npix = (uint64_t)mc.W * mc.H;
  uint32_t cov = 0, cs = 0;
  for (uint32_t k = 0; k < 32; k++) {
    uint64_t px = (uint64_t)w * 32 + k;
    if (px >= npix) break;
    uint32_t r = (uint32_t)(px / mc.W), c = (uint32_t)(px - (uint64_t)r * mc.W);
    if ((rows[(size_t)r * wordsPerRow + (c >> 5)] >> (c & 31)) & 1u) cov |= 1u << k;
    else if (maskConstantSky(mc, c, r)) cs |= 1u << k;
  }
  bits[w] = cov; sky[w] = cs;
}

}  // namespace

size_t maskRowWords(uint32_t W, uint32_t H) { return (size_t)H * (((size_t)W + 31) / 32); }

hipError_t buildCoverageMaskGpu(hipStream_t stream, const float* dVerts9, uint32_t T, const Camera& cam, uint32_t W, uint32_t H,
                                double scale, uint32_t* dRows, uint32_t* dBigList, uint32_t* dFlags, uint32_t* dBits, uint32_t* dSky,
                                bool* usable) {
  *usable = false;
  MaskCam mc;
  if (!maskCamSetup(cam, W, H, scale, mc)) return hipSuccess;
  const uint32_t wordsPerRow = (W + 31) / 32;
  const uint32_t words = (uint32_t)(((uint64_t)W * H + 31) / 32);
  hipError_t e;
  if ((e = hipMemsetAsync(dRows, 0, maskRowWords(W, H) * 4, stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(dFlags, 0, 8, stream)) != hipSuccess) return e;
  if (T) {
    hipLaunchKernelGGL(k_mask_triangles, dim3((T + 255) / 256), dim3(256), 0, stream, mc, dVerts9, T, dRows, wordsPerRow, dFlags, dBigList);
    hipLaunchKernelGGL(k_mask_big_triangles, dim3(1024), dim3(256), 0, stream, mc, dVerts9, dRows, wordsPerRow, dFlags, dBigList);
  }
  hipLaunchKernelGGL(k_mask_pack, dim3((words + 255) / 256), dim3(256), 0, stream, mc, dRows, wordsPerRow, dBits, dSky, words);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  uint32_t behind = 0;
  if ((e = hipMemcpyAsync(&behind, dFlags, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  *usable = behind == 0;
  return hipSuccess;
}

}  // namespace pt
