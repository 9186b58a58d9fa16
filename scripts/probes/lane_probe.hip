// Micro-benchmark: what a dependent gather costs a wave when only K of its 64 lanes take part, and how that depends on
// the number of 16-byte loads a lane makes from its 128-byte record (1, 2 or 6) and on where the table lives (2 MB:
// the L2 of one XCD holds it; 128 MB: the memory-side cache; 2 GB: HBM).  32 waves per CU, every lane follows its own
// chain of records.  If a load instruction costs the same whatever K is, lane utilisation is what a traversal from
// global memory has to buy; if it costs per participating lane, fewer requests per ray is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15; return x; }

template <int LOADS>
__global__ void __launch_bounds__(512) k_probe(const float4* __restrict__ table, uint32_t recMask, uint32_t iters, uint32_t K, float* out) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  uint32_t key = mix(tid + 1);
  float acc = 0.0f;
  if (lane < K) {
    for (uint32_t it = 0; it < iters; it++) {
      const float4* r = table + (size_t)(key & recMask) * 8;
      float4 a = r[0];
      float s = a.x;
      if (LOADS >= 2) { float4 b = r[1]; s += b.y; }
      if (LOADS >= 6) { float4 c = r[2], d = r[3], e = r[4], f = r[5]; s += c.x + d.y + e.z + f.w; }
      acc += s;
      key = mix(key + __float_as_uint(s) + it);
    }
  }
  out[tid] = acc;
}

int main() {
  float4* table; float* out;
  const size_t maxBytes = (size_t)2 << 30;
  hipMalloc(&table, maxBytes); hipMemset(table, 0, maxBytes);
  const int blocks = 256 * 4, threads = 512;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const uint32_t iters = 400;
  const size_t sizes[3] = {(size_t)2 << 20, (size_t)128 << 20, (size_t)2 << 30};
  for (int sz = 0; sz < 3; sz++)
    for (int loads : {1, 2, 6})
      for (uint32_t K : {8u, 16u, 32u, 64u}) {
        const uint32_t recMask = (uint32_t)(sizes[sz] / 128 - 1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
          hipEventRecord(a);
          if (loads == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(threads), 0, 0, table, recMask, iters, K, out);
          if (loads == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(threads), 0, 0, table, recMask, iters, K, out);
          if (loads == 6) hipLaunchKernelGGL(k_probe<6>, dim3(blocks), dim3(threads), 0, 0, table, recMask, iters, K, out);
          hipEventRecord(b); hipEventSynchronize(b);
          hipEventElapsedTime(&ms, a, b);
        }
        const double waveSteps = (double)blocks * threads / 64 * iters, laneSteps = waveSteps * K;
        printf("table %4zu MB  loads %d  K %2u: %7.3f ms  %6.2f G wave-steps/s  %7.2f G lane-records/s  %6.1f cycles/CU per wave-step\n",
               sizes[sz] >> 20, loads, K, ms, waveSteps / ms / 1e6, laneSteps / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / waveSteps);
      }
  return 0;
}
