// Micro-benchmark: how fast can a wave gather 64-byte records from a 1 GB table at random addresses?
//   mode 0: each lane loads its own record with four 16-byte loads (what the HBM traversal does per step)
//   mode 1: each lane loads 32 bytes of its own record (two loads)
//   mode 2: the four lanes of a quad load one record together (16 bytes each), four records in turn
//   mode 3: each lane loads 16 bytes of its own record (one load)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15; return x; }

template <int MODE>
__global__ void __launch_bounds__(512) k_gather(const float4* __restrict__ table, uint32_t recMask, uint32_t iters, float* out) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, c = lane & 3;
  uint32_t key = mix(tid + 1);
  float acc = 0.0f;
  for (uint32_t it = 0; it < iters; it++) {
    uint32_t rec = key & recMask;
    if (MODE == 0) {
      float4 a = table[rec * 4], b = table[rec * 4 + 1], d = table[rec * 4 + 2], e = table[rec * 4 + 3];
      acc += a.x + b.y + d.z + e.w;
      key = mix(key + __float_as_uint(a.x));
    } else if (MODE == 1) {
      float4 a = table[rec * 4], b = table[rec * 4 + 1];
      acc += a.x + b.y;
      key = mix(key + __float_as_uint(a.x));
    } else if (MODE == 3) {
      float4 a = table[rec * 4];
      acc += a.x;
      key = mix(key + __float_as_uint(a.x));
    } else {
      float s = 0.0f;
      #pragma unroll
      for (int r = 0; r < 4; r++) {
        uint32_t nr = (uint32_t)__shfl((int)rec, (lane & ~3u) + r);
        float4 v = table[nr * 4 + c];
        s += v.x;
      }
      acc += s;
      key = mix(key + __float_as_uint(s));
    }
  }
  out[tid] = acc;
}

int main() {
  const size_t recs = 1u << 24;                        // 16 M records x 64 B = 1 GB
  float4* table; float* out;
  hipMalloc(&table, recs * 64); hipMemset(table, 0, recs * 64);
  const int blocks = 256 * 4, threads = 512;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const uint32_t iters = 200;
  for (int mode = 0; mode < 4; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(threads), 0, 0, table, (uint32_t)recs - 1, iters, out);
      if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, table, (uint32_t)recs - 1, iters, out);
      if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(threads), 0, 0, table, (uint32_t)recs - 1, iters, out);
      if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(threads), 0, 0, table, (uint32_t)recs - 1, iters, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double gathers = (double)blocks * threads * iters;
      if (rep) printf("mode %d: %.3f ms, %.2f G records/s per lane-gather\n", mode, ms, gathers / ms / 1e6);
    }
  }
  return 0;
}
