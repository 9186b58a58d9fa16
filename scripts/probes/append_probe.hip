// Micro-benchmark of shade's memory pattern: 1024 workgroups, each reading consecutive 32-byte records of its own
// range and appending NA 16-byte streams to its own segment of NA arrays (compacted by a per-wave ballot, as the
// queue appends are), optionally plus a 16-byte store scattered by a per-record slot.  No arithmetic to speak of:
// what it reaches is the ceiling of this access pattern.
//   append_probe <records per workgroup> <threads>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15; return x; }

template <int NA, bool SCATTER>
__global__ void k_append(const float4* __restrict__ inA, const float4* __restrict__ inB, float4* __restrict__ out, float4* __restrict__ lacc,
                         uint32_t perBlock, size_t arrayStride, uint32_t keepMask) {
  __shared__ uint32_t cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  size_t base = (size_t)blockIdx.x * perBlock;
  for (uint32_t i = threadIdx.x; i < perBlock; i += blockDim.x) {
    float4 a = inA[base + i], b = inB[base + i];
    uint32_t slot = __float_as_uint(b.z);
    bool keep = (mix(slot) & 0xFFu) < keepMask;                       // ~85 % of the records emit
    unsigned long long m = __ballot(keep);
    uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t wbase = 0;
    if ((threadIdx.x & 63) == 0 && m) wbase = atomicAdd(&cnt, (uint32_t)__popcll(m));
    wbase = __builtin_amdgcn_readfirstlane(wbase);
    if (keep) {
      size_t e = base + wbase + rank;
#pragma unroll
      for (int k = 0; k < NA; k++) out[k * arrayStride + e] = make_float4(a.x + k, a.y, b.x, b.y);
    } else if (SCATTER) lacc[slot] = a;
  }
}

int main(int argc, char** argv) {
  const uint32_t perBlock = argc > 1 ? (uint32_t)atoi(argv[1]) : 120000u, threads = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
  const uint32_t blocks = 1024;
  const size_t n = (size_t)blocks * perBlock;
  float4 *inA, *inB, *out, *lacc;
  hipMalloc(&inA, n * 16); hipMalloc(&inB, n * 16); hipMalloc(&out, n * 16 * 4); hipMalloc(&lacc, n * 2 * 16);
  hipMemset(inA, 0, n * 16);
  {  // slots: roughly increasing with gaps, as hits of consecutive view rays are
    uint32_t* h = (uint32_t*)malloc(n * 16);
    for (size_t i = 0; i < n; i++) { h[4 * i] = 0; h[4 * i + 1] = 0; h[4 * i + 2] = (uint32_t)(i * 2 - (i % 7 == 0)); h[4 * i + 3] = 0; }
    hipMemcpy(inB, h, n * 16, hipMemcpyHostToDevice); free(h);
  }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto kernel, int na, bool scatter, uint32_t keepMask) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a);
      hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, inA, inB, out, lacc, perBlock, n, keepMask);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    double keep = keepMask / 256.0;
    double bytes = (double)n * (32.0 + keep * 16.0 * na + (scatter ? (1.0 - keep) * 16.0 : 0.0));
    printf("%-34s %8.3f ms  %7.1f GB/s  (%.2f GB)\n", name, best, bytes / best / 1e6, bytes / 1e9);
  };
  run("read only", k_append<0, false>, 0, false, 218);
  run("1 stream", k_append<1, false>, 1, false, 218);
  run("2 streams", k_append<2, false>, 2, false, 218);
  run("4 streams", k_append<4, false>, 4, false, 218);
  run("4 streams, all lanes", k_append<4, false>, 4, false, 256);
  run("4 streams + scattered 16 B", k_append<4, true>, 4, true, 218);
  return 0;
}
