// Micro-benchmark: throughput of device-scope global atomics on MI355X (informs binning designs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <bool RET>
__global__ void atom_kernel(uint32_t* counters, uint32_t ncounters, uint32_t n, uint32_t* sink, int local) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // local != 0: neighbouring threads hit neighbouring counters (a Gaussian's tiles); else random
  const uint32_t c = local ? (hash(i >> 2) + (i & 3)) % ncounters : hash(i) % ncounters;
  if (RET) { const uint32_t r = atomicAdd(&counters[c], 1u); if (r == 0xffffffffu) sink[0] = r; }
  else atomicAdd(&counters[c], 1u);
}
__global__ void store_kernel(uint32_t* out, uint32_t n, uint32_t span) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[hash(i) % span] = i;
}

// the same stores, but a workgroup only writes into the eighth of the span that belongs to its XCD (workgroup b runs
// on XCD b % 8): every 128-byte line is then assembled in ONE L2 instead of being written back in pieces by eight
__global__ void store_xcd_kernel(uint32_t* out, uint32_t n, uint32_t span) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t part = span / 8;
  if (i < n) out[(blockIdx.x & 7) * part + hash(i) % part] = i;
}

// the XCD-local stores in RUNS: `run` neighbouring lanes write neighbouring words (what a scatter that first groups its
// chunk's instances by tile in LDS would issue: a tile's entries of one chunk are contiguous)
__global__ void store_xcd_runs_kernel(uint32_t* out, uint32_t n, uint32_t span, uint32_t run) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t part = span / 8;
  if (i < n) out[(blockIdx.x & 7) * part + (hash(i / run) % (part / run)) * run + i % run] = i;
}

int main() {
  const uint32_t n = 2665429;
  uint32_t *cnt, *sink;
  CK(hipMalloc(&cnt, 64u << 20)); CK(hipMalloc(&sink, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const uint32_t sizes[] = {8160, 250000, 2665429};
  for (uint32_t nc : sizes) for (int local = 0; local < 2; ++local) for (int ret = 0; ret < 2; ++ret) {
    CK(hipMemset(cnt, 0, 64u << 20));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(a));
      if (ret) hipLaunchKernelGGL(atom_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, nc, n, sink, local);
      else hipLaunchKernelGGL(atom_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, nc, n, sink, local);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("atomics n=%u counters=%u local=%d returning=%d : %.1f us  (%.1f G/s)\n", n, nc, local, ret, best * 1e3, n / best * 1e-6);
  }
  for (uint32_t span : {2665429u, 8u << 20}) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(store_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, n, span);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("random 4B stores n=%u span=%u : %.1f us\n", n, span, best * 1e3);
  }
  for (uint32_t span : {2665429u, 8u << 20}) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(store_xcd_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, n, span);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("random 4B stores, XCD-local eighths n=%u span=%u : %.1f us\n", n, span, best * 1e3);
  }
  for (uint32_t run : {1u, 2u, 4u, 8u, 16u}) {
    const uint32_t span = 2665429u;
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(store_xcd_runs_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, n, span, run);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("XCD-local stores in runs of %u words n=%u span=%u : %.1f us\n", run, n, span, best * 1e3);
  }
  return 0;
}
