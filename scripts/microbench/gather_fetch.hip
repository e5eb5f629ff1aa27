// Micro-benchmark: what do FETCH_SIZE / WRITE_SIZE count for the access patterns of the compositing kernels?
// MI355X_MICROARCH.md calibrates "HBM bytes = 2 * FETCH_SIZE KiB" for wide streaming reads (gfx950 counts a 128-byte
// request as 64).  K6 / K7 gather one 64-byte record per tile instance from a table far larger than the L2s, and K7
// stores 40-byte records as five 8-byte stores per lane to scattered slots.  Three kernels with KNOWN byte counts:
//   stream_kernel   reads n * 16 bytes contiguously (float4 per lane)            -> calibration point of the guide
//   gather64_kernel reads n records of 64 bytes (4 x float4) at random, 64-byte aligned places of a 1 GiB table
//   scatter40_kernel writes n records of 40 bytes (5 x float2) at random 40-byte slots
// Run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  (and WRITE_SIZE) and compare the counters with the bytes printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void stream_kernel(const float4* __restrict__ in, uint32_t n, float* sink) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = in[i];
  if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = v.x;
}
__global__ void gather64_kernel(const float4* __restrict__ table, uint32_t records, uint32_t n, float* sink) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4* r = table + (size_t)(hash(i) % records) * 4;
  const float4 a = r[0], b = r[1], c = r[2], d = r[3];
  if (a.x + b.y + c.z + d.w == 12345.678f) sink[0] = a.x;
}
__global__ void scatter40_kernel(float2* __restrict__ out, uint32_t slots, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2* r = out + (size_t)(hash(i) % slots) * 5;
#pragma unroll
  for (int k = 0; k < 5; ++k) r[k] = make_float2((float)i, (float)k);
}

int main() {
  const uint32_t n = 4u << 20;                 // 4 Mi accesses
  const uint32_t records = 16u << 20;          // 16 Mi x 64 B = 1 GiB table
  float4* table; float* sink; float2* out;
  CK(hipMalloc(&table, (size_t)records * 64));
  CK(hipMalloc(&out, (size_t)records * 40));
  CK(hipMalloc(&sink, 256));
  CK(hipMemset(table, 0, (size_t)records * 64));
  CK(hipMemset(out, 0, (size_t)records * 40));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream_kernel, dim3(n / 256), dim3(256), 0, 0, table, n, sink);
    hipLaunchKernelGGL(gather64_kernel, dim3(n / 256), dim3(256), 0, 0, table, records, n, sink);
    hipLaunchKernelGGL(scatter40_kernel, dim3(n / 256), dim3(256), 0, 0, out, records, n);
  }
  CK(hipDeviceSynchronize());
  printf("stream_kernel    reads  %.1f MiB per launch\n", n * 16.0 / (1 << 20));
  printf("gather64_kernel  reads  %.1f MiB per launch (random 64-byte records of a 1 GiB table)\n", n * 64.0 / (1 << 20));
  printf("scatter40_kernel writes %.1f MiB per launch (random 40-byte records as 5 x 8-byte stores)\n", n * 40.0 / (1 << 20));
  return 0;
}
