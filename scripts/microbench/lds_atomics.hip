// Micro-benchmark: cost of wave-level sums through LDS float atomics (ds_add_f32) versus the register reduction
// used by render_bwd_packed_kernel.  One wave per workgroup (as the compositing kernels), 8160 workgroups; every
// iteration adds NV values from the `active` first / scattered lanes into one LDS row, optionally interleaved with
// FILL dependent-free FMAs per iteration (stand-in for the kernel's own vector work).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NV, int FILL>
__global__ __launch_bounds__(64) void lds_atomic_kernel(float* out, int iters, int active, int scattered) {
  __shared__ float acc[64 * 12];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 12; i += 64) acc[i] = 0.f;
  __syncthreads();
  const bool on = scattered ? ((lane * 7 + 3) % 64) < active : lane < active;
  float f[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = 1.0f + lane * 0.001f + k;
  const float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
    float* row = acc + (it & 63) * 12;
#pragma unroll
    for (int r = 0; r < FILL / 8; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = __builtin_fmaf(f[k], 1.0000001f, 0.5f);
    }
    if (on) {
#pragma unroll
      for (int k = 0; k < NV; ++k) atomicAdd(&row[k], v + k + f[k & 7] * 1e-30f);
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = lane; i < 64 * 12; i += 64) s += acc[i];
#pragma unroll
  for (int k = 0; k < 8; ++k) s += f[k] * 1e-30f;
  out[blockIdx.x * 64 + lane] = s;
}

template <int NV, int FILL>
static int run(float* out, int active, int scattered) {
  const int iters = 327, blocks = 8160;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((lds_atomic_kernel<NV, FILL>), dim3(blocks), dim3(64), 0, 0, out, iters, active, scattered);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  printf("NV=%2d FILL=%3d active=%2d scattered=%d : %8.1f us  (%.1f ns per iteration per CU-slot)\n", NV, FILL, active,
         scattered, best * 1e3, best * 1e6 / (iters * (blocks / 256.0)));
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, 8160 * 64 * 4));
  for (int sc = 0; sc < 2; ++sc)
    for (int act : {0, 1, 4, 8, 12, 16, 32, 64}) { if (run<10, 0>(out, act, sc)) return 1; }
  for (int act : {0, 8, 12, 16}) { if (run<10, 120>(out, act, 1)) return 1; }
  for (int act : {0, 8, 12, 16}) { if (run<13, 120>(out, act, 1)) return 1; }   // 1.3 live pairs on average
  if (run<0, 120>(out, 0, 0)) return 1;
  if (run<0, 168>(out, 0, 0)) return 1;      // the vector work of today's kernel with its register reduction
  return 0;
}
