// Micro-benchmark: vector-ALU issue rate on gfx950 (MI355X).  Calibrates the "VALU roof" bench.py quotes for the
// compositing kernels (K6 / K7): how many wave64 VALU instructions per second the chip issues for the instruction kinds
// those kernels are made of, as a function of the waves resident per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_bench valu_issue.hip && ./valu_issue_bench
//
// Every wave runs ITER iterations of a block of 32 instructions over 8 independent accumulators (no instruction
// depends on the previous three), written in inline asm so that the compiler neither folds nor reorders them.
// Reported per (instruction kind, waves per SIMD): wall time (hipEvents), G wave-instructions/s for the whole chip,
// and instructions per SIMD per cycle at the measured shader clock (s_memtime delta / wall time).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
enum Kind { FMA = 0, PK_FMA, PK_MUL, EXP, RCP, CNDMASK, DPP_ADD, PERMLANE_SWAP, MIX_K7, CMP_VCC, CMP_SGPR, CND_SGPR_MIX, FMA_CND_MIX,
            FMA_CMP_MIX, MAXF, FMA_CLAMP, PK_ADD, MOV, NKIND };
static const char* kNames[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32",
                                    "v_cndmask_b32", "v_add_f32 row_shr:1 (DPP)", "v_permlane32_swap",
                                    "K7 mix (8 pk_fma : 2 exp : 2 rcp : 4 cndmask : 16 fma)",
                                    "v_cmp_ge_f32 -> vcc", "v_cmp_ge_f32_e64 -> s[n:n+1]",
                                    "1 v_cndmask_e64 (sgpr mask) : 3 v_fma", "1 v_cndmask (vcc) : 3 v_fma",
                                    "1 v_cmp_e64 -> sgpr : 3 v_fma", "v_max_f32", "v_fma_f32 clamp", "v_pk_add_f32", "v_mov_b32"};

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

template <int KIND>
__global__ __launch_bounds__(256) void issue_kernel(float* out, uint64_t* cycles, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
        a7 = a0 + 7.f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float x = 0.999f, y = 1e-3f;
  const f2 px = {0.999f, 0.998f}, py = {1e-3f, 2e-3f};
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == FMA) {
      REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (KIND == PK_FMA) {
      REP4(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px), "v"(py));)
    } else if (KIND == PK_MUL) {
      REP4(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px));)
    } else if (KIND == EXP) {
      REP4(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == RCP) {
      REP4(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == CNDMASK) {
      REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");)
    } else if (KIND == DPP_ADD) {
      REP4(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == PERMLANE_SWAP) {
      REP4(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                        "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == CMP_VCC) {
      REP4(asm volatile("v_cmp_ge_f32 vcc, %0, %8\n v_cmp_ge_f32 vcc, %1, %8\n v_cmp_ge_f32 vcc, %2, %8\n v_cmp_ge_f32 vcc, %3, %8\n"
                        "v_cmp_ge_f32 vcc, %4, %8\n v_cmp_ge_f32 vcc, %5, %8\n v_cmp_ge_f32 vcc, %6, %8\n v_cmp_ge_f32 vcc, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");)
    } else if (KIND == CMP_SGPR) {
      REP4(asm volatile("v_cmp_ge_f32_e64 s[20:21], %0, %8\n v_cmp_ge_f32_e64 s[22:23], %1, %8\n v_cmp_ge_f32_e64 s[24:25], %2, %8\n v_cmp_ge_f32_e64 s[26:27], %3, %8\n"
                        "v_cmp_ge_f32_e64 s[28:29], %4, %8\n v_cmp_ge_f32_e64 s[30:31], %5, %8\n v_cmp_ge_f32_e64 s[32:33], %6, %8\n v_cmp_ge_f32_e64 s[34:35], %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x)
                        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");)
    } else if (KIND == CND_SGPR_MIX) {
      REP4(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_cndmask_b32_e64 %4, %4, %8, s[22:23]\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (KIND == FMA_CND_MIX) {
      REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");)
    } else if (KIND == FMA_CMP_MIX) {
      REP4(asm volatile("v_cmp_ge_f32_e64 s[20:21], %0, %8\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_cmp_ge_f32_e64 s[22:23], %4, %8\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y)
                        : "s20", "s21", "s22", "s23");)
    } else if (KIND == MAXF) {
      REP4(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                        "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (KIND == FMA_CLAMP) {
      REP4(asm volatile("v_fma_f32 %0, %0, %8, %9 clamp\n v_fma_f32 %1, %1, %8, %9 clamp\n v_fma_f32 %2, %2, %8, %9 clamp\n v_fma_f32 %3, %3, %8, %9 clamp\n"
                        "v_fma_f32 %4, %4, %8, %9 clamp\n v_fma_f32 %5, %5, %8, %9 clamp\n v_fma_f32 %6, %6, %8, %9 clamp\n v_fma_f32 %7, %7, %8, %9 clamp\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (KIND == PK_ADD) {
      REP4(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px));)
    } else if (KIND == MOV) {
      REP4(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                        "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else {   // the instruction mix of K7's live path, 32 instructions
      asm volatile("v_pk_fma_f32 %8, %8, %16, %17\n v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n"
                   "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                   "v_pk_fma_f32 %12, %12, %16, %17\n v_pk_fma_f32 %13, %13, %16, %17\n v_pk_fma_f32 %14, %14, %16, %17\n v_pk_fma_f32 %15, %15, %16, %17\n"
                   "v_cndmask_b32 %4, %4, %18, vcc\n v_cndmask_b32 %5, %5, %18, vcc\n v_cndmask_b32 %6, %6, %18, vcc\n v_cndmask_b32 %7, %7, %18, vcc\n"
                   "v_fma_f32 %0, %0, %18, %19\n v_fma_f32 %1, %1, %18, %19\n v_fma_f32 %2, %2, %18, %19\n v_fma_f32 %3, %3, %18, %19\n"
                   "v_fma_f32 %4, %4, %18, %19\n v_fma_f32 %5, %5, %18, %19\n v_fma_f32 %6, %6, %18, %19\n v_fma_f32 %7, %7, %18, %19\n"
                   "v_fma_f32 %0, %0, %18, %19\n v_fma_f32 %1, %1, %18, %19\n v_fma_f32 %2, %2, %18, %19\n v_fma_f32 %3, %3, %18, %19\n"
                   "v_fma_f32 %4, %4, %18, %19\n v_fma_f32 %5, %5, %18, %19\n v_fma_f32 %6, %6, %18, %19\n v_fma_f32 %7, %7, %18, %19\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                     "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                   : "v"(px), "v"(py), "v"(x), "v"(y) : "vcc");
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (r == 12345.678f) out[0] = r;                    // keeps everything alive, never true in practice
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
static int run(float* out, uint64_t* cyc, int cus, hipEvent_t a, hipEvent_t b) {
  const int iters = 4096;
  for (int wps : {1, 2, 4, 5, 8}) {
    // one workgroup of 256 lanes = one wave on each of a CU's four SIMDs; wps workgroups per CU
    const int blocks = cus * wps;
    float best = 1e9f;
    uint64_t cycles = 0;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.5f);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) { best = ms; CK(hipMemcpy(&cycles, cyc, 8, hipMemcpyDeviceToHost)); }
    }
    const double inst = (double)blocks * 4 * iters * 32;       // wave-instructions issued
    const double per_simd_clk = (double)iters * 32 * wps / (double)cycles;   // issued by one SIMD per counter tick
    printf("%-58s waves/SIMD=%d  %8.1f us  %8.1f G wave-inst/s  counter ticks %llu  (%.3f inst/SIMD/tick, tick rate %.1f MHz)\n",
           kNames[KIND], wps, best * 1e3, inst / best * 1e-6, (unsigned long long)cycles, per_simd_clk,
           (double)cycles / best * 1e-3);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %.0f MHz (reported)\n", prop.gcnArchName, cus, prop.clockRate * 1e-3);
  float* out;
  uint64_t* cyc;
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&cyc, 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  if (run<FMA>(out, cyc, cus, a, b)) return 1;
  if (run<PK_FMA>(out, cyc, cus, a, b)) return 1;
  if (run<PK_MUL>(out, cyc, cus, a, b)) return 1;
  if (run<EXP>(out, cyc, cus, a, b)) return 1;
  if (run<RCP>(out, cyc, cus, a, b)) return 1;
  if (run<CNDMASK>(out, cyc, cus, a, b)) return 1;
  if (run<DPP_ADD>(out, cyc, cus, a, b)) return 1;
  if (run<PERMLANE_SWAP>(out, cyc, cus, a, b)) return 1;
  if (run<MIX_K7>(out, cyc, cus, a, b)) return 1;
  if (run<CMP_VCC>(out, cyc, cus, a, b)) return 1;
  if (run<CMP_SGPR>(out, cyc, cus, a, b)) return 1;
  if (run<CND_SGPR_MIX>(out, cyc, cus, a, b)) return 1;
  if (run<FMA_CND_MIX>(out, cyc, cus, a, b)) return 1;
  if (run<FMA_CMP_MIX>(out, cyc, cus, a, b)) return 1;
  if (run<MAXF>(out, cyc, cus, a, b)) return 1;
  if (run<FMA_CLAMP>(out, cyc, cus, a, b)) return 1;
  if (run<PK_ADD>(out, cyc, cus, a, b)) return 1;
  if (run<MOV>(out, cyc, cus, a, b)) return 1;
  return 0;
}
