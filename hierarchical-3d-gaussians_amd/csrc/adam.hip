// K13: fused row-sparse Adam.  Replaces the ~12 torch kernels per parameter tensor of the
// reference's optimiser step (scene/OurAdam.py:249-337: gather rows -> update -> scatter rows) with ONE launch
// over all parameter tensors of the model.  Semantics of _single_tensor_adam (non-capturable, no amsgrad):
//     g' = g + weight_decay * p
//     m  = beta1 * m + (1 - beta1) * g'
//     v  = beta2 * v + (1 - beta2) * g' * g'
//     p  = p - (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps)
// applied to the rows listed in `rows` (train_single.py:171-174: rows whose opacity gradient is non-zero), or to the
// rows whose `row_mask_grad` entry is non-zero (the same selection evaluated in-kernel: no nonzero(), no host
// sync), or to every row (relevant.size(0) == 0 -> _single_tensor_adam2, scene/OurAdam.py:207-222).
//
// HBM-bound streaming: 16 B read + 12 B written per updated element; one thread per element, consecutive
// threads walk consecutive floats of a row so every access is a contiguous run of row_len floats.
#include "common.h"

namespace hgs {
namespace {

constexpr int kMaxTensors = HGS_ADAM_MAX_TENSORS;

struct AdamLaunch {
  hgs_adam_tensor t[kMaxTensors];
  uint32_t first_block[kMaxTensors + 1];   // block range of every tensor
};

template <int MODE, typename IDX>   // MODE: 0 dense, 1 row list, 2 row mask; IDX: element index type
__global__ __launch_bounds__(256) void adam_kernel(AdamLaunch L, int n_tensors, const int64_t* __restrict__ rows,
                                                   int64_t n_rows, const float* __restrict__ row_mask_grad) {
  int ti = 0;
#pragma unroll
  for (int k = 1; k < kMaxTensors; ++k)
    if (k < n_tensors && blockIdx.x >= L.first_block[k]) ti = k;
  const hgs_adam_tensor& T = L.t[ti];
  const IDX e = (IDX)(blockIdx.x - L.first_block[ti]) * 256 + threadIdx.x;   // element among selected rows
  const IDX total = (IDX)n_rows * (IDX)T.row_len;
  if (e >= total) return;
  const IDX r = e / (IDX)T.row_len;
  const int c = (int)(e - r * (IDX)T.row_len);
  int64_t row = (int64_t)r;
  if (MODE == 1) row = rows[r];
  if (MODE == 2 && row_mask_grad[r] == 0.0f) return;
  const int64_t idx = row * T.row_len + c;
  float g = T.grad[idx];
  const float p = T.param[idx];
  if (T.weight_decay != 0.0f) g = fmaf(T.weight_decay, p, g);
  const float m = fmaf(T.one_minus_beta1, g, T.exp_avg[idx] * T.beta1);
  const float v = fmaf(T.one_minus_beta2 * g, g, T.exp_avg_sq[idx] * T.beta2);
  const float denom = sqrtf(v) / T.bias_correction2_sqrt + T.eps;
  T.exp_avg[idx] = m;
  T.exp_avg_sq[idx] = v;
  T.param[idx] = fmaf(-T.step_size, m / denom, p);
}

}  // namespace
}  // namespace hgs

using namespace hgs;

extern "C" int hgs_adam_step(const hgs_adam_tensor* tensors, int32_t n_tensors, int64_t P, const int64_t* rows,
                             int64_t n_rows, const float* row_mask_grad, hgs_stream_t stream, int device) {
  if (n_tensors <= 0) return HGS_OK;
  if (!tensors || n_tensors > kMaxTensors) { set_error("adam: 1..%d tensors per call", kMaxTensors); return HGS_ERR_INVALID; }
  if (rows && row_mask_grad) { set_error("adam: pass a row list or a row mask, not both"); return HGS_ERR_INVALID; }
  const int64_t nsel = rows ? n_rows : P;
  if (nsel <= 0) return HGS_OK;
  AdamLaunch L;
  uint64_t nb = 0;
  bool wide = false;
  for (int k = 0; k < n_tensors; ++k) {
    const hgs_adam_tensor& t = tensors[k];
    wide = wide || (nsel * (int64_t)t.row_len >= (int64_t)0x7fffff00);
    if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || t.row_len <= 0) {
      set_error("adam: tensor %d has a null pointer or row_len <= 0", k);
      return HGS_ERR_INVALID;
    }
    L.t[k] = t;
    L.first_block[k] = (uint32_t)nb;
    nb += (uint64_t)((nsel * t.row_len + 255) / 256);
    if (nb > 0x7fffffffull) { set_error("adam: too many elements for one launch"); return HGS_ERR_INVALID; }
  }
  for (int k = n_tensors; k <= kMaxTensors; ++k) L.first_block[k] = (uint32_t)nb;
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int mode = rows ? 1 : (row_mask_grad ? 2 : 0);
  void (*kern)(AdamLaunch, int, const int64_t*, int64_t, const float*) =
      wide ? (mode == 1 ? adam_kernel<1, int64_t> : mode == 2 ? adam_kernel<2, int64_t> : adam_kernel<0, int64_t>)
           : (mode == 1 ? adam_kernel<1, uint32_t> : mode == 2 ? adam_kernel<2, uint32_t> : adam_kernel<0, uint32_t>);
  hipLaunchKernelGGL(kern, dim3((uint32_t)nb), dim3(256), 0, s, L, n_tensors, rows, nsel, row_mask_grad);
  HGS_LAUNCH_CHECK("adam", s, false);
  return HGS_OK;
}
