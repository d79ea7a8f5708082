// Memory-bound ViT glue ops for the bf16 forward (vit/dinov2.py): LayerNorm fused with the residual
// add that precedes it.  PyTorch-ROCm runs `x = x + y` and `layer_norm(x)` as two kernels that
// together move 6 row-passes at ~1.3 TB/s; here one wave owns one token row (C bf16 = 768 B at
// ViT-S), loads it with 16-byte vectors, keeps it in registers, and writes the new residual stream
// and the normalised row in one pass (4 row-passes at HBM speed).
//
// Numerics follow the unfused PyTorch sequence: the sum is rounded to bf16 first (that is what the
// next op would have read), statistics are float32 two-pass over the rounded values
// (var = E[(x-mean)^2], biased), y = (x-mean) * rsqrt(var + eps) * gamma + beta rounded to bf16.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

typedef uint16_t bf16_t;
constexpr int kMaxChunksPerLane = 4;  // C <= 64 lanes * 4 chunks * 8 = 2048

__device__ inline float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ inline bf16_t f2bf(float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ inline float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct alignas(16) Vec8 { bf16_t e[8]; };

template <bool HAS_RES, bool WRITE_SUM>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ res,
                                                            const bf16_t* __restrict__ gamma,
                                                            const bf16_t* __restrict__ beta, float eps,
                                                            int rows, int C, bf16_t* __restrict__ sum_out,
                                                            bf16_t* __restrict__ y_out, int drop_group) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const size_t base = (size_t)row * C;
  // drop_group > 0: rows come in groups of drop_group (one image's tokens); the first row of every group (the class
  // token) is not needed and y is written densely without it: y row = row - group - 1
  size_t ybase = base;
  if (drop_group > 0) {
    const int grp = row / drop_group;
    if (row - grp * drop_group == 0) return;
    ybase = (size_t)(row - grp - 1) * C;
  }
  float v[kMaxChunksPerLane][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunksPerLane; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nchunk) {
      const Vec8 a = *(const Vec8*)(x + base + ch * 8);
      Vec8 b;
      if (HAS_RES) b = *(const Vec8*)(res + base + ch * 8);
      Vec8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t = bf2f(a.e[i]);
        if (HAS_RES) {
          o.e[i] = f2bf(t + bf2f(b.e[i]));   // the residual stream is stored in bf16
          t = bf2f(o.e[i]);
        }
        v[k][i] = t;
        s += t;
      }
      if (HAS_RES && WRITE_SUM) *(Vec8*)(sum_out + base + ch * 8) = o;
    }
  }
  const float mean = wsum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunksPerLane; ++k) {
    if (lane + 64 * k < nchunk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wsum(q) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < kMaxChunksPerLane; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nchunk) {
      const Vec8 g = *(const Vec8*)(gamma + ch * 8);
      const Vec8 bb = *(const Vec8*)(beta + ch * 8);
      Vec8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.e[i] = f2bf((v[k][i] - mean) * rstd * bf2f(g.e[i]) + bf2f(bb.e[i]));
      *(Vec8*)(y_out + ybase + ch * 8) = o;
    }
  }
}

// out = gelu(x) (exact, erf) elementwise over bf16 — used when the GEMM library's epilogue cannot
// apply the exact GELU; kept for completeness of the fused-op set.
}  // namespace

extern "C" {

static int add_layernorm_launch(const void* x, const void* residual_or_null, const void* gamma, const void* beta, float eps,
                                int rows, int C, void* sum_out_or_null, void* y_out, int drop_group, vc_stream_t stream);

int vc_add_layernorm_bf16(const void* x, const void* residual_or_null, const void* gamma, const void* beta,
                          float eps, int rows, int C, void* sum_out_or_null, void* y_out, vc_stream_t stream) {
  return add_layernorm_launch(x, residual_or_null, gamma, beta, eps, rows, C, sum_out_or_null, y_out, 0, stream);
}

int vc_layernorm_drop_first_bf16(const void* x, const void* gamma, const void* beta, float eps, int n_groups, int group_rows,
                                 int C, void* y_out, vc_stream_t stream) {
  if (n_groups < 0 || group_rows < 2) return VC_ERR_INVALID_ARG;
  return add_layernorm_launch(x, nullptr, gamma, beta, eps, n_groups * group_rows, C, nullptr, y_out, group_rows, stream);
}

static int add_layernorm_launch(const void* x, const void* residual_or_null, const void* gamma, const void* beta, float eps,
                                int rows, int C, void* sum_out_or_null, void* y_out, int drop_group, vc_stream_t stream) {
  if (!x || !gamma || !beta || !y_out || rows < 0 || C <= 0) return VC_ERR_INVALID_ARG;
  if (C % 8 != 0 || C > 64 * kMaxChunksPerLane * 8) return VC_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)y_out) | ((uintptr_t)gamma) | ((uintptr_t)beta) |
       ((uintptr_t)residual_or_null) | ((uintptr_t)sum_out_or_null)) % 16 != 0)
    return VC_ERR_INVALID_ARG;
  if (rows == 0) return VC_OK;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t *px = (const bf16_t*)x, *pr = (const bf16_t*)residual_or_null, *pg = (const bf16_t*)gamma,
               *pb = (const bf16_t*)beta;
  bf16_t *ps = (bf16_t*)sum_out_or_null, *py = (bf16_t*)y_out;
  if (!pr)
    hipLaunchKernelGGL((add_layernorm_kernel<false, false>), grid, block, 0, s, px, pr, pg, pb, eps, rows, C, ps, py, drop_group);
  else if (ps)
    hipLaunchKernelGGL((add_layernorm_kernel<true, true>), grid, block, 0, s, px, pr, pg, pb, eps, rows, C, ps, py, drop_group);
  else
    hipLaunchKernelGGL((add_layernorm_kernel<true, false>), grid, block, 0, s, px, pr, pg, pb, eps, rows, C, ps, py, drop_group);
  return vc::check_launch();
}

}  // extern "C"
