// Two-view geometric verification, the scoring half (gfx950): inlier counts of many fundamental-matrix or
// homography hypotheses against the putative matches of many image pairs, and the inlier mask of one model
// per pair.  This is the step that follows descriptor matching inside pycolmap.match_exhaustive and fills the
// `two_view_geometries` table the reference's metrics read (vit_colmap/utils/metrics.py:207-243).
// Hypotheses come from 8x8 linear solves on the host side of the ABI (vit_colmap_amd/matching/two_view.py);
// what is O(pairs x hypotheses x matches) runs here.  Specification: oracle/two_view_oracle.py (float32
// arithmetic in exactly this operation order; the library is built with -ffp-contract=off).
//
// Residuals (no division, so host oracle and device agree bit for bit):
//   F: Sampson error  (x2' F x1)^2 <= t^2 * (|F x1|_xy^2 + |F' x2|_xy^2)
//   H: forward transfer error  |p_xy - x2 p_w|^2 <= t^2 * p_w^2  with p = H x1, p_w != 0
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

__device__ __forceinline__ bool inlier_f(const float (&m)[9], float x1, float y1, float x2, float y2, float t2) {
  const float fx0 = m[0] * x1 + m[1] * y1 + m[2];
  const float fx1 = m[3] * x1 + m[4] * y1 + m[5];
  const float fx2 = m[6] * x1 + m[7] * y1 + m[8];
  const float ft0 = m[0] * x2 + m[3] * y2 + m[6];
  const float ft1 = m[1] * x2 + m[4] * y2 + m[7];
  const float c = x2 * fx0 + y2 * fx1 + fx2;
  const float den = fx0 * fx0 + fx1 * fx1 + ft0 * ft0 + ft1 * ft1;
  return c * c <= t2 * den;          // NaN hypotheses compare false
}

__device__ __forceinline__ bool inlier_h(const float (&m)[9], float x1, float y1, float x2, float y2, float t2) {
  const float p0 = m[0] * x1 + m[1] * y1 + m[2];
  const float p1 = m[3] * x1 + m[4] * y1 + m[5];
  const float pw = m[6] * x1 + m[7] * y1 + m[8];
  const float dx = p0 - x2 * pw;
  const float dy = p1 - y2 * pw;
  return pw != 0.f && dx * dx + dy * dy <= t2 * (pw * pw);
}

// grid (n_pairs, hypothesis groups); 4 waves per workgroup, one hypothesis per wave and round, lanes over matches
__global__ __launch_bounds__(256) void two_view_score_kernel(const float4* __restrict__ pts, const int32_t* __restrict__ offsets,
                                                             const float* __restrict__ hyp, int K, int model, float t2,
                                                             int32_t* __restrict__ counts) {
  const int p = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = offsets[p], hi = offsets[p + 1];
  for (int k = blockIdx.y * 4 + wave; k < K; k += gridDim.y * 4) {
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = hyp[((size_t)p * K + k) * 9 + i];
    int n = 0;
    for (int base = lo; base < hi; base += 64) {   // whole waves: the ballot needs every lane (base is wave-uniform)
      const int i = base + lane;
      bool in = false;
      if (i < hi) {
        const float4 q = pts[i];
        in = model == 0 ? inlier_f(m, q.x, q.y, q.z, q.w, t2) : inlier_h(m, q.x, q.y, q.z, q.w, t2);
      }
      n += __popcll(__ballot(in));
    }
    if (lane == 0) counts[(size_t)p * K + k] = n;
  }
}

__global__ __launch_bounds__(256) void two_view_mask_kernel(const float4* __restrict__ pts, const int32_t* __restrict__ offsets,
                                                            const float* __restrict__ model9, int model, float t2,
                                                            uint8_t* __restrict__ mask) {
  const int p = blockIdx.x;
  const int lo = offsets[p], hi = offsets[p + 1];
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = model9[(size_t)p * 9 + i];
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float4 q = pts[i];
    mask[i] = (model == 0 ? inlier_f(m, q.x, q.y, q.z, q.w, t2) : inlier_h(m, q.x, q.y, q.z, q.w, t2)) ? 1 : 0;
  }
}

}  // namespace

extern "C" {

int vc_two_view_score(const float* pts, const int32_t* offsets, int n_pairs, const float* hypotheses, int n_hyp,
                      int model, float max_error, int32_t* out_counts, vc_stream_t stream) {
  if (n_pairs < 0 || n_hyp < 0 || (model != VC_MODEL_FUNDAMENTAL && model != VC_MODEL_HOMOGRAPHY)) return VC_ERR_INVALID_ARG;
  if (n_pairs == 0 || n_hyp == 0) return VC_OK;
  if (!pts || !offsets || !hypotheses || !out_counts || !(max_error >= 0.f)) return VC_ERR_INVALID_ARG;
  if (((uintptr_t)pts) % 16 != 0) return VC_ERR_INVALID_ARG;
  if (n_pairs > 65535 * 32) return VC_ERR_UNSUPPORTED;
  const int groups = n_hyp >= 64 ? 16 : (n_hyp + 3) / 4;
  hipLaunchKernelGGL(two_view_score_kernel, dim3(n_pairs, groups), dim3(256), 0, (hipStream_t)stream, (const float4*)pts,
                     offsets, hypotheses, n_hyp, model, max_error * max_error, out_counts);
  return vc::check_launch();
}

int vc_two_view_inliers(const float* pts, const int32_t* offsets, int n_pairs, const float* models, int model,
                        float max_error, uint8_t* out_mask, vc_stream_t stream) {
  if (n_pairs < 0 || (model != VC_MODEL_FUNDAMENTAL && model != VC_MODEL_HOMOGRAPHY)) return VC_ERR_INVALID_ARG;
  if (n_pairs == 0) return VC_OK;
  if (!pts || !offsets || !models || !out_mask || !(max_error >= 0.f)) return VC_ERR_INVALID_ARG;
  if (((uintptr_t)pts) % 16 != 0) return VC_ERR_INVALID_ARG;
  hipLaunchKernelGGL(two_view_mask_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, (const float4*)pts, offsets,
                     models, model, max_error * max_error, out_mask);
  return vc::check_launch();
}

}  // extern "C"
