// Keypoints and descriptors from the dense head outputs of the trainable model, batched over images.
//
// Replaces the post-model part of TrainableViTExtractor._run_inference
// (reference vit_colmap/features/trainable_vit_extractor.py:170-267, _simple_nms :114-138);
// specification: oracle/trainable_oracle.py.  Inputs are what the model emits: a 4-channel map
// (score logit, dx, dy, orientation) and a unit-norm descriptor map, both at 1/4 resolution.
//
//   heat_sigmoid_kernel   score = float32(1 / (1 + exp(-float64(logit)))) for every cell (the
//                         correctly rounded sigmoid: device and oracle agree on every input)
//   heat_select_kernel    one 1024-thread workgroup per image:
//                           1. (2r+1)^2 window maximum (out-of-map cells do not take part, i.e.
//                              -inf padding) -> cell is a candidate iff score == max and score > threshold;
//                              key = score bits + 1 (scores are >= 0, so the bits order like the
//                              values), 0 for non-candidates
//                           2. radix select (4 passes of 8 bits, LDS histogram) of the k-th largest key
//                           3. keys above the threshold key are collected unordered, keys equal to it
//                              by an ordered compaction in position order (ties go to the lower position)
//                           4. bitonic sort of the (key, ~position) pairs — in LDS up to 4096 keypoints, in the
//                              workspace (L2 resident) above that (the reference's pipeline asks for 20 480)
//                           5. keypoint rows (x, y, 1, orientation, score, 0), descriptor gather +
//                              (d + 1) * 127.5 truncating quantiser, one wave per keypoint
//
// Integer stages are exact; the float arithmetic of the coordinates and of the quantiser is
// written in the reference's operation order and built with -ffp-contract=off, so every output
// is bit-identical to the oracle's.  A map is never resident in LDS (a 1600x1200 image has
// 120 000 cells), it is re-read from L2 in each pass: 7 passes x 4 bytes per cell.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

constexpr int kHeatThreads = 1024;
constexpr int kHeatLdsK = 4096;      // selection list held in LDS up to this many keypoints
constexpr int kHeatMaxK = 1 << 16;

inline int pow2_at_least(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__global__ __launch_bounds__(256) void heat_sigmoid_kernel(const float* __restrict__ kp_map, int cells,
                                                           float* __restrict__ scores) {
  const int i = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
  if (i >= cells) return;
  const double x = (double)kp_map[(size_t)img * 4 * cells + i];
  scores[(size_t)img * cells + i] = (float)(1.0 / (1.0 + exp(-x)));
}

__global__ __launch_bounds__(kHeatThreads) void heat_select_kernel(
    const float* __restrict__ kp_map, const float* __restrict__ scores_g, uint32_t* __restrict__ keys_g,
    const float* __restrict__ desc, long long desc_is, long long desc_cs, long long desc_ps, int D, int H, int W,
    int radius, float thr, int kmax, float sx, float sy, float x_max, float y_max, float* __restrict__ out_kp,
    uint8_t* __restrict__ out_desc, int32_t* __restrict__ out_count, unsigned long long* __restrict__ sel_g, int sel_cap) {
  __shared__ uint32_t hist[256];
  __shared__ unsigned long long sel_l[kHeatLdsK];
  __shared__ int s_wave_tot[kHeatThreads / 64];
  __shared__ int s_cnt, s_valid;
  __shared__ uint32_t s_prefix;
  __shared__ int s_rem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.x, cells = H * W;
  const float* sc = scores_g + (size_t)img * cells;
  uint32_t* keys = keys_g + (size_t)img * cells;
  const float* maps = kp_map + (size_t)img * 4 * cells;
  unsigned long long* sel = sel_g ? sel_g + (size_t)blockIdx.x * sel_cap : sel_l;

  // ---- 1. candidates -----------------------------------------------------------------------
  if (tid == 0) { s_valid = 0; s_cnt = 0; }
  __syncthreads();
  int my_valid = 0;
  for (int i = tid; i < cells; i += kHeatThreads) {
    const int y = i / W, x = i - y * W;
    const float s = sc[i];
    float m = s;
    const int y0 = max(0, y - radius), y1 = min(H - 1, y + radius), x0 = max(0, x - radius), x1 = min(W - 1, x + radius);
    for (int yy = y0; yy <= y1; ++yy)
      for (int xx = x0; xx <= x1; ++xx) m = fmaxf(m, sc[yy * W + xx]);
    const bool valid = (s == m) && (s > thr);
    keys[i] = valid ? __float_as_uint(s) + 1u : 0u;
    my_valid += valid ? 1 : 0;
  }
  if (my_valid) atomicAdd(&s_valid, my_valid);
  __syncthreads();
  const int n_valid = s_valid;
  const int take = min(n_valid, kmax);
  float* kp_rows = out_kp + (size_t)img * kmax * 6;
  uint8_t* d_rows = out_desc + (size_t)img * kmax * D;
  if (tid == 0) out_count[img] = take;

  if (take > 0) {
    // ---- 2. radix select: the take-th largest key -------------------------------------------
    if (tid == 0) { s_prefix = 0; s_rem = take; }
    uint32_t mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int b = tid; b < 256; b += kHeatThreads) hist[b] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      for (int i = tid; i < cells; i += kHeatThreads) {
        const uint32_t k = keys[i];
        if (k != 0u && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int rem = s_rem, b = 255;
        for (; b > 0; --b) {
          const int hcount = (int)hist[b];
          if (hcount >= rem) break;
          rem -= hcount;
        }
        s_rem = rem;                                   // how many keys with this digit (and prefix) are still wanted
        s_prefix = prefix | ((uint32_t)b << shift);
      }
      mask |= 255u << shift;
      __syncthreads();
    }
    const uint32_t T = s_prefix;
    const int want_eq = s_rem;                         // >= 1 keys equal to T, lowest positions first
    // ---- 3. collect ---------------------------------------------------------------------------
    for (int i = tid; i < cells; i += kHeatThreads) {
      const uint32_t k = keys[i];
      if (k > T) {
        const int j = atomicAdd(&s_cnt, 1);
        sel[j] = ((unsigned long long)k << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
      }
    }
    __syncthreads();
    const int n_gt = s_cnt;                            // == take - want_eq
    int taken = 0;
    for (int i0 = 0; i0 < cells && taken < want_eq; i0 += kHeatThreads) {   // uniform loop
      const int i = i0 + tid;
      const bool eq = i < cells && keys[i] == T;
      const unsigned long long bal = __ballot(eq);
      if (lane == 0) s_wave_tot[wave] = __popcll(bal);
      __syncthreads();
      int before = taken, tot = 0;
      for (int w = 0; w < kHeatThreads / 64; ++w) { if (w < wave) before += s_wave_tot[w]; tot += s_wave_tot[w]; }
      if (eq) {
        const int r = before + __popcll(bal & ((1ull << lane) - 1ull));
        if (r < want_eq) sel[n_gt + r] = ((unsigned long long)T << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
      }
      taken += tot;
      __syncthreads();
    }
    // ---- 4. bitonic sort, descending (score desc, position asc) -------------------------------
    int P = 1;
    while (P < take) P <<= 1;
    for (int i = take + tid; i < P; i += kHeatThreads) sel[i] = 0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < P; t += kHeatThreads) {
          const int u = t ^ j;
          if (u > t) {
            const unsigned long long a = sel[t], b = sel[u];
            const bool desc_block = (t & k) == 0;
            if (desc_block ? a < b : a > b) { sel[t] = b; sel[u] = a; }
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  // ---- 5. outputs ------------------------------------------------------------------------------
  for (int r = tid; r < kmax; r += kHeatThreads) {
    float row[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < take) {
      const unsigned long long e = sel[r];
      const int pos = (int)(0xffffffffu - (uint32_t)(e & 0xffffffffull));
      const int y = pos / W, x = pos - y * W;
      const float fx = ((((float)x + maps[cells + pos]) + 0.5f) * 4.0f) * sx;          // trainable_vit_extractor.py:226-231
      const float fy = ((((float)y + maps[2 * cells + pos]) + 0.5f) * 4.0f) * sy;
      row[0] = fminf(fmaxf(fx, 0.f), x_max);                                            // :234-235
      row[1] = fminf(fmaxf(fy, 0.f), y_max);
      row[2] = 1.0f;
      row[3] = maps[3 * cells + pos];
      row[4] = __uint_as_float((uint32_t)(e >> 32) - 1u);
      row[5] = 0.0f;
    }
    for (int c = 0; c < 6; ++c) kp_rows[(size_t)r * 6 + c] = row[c];
  }
  const float* dimg = desc + (size_t)img * desc_is;
  for (int r = wave; r < kmax; r += kHeatThreads / 64) {
    if (r < take) {
      const int pos = (int)(0xffffffffu - (uint32_t)(sel[r] & 0xffffffffull));
      const float* dp = dimg + (size_t)pos * desc_ps;
      for (int c = lane; c < D; c += 64) {
        const float q = (dp[(size_t)c * desc_cs] + 1.0f) * 127.5f;                        // :265-267
        d_rows[(size_t)r * D + c] = (uint8_t)fminf(fmaxf(q, 0.f), 255.f);
      }
    } else {
      for (int c = lane; c < D; c += 64) d_rows[(size_t)r * D + c] = 0;                   // whole blocks for the matcher
    }
  }
}

}  // namespace

extern "C" {

size_t vc_heatmap_workspace_bytes(int n_images, int H, int W, int kmax) {
  if (n_images <= 0 || H <= 0 || W <= 0 || kmax <= 0 || kmax > kHeatMaxK) return 0;
  size_t b = (size_t)n_images * H * W * 8;   // float32 scores + uint32 keys
  if (kmax > kHeatLdsK) b += (size_t)n_images * pow2_at_least(kmax) * 8;   // selection list
  return b;
}

int vc_heatmap_keypoints(const float* kp_map, const float* desc_map, long long desc_image_stride,
                         long long desc_channel_stride, long long desc_pixel_stride, int n_images, int H, int W, int D,
                         int nms_radius, float score_threshold, int kmax, float scale_x, float scale_y, float x_max,
                         float y_max, void* workspace, float* out_keypoints, uint8_t* out_desc, int32_t* out_count,
                         vc_stream_t stream) {
  if (!kp_map || !desc_map || !workspace || !out_keypoints || !out_desc || !out_count) return VC_ERR_INVALID_ARG;
  if (((uintptr_t)workspace) % 16 != 0) return VC_ERR_INVALID_ARG;
  if (n_images < 0 || H <= 0 || W <= 0 || D <= 0 || nms_radius < 0 || kmax <= 0) return VC_ERR_INVALID_ARG;
  if (kmax > kHeatMaxK || (long long)H * W > (1ll << 30)) return VC_ERR_UNSUPPORTED;
  if (n_images == 0) return VC_OK;
  const int cells = H * W;
  float* scores = (float*)workspace;
  uint32_t* keys = (uint32_t*)(scores + (size_t)n_images * cells);
  unsigned long long* sel_g = kmax > kHeatLdsK ? (unsigned long long*)(keys + (size_t)n_images * cells) : nullptr;
  hipLaunchKernelGGL(heat_sigmoid_kernel, dim3((cells + 255) / 256, n_images), dim3(256), 0, (hipStream_t)stream, kp_map,
                     cells, scores);
  hipLaunchKernelGGL(heat_select_kernel, dim3(n_images), dim3(kHeatThreads), 0, (hipStream_t)stream, kp_map, scores, keys,
                     desc_map, desc_image_stride, desc_channel_stride, desc_pixel_stride, D, H, W, nms_radius,
                     score_threshold, kmax, scale_x, scale_y, x_max, y_max, out_keypoints, out_desc, out_count, sel_g,
                     pow2_at_least(kmax));
  return vc::check_launch();
}

}  // extern "C"
