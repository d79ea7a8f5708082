// Image preprocessing for the ViT: uint8 BGR frames -> resized, RGB, ImageNet-normalised model
// input, written either as NCHW or directly in patch order (B, Hp*Wp, 3*14*14) so that the patch
// embedding is a single GEMM.
//
// Replaces reference vit_colmap/features/vit_extractor.py:117-132:
//   cv2.cvtColor(BGR2RGB); floor H, W to multiples of 14; cv2.resize(INTER_LINEAR) if they
//   changed; ToTensor (/255); Normalize(mean, std).
// The resize restates OpenCV's 8-bit bilinear path [recalled: cv2 is not installed here, so this
// step is PARITY UNPINNED against cv2 itself]: half-pixel centres, coefficients rounded to 11-bit
// fixed point, horizontal pass in int32, vertical pass
//   ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// Specification and bit-exact target: oracle/preprocess_oracle.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

constexpr int kPatch = 14;
constexpr int kPatchPad = VC_PATCH_K_PADDED;   // 588 elements padded to a multiple of the GEMM K step

struct Coef {  // one source index and its two 11-bit weights
  int s0, s1;
  int a0, a1;
};

__device__ inline Coef linear_coef(int d, int src, double scale) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src - 1) { f = 0.f; s = src - 1; }
  Coef c;
  c.s0 = s;
  c.s1 = min(s + 1, src - 1);
  // saturate_cast<short>(x * 2048): round half to even
  c.a0 = (int)rintf((1.f - f) * 2048.f);
  c.a1 = (int)rintf(f * 2048.f);
  return c;
}

__device__ inline uint16_t f32_to_bf16(float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <typename OutT>
__device__ inline void store(OutT* p, float v);
template <>
__device__ inline void store<float>(float* p, float v) { *p = v; }
template <>
__device__ inline void store<uint16_t>(uint16_t* p, float v) { *p = f32_to_bf16(v); }

template <typename OutT>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ img, int h, int w, int oh,
                                                         int ow, int layout, OutT* __restrict__ out,
                                                         uint8_t* __restrict__ resized_dbg) {
  const int n = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= oh * ow) return;
  const int y = idx / ow, x = idx - y * ow;
  const uint8_t* src = img + (size_t)n * h * w * 3;
  int px[3];
  if (oh == h && ow == w) {
    for (int c = 0; c < 3; ++c) px[c] = src[((size_t)y * w + x) * 3 + c];
  } else {
    const Coef cx = linear_coef(x, w, (double)w / (double)ow);
    const Coef cy = linear_coef(y, h, (double)h / (double)oh);
    for (int c = 0; c < 3; ++c) {
      const int r0 = src[((size_t)cy.s0 * w + cx.s0) * 3 + c] * cx.a0 + src[((size_t)cy.s0 * w + cx.s1) * 3 + c] * cx.a1;
      const int r1 = src[((size_t)cy.s1 * w + cx.s0) * 3 + c] * cx.a0 + src[((size_t)cy.s1 * w + cx.s1) * 3 + c] * cx.a1;
      px[c] = (((cy.a0 * (r0 >> 4)) >> 16) + ((cy.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
    }
  }
  if (resized_dbg)
    for (int c = 0; c < 3; ++c) resized_dbg[(((size_t)n * oh + y) * ow + x) * 3 + c] = (uint8_t)px[c];
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const int hp = oh / kPatch, wp = ow / kPatch;
  for (int c = 0; c < 3; ++c) {
    const float t = (float)px[2 - c] / 255.0f;  // channel c of RGB is channel 2-c of BGR
    const float v = (t - mean[c]) / stdv[c];
    size_t o;
    if (layout == VC_LAYOUT_NCHW) {
      o = (((size_t)n * 3 + c) * oh + y) * ow + x;
    } else {
      const int py = y / kPatch, dy = y - py * kPatch, pxx = x / kPatch, dx = x - pxx * kPatch;
      const int stride = layout == VC_LAYOUT_PATCHES_PAD ? kPatchPad : 3 * kPatch * kPatch;
      o = ((size_t)n * hp * wp + (size_t)py * wp + pxx) * stride + (c * kPatch + dy) * kPatch + dx;
      if (layout == VC_LAYOUT_PATCHES_PAD && c == 0 && dy == 0 && dx == 0) {
        // the patch's first pixel also clears the K padding (elements 588 .. 639)
        OutT* pad = out + ((size_t)n * hp * wp + (size_t)py * wp + pxx) * stride + 3 * kPatch * kPatch;
        for (int i = 0; i < kPatchPad - 3 * kPatch * kPatch; ++i) store<OutT>(pad + i, 0.0f);
      }
    }
    store<OutT>(out + o, v);
  }
}

// The ViT-S path's layout (bf16 patches padded to 640 elements), one wave per patch: the 14 + 14 interpolation coefficients
// of the patch are computed once (float64 source coordinate as in the generic kernel), every lane produces ten of the
// 588 elements into a 1280-byte LDS image of the patch row, and the row leaves as 80 coalesced 16-byte stores (the
// generic kernel's thread-per-pixel mapping writes 2-byte elements in 28-byte runs).  Same arithmetic, same bits.
// The patch's source window (about 16 rows x 16 pixels x 3 bytes) is staged in LDS with dword loads and the taps read
// from there; s0 / s1 of the coefficient tables then hold byte offsets into the window.
constexpr int kWinRowBytes = 96, kWinRows = 24;
__global__ __launch_bounds__(256) void preprocess_patches_kernel(const uint8_t* __restrict__ img, int n_images, int h, int w, int oh,
                                                                 int ow, double scale_x, double scale_y,
                                                                 uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t row_l[4][kPatchPad];
  __shared__ __attribute__((aligned(16))) uint8_t win_l[4][kWinRows * kWinRowBytes];
  __shared__ Coef coef_l[4][2][16];
  // normalised value of every (channel, 8-bit pixel) pair, by the generic kernel's expressions (768 entries per workgroup
  // instead of two IEEE float divisions per element)
  __shared__ uint16_t norm_l[3][256];
  {
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int i = threadIdx.x; i < 768; i += 256) {
      const int c = i >> 8;
      const float t = (float)(i & 255) / 255.0f;
      norm_l[c][i & 255] = f32_to_bf16((t - mean[c]) / stdv[c]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y, hp = oh / kPatch, wp = ow / kPatch;
  const int patch = blockIdx.x * 4 + wave;
  const bool live = patch < hp * wp;
  const int pc = live ? patch : hp * wp - 1;
  const int py = pc / wp, pxx = pc - py * wp;
  // source window of the patch: the coefficients are monotonic in the output coordinate, so the first / last of the 14
  // give its bounds (every lane evaluates these four: wave-uniform values without a round trip through LDS)
  const int xs0 = linear_coef(pxx * kPatch, w, scale_x).s0, xs1 = linear_coef(pxx * kPatch + 13, w, scale_x).s1;
  const int ys0 = linear_coef(py * kPatch, h, scale_y).s0, ys1 = linear_coef(py * kPatch + 13, h, scale_y).s1;
  const int nxb = (xs1 - xs0 + 1) * 3, ny = ys1 - ys0 + 1;
  const bool use_win = nxb + 6 <= kWinRowBytes && ny <= kWinRows;       // wave-uniform
  const size_t img_bytes = (size_t)h * w * 3, all_bytes = img_bytes * n_images;
  const size_t img_off = (size_t)n * img_bytes;
  const uint8_t* src = img + img_off;
  const uint32_t off3 = (uint32_t)(img_off & 3);
  // coefficient tables; with a window, s0 / s1 become byte offsets into it (x: 3 (s - xs0); y: row start + misalignment)
  if (lane < 14) {
    Coef c = linear_coef(pxx * kPatch + lane, w, scale_x);
    if (use_win) { c.s0 = (c.s0 - xs0) * 3; c.s1 = (c.s1 - xs0) * 3; }
    coef_l[wave][0][lane] = c;
  } else if (lane >= 16 && lane < 30) {
    Coef c = linear_coef(py * kPatch + lane - 16, h, scale_y);
    if (use_win) {
      const uint32_t a0 = (off3 + ((uint32_t)c.s0 * (uint32_t)w + (uint32_t)xs0) * 3u) & 3u;   // mod 4: 32-bit wrap is harmless
      const uint32_t a1 = (off3 + ((uint32_t)c.s1 * (uint32_t)w + (uint32_t)xs0) * 3u) & 3u;
      c.s0 = (c.s0 - ys0) * kWinRowBytes + (int)a0;
      c.s1 = (c.s1 - ys0) * kWinRowBytes + (int)a1;
    }
    coef_l[wave][1][lane - 16] = c;
  }
  uint8_t* const win = win_l[wave];
  if (use_win) {
    // the window, ~4 dword loads per lane (the generic kernel issues 12 byte loads per output pixel)
    const int dw_row = (nxb + 6) >> 2;                                  // dwords per row incl. worst-case misalignment
    for (int idx = lane; idx < ny * dw_row; idx += 64) {
      const int row = idx / dw_row, j = idx - row * dw_row;
      const size_t rb = img_off + ((size_t)(ys0 + row) * w + xs0) * 3;  // byte offset of the row's first tap in the batch
      const size_t a = (rb & ~(size_t)3) + (size_t)j * 4;               // img is 4-byte aligned (host check)
      uint32_t v;
      if (a + 4 <= all_bytes) {
        v = *(const uint32_t*)(img + a);
      } else {                                                          // last bytes of the batch
        v = 0;
        for (int k = 0; k < 4; ++k)
          if (a + k < all_bytes) v |= (uint32_t)img[a + k] << (8 * k);
      }
      *(uint32_t*)(win + row * kWinRowBytes + j * 4) = v;
    }
  }
  __syncthreads();
  for (int e = lane; e < kPatchPad; e += 64) {
    uint16_t v16 = 0;                                   // elements 588 .. 639: the K padding
    if (e < 3 * kPatch * kPatch) {
      const int c = e / (kPatch * kPatch), rem = e - c * (kPatch * kPatch), dy = rem / kPatch, dx = rem - dy * kPatch;
      const Coef cx = coef_l[wave][0][dx], cy = coef_l[wave][1][dy];
      const int cc = 2 - c;                             // channel c of RGB is channel 2-c of BGR
      int p00, p01, p10, p11;
      if (use_win) {
        p00 = win[cy.s0 + cx.s0 + cc]; p01 = win[cy.s0 + cx.s1 + cc];
        p10 = win[cy.s1 + cx.s0 + cc]; p11 = win[cy.s1 + cx.s1 + cc];
      } else {                                          // strong down-scaling: taps from global memory
        p00 = src[((size_t)cy.s0 * w + cx.s0) * 3 + cc]; p01 = src[((size_t)cy.s0 * w + cx.s1) * 3 + cc];
        p10 = src[((size_t)cy.s1 * w + cx.s0) * 3 + cc]; p11 = src[((size_t)cy.s1 * w + cx.s1) * 3 + cc];
      }
      const int r0 = p00 * cx.a0 + p01 * cx.a1;
      const int r1 = p10 * cx.a0 + p11 * cx.a1;
      const int px = (((cy.a0 * (r0 >> 4)) >> 16) + ((cy.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
      v16 = norm_l[c][px];
    }
    row_l[wave][e] = v16;
  }
  __syncthreads();
  if (!live) return;
  uint8_t* dst = (uint8_t*)(out + ((size_t)n * hp * wp + patch) * kPatchPad);
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  for (int q = lane; q < kPatchPad * 2 / 16; q += 64) *(v4u*)(dst + q * 16) = *(const v4u*)((const uint8_t*)row_l[wave] + q * 16);
}

}  // namespace

extern "C" {

int vc_preprocess_u8(const uint8_t* images_bgr, int n_images, int h, int w, int out_h, int out_w, int out_dtype,
                     int layout, void* out, uint8_t* resized_bgr_or_null, vc_stream_t stream) {
  if (!images_bgr || !out || n_images < 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0) return VC_ERR_INVALID_ARG;
  if (out_dtype != VC_DTYPE_F32 && out_dtype != VC_DTYPE_BF16) return VC_ERR_INVALID_ARG;
  if (layout != VC_LAYOUT_NCHW && layout != VC_LAYOUT_PATCHES && layout != VC_LAYOUT_PATCHES_PAD) return VC_ERR_INVALID_ARG;
  if (layout != VC_LAYOUT_NCHW && (out_h % kPatch != 0 || out_w % kPatch != 0)) return VC_ERR_INVALID_ARG;
  if (n_images == 0) return VC_OK;
  if (layout == VC_LAYOUT_PATCHES_PAD && out_dtype == VC_DTYPE_BF16 && !resized_bgr_or_null && ((uintptr_t)out) % 16 == 0 &&
      ((uintptr_t)images_bgr) % 4 == 0) {
    const int n_patches = (out_h / kPatch) * (out_w / kPatch);
    hipLaunchKernelGGL(preprocess_patches_kernel, dim3((n_patches + 3) / 4, n_images), dim3(256), 0, (hipStream_t)stream, images_bgr,
                       n_images, h, w, out_h, out_w, (double)w / (double)out_w, (double)h / (double)out_h, (uint16_t*)out);
    return vc::check_launch();
  }
  const dim3 grid((out_h * out_w + 255) / 256, n_images);
  if (out_dtype == VC_DTYPE_F32)
    hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, images_bgr, h, w, out_h,
                       out_w, layout, (float*)out, resized_bgr_or_null);
  else
    hipLaunchKernelGGL(preprocess_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, images_bgr, h, w,
                       out_h, out_w, layout, (uint16_t*)out, resized_bgr_or_null);
  return vc::check_launch();
}

}  // extern "C"
