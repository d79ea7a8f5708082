// bf16 linear layer with fused epilogue for the DINOv2 blocks (the model behind reference
// vit_colmap/features/vit_extractor.py:135-146):  out = epi(x W^T + b)  with
//   epi = identity (qkv), exact-erf GELU (fc1), or "+ residual" (attn.proj / fc2 — the residual
//   stream update), so the activation and the residual add never make their own pass over HBM.
//
// x [M][K] bf16 (token rows), W [N][K] bf16 (torch Linear layout: one output feature per row),
// out [M][N] bf16.  M is arbitrary (76 550 at 50 images x 1531 tokens), N % 128 == 0, K % 64 == 0.
//
// One workgroup = 4 waves = one 128 (tokens) x 128 (features) output tile; two workgroups per CU
// (64 KiB of LDS, <= 128 VGPRs each) so that one tile's epilogue VALU work runs under the other's
// MFMAs.  Per 64-deep K step both operand tiles (128 rows x 128 B) go global -> LDS by LDS-DMA
// (1 KiB per wave instruction, destination lane-linear), double buffered one step ahead; the bank
// swizzle (16-byte chunk ^ (row & 7), conflict-free for the ds_read_b128 column slices) is applied
// to the per-lane SOURCE address.  The product is computed transposed,
//   D^T[feature][token] = W_tile (A operand) x x_tile^T (B operand),  v_mfma_f32_16x16x32_bf16,
// so a lane ends up with 4 CONSECUTIVE FEATURES of one token: bias, GELU and residual are applied
// to 4-vectors and the bf16 result leaves as 8-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BM = 128;              // token rows per tile
constexpr int BN = 128;              // output features per tile
constexpr int BK = 64;               // K step: 64 bf16 = one 128-byte LDS row
constexpr int kImg = 128 * 128;      // bytes of one operand image (128 rows x 128 B)
constexpr int kStage = 2 * kImg;     // [x image | W image]

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESIDUAL = 2, EPI_PATCH = 3 };

// exact GELU, 0.5 x (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7):
// with q = (1 - erf|z|) = poly(t) t exp(-z^2), t = 1 / (1 + p|z|):  gelu = max(x, 0) - 0.5 |x| q.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
  float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float zl = x * 0.84932180028801907f;   // x sqrt(log2(e) / 2):  exp(-x^2/2) = exp2(-zl^2)
  const float e = __builtin_amdgcn_exp2f(-(zl * zl));
  const float q = p * t * e;
  return __builtin_fmaf(-0.5f * ax, q, __builtin_fmaxf(x, 0.0f));
}

// The same on a pair of values with 2-wide packed float32 operations (v_pk_fma_f32 / v_pk_mul_f32):
// beside a partner wave that owns the matrix pipe, a lone wave's VALU stream is issue-bound, and a packed
// instruction does twice the work per issue slot.  Only rcp / exp2 / abs / max stay one value at a time.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t gelu_erf2(v2f_t x) {
  const v2f_t ax = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  const v2f_t one = {1.0f, 1.0f};
  const v2f_t den = __builtin_elementwise_fma(ax, (v2f_t){0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f}, one);
  const v2f_t t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  v2f_t p = __builtin_elementwise_fma(t, (v2f_t){1.061405429f, 1.061405429f}, (v2f_t){-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, (v2f_t){1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, (v2f_t){-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, (v2f_t){0.254829592f, 0.254829592f});
  const v2f_t zl = x * (v2f_t){0.84932180028801907f, 0.84932180028801907f};
  const v2f_t z2 = zl * zl;
  const v2f_t e = {__builtin_amdgcn_exp2f(-z2[0]), __builtin_amdgcn_exp2f(-z2[1])};
  const v2f_t q = p * t * e;
  const v2f_t pos = {__builtin_fmaxf(x[0], 0.0f), __builtin_fmaxf(x[1], 0.0f)};
  return __builtin_elementwise_fma(ax * (v2f_t){-0.5f, -0.5f}, q, pos);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const __bf16* __restrict__ X, const __bf16* __restrict__ W,
                                                      const __bf16* __restrict__ bias,
                                                      const __bf16* __restrict__ res, __bf16* __restrict__ out,
                                                      int M, int N, int K, int n_tiles_n, int n_tiles,
                                                      int group) {
  // EPI_PATCH: row m = (image b, token t) with b = m / group; the result goes to row m + b + 1 of `out`
  // (one class-token row per image is skipped) and `res` is the position embedding, indexed by 1 + t.
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2][kStage];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware tile order: the 8 XCDs take workgroups round-robin, so give each XCD a contiguous
  // run of tiles (feature tiles fastest: the tiles an XCD works on at one time share x rows in its L2)
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int q = n_tiles >> 3, r = n_tiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / n_tiles_n, tn = tile - tm * n_tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDS-DMA staging: a stage is 32 pieces of 1 KiB (8 rows x 128 B); wave w issues x pieces
  // 4w..4w+3 and W pieces 4w..4w+3.  Lane l fills LDS row (l >> 3), chunk (l & 7) of its piece with
  // source chunk (l & 7) ^ (l >> 3)  [row & 7 == l >> 3 because pieces start at multiples of 8 rows].
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0][0]);
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  const __bf16* xsrc[4];
  const __bf16* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + prow;
    xsrc[i] = X + (size_t)min(m0 + row, M - 1) * K + pchunk * 8;
    wsrc[i] = W + (size_t)(n0 + row) * K + pchunk * 8;
  }
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const __bf16* src = (i < 4 ? xsrc[i] : wsrc[i - 4]) + kt * BK;
      const uint32_t dst = lds0 + (uint32_t)buf * kStage + (uint32_t)(i >> 2) * kImg + (uint32_t)(wave * 4 + (i & 3)) * 1024u;
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src), "s"(dst)
          : "memory");
    }
  };

  v4f acc[4][4];   // [feature block][token block]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane reads row (l & 15) of a 16-row block, 16-byte chunk 4 ks + (l >> 4)
  const int fr = lane & 15, fq = lane >> 4;
  uint32_t xoff[2], woff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ks * 4 + fq;
    // rows of all 16-row blocks share (row & 7) == (fr & 7): one swizzled chunk offset per k-substep
    xoff[ks] = (uint32_t)(wm * 64 + fr) * 128u + (uint32_t)((ch ^ (fr & 7)) << 4);
    woff[ks] = (uint32_t)kImg + (uint32_t)(wn * 64 + fr) * 128u + (uint32_t)((ch ^ (fr & 7)) << 4);
  }

  const int nk = K / BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of step kt have landed; after the barrier everyone's have, and everyone
    // has finished reading the other buffer (step kt-1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const uint8_t* st = lds[kt & 1];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v8bf wf[4], xf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) wf[a] = *(const v8bf*)(st + woff[ks] + a * (16 * 128));
#pragma unroll
      for (int b = 0; b < 4; ++b) xf[b] = *(const v8bf*)(st + xoff[ks] + b * (16 * 128));
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
  }

  // ---- epilogue: lane (token = l & 15 of block b, features 4 (l >> 4) .. +3 of block a) -------------
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = n0 + wn * 64 + a * 16 + fq * 4;
    const v4bf bv = *(const v4bf*)(bias + n);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = m0 + wm * 64 + b * 16 + fr;
      if (m < M) {
        size_t o = (size_t)m * N + n;
        size_t ro = o;
        if (EPI == EPI_PATCH) {
          const int b_img = m / group;
          o = (size_t)(m + b_img + 1) * N + n;
          ro = (size_t)(m - b_img * group + 1) * N + n;
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j] + (float)bv[j];
        if (EPI == EPI_GELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        }
        if (EPI == EPI_RESIDUAL || EPI == EPI_PATCH) {
          const v4bf rv = *(const v4bf*)(res + ro);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
        }
        v4bf ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = (__bf16)v[j];
        *(v4bf*)(out + o) = ov;
      }
    }
  }
}


// =====================================================================================================
// 256 x 256 tile for the wide layers (ViT-B / ViT-L: n_out % 256 == 0).  The 128 x 128 kernel above moves (128 + 128) x 64 x 2
// bytes through L2 -> LDS per 128 x 128 x 64 MACs, two workgroups per CU do so independently, and every K step ends in a
// vmcnt(0) + barrier: measured bound by that at ~28 % of the matrix peak (the library reaches 35-50 % on the
// ViT-B shapes).  This kernel follows the guide's 256^2 eight-phase structure (cdna_hip_programming.md §5):
//   * ONE workgroup of 8 waves per CU owns a 256 x 256 tile: waves 2 (token halves, `wr`) x 4 (feature quarters, `wc`); per
//     wave D^T[64 features][128 tokens] = 4 x 8 accumulator tiles of v_mfma_f32_16x16x32_bf16 (128 VGPRs);
//   * a 64-deep K tile is computed in FOUR PHASES, one 32-feature x 64-token quadrant (16 MFMAs) each, in the order
//     (f0,t0) (f1,t0) (f1,t1) (f0,t1): a phase re-reads from LDS only the operand half that changes (W half: 4 ds_read_b128,
//     x half: 8), the W0 fragments stay in registers for the fourth;
//   * the two waves of a SIMD (w and w + 4) run half a phase apart: between two barriers one of them is in its MFMA cluster,
//     the other in its memory cluster (fragment reads + its LDS-DMA pieces) — the matrix pipe always has a wave that does
//     nothing but feed it (two barriers per phase, waves 4-7 one barrier late);
//   * staging is cut into UNITS of 16 KiB = the rows one phase starts to need: X0 (token rows [0,64) of both wave rows), W0
//     (feature rows [0,32) of all four wave columns), W1, X1, in that order, one unit (2 pieces of 1 KiB per wave) issued
//     per phase, SIX units ahead, into 2 buffers x 4 units = 128 KiB; one counted wait per K tile (vmcnt(4): two units
//     stay in flight across the barriers) in the fourth phase retires the next K tile.  Hazards, with the half-phase lag
//     counted in: a unit is first read two barriers after every wave's wait for it, and restaged at least TWO phases after
//     its last fragment read was issued (X0: read in phase 0, restaged in phase 2; W0 0 / 3; W1 1 / 0 of the next K tile;
//     X1 2 / 1), so the lgkmcnt(0) for a phase's reads can sit behind the barrier, at the head of the MFMA cluster, where
//     the read latency overlaps the partner's last MFMAs instead of lengthening the memory cluster.
constexpr int G2M = 256, G2N = 256;
constexpr int kUnit2 = 128 * 128;         // 16 KiB: 128 rows x 64 k
constexpr int kBuf2 = 4 * kUnit2;         // X0 | W0 | W1 | X1
constexpr int G256_LDS = 2 * kBuf2;       // 128 KiB
constexpr int kAhead2 = 6;                // units issued ahead of the phase that runs

// CONV: the same kernel as an implicit GEMM over a channels-last image batch x [batch][cH][cW][cC] (+ one trailing row of cC
// zeros, row M): output row m = (b, y, x) gathers kh x ckw taps, K = taps * cC, k = (tap, channel); tap t reads the row of
// pixel (y + cdy0 + t / ckw, x + cdx0 + t % ckw) of the same image, or the zero row when that pixel is outside it.  A K tile
// (64 channels of one tap) is the same 1 KiB pieces as before — the tap only moves the scalar base, and a lane whose pixel
// is outside points at the zero row instead (one v_cndmask per piece).  W [N][K] with k in the same (tap, channel) order.
// cpar >= 0: the result is one parity class of a stride-2 transposed convolution — row (b, y, x) goes to pixel
// (2y + (cpar >> 1), 2x + (cpar & 1)) of a [batch][2 cH][2 cW][N] output (the four classes interleave there, no copy).
template <int EPI, bool CONV>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const __bf16* __restrict__ X, const __bf16* __restrict__ W,
                                                         const __bf16* __restrict__ bias, const __bf16* __restrict__ res,
                                                         __bf16* __restrict__ out, int M, int N, int K, int n_tiles_n,
                                                         int n_tiles, int cH, int cW, int cC, int ckw, int cdy0, int cdx0,
                                                         int cpar) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds2[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = K / BK;
  const int n_units = 4 * nk;
  // Persistent workgroups (one per CU) walk the tile list.  (Measured and dropped: starting them up to one tile-time apart so
  // that epilogues do not coincide — the output writes are not a shared-HBM burst problem; every start delay came back as
  // added time, 1268 -> 1344 us per ViT-B layer.)
  for (int slot = blockIdx.x; slot < n_tiles; slot += gridDim.x) {
  int tile;
  {
    const int xcd = slot & 7, idx = slot >> 3;
    const int q = n_tiles >> 3, r = n_tiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / n_tiles_n, tn = tile - tm * n_tiles_n;
  const int m0 = tm * G2M, n0 = tn * G2N;

  // ---- staging: unit u = 4 t + j of K tile t; j = 0: X0, 1: W0, 2: W1, 3: X1.  Wave w issues pieces 2w, 2w+1 (8 unit rows each).
  // unit row rho -> tile row:  X_h: rho < 64 ? rho + 64 h : 128 + (rho - 64) + 64 h;   W_h: (rho >> 5) * 64 + 32 h + (rho & 31)
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds2[0]);
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  uint32_t voffx[2][2], voffw[2];          // per-lane byte offsets from X / W (+ k offset of the K tile added as a scalar)
  uint32_t vtaps[2][2];                    // CONV: bit t set = tap t of this lane's pixel lies inside the image
  const int n_taps = CONV ? K / cC : 1, tiles_per_tap = CONV ? cC / BK : 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rho = (wave * 2 + i) * 8 + prow;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int xr = (rho < 64 ? rho : 128 + (rho - 64)) + 64 * h;
      if (CONV) {
        const int m = m0 + xr, hw = cH * cW;
        const int bimg = m / hw, rem = m - bimg * hw, py = rem / cW, px = rem - py * cW;
        uint32_t bits = 0;
        for (int t = 0; t < n_taps; ++t) {
          const int yy = py + cdy0 + t / ckw, xx = px + cdx0 + t % ckw;
          if (m < M && yy >= 0 && yy < cH && xx >= 0 && xx < cW) bits |= 1u << t;
        }
        vtaps[h][i] = bits;
        voffx[h][i] = (uint32_t)(((size_t)min(m, M - 1) * cC + pchunk * 8) * 2);
      } else {
        voffx[h][i] = (uint32_t)(((size_t)min(m0 + xr, M - 1) * K + pchunk * 8) * 2 - (size_t)min(m0, M - 1) * K * 2);
      }
    }
    const int wrow = (rho >> 5) * 64 + (rho & 31);
    voffw[i] = (uint32_t)(((size_t)wrow * K + pchunk * 8) * 2);
  }
  const char* const xbase = CONV ? (const char*)X : (const char*)(X + (size_t)min(m0, M - 1) * K);
  const char* const wbase = (const char*)(W + (size_t)n0 * K);
  const size_t whalf = (size_t)32 * K * 2;
  auto issue = [&](int u) {
    if (u >= n_units) return;
    const int t = u >> 2, j = u & 3;
    const bool is_x = j == 0 || j == 3;
    const char* sb = is_x ? xbase + (size_t)t * (BK * 2) : wbase + (size_t)t * (BK * 2) + (j == 2 ? whalf : 0);
    int tap = 0;
    uint32_t zoff = 0, spos = 0;           // CONV: offset (from sb) of this lane's 16 bytes in the zero row; forward part of the tap shift
    if (CONV && is_x) {
      tap = t / tiles_per_tap;
      const long long shift = ((long long)(cdy0 + tap / ckw) * cW + (cdx0 + tap % ckw)) * cC * 2;   // bytes, may be negative
      // a backward shift moves the scalar base (the per-lane offsets are unsigned), a forward one is added per lane: the
      // zero row (behind the batch) stays reachable from the base however small the batch is
      const long long sneg = shift < 0 ? shift : 0;
      spos = (uint32_t)(shift - sneg);
      sb = xbase + sneg + (long long)(t - tap * tiles_per_tap) * (BK * 2);
      zoff = (uint32_t)((long long)M * cC * 2 - sneg) + (uint32_t)pchunk * 16u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint32_t vo = j == 0 ? voffx[0][i] : (j == 3 ? voffx[1][i] : voffw[i]);
      if (CONV && is_x) vo = ((j == 0 ? vtaps[0][i] : vtaps[1][i]) >> tap) & 1u ? vo + spos : zoff;
      const uint32_t dst = lds0 + (uint32_t)(t & 1) * kBuf2 + (uint32_t)j * kUnit2 + (uint32_t)(wave * 2 + i) * 1024u;
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, %2\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(vo), "s"(sb), "s"(dst)
          : "memory");
    }
  };

  v4f acc[4][8];   // [feature block of 16][token block of 16]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};

  // fragment addressing inside a unit: lane reads unit row (block * 16 + fr), 16-byte chunk (4 ks + fq) ^ (row & 7)
  const int fr = lane & 15, fq = lane >> 4;
  uint32_t xoff[2], woff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ks * 4 + fq;
    xoff[ks] = (uint32_t)(wr * 64 + fr) * 128u + (uint32_t)((ch ^ (fr & 7)) << 4);
    woff[ks] = (uint32_t)(wc * 32 + fr) * 128u + (uint32_t)((ch ^ (fr & 7)) << 4);
  }
  v8bf xf[2][4], wf[2][2][2];   // x fragments of the current token half [ks][block]; W fragments [feature half][ks][block]

  auto read_x = [&](const uint8_t* buf, int th) {
    const uint8_t* u = buf + (th == 0 ? 0 : 3) * kUnit2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < 4; ++b) xf[ks][b] = *(const v8bf*)(u + xoff[ks] + b * 2048);
  };
  auto read_w = [&](const uint8_t* buf, int fh) {
    const uint8_t* u = buf + (1 + fh) * kUnit2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < 2; ++a) wf[fh][ks][a] = *(const v8bf*)(u + woff[ks] + a * 2048);
  };
  auto barrier = []() { asm volatile("s_barrier" ::: "memory"); };
#define G2_MFMA(FH, TH)                                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                                         \
  __builtin_amdgcn_s_setprio(1);                                                                             \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
  _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                              \
  _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                              \
    acc[2 * (FH) + a][4 * (TH) + b] =                                                                        \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[FH][ks][a], xf[ks][b], acc[2 * (FH) + a][4 * (TH) + b], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);

  // ---- prologue: K tile 0 and three units of K tile 1 in flight, K tile 0 landed -----------------------------------
#pragma unroll
  for (int u = 0; u < kAhead2; ++u) issue(u);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();
  if (wr == 1) barrier();   // waves 4-7 run one barrier (half a phase) behind their SIMD partners

  for (int t = 0; t < nk; ++t) {
    const uint8_t* buf = lds2 + (size_t)(t & 1) * kBuf2;
    const int g = 4 * t;
    // phase 0: quadrant (f0, t0)
    read_w(buf, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(buf, 0);
    issue(g + kAhead2);
    barrier();
    G2_MFMA(0, 0)
    barrier();
    // phase 1: quadrant (f1, t0)
    read_w(buf, 1);
    issue(g + 1 + kAhead2);
    barrier();
    G2_MFMA(1, 0)
    barrier();
    // phase 2: quadrant (f1, t1)
    read_x(buf, 1);
    issue(g + 2 + kAhead2);
    barrier();
    G2_MFMA(1, 1)
    barrier();
    // phase 3: quadrant (f0, t1); the one wait of the K tile: everything but the two youngest units has landed, i.e. all of
    // K tile t + 1 (its first reads come two barriers later, behind every wave's wait)
    issue(g + 3 + kAhead2);
    if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    G2_MFMA(0, 1)
    barrier();
  }
  if (wr == 0) barrier();   // (same number of barriers in both halves)
#undef G2_MFMA

#ifdef VC_G2_PROBE_NOEPI   // timing probe: accumulators kept alive, nothing stored
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) asm volatile("" :: "v"(acc[a][b]));
  asm volatile("s_barrier" ::: "memory");
  continue;
#endif
  // ---- epilogue ------------------------------------------------------------------------------------------------------------
  // A lane holds, per (feature block a, token block b), 4 consecutive features of token 16 b + fr: 8 bytes of an output row.
  // Stored from that layout, a store instruction touches 16 to 64 rows with 8 to 16 bytes each, and the epilogue cost 40-65 %
  // of a K = 768 tile's main loop (stores are bound by the lines they touch: same bytes in contiguous KiB ran 100 us per
  // ViT-B layer faster).  Every wave therefore turns its 128 x 64 block through its own 16 KiB of the (now idle) staging
  // buffers — all fragment reads of the tile are behind the last barrier: 8-byte writes from the accumulator layout, 16-byte
  // reads of [8 rows][128 B] per instruction, so each store instruction writes 8 whole 128-byte lines.  The residual takes the
  // same way in: whole-line loads, 16-byte LDS writes, and each lane picks up its own 8 bytes in the accumulator layout, adds
  // in float32 and overwrites them with the rounded result (nothing is rounded before the residual add).  16-byte slots are
  // XORed with (row >> 1) & 7: the 16-byte reads are conflict-free in the hardware's lane groups ({0-3, 12-15, 20-27}, ...:
  // 16 distinct slots of the 256-byte bank row each); the 8-byte writes keep a 2-way conflict (rows 2k and 2k + 1 of a
  // 16-lane group share a slot), which a ds_write_b64's own issue time nearly covers.
  typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
  typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
  typedef __bf16 v2bf16 __attribute__((ext_vector_type(2)));
  uint8_t* const ep = lds2 + (size_t)wave * 16384;
  const int lrow = lane >> 3, lslot = lane & 7;          // whole-line view: row 8 i + lrow, 16-byte slot lslot
  auto ep_at = [&](int row, int byte) { return ep + row * 128 + (byte ^ (((row >> 1) & 7) << 4)); };
  const size_t col0 = (size_t)n0 + wc * 64;
  if (EPI == EPI_RESIDUAL) {
    v4u32 rr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      rr[i] = *(const v4u32*)(res + (size_t)min(m0 + wr * 128 + 8 * i + lrow, M - 1) * N + col0 + lslot * 8);
#pragma unroll
    for (int i = 0; i < 16; ++i) *(v4u32*)ep_at(8 * i + lrow, lslot * 16) = rr[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const v4bf bv = *(const v4bf*)(bias + col0 + a * 16 + fq * 4);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      uint8_t* const pp = ep_at(16 * b + fr, 32 * a + 8 * fq);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j] + (float)bv[j];
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const v2f_t g = gelu_erf2((v2f_t){v[j], v[j + 1]});
          v[j] = g[0];
          v[j + 1] = g[1];
        }
      }
      if (EPI == EPI_RESIDUAL) {
        const v2u32 r8 = *(const v2u32*)pp;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          v[2 * j] += __uint_as_float(r8[j] << 16);
          v[2 * j + 1] += __uint_as_float(r8[j] & 0xffff0000u);
        }
      }
      const v2bf16 q0 = {(__bf16)v[0], (__bf16)v[1]}, q1 = {(__bf16)v[2], (__bf16)v[3]};
      *(v2u32*)pp = (v2u32){*(const unsigned int*)&q0, *(const unsigned int*)&q1};
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (CONV && cpar >= 0) {
    // a lane's 16 rows are 8 pixels apart: one division for the first, then carries
    int mm = m0 + wr * 128 + lrow;
    int pb = mm / (cH * cW), py = (mm - pb * cH * cW) / cW, px = mm - pb * cH * cW - py * cW;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 8 * i + lrow;
      const v4u32 o = *(const v4u32*)ep_at(row, lslot * 16);
      const size_t orow = ((size_t)pb * 2 * cH + 2 * py + (cpar >> 1)) * (2 * cW) + 2 * px + (cpar & 1);
      if (mm < M) *(v4u32*)(out + orow * N + col0 + lslot * 8) = o;
      mm += 8;
      px += 8;
      while (px >= cW) { px -= cW; if (++py == cH) { py = 0; ++pb; } }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 8 * i + lrow, m = m0 + wr * 128 + row;
      const v4u32 o = *(const v4u32*)ep_at(row, lslot * 16);
      if (m < M) *(v4u32*)(out + (size_t)m * N + col0 + lslot * 8) = o;
    }
  }
  asm volatile("s_barrier" ::: "memory");   // every wave is past its last fragment read of this tile before the next tile's copies land
  }  // tiles of this workgroup
}

// =====================================================================================================
// x-stationary kernel for K = 384 (every Linear of ViT-S that reads the 384-wide residual stream or the
// attention output: qkv, proj, fc1).  The staged kernel above moves (128 + 128) x K bytes through
// L2 -> LDS per 128 x 128 tile and is bound by that traffic (measured ~10 TB/s aggregate at 30 % MFMA
// utilisation).  Here a wave keeps ITS 32 token rows resident in registers as MFMA B fragments for the
// whole launch (24 k-steps x 4 VGPRs = 96 VGPRs) and only W streams, once per 256-row workgroup:
//   * W is pre-tiled on the host into fragment order (vc_linear_xs_prepare): piece (nb, ks) is the 1 KiB
//     A operand of v_mfma_f32_32x32x16_bf16 for features 32 nb .. +32, k 16 ks .. +16, so a stage
//     (one 32-feature block, 24 pieces) is 24 perfectly coalesced LDS-DMA instructions and a fragment
//     read is ds_read_b128 at lane x 16 (conflict free, no swizzle);
//   * 8 waves x 32 rows = 256 token rows per workgroup, one persistent workgroup per CU; the
//     (row tile, feature block) stages are split evenly over the grid in row-tile-major order, so a
//     workgroup reloads its x rows at most a few times and the grid finishes together (no tile-count
//     quantisation: 14 400 stages over 256 CUs at fc1);
//   * ring of 3 stages (72 KiB), filled 2 stages ahead, one barrier per stage (24 MFMAs per wave);
//   * all 8 waves run the same stream: the epilogue of block i-1 (exact GELU, residual, bf16 pack,
//     transpose, store) is cut into slices issued between the 24 MFMAs of block i (see the epilogue
//     comment for why the alternative — partner waves in anti-phase — serialises on this hardware);
//   * LayerNorm is fused into the x load: a lane and its partner (l ^ 32) hold one whole row, the
//     statistics are two-pass float32 in registers, gamma is folded into W and beta into the bias by
//     the prepare step, so the normalised activations never exist in memory; the bias is the
//     accumulator's initial value;
//   * D^T orientation (features on registers, token on the lane); v_permlane32_swap + a wave-private
//     LDS transposer turn a block into 16 rows x 64 contiguous bytes per store instruction.
// Measured (76 550 rows, 1x MI355X): qkv+LN 100 us, proj+residual 46 us, fc1+LN+GELU 156 us; ablations
// (tools/xs_variants.sh): MFMA + ring alone 64 / 32 / 80 us, the x reloads ~10 us, stores ~10-20 us.
constexpr int XK = 384;
constexpr int XKS = XK / 16;                 // 24 k-steps of v_mfma_f32_32x32x16_bf16
constexpr int XStage = XKS * 1024;           // 24 KiB: one 32-feature block of W in fragment order
constexpr int XNS = 3;                       // ring slots
constexpr int XPF = XNS - 1;                 // stages in flight
constexpr int XMaxN = 4096;                  // bias staging area (floats)
constexpr int XRows = 256;                   // token rows per workgroup (8 waves x 32)
constexpr int XChunk = 32 * 128;             // x staging: 32 rows x 64 k (128-byte rows), per wave, double buffered
constexpr int XOffBias = XNS * XStage;
constexpr int XOffStage = XOffBias + XMaxN * 4;
constexpr int XLds = XOffStage + 8 * 2 * XChunk;   // 72 + 16 + 64 = 152 KiB

typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// GELU by table (EPI_GELU with a table pointer).  float32 VALU work does not overlap with the matrix pipe on
// this part, integer VALU work and LDS reads do (tools/overlap_probe.hip), and the exact GELU is ~150 float
// instructions per block and wave.  The standard bf16 pipeline evaluates gelu on the bf16-ROUNDED pre-activation
// and rounds the result to bf16, i.e. it is a function from 16 bits to 16 bits: for 2^-14 <= |x| < 8 it is looked
// up in LDS (17 exponents x 128 mantissas x 2 signs = 4352 entries of 2 bytes, entry [k][sign] with
// k = (bits & 0x7fff) - kGtLo), the index arithmetic is integer work.  A wave that holds a value outside that
// range (|x| >= 8, |x| < 2^-14, NaN) takes the float path for the block (rare).  Table values: 0.5 x (1 + erff(x / sqrt 2))
// in float32, rounded to nearest-even — what torch's bf16 gelu computes.
constexpr uint32_t kGtLo = (127 - 14) << 7;                 // bf16 bits of 2^-14
constexpr uint32_t kGtN = 17 * 128;                         // magnitudes covered: [2^-14, 8)
constexpr int kGtBytes = (int)kGtN * 2 * 2;                 // 8704

__device__ __forceinline__ uint32_t f32_to_bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__global__ void gelu_table_kernel(uint16_t* __restrict__ tab) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kGtN * 2) return;
  const uint32_t k = i >> 1, sign = i & 1;
  const float x = __uint_as_float(((sign << 15) | (kGtLo + k)) << 16);
  const float y = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
  tab[i] = (uint16_t)f32_to_bf16_rne(y);
}

#ifndef VC_XS_ISSUE0
#define VC_XS_ISSUE0 1        // MFMA behind which the first piece of stage i + 2 is issued ...
#define VC_XS_ISSUE_STEP 8    // ... and the distance to the next (the last one stays ahead of the result stores at slice 19)
#endif
template <int EPI, bool LN, bool GT>
__global__ __launch_bounds__(512, 1) void xs_kernel(const __bf16* __restrict__ X, const uint8_t* __restrict__ Wp,
                                                    const float* __restrict__ biasf, const __bf16* res,
                                                    __bf16* out, int M, int N, int n_nb, int n_stages, float eps,
                                                    const uint16_t* __restrict__ gelu_tab) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[XLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int G = gridDim.x, bid = blockIdx.x;
  const int s0 = (int)((long long)n_stages * bid / G), s1 = (int)((long long)n_stages * (bid + 1) / G);
  const int n = s1 - s0;
  if (n <= 0) return;

  float* bias_l = (float*)(lds + XOffBias);
  for (int i = tid; i < N; i += 512) bias_l[i] = biasf[i];   // visible after the first stage barrier
  // GELU table behind the bias (the host checks that both fit the 16 KiB area)
  uint8_t* const gt_l = lds + XOffBias + ((N * 4 + 15) & ~15);
  if (GT)
    for (int i = tid; i < kGtBytes / 4; i += 512) ((uint32_t*)gt_l)[i] = ((const uint32_t*)gelu_tab)[i];

  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0]);
  // ---- producer: wave w issues pieces 3w..3w+2 of every stage ------------------------------------
  int inb = s0 % n_nb, islot = 0, ileft = n;   // feature block / ring slot / stages left to issue
  // One piece: wave-uniform base in SGPRs + one per-lane offset (with per-piece 64-bit VGPR addresses the compiler keeps
  // three pairs alive across the stage loop).  The last piece of a stage advances the stream.
  const uint32_t lane_off16 = (uint32_t)lane * 16u;
  auto issue_piece = [&](int i) {
    const uint8_t* sbase = Wp + (size_t)inb * XStage + (size_t)(wave * 3 + i) * 1024;
    const uint32_t dst = lds0 + (uint32_t)islot * XStage + (uint32_t)(wave * 3 + i) * 1024u;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off16), "s"(sbase), "s"(dst)
        : "memory");
    if (i == 2) {
      // past the end the last stage is staged again into a slot nobody reads (keeps the vmcnt counts uniform)
      if (ileft > 1) { --ileft; inb = inb + 1 == n_nb ? 0 : inb + 1; }
      islot = islot + 1 == XNS ? 0 : islot + 1;
    }
  };
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_piece(i);
  };
#pragma unroll
  for (int j = 0; j < XPF; ++j) issue();

  // ---- x rows of this wave as B fragments ---------------------------------------------------------
  // Fragment-shaped global loads (32 rows x 32 B per instruction) run at ~5 B/clk/CU, and a staged copy
  // through LDS-DMA is limited by the bytes the staging area lets a wave keep in flight.  So the 32 rows
  // are fetched in whole 128-byte lines straight into registers — 24 loads of 8 rows x 128 B, all in
  // flight at once (the registers that will hold the fragments are the landing zone) — and then turned
  // into fragments 64 k at a time through a 4 KiB wave-private LDS transposer ([32 rows][128 B], 16-byte
  // chunk ^ ((row >> 1) & 7): conflict free for the 8-row writes and the 32-row column-slice reads).
  // DS operations of one wave execute in order, so the transposer needs no waits between chunks.
  v8bf xf[XKS];
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int prow = lane >> 3;
  auto x_issue = [&](int mt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(mt * XRows + wave * 32 + 8 * p + prow, M - 1);
      const uint8_t* src = (const uint8_t*)X + (size_t)m * (XK * 2) + (lane & 7) * 16;
#pragma unroll
      for (int c = 0; c < XK / 64; ++c) *(v4u*)&xf[4 * c + p] = *(const v4u*)(src + c * 128);   // raw lines land in xf itself
    }
  };
  auto x_finish = [&]() {
    uint8_t* const tp = lds + XOffStage + wave * (2 * XChunk);
    const int fsw = (r >> 1) & 7;
#pragma unroll
    for (int c = 0; c < XK / 64; ++c) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = 8 * p + prow;
        *(v4u*)(tp + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4)) = *(const v4u*)&xf[4 * c + p];
      }
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        xf[4 * c + k4] = *(const v8bf*)(tp + r * 128 + (((2 * k4 + h) ^ fsw) << 4));
    }
    if (LN) {
      // two-pass float32 statistics with packed (2-wide) VALU ops; the passes re-expand the bf16 pairs
      // (shift / mask) instead of keeping 192 floats live
      auto expand = [](uint32_t u) { return (v2f){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; };
      v2f s2 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < XKS; ++ks) {
        const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
        for (int j = 0; j < 4; ++j) s2 += expand(u[j]);
      }
      float sum = s2[0] + s2[1];
      sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.0f / XK);
      const v2f mean2 = {mean, mean};
#pragma unroll
      for (int ks = 0; ks < XKS; ++ks) asm volatile("" : "+v"(xf[ks]));
      v2f q2 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < XKS; ++ks) {
        const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const v2f d = expand(u[j]) - mean2; q2 = __builtin_elementwise_fma(d, d, q2); }
      }
      float q = q2[0] + q2[1];
      q += __shfl_xor(q, 32);
      const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / XK) + eps);
      const v2f rstd2 = {rstd, rstd};
#pragma unroll
      for (int ks = 0; ks < XKS; ++ks) asm volatile("" : "+v"(xf[ks]));
#pragma unroll
      for (int ks = 0; ks < XKS; ++ks) {
        const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const v2f y = (expand(u[j]) - mean2) * rstd2;
          xf[ks][2 * j] = (__bf16)y[0];
          xf[ks][2 * j + 1] = (__bf16)y[1];
        }
      }
    }
  };

  // ---- epilogue of one 32 (features) x 32 (tokens) block.  MFMA side: lane = token r, half h, register
  // 4g + j = feature 8g + 4h + j.  A row-per-lane global access (32 rows per instruction) is issue-bound in
  // the texture path, so both the residual read and the result write go through a wave-private transposer
  // in the (idle) x staging area — [32 tokens][64 B], 16-byte chunk ^ ((token >> 2) & 3), conflict free both
  // ways — and touch memory as lane -> (token l >> 2 (+16), 16-byte chunk l & 3): 16 rows x 64 contiguous
  // bytes per instruction.  DS operations of one wave execute in order, so no wait separates write and read.
  // Results leave through a raw buffer store: rows past M fall outside num_records and are dropped by the
  // hardware, so every epilogue issues exactly two store instructions (the vmcnt bookkeeping below counts them).
  //
  // The epilogue of block i-1 is cut into 24 slices that are issued BETWEEN the 24 MFMAs of block i, in the
  // same wave (all 8 waves run the same stream, no wave roles).  tools/overlap_probe.hip: float32 VALU work does
  // not overlap with the matrix pipe on this part however it is arranged (integer VALU work does), so this
  // costs the same as running the epilogue as a cluster — time = MFMA time + float VALU time — and is kept
  // for its simplicity; the lever for this kernel is fewer float instructions per output.
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)M * N * 2), 0x00020000);
  uint8_t* const tr_out = lds + XOffStage + wave * (2 * XChunk);
  uint8_t* const tr_res = tr_out + 2048;
  const int crow = lane >> 2, cch = lane & 3;
  const int rsw = (r >> 2) & 3;
  v4u resq[2];
  auto load_res = [&](int m_base, int nb) {
    if (EPI == EPI_RESIDUAL) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        resq[q] = *(const v4u*)(res + (size_t)min(m_base + crow + 16 * q, M - 1) * N + nb * 32 + cch * 8);
    }
  };
  // state that flows from slice to slice
  v16f pa;          // the pending block's accumulators
  float ev[16];     // ... after GELU / residual
  uint32_t ep[8];   // ... packed to bf16 pairs
  v4u eo[2];        // ... transposed for the store
  int pm_base = 0, pnb = 0;
  auto res_to_lds = [&]() {   // the pending block's residual (loaded an iteration ago) enters the transposer
    if (EPI == EPI_RESIDUAL) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = crow + 16 * q;
        *(v4u*)(tr_res + row * 64 + ((cch ^ ((row >> 2) & 3)) << 4)) = resq[q];
      }
    }
  };
  uint32_t gq[8];        // table GELU: looked-up halves in flight (ring of 4 packed registers)
  uint32_t gres[8];      // ... results, packed bf16 pairs, parked until the range check
  uint32_t gbad = 0;     // ... largest table index seen (out-of-range detector)
  auto slice = [&](int sl) {
#ifdef VC_XS_NOEPI
    if (sl == 23 && pa[0] == 12345.678f && pa[7] == 1.25f) out[pnb] = (__bf16)pa[1];
    return;
#endif
    if (EPI == EPI_GELU && GT) {
      // slices 0-3: pre-activations -> bf16 pairs (ep), the only float work; 4-11: table index + LDS gather of
      // packed register sl-4; 7-14: results packed back; 15: range check (+ float path); 16/17/19/22: exchange,
      // transposer write / read, store
      typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
      if (sl < 4) {
        const int g = sl;
        const v2bf lo = {(__bf16)pa[4 * g], (__bf16)pa[4 * g + 1]}, hi = {(__bf16)pa[4 * g + 2], (__bf16)pa[4 * g + 3]};
        ep[2 * g] = *(const uint32_t*)&lo;
        ep[2 * g + 1] = *(const uint32_t*)&hi;
        if (sl == 0) gbad = 0;
      }
      if (sl >= 4 && sl < 12) {
        const int qd = sl - 4;
        const uint32_t b0 = ep[qd] & 0xffffu, b1 = ep[qd] >> 16;
        const uint32_t k0 = (b0 & 0x7fffu) - kGtLo, k1 = (b1 & 0x7fffu) - kGtLo;   // wraps to a huge value below the range
        gbad = max(gbad, max(k0, k1));
        const uint32_t a0 = min(k0, kGtN - 1) * 4 + ((b0 >> 15) << 1), a1 = min(k1, kGtN - 1) * 4 + ((b1 >> 15) << 1);
        gq[2 * (qd & 3)] = *(const uint16_t*)(gt_l + a0);
        gq[2 * (qd & 3) + 1] = *(const uint16_t*)(gt_l + a1);
      }
      if (sl >= 7 && sl < 15) {        // three slices (~100+ cycles) after the gather was issued
        const int qd = sl - 7;
        gres[qd] = gq[2 * (qd & 3)] | (gq[2 * (qd & 3) + 1] << 16);
      }
      if (sl == 15) {
        if (__any(gbad >= kGtN)) {       // a value outside the table's range somewhere in the wave: float path for the block
#pragma unroll
          for (int qd = 0; qd < 8; ++qd) {
            const v2f_t xin = {__uint_as_float(ep[qd] << 16), __uint_as_float(ep[qd] & 0xffff0000u)};
            const v2f_t y = gelu_erf2(xin);
            const v2bf o = {(__bf16)y[0], (__bf16)y[1]};
            ep[qd] = *(const uint32_t*)&o;
          }
        } else {
#pragma unroll
          for (int qd = 0; qd < 8; ++qd) ep[qd] = gres[qd];
        }
      }
      if (sl == 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const auto sw = __builtin_amdgcn_permlane32_swap(ep[i], ep[i + 4], false, false);
          ep[i] = sw[0];
          ep[i + 4] = sw[1];
        }
      }
      if (sl == 17) {
        *(v4u*)(tr_out + r * 64 + (((2 * h) ^ rsw) << 4)) = (v4u){ep[0], ep[1], ep[4], ep[5]};
        *(v4u*)(tr_out + r * 64 + (((2 * h + 1) ^ rsw) << 4)) = (v4u){ep[2], ep[3], ep[6], ep[7]};
      }
      if (sl == 19) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int row = crow + 16 * q;
          eo[q] = *(const v4u*)(tr_out + row * 64 + ((cch ^ ((row >> 2) & 3)) << 4));
        }
      }
      if (sl == 22) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int row = crow + 16 * q;
          __builtin_amdgcn_raw_buffer_store_b128(eo[q], out_rs, (int)(((size_t)(pm_base + row) * N + pnb * 32 + cch * 8) * 2), 0, 0);
        }
      }
      return;
    }
    if (sl < 8) {                       // values 2 sl, 2 sl + 1
      const int j0 = 2 * sl;
      if (EPI == EPI_GELU) {
        const v2f_t y = gelu_erf2((v2f_t){pa[j0], pa[j0 + 1]});
        ev[j0] = y[0];
        ev[j0 + 1] = y[1];
      } else if (EPI == EPI_RESIDUAL) {
        if ((sl & 1) == 0) {            // one 8-byte read serves values 4g .. 4g + 3
          const int g = sl >> 1;
          const v4bf rv = *(const v4bf*)(tr_res + r * 64 + ((g ^ rsw) << 4) + 8 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) ev[4 * g + j] = pa[4 * g + j] + (float)rv[j];
        }
      } else {
        ev[j0] = pa[j0];
        ev[j0 + 1] = pa[j0 + 1];
      }
    } else if (sl < 12) {               // pack group g
      const int g = sl - 8;
      typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
      const v2bf lo = {(__bf16)ev[4 * g], (__bf16)ev[4 * g + 1]}, hi = {(__bf16)ev[4 * g + 2], (__bf16)ev[4 * g + 3]};
      ep[2 * g] = *(const uint32_t*)&lo;
      ep[2 * g + 1] = *(const uint32_t*)&hi;
    } else if (sl == 12) {
      // exchange with the partner lane (l ^ 32): h = 0 ends with features 0..15, h = 1 with 16..31
#pragma unroll
      for (int i = 0; i < 4; ++i) {     // i = 0,1: group 0 <-> group 2;  i = 2,3: group 1 <-> group 3
        const auto sw = __builtin_amdgcn_permlane32_swap(ep[i], ep[i + 4], false, false);
        ep[i] = sw[0];
        ep[i + 4] = sw[1];
      }
    } else if (sl == 13) {
      *(v4u*)(tr_out + r * 64 + (((2 * h) ^ rsw) << 4)) = (v4u){ep[0], ep[1], ep[4], ep[5]};
      *(v4u*)(tr_out + r * 64 + (((2 * h + 1) ^ rsw) << 4)) = (v4u){ep[2], ep[3], ep[6], ep[7]};
    } else if (sl == 15) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = crow + 16 * q;
        eo[q] = *(const v4u*)(tr_out + row * 64 + ((cch ^ ((row >> 2) & 3)) << 4));
      }
    } else if (sl == 19) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = crow + 16 * q;
#ifdef VC_XS_NOSTORE
        if (eo[q][0] == 0x12345678u && eo[q][3] == 0x9abcdef0u)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(eo[q], out_rs, (int)(((size_t)(pm_base + row) * N + pnb * 32 + cch * 8) * 2), 0, 0);
      }
    }
  };

#ifdef VC_XS_STAMP
  // diagnostic build only: per-wave cycle totals of the loop phases, written over the start of `out`
  // after the last real store (tools/stamp_xs.py); no output value is computed from them
  uint32_t st_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t st_t0 = __builtin_amdgcn_s_memtime();
  uint64_t st_prev = st_t0;
#define XS_STAMP(k) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_c[k] += (uint32_t)(t_ - st_prev); st_prev = t_; }
#else
#define XS_STAMP(k)
#endif
  v16f acc;
  int mt_cur = -1, slot = 0;
  int mt = s0 / n_nb, nb = s0 - mt * n_nb;
  for (int i = 0; i < n; ++i) {
#if defined(VC_XS_STAMP) && defined(VC_XS_STAMP_STEADY)
    if (i == 2) {   // steady-state totals: drop the pipeline fill and the first row tile's load (st_c[6] = stages counted)
#pragma unroll
      for (int k = 0; k < 6; ++k) st_c[k] = 0;
    }
#endif
    // This wave's pieces of stage i (issued BETWEEN the MFMAs of iteration i-2) have landed once only operations issued
    // after them are pending.  Per iteration a wave issues (residual epilogue) 2 loads, then inside the MFMA loop its 3
    // pieces and the 2 result stores of the pending block: iteration i-1 alone accounts for 5 (+2) younger operations in
    // every case — steady state, behind a row-tile switch (whose x loads drain everything older anyway) and while the
    // pipeline fills from iteration 2 on; the stores of iteration i-2 that follow its last piece are not counted, so the
    // wait is never too weak.  Counting keeps a wave from stalling on the write acknowledgement of its latest stores.
    if (i >= 2) {
      if (EPI == EPI_RESIDUAL) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    XS_STAMP(0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    XS_STAMP(1)
    const bool pending = i > 0;
    if (pending) res_to_lds();
#ifdef VC_XS_NOXRELOAD
    const bool reload = mt_cur < 0;
#else
    const bool reload = mt != mt_cur;
#endif
    // (stage i + 2 goes into the slot read during iteration i - 1: its three pieces are issued between this
    // iteration's MFMAs — in a block behind the barrier they held every wave for 150-300 cycles with the matrix pipe idle)
    const int m_base = mt * XRows + wave * 32;
    if (reload) {
      // new row tile: the pending block cannot ride under this block's MFMAs (its x loads come first)
      x_issue(mt);
      if (pending) {
#pragma unroll
        for (int sl = 0; sl < XKS; ++sl) slice(sl);
      }
      x_finish();
      mt_cur = mt;
    }
    XS_STAMP(2)
    load_res(m_base, nb);
    const uint8_t* st = lds + slot * XStage + lane * 16;
    {
      // the bias is the accumulator's initial value: register 4g + j <- bias[32 nb + 8g + 4h + j]
#ifdef VC_XS_NOBIAS
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#else
      const float* bl = bias_l + nb * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4f bv = *(const v4f*)(bl + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = bv[j];
      }
#endif
    }
    {
      // W fragments are read XRD k-steps ahead of the MFMA that consumes them (LDS latency ~ 4 MFMAs);
      // sched_barrier pins [MFMA, fragment read, epilogue slice] per k-step, the compiler places the waits
#ifndef VC_XS_RD
#define VC_XS_RD 8
#endif
      constexpr int XRD = (EPI == EPI_GELU && GT) ? 6 : VC_XS_RD;   // the table epilogue needs the registers
      v8bf wf[XKS];
#pragma unroll
      for (int ks = 0; ks < XRD; ++ks) wf[ks] = *(const v8bf*)(st + ks * 1024);
      __builtin_amdgcn_sched_barrier(0);
      if (pending && !reload) {
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], acc, 0, 0, 0);
          if (ks + XRD < XKS) wf[ks + XRD] = *(const v8bf*)(st + (ks + XRD) * 1024);
          slice(ks);
          if (ks == VC_XS_ISSUE0 || ks == VC_XS_ISSUE0 + VC_XS_ISSUE_STEP || ks == VC_XS_ISSUE0 + 2 * VC_XS_ISSUE_STEP)
            issue_piece((ks - VC_XS_ISSUE0) / VC_XS_ISSUE_STEP);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], acc, 0, 0, 0);
          if (ks + XRD < XKS) wf[ks + XRD] = *(const v8bf*)(st + (ks + XRD) * 1024);
          if (ks == VC_XS_ISSUE0 || ks == VC_XS_ISSUE0 + VC_XS_ISSUE_STEP || ks == VC_XS_ISSUE0 + 2 * VC_XS_ISSUE_STEP)
            issue_piece((ks - VC_XS_ISSUE0) / VC_XS_ISSUE_STEP);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    XS_STAMP(3)
    pa = acc;
    pm_base = m_base;
    pnb = nb;
    slot = slot + 1 == XNS ? 0 : slot + 1;
    if (++nb == n_nb) { nb = 0; ++mt; }
  }
  // the last block's epilogue has no MFMAs to ride under
  res_to_lds();
#pragma unroll
  for (int sl = 0; sl < XKS; ++sl) slice(sl);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped refills before the LDS is released
#ifdef VC_XS_STAMP
#ifdef VC_XS_STAMP_STEADY
  st_c[6] = (uint32_t)(n > 2 ? n - 2 : 0);
#else
  XS_STAMP(6)
#endif
  st_c[7] = (uint32_t)(__builtin_amdgcn_s_memtime() - st_t0);
  if (lane < 8) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v = lane == k ? st_c[k] : v;
    ((uint32_t*)out)[(bid * 8 + wave) * 8 + lane] = v;
  }
#endif
}

// W [N][K] float32 (+ optional LayerNorm gamma / beta [K] folded in) -> fragment-ordered bf16 + float32 bias.
__global__ __launch_bounds__(256) void xs_prepare_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int N, __bf16* __restrict__ Wp, float* __restrict__ biasf) {
  // one wave per output feature n
  const int lane = threadIdx.x & 63;
  const int nfeat = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (nfeat >= N) return;
  const int nb = nfeat >> 5, rr = nfeat & 31;
  float dot = 0.f;
  for (int k = lane; k < XK; k += 64) {
    const float w = W[(size_t)nfeat * XK + k];
    const float wg = gamma ? w * gamma[k] : w;
    if (beta) dot += w * beta[k];
    const int ks = k >> 4, hh = (k >> 3) & 1, j = k & 7;
    Wp[(((size_t)nb * XKS + ks) * 64 + hh * 32 + rr) * 8 + j] = (__bf16)wg;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  if (lane == 0) biasf[nfeat] = (b ? b[nfeat] : 0.f) + dot;
}


// =====================================================================================================
// Fused MLP for 384-wide models:  x += fc2(gelu(fc1(LayerNorm(x))))  in ONE kernel, so the hidden tensor
// (76 550 x 1536 bf16 = 235 MB, written by fc1 and read back by fc2: a third of a block's HBM traffic) never
// exists and fc2 no longer stages both operands through L2 -> LDS.  Structure = the x-stationary kernel with a
// second product per stage:
//   * a wave keeps its 32 normalised token rows as B fragments (96 VGPRs) AND the 384 x 32 output tile of fc2 as
//     12 accumulator tiles (192 VGPRs): one wave per SIMD (4 waves, 128 rows per workgroup, 512-register budget);
//   * a stage is one 32-wide chunk of the hidden layer: 24 pieces of W1 (as in xs_kernel) + 24 pieces of W2
//     (12 output blocks x 2 k-steps), 48 KiB, ring of 2;
//   * per stage: H^T(32 hidden x 32 tokens) = W1c X^T (24 MFMAs, bias as initial value) -> GELU by the LDS table
//     (integer work) -> the bf16 pairs ARE the B operand of Out^T += W2c^T-tile x G^T (24 MFMAs; accumulator tile as
//     the next MFMA's operand: W2's k order is permuted by the prepare step to match the register order);
//   * after the last chunk of a row tile: + fc2 bias + residual (the tile's own x rows, read back), stored in place.
// Row tiles are dealt round-robin to one persistent workgroup per CU.
constexpr int MStage = 2 * XStage;                 // 48 KiB: [W1 chunk | W2 chunk]
constexpr int MNS = 2;
constexpr int MOffB1 = MNS * MStage;               // fc1 bias (<= 2048 floats)
constexpr int MOffGt = MOffB1 + 2048 * 4;          // GELU table (8704 B)
constexpr int MOffB2 = MOffGt + 9216;              // fc2 bias (384 floats)
constexpr int MOffTr = MOffB2 + 2048;              // per-wave transposers, 4 x 4 KiB
constexpr int MLds = MOffTr + 4 * XChunk;          // 96 + 8 + 9 + 2 + 16 = 131 KiB

__global__ __launch_bounds__(256, 1) void mlp_kernel(__bf16* __restrict__ X, const uint8_t* __restrict__ Wm,
                                                     const float* __restrict__ b1f, const float* __restrict__ b2f,
                                                     int M, int n_chunks, int n_tiles, float eps,
                                                     const uint16_t* __restrict__ gelu_tab) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[MLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x, bid = blockIdx.x;
  const int my_tiles = bid < n_tiles ? (n_tiles - bid + G - 1) / G : 0;
  const int n = my_tiles * n_chunks;
  if (n <= 0) return;

  float* const b1_l = (float*)(lds + MOffB1);
  float* const b2_l = (float*)(lds + MOffB2);
  uint8_t* const gt_l = lds + MOffGt;
  for (int i = tid; i < n_chunks * 32; i += 256) b1_l[i] = b1f[i];
  for (int i = tid; i < XK; i += 256) b2_l[i] = b2f[i];
  for (int i = tid; i < kGtBytes / 4; i += 256) ((uint32_t*)gt_l)[i] = ((const uint32_t*)gelu_tab)[i];

  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0]);
  // producer: wave w issues pieces 12w .. 12w+11 of every stage
  auto issue = [&](int chunk, int slot) {
    const uint8_t* src = Wm + (size_t)chunk * MStage + (size_t)(wave * 12) * 1024 + lane * 16;
    const uint32_t dst = lds0 + (uint32_t)slot * MStage + (uint32_t)(wave * 12) * 1024u;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src + i * 1024), "s"(dst + (uint32_t)i * 1024u)
          : "memory");
    }
  };

  // x rows of this wave: whole lines into registers, transposed to fragments through LDS, LayerNorm (see xs_kernel)
  v8bf xf[XKS];
  typedef float v2f __attribute__((ext_vector_type(2)));
  uint8_t* const tp = lds + MOffTr + wave * XChunk;
  const int prow = lane >> 3;
  auto load_x = [&](int m0w) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(m0w + 8 * p + prow, M - 1);
      const uint8_t* src = (const uint8_t*)X + (size_t)m * (XK * 2) + (lane & 7) * 16;
#pragma unroll
      for (int c = 0; c < XK / 64; ++c) *(v4u*)&xf[4 * c + p] = *(const v4u*)(src + c * 128);
    }
    const int fsw = (r >> 1) & 7;
#pragma unroll
    for (int c = 0; c < XK / 64; ++c) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = 8 * p + prow;
        *(v4u*)(tp + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4)) = *(const v4u*)&xf[4 * c + p];
      }
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        xf[4 * c + k4] = *(const v8bf*)(tp + r * 128 + (((2 * k4 + h) ^ fsw) << 4));
    }
    auto expand = [](uint32_t u) { return (v2f){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; };
    v2f s2 = {0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < XKS; ++ks) {
      const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
      for (int j = 0; j < 4; ++j) s2 += expand(u[j]);
    }
    float sum = s2[0] + s2[1];
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / XK);
    const v2f mean2 = {mean, mean};
#pragma unroll
    for (int ks = 0; ks < XKS; ++ks) asm volatile("" : "+v"(xf[ks]));
    v2f q2 = {0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < XKS; ++ks) {
      const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const v2f d = expand(u[j]) - mean2; q2 = __builtin_elementwise_fma(d, d, q2); }
    }
    float q = q2[0] + q2[1];
    q += __shfl_xor(q, 32);
    const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / XK) + eps);
    const v2f rstd2 = {rstd, rstd};
#pragma unroll
    for (int ks = 0; ks < XKS; ++ks) asm volatile("" : "+v"(xf[ks]));
#pragma unroll
    for (int ks = 0; ks < XKS; ++ks) {
      const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const v2f y = (expand(u[j]) - mean2) * rstd2;
        xf[ks][2 * j] = (__bf16)y[0];
        xf[ks][2 * j + 1] = (__bf16)y[1];
      }
    }
  };

  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(X, 0, (int)((size_t)M * XK * 2), 0x00020000);
  const int crow = lane >> 2, cch = lane & 3;
  const int rsw = (r >> 2) & 3;
  uint8_t* const tr_out = tp;
  uint8_t* const tr_res = tp + 2048;

  v16f oacc[12];
  int tile = bid, chunk = 0, slot = 0;
  issue(0, 0);
  for (int i = 0; i < n; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (i + 1 < n) issue(chunk + 1 == n_chunks ? 0 : chunk + 1, slot ^ 1);   // into the slot read during iteration i-1
    const int m0w = tile * 128 + wave * 32;
    if (chunk == 0) {
      load_x(m0w);
#pragma unroll
      for (int ob = 0; ob < 12; ++ob)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[ob][j] = 0.f;
    }
    const uint8_t* st = lds + slot * MStage + lane * 16;
    // ---- fc1 chunk: H^T = W1c X^T, bias as the initial value --------------------------------------------
    v16f hacc;
    {
      const float* bl = b1_l + chunk * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4f bv = *(const v4f*)(bl + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) hacc[4 * g + j] = bv[j];
      }
    }
    {
      constexpr int RD = 8;
      v8bf wf[XKS];
#pragma unroll
      for (int ks = 0; ks < RD; ++ks) wf[ks] = *(const v8bf*)(st + ks * 1024);
#pragma unroll
      for (int ks = 0; ks < XKS; ++ks) {
#ifdef VC_MLP_NOFC1
        if (ks > 0) { asm volatile("" :: "v"(wf[ks])); if (ks + RD < XKS) wf[ks + RD] = *(const v8bf*)(st + (ks + RD) * 1024); continue; }
#endif
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], hacc, 0, 0, 0);
        if (ks + RD < XKS) wf[ks + RD] = *(const v8bf*)(st + (ks + RD) * 1024);
      }
    }
    // ---- GELU: bf16 pairs -> table (float path for a wave that holds a value outside the table) -----------
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    uint32_t gp[8];
    {
      uint32_t pb[8], gbad = 0;
#pragma unroll
      for (int qd = 0; qd < 8; ++qd) {
        const v2bf pr = {(__bf16)hacc[2 * qd], (__bf16)hacc[2 * qd + 1]};
        pb[qd] = *(const uint32_t*)&pr;
      }
      uint32_t g0[8], g1[8];
#pragma unroll
      for (int qd = 0; qd < 8; ++qd) {
        const uint32_t b0 = pb[qd] & 0xffffu, b1 = pb[qd] >> 16;
        const uint32_t k0 = (b0 & 0x7fffu) - kGtLo, k1 = (b1 & 0x7fffu) - kGtLo;
        gbad = max(gbad, max(k0, k1));
        g0[qd] = *(const uint16_t*)(gt_l + min(k0, kGtN - 1) * 4 + ((b0 >> 15) << 1));
        g1[qd] = *(const uint16_t*)(gt_l + min(k1, kGtN - 1) * 4 + ((b1 >> 15) << 1));
      }
#ifdef VC_MLP_NOGELU
      if (false) {
#else
      if (__any(gbad >= kGtN)) {
#endif
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) {
          const v2f_t y = gelu_erf2((v2f_t){__uint_as_float(pb[qd] << 16), __uint_as_float(pb[qd] & 0xffff0000u)});
          const v2bf o = {(__bf16)y[0], (__bf16)y[1]};
          gp[qd] = *(const uint32_t*)&o;
        }
      } else {
#pragma unroll
#ifdef VC_MLP_NOGELU
        for (int qd = 0; qd < 8; ++qd) gp[qd] = pb[qd];
#else
        for (int qd = 0; qd < 8; ++qd) gp[qd] = g0[qd] | (g1[qd] << 16);
#endif
      }
    }
    // ---- fc2 partial: Out^T[12 x 32 features][32 tokens] += W2c G^T (registers 8s..8s+7 of the hidden tile are k-step s)
    {
      const v4u gA = {gp[0], gp[1], gp[2], gp[3]}, gB = {gp[4], gp[5], gp[6], gp[7]};
      const v8bf g_s0 = *(const v8bf*)&gA, g_s1 = *(const v8bf*)&gB;
      const uint8_t* st2 = st + XStage;
      constexpr int RD = 8;
      v8bf wf[24];
#pragma unroll
      for (int q = 0; q < RD; ++q) wf[q] = *(const v8bf*)(st2 + q * 1024);
#pragma unroll
      for (int q = 0; q < 24; ++q) {
#ifdef VC_MLP_NOFC2
        if (q > 0) { asm volatile("" :: "v"(wf[q])); continue; }
#endif
        oacc[q >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q], (q & 1) ? g_s1 : g_s0, oacc[q >> 1], 0, 0, 0);
        if (q + RD < 24) wf[q + RD] = *(const v8bf*)(st2 + (q + RD) * 1024);
      }
    }
    // ---- last chunk of the row tile: + fc2 bias + residual, in place ----------------------------------------
    if (chunk == n_chunks - 1) {
#pragma unroll
      for (int ob = 0; ob < 12; ++ob) {
        v4u resq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
          resq[q] = *(const v4u*)(X + (size_t)min(m0w + crow + 16 * q, M - 1) * XK + ob * 32 + cch * 8);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int row = crow + 16 * q;
          *(v4u*)(tr_res + row * 64 + ((cch ^ ((row >> 2) & 3)) << 4)) = resq[q];
        }
        uint32_t ep[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const v4bf rv = *(const v4bf*)(tr_res + r * 64 + ((g ^ rsw) << 4) + 8 * h);
          const v4f bv = *(const v4f*)(b2_l + ob * 32 + 8 * g + 4 * h);
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = oacc[ob][4 * g + j] + bv[j] + (float)rv[j];
          const v2bf lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
          ep[2 * g] = *(const uint32_t*)&lo;
          ep[2 * g + 1] = *(const uint32_t*)&hi;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const auto sw = __builtin_amdgcn_permlane32_swap(ep[k], ep[k + 4], false, false);
          ep[k] = sw[0];
          ep[k + 4] = sw[1];
        }
        *(v4u*)(tr_out + r * 64 + (((2 * h) ^ rsw) << 4)) = (v4u){ep[0], ep[1], ep[4], ep[5]};
        *(v4u*)(tr_out + r * 64 + (((2 * h + 1) ^ rsw) << 4)) = (v4u){ep[2], ep[3], ep[6], ep[7]};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int row = crow + 16 * q;
          const v4u o = *(const v4u*)(tr_out + row * 64 + ((cch ^ ((row >> 2) & 3)) << 4));
          __builtin_amdgcn_raw_buffer_store_b128(o, out_rs, (int)(((size_t)(m0w + row) * XK + ob * 32 + cch * 8) * 2), 0, 0);
        }
      }
      tile += G;
    }
    slot ^= 1;
    chunk = chunk + 1 == n_chunks ? 0 : chunk + 1;
  }
}

// Role-split form of the fused MLP (the one that is launched): 8 waves, two per SIMD.  Waves 0-3 ("A") keep the
// normalised x rows and compute H^T chunk by chunk + the GELU; waves 4-7 ("B") keep the 384 x 32 output accumulators
// and run fc2 one stage behind, taking the 32 x 32 bf16 activation tile of their partner (same SIMD, same 32 token rows)
// through a 2 KiB LDS buffer.  Compared with one wave doing both (mlp_kernel above, kept for reference): both register
// sets fit 256 VGPRs, so a SIMD holds an fc1 wave and an fc2 wave whose MFMA streams share the matrix pipe, the GELU's
// integer work runs beside the partner's MFMAs, and the LDS-DMA pieces of a stage are issued by 8 waves instead of 4
// (a piece occupies its wave for ~180 cycles: with 12 pieces per wave the staging alone took 2170 cycles per stage).
// Stage i in the ring = [W1 chunk c_i | W2 chunk c_{i-1}] so both halves are consumed in iteration i.
constexpr int M2OffB1 = MNS * MStage;                // fc1 bias, 1536 floats
constexpr int M2OffGt = M2OffB1 + 1536 * 4;          // GELU table (8704 B)
constexpr int M2OffB2 = M2OffGt + 8704;              // fc2 bias (384 floats)
constexpr int M2OffTrA = M2OffB2 + 1536;             // x transposers of the A waves, 4 x 4 KiB
constexpr int M2OffTrB = M2OffTrA + 4 * XChunk;      // out / residual transposers of the B waves, 4 x 4 KiB
constexpr int M2OffG = M2OffTrB + 4 * XChunk;        // activation hand-off: [2 parities][4 pairs][2 KiB]
constexpr int M2Lds = M2OffG + 2 * 4 * 2048;         // 96 + 16 + 16 + 16 + 16 = 160 KiB
static_assert(M2Lds <= 160 * 1024, "fused MLP: LDS budget");

#ifndef VC_MLP_RD
#define VC_MLP_RD 6          // weight fragments read ahead of the MFMA that uses them
#endif
#ifndef VC_MLP_NA
#define VC_MLP_NA 3           // LDS-DMA pieces of a stage issued by each fc1 wave (each fc2 wave issues 12 - NA): the fc1 role is the longer one
#endif
#ifndef VC_MLP_ISSUE0
#define VC_MLP_ISSUE0 1       // first MFMA behind which a piece of the next stage is issued ...
#define VC_MLP_ISSUE_STEP 3   // ... and the distance to the next one (sweep: 1+3k 228 us, 1+2k 229, 0+k 233, 1+4k 235)
#endif
__global__ __launch_bounds__(512, 2) void mlp2_kernel(__bf16* __restrict__ X, const uint8_t* __restrict__ Wm,
                                                      const float* __restrict__ b1f, const float* __restrict__ b2f,
                                                      int M, int n_chunks, int n_tiles, float eps,
                                                      const uint16_t* __restrict__ gelu_tab) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[M2Lds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool roleB = wave >= 4;
  const int pairw = wave & 3;                       // row block of the pair inside the 128-row tile
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x, bid = blockIdx.x;
  const int my_tiles = bid < n_tiles ? (n_tiles - bid + G - 1) / G : 0;
  const int n = my_tiles * n_chunks;                // fc1 stages; fc2 runs one iteration behind
  if (n <= 0) return;

  float* const b1_l = (float*)(lds + M2OffB1);
  float* const b2_l = (float*)(lds + M2OffB2);
  uint8_t* const gt_l = lds + M2OffGt;
  for (int i = tid; i < n_chunks * 32; i += 512) b1_l[i] = b1f[i];
  for (int i = tid; i < XK; i += 512) b2_l[i] = b2f[i];
  for (int i = tid; i < kGtBytes / 4; i += 512) ((uint32_t*)gt_l)[i] = ((const uint32_t*)gelu_tab)[i];

  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0]);
  // producer: waves 0-3 issue the W1 half of a stage (chunk c1), waves 4-7 the W2 half (chunk c2), six pieces each.
  // One piece: wave-uniform base in SGPRs + one per-lane offset
  // shared by all pieces — with a 64-bit address per piece in VGPRs the compiler kept six pairs alive, spilled them,
  // and every reload in the stage loop waited for vmcnt(0), i.e. for the copy just issued
  const uint32_t lane_off = (uint32_t)lane * 16u;
  constexpr int NA = VC_MLP_NA, NB = 12 - NA;       // pieces per fc1 / fc2 wave and stage (48 pieces: 24 of W1, then 24 of W2)
  auto issue_piece = [&](int c1, int c2, int slot, int i) {
    const int role = wave >> 2, q = wave & 3;
    if (i >= (role ? NB : NA)) return;
    const int id = role ? 4 * NA + q * NB + i : q * NA + i;
    const int half = id >= 24, idx = id - 24 * half;
    const int c = half ? c2 : c1;
    const uint8_t* sbase = Wm + (size_t)c * MStage + (size_t)half * XStage + (size_t)idx * 1024;
    const uint32_t dst = lds0 + (uint32_t)slot * MStage + (uint32_t)half * XStage + (uint32_t)idx * 1024u;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(sbase), "s"(dst)
        : "memory");
  };
  auto issue = [&](int c1, int c2, int slot) {
#pragma unroll
    for (int i = 0; i < (NA > NB ? NA : NB); ++i) issue_piece(c1, c2, slot, i);
  };
  typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
  typedef float v2f __attribute__((ext_vector_type(2)));

  // Iteration i (0 .. n+1) works on ring stage i = [W1 chunk of fc1 stage i | W2 chunk of fc1 stage i-2]:
  //   A waves: MFMAs of fc1 stage i, with the GELU of stage i-1 (integer work + LDS gathers, which overlap with MFMAs
  //            in the same wave) issued in slices between them; the activation tile of stage i-1 is handed over at the end;
  //   B waves: fc2 partial product with the activation tile of stage i-2.
  int chunk = 0, slot = 0, tile = bid;
  auto chunk_of = [&](int j) { return j % n_chunks; };
  auto issue_stage = [&](int j, int slot_) {        // stage j; clamped chunks stage data nobody reads
    const int c1 = chunk_of(min(max(j, 0), n - 1)), c2 = chunk_of(min(max(j - 2, 0), n - 1));
    issue(c1, c2, slot_);
  };
#ifdef VC_MLP_STAMP
  uint32_t st_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t st_t0 = __builtin_amdgcn_s_memtime();
  uint64_t st_prev = st_t0;
#define MS(k_) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_c[k_] += (uint32_t)(t_ - st_prev); st_prev = t_; }
#else
#define MS(k_)
#endif
  issue_stage(0, 0);
  if (!roleB) {
    // =========================== A: x rows, fc1, GELU ======================================================
    v8bf xf[XKS];
    uint8_t* const tp = lds + M2OffTrA + pairw * XChunk;
    const int prow = lane >> 3;
    v16f hprev;                                      // pre-activations of stage i-1
    uint32_t pb[8], gq[8], gres[8], gbad = 0;        // GELU state flowing from slice to slice
    auto gelu_slice = [&](int sl, int parity) {
      if (sl < 4) {                                  // bf16 pairs of the pre-activations (the only float work)
        const int g = sl;
        const v2bf lo = {(__bf16)hprev[4 * g], (__bf16)hprev[4 * g + 1]}, hi = {(__bf16)hprev[4 * g + 2], (__bf16)hprev[4 * g + 3]};
        pb[2 * g] = *(const uint32_t*)&lo;
        pb[2 * g + 1] = *(const uint32_t*)&hi;
        if (sl == 0) gbad = 0;
      }
      if (sl >= 4 && sl < 12) {                      // table index + LDS gather of packed register sl-4
        const int qd = sl - 4;
        const uint32_t b0 = pb[qd] & 0xffffu, b1 = pb[qd] >> 16;
        const uint32_t k0 = (b0 & 0x7fffu) - kGtLo, k1 = (b1 & 0x7fffu) - kGtLo;
        gbad = max(gbad, max(k0, k1));
        gq[2 * (qd & 3)] = *(const uint16_t*)(gt_l + min(k0, kGtN - 1) * 4 + ((b0 >> 15) << 1));
        gq[2 * (qd & 3) + 1] = *(const uint16_t*)(gt_l + min(k1, kGtN - 1) * 4 + ((b1 >> 15) << 1));
      }
      if (sl >= 7 && sl < 15) {                      // three slices after the gather was issued
        const int qd = sl - 7;
        gres[qd] = gq[2 * (qd & 3)] | (gq[2 * (qd & 3) + 1] << 16);
      }
      if (sl == 15) {
        if (__any(gbad >= kGtN)) {                   // a value outside the table somewhere in the wave: float path
#pragma unroll
          for (int qd = 0; qd < 8; ++qd) {
            const v2f_t y = gelu_erf2((v2f_t){__uint_as_float(pb[qd] << 16), __uint_as_float(pb[qd] & 0xffff0000u)});
            const v2bf o = {(__bf16)y[0], (__bf16)y[1]};
            gres[qd] = *(const uint32_t*)&o;
          }
        }
      }
      if (sl == 16) {                                // hand the activation tile to the partner (k-step 0, k-step 1)
        uint8_t* const gb = lds + M2OffG + (parity * 4 + pairw) * 2048 + lane * 16;
        *(v4u*)gb = (v4u){gres[0], gres[1], gres[2], gres[3]};
        *(v4u*)(gb + 1024) = (v4u){gres[4], gres[5], gres[6], gres[7]};
      }
    };
    for (int i = 0; i <= n + 1; ++i) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      MS(0)
      // The six pieces of the next stage are issued BETWEEN this stage's MFMAs, one every four (round 2: issued in a
      // block they held the wave for 300-650 cycles with its matrix pipe idle; in the fc2 waves, where the block sat
      // behind the MFMAs, the copies' whole L2 latency was then waited for at the top of the next stage — ~750 cycles
      // of every stage).  Stages without MFMAs issue them at once.
      const bool do_issue = i <= n;
      const int nc1 = chunk_of(min(max(i + 1, 0), n - 1)), nc2 = chunk_of(min(max(i - 1, 0), n - 1));
      if (do_issue && !(i < n && i > 0)) issue_stage(i + 1, slot ^ 1);
      MS(1)
      if (i < n) {
        if (chunk == 0) {
          const int m0w = tile * 128 + pairw * 32;
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int m = min(m0w + 8 * p + prow, M - 1);
            const uint8_t* src = (const uint8_t*)X + (size_t)m * (XK * 2) + (lane & 7) * 16;
#pragma unroll
            for (int c = 0; c < XK / 64; ++c) *(v4u*)&xf[4 * c + p] = *(const v4u*)(src + c * 128);
          }
          const int fsw = (r >> 1) & 7;
#pragma unroll
          for (int c = 0; c < XK / 64; ++c) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const int row = 8 * p + prow;
              *(v4u*)(tp + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4)) = *(const v4u*)&xf[4 * c + p];
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
              xf[4 * c + k4] = *(const v8bf*)(tp + r * 128 + (((2 * k4 + h) ^ fsw) << 4));
          }
          auto expand = [](uint32_t u) { return (v2f){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; };
          v2f s2 = {0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < XKS; ++ks) {
            const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
            for (int j = 0; j < 4; ++j) s2 += expand(u[j]);
          }
          float sum = s2[0] + s2[1];
          sum += __shfl_xor(sum, 32);
          const float mean = sum * (1.0f / XK);
          const v2f mean2 = {mean, mean};
#pragma unroll
          for (int ks = 0; ks < XKS; ++ks) asm volatile("" : "+v"(xf[ks]));
          v2f q2 = {0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < XKS; ++ks) {
            const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const v2f d = expand(u[j]) - mean2; q2 = __builtin_elementwise_fma(d, d, q2); }
          }
          float q = q2[0] + q2[1];
          q += __shfl_xor(q, 32);
          const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / XK) + eps);
          const v2f rstd2 = {rstd, rstd};
#pragma unroll
          for (int ks = 0; ks < XKS; ++ks) asm volatile("" : "+v"(xf[ks]));
#pragma unroll
          for (int ks = 0; ks < XKS; ++ks) {
            const v4u u = *(const v4u*)&xf[ks];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const v2f y = (expand(u[j]) - mean2) * rstd2;
              xf[ks][2 * j] = (__bf16)y[0];
              xf[ks][2 * j + 1] = (__bf16)y[1];
            }
          }
        }
        MS(2)
        const uint8_t* st = lds + slot * MStage + lane * 16;
        v16f hacc;
        {
          const float* bl = b1_l + chunk * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const v4f bv = *(const v4f*)(bl + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) hacc[4 * g + j] = bv[j];
          }
        }
        {
          constexpr int RD = VC_MLP_RD;
          v8bf wf[XKS];
#pragma unroll
          for (int ks = 0; ks < RD; ++ks) wf[ks] = *(const v8bf*)(st + ks * 1024);
          __builtin_amdgcn_sched_barrier(0);
#ifndef VC_MLP_NO_PRIO_A
          // the fc1 role is the longer one (its GELU slices and x load ride in this phase): it gets the issue priority on
          // its SIMD — 238 -> 230 us per layer; raising the fc2 role instead loses
          __builtin_amdgcn_s_setprio(1);
#endif
          if (i > 0) {
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) {
              hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], hacc, 0, 0, 0);
              if (ks + RD < XKS) wf[ks + RD] = *(const v8bf*)(st + (ks + RD) * 1024);
              gelu_slice(ks, (i - 1) & 1);
              if (ks >= VC_MLP_ISSUE0 && ks < VC_MLP_ISSUE0 + NA * VC_MLP_ISSUE_STEP && (ks - VC_MLP_ISSUE0) % VC_MLP_ISSUE_STEP == 0 && do_issue)
                issue_piece(nc1, nc2, slot ^ 1, (ks - VC_MLP_ISSUE0) / VC_MLP_ISSUE_STEP);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) {
              hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], hacc, 0, 0, 0);
              if (ks + RD < XKS) wf[ks + RD] = *(const v8bf*)(st + (ks + RD) * 1024);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
#ifndef VC_MLP_NO_PRIO_A
        __builtin_amdgcn_s_setprio(0);
#endif
        hprev = hacc;
        MS(3)
        if (++chunk == n_chunks) { chunk = 0; tile += G; }
      } else if (i == n) {
        // the last stage's GELU has no MFMAs to ride under
#pragma unroll
        for (int sl = 0; sl < XKS; ++sl) gelu_slice(sl, (i - 1) & 1);
        MS(4)
      }
      slot ^= 1;
    }
  } else {
    // =========================== B: fc2 accumulators, two stages behind, tile epilogue ==========================
    v16f oacc[12];
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(X, 0, (int)((size_t)M * XK * 2), 0x00020000);
    const int crow = lane >> 2, cch = lane & 3;
    const int rsw = (r >> 2) & 3;
    uint8_t* const tr_out = lds + M2OffTrB + pairw * XChunk;
    uint8_t* const tr_res = tr_out + 2048;
    int pchunk = 0;                                  // chunk consumed in this iteration (that of fc1 stage i-2)
    for (int i = 0; i <= n + 1; ++i) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      MS(0)
      const bool do_issue = i <= n;
      const int nc1 = chunk_of(min(max(i + 1, 0), n - 1)), nc2 = chunk_of(min(max(i - 1, 0), n - 1));
      if (i >= 2) {
        if (pchunk == 0) {
#pragma unroll
          for (int ob = 0; ob < 12; ++ob)
#pragma unroll
            for (int j = 0; j < 16; ++j) oacc[ob][j] = 0.f;
        }
        const uint8_t* gb = lds + M2OffG + ((i & 1) * 4 + pairw) * 2048 + lane * 16;   // parity of stage i-2
        const v8bf g_s0 = *(const v8bf*)gb, g_s1 = *(const v8bf*)(gb + 1024);
        const uint8_t* st2 = lds + slot * MStage + XStage + lane * 16;
        {
          constexpr int RD = VC_MLP_RD;
          v8bf wf[24];
#pragma unroll
          for (int q = 0; q < RD; ++q) wf[q] = *(const v8bf*)(st2 + q * 1024);
#ifdef VC_MLP_PRIO_B
          __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
          for (int q = 0; q < 24; ++q) {
            oacc[q >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q], (q & 1) ? g_s1 : g_s0, oacc[q >> 1], 0, 0, 0);
            if (q + RD < 24) wf[q + RD] = *(const v8bf*)(st2 + (q + RD) * 1024);
#ifndef VC_MLP_BSTEP
#define VC_MLP_BSTEP 2
#endif
            constexpr int BSTEP = NB <= 6 ? VC_MLP_ISSUE_STEP : VC_MLP_BSTEP;
            if (q >= VC_MLP_ISSUE0 && q < VC_MLP_ISSUE0 + NB * BSTEP && (q - VC_MLP_ISSUE0) % BSTEP == 0 && do_issue)
              issue_piece(nc1, nc2, slot ^ 1, (q - VC_MLP_ISSUE0) / BSTEP);   // (see the A waves)
          }
        }
#ifdef VC_MLP_PRIO_B
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef VC_MLP_STAMP
        asm volatile("s_nop 0" :: "v"(oacc[11]));
#endif
        MS(3)
        if (pchunk == n_chunks - 1) {
          const int m0w = tile * 128 + pairw * 32;
#pragma unroll 1
          for (int ob = 0; ob < 12; ++ob) {
            v4u resq[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
              resq[q] = *(const v4u*)(X + (size_t)min(m0w + crow + 16 * q, M - 1) * XK + ob * 32 + cch * 8);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int row = crow + 16 * q;
              *(v4u*)(tr_res + row * 64 + ((cch ^ ((row >> 2) & 3)) << 4)) = resq[q];
            }
            v16f oa;
            switch (ob) {   // dynamic index into the register tiles
#define VC_OB(k) case k: oa = oacc[k]; break;
              VC_OB(0) VC_OB(1) VC_OB(2) VC_OB(3) VC_OB(4) VC_OB(5) VC_OB(6) VC_OB(7) VC_OB(8) VC_OB(9) VC_OB(10)
              default: oa = oacc[11]; break;
#undef VC_OB
            }
            uint32_t ep[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const v4bf rv = *(const v4bf*)(tr_res + r * 64 + ((g ^ rsw) << 4) + 8 * h);
              const v4f bv = *(const v4f*)(b2_l + ob * 32 + 8 * g + 4 * h);
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = oa[4 * g + j] + bv[j] + (float)rv[j];
              const v2bf lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
              ep[2 * g] = *(const uint32_t*)&lo;
              ep[2 * g + 1] = *(const uint32_t*)&hi;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const auto sw = __builtin_amdgcn_permlane32_swap(ep[k], ep[k + 4], false, false);
              ep[k] = sw[0];
              ep[k + 4] = sw[1];
            }
            *(v4u*)(tr_out + r * 64 + (((2 * h) ^ rsw) << 4)) = (v4u){ep[0], ep[1], ep[4], ep[5]};
            *(v4u*)(tr_out + r * 64 + (((2 * h + 1) ^ rsw) << 4)) = (v4u){ep[2], ep[3], ep[6], ep[7]};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int row = crow + 16 * q;
              const v4u o = *(const v4u*)(tr_out + row * 64 + ((cch ^ ((row >> 2) & 3)) << 4));
              __builtin_amdgcn_raw_buffer_store_b128(o, out_rs, (int)(((size_t)(m0w + row) * XK + ob * 32 + cch * 8) * 2), 0, 0);
            }
          }
          tile += G;
        }
        if (++pchunk == n_chunks) pchunk = 0;
        MS(5)
      }
      if (do_issue && i < 2) issue_stage(i + 1, slot ^ 1);   // (no MFMAs to issue them between)
      MS(1)
      slot ^= 1;
    }
  }
#ifdef VC_MLP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  st_c[7] = (uint32_t)(__builtin_amdgcn_s_memtime() - st_t0);
  if (lane < 8) {
    uint32_t v = 0;
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) v = lane == kq ? st_c[kq] : v;
    ((uint32_t*)X)[(bid * 8 + wave) * 8 + lane] = v;   // diagnostic build: overwrites the first rows of x
  }
#endif
}

// MLP weights -> stage order: chunk c = [24 pieces of W1' (features 32c..+32, LayerNorm gamma folded) | 24 pieces of W2
// (piece 2 ob + s: output features 32 ob..+32, hidden 32c + 16s + 8(j>>2) + 4h + (j&3) in element j of lane 32h + r)].
__global__ __launch_bounds__(256) void mlp_prepare_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ W2, const float* __restrict__ b2,
                                                          int n_hid, __bf16* __restrict__ Wm, float* __restrict__ b1f,
                                                          float* __restrict__ b2f) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);   // hidden feature (fc1 part), then output feature (fc2 part)
  if (unit < n_hid) {
    const int c = unit >> 5, rr = unit & 31;
    float dot = 0.f;
    for (int k = lane; k < XK; k += 64) {
      const float w = W1[(size_t)unit * XK + k];
      if (beta) dot += w * beta[k];
      const int ks = k >> 4, hh = (k >> 3) & 1, j = k & 7;
      Wm[(size_t)c * (MStage / 2) + ((size_t)ks * 64 + hh * 32 + rr) * 8 + j] = (__bf16)(gamma ? w * gamma[k] : w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (lane == 0) b1f[unit] = (b1 ? b1[unit] : 0.f) + dot;
  } else if (unit < n_hid + XK) {
    const int f = unit - n_hid, ob = f >> 5, rr = f & 31;
    for (int hd = lane; hd < n_hid; hd += 64) {
      const int c = hd >> 5, e = hd & 31, s = e >> 4, e16 = e & 15;
      // e16 = 8 (j >> 2) + 4 h + (j & 3)
      const int hh = (e16 >> 2) & 1, j = ((e16 >> 3) << 2) | (e16 & 3);
      Wm[(size_t)c * (MStage / 2) + (XStage / 2) + ((size_t)(2 * ob + s) * 64 + hh * 32 + rr) * 8 + j] = (__bf16)W2[(size_t)f * n_hid + hd];
    }
    if (lane == 0) b2f[f] = b2 ? b2[f] : 0.f;
  }
}

}  // namespace

extern "C" {

int vc_linear_bf16(const void* x, const void* weight, const void* bias, const void* residual_or_null, void* out,
                   int rows, int n_out, int k_in, int epilogue, vc_stream_t stream) {
  if (!x || !weight || !bias || !out || rows < 0 || n_out <= 0 || k_in <= 0) return VC_ERR_INVALID_ARG;
  if (epilogue < EPI_BIAS || epilogue > EPI_RESIDUAL) return VC_ERR_INVALID_ARG;
  if ((epilogue == EPI_RESIDUAL) != (residual_or_null != nullptr)) return VC_ERR_INVALID_ARG;
  if (n_out % BN != 0 || k_in % BK != 0) return VC_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)weight) | ((uintptr_t)bias) | ((uintptr_t)residual_or_null) | ((uintptr_t)out)) % 16 != 0)
    return VC_ERR_INVALID_ARG;
  if (rows == 0) return VC_OK;
  hipStream_t s = (hipStream_t)stream;
  const __bf16 *px = (const __bf16*)x, *pw = (const __bf16*)weight, *pb = (const __bf16*)bias,
               *pr = (const __bf16*)residual_or_null;
  __bf16* po = (__bf16*)out;
  static const int tile256 = [] { const char* e = getenv("VITCOLMAP_GEMM_TILE"); return e ? atoi(e) : 256; }();   // developer A/B: 128
  int cus256 = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus256, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus256 <= 0)
      cus256 = 256;
  }
  // wide layers with enough 256 x 256 tiles to occupy at least half the CUs (one workgroup per CU); below that the 128 x 128
  // tile, two workgroups per CU, fills the chip better (a single image of ViT-B is 18-72 large tiles)
  static const int min_fill_pct = [] { const char* e = getenv("VITCOLMAP_GEMM256_MIN_FILL"); return e ? atoi(e) : 50; }();
  if (n_out % G2N == 0 && tile256 == 256 &&
      (long long)((rows + G2M - 1) / G2M) * (n_out / G2N) * 100 >= (long long)cus256 * min_fill_pct) {
    const int tiles_m = (rows + G2M - 1) / G2M, tiles_n = n_out / G2N;
    const long long nt = (long long)tiles_m * tiles_n;
    if (nt > 0x7fffffffLL) return VC_ERR_UNSUPPORTED;
    const size_t smem = (size_t)G256_LDS;
    static vc::PerDeviceOnce configured;
    if (int st = configured.run([] {
          hipError_t r = hipFuncSetAttribute((const void*)gemm256_kernel<EPI_BIAS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          if (r == hipSuccess) r = hipFuncSetAttribute((const void*)gemm256_kernel<EPI_GELU, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          if (r == hipSuccess) r = hipFuncSetAttribute((const void*)gemm256_kernel<EPI_RESIDUAL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          return r;
        }))
      return st;
    const int cus = cus256;
    const dim3 grid((unsigned)(nt < cus ? nt : cus)), block(512);
    switch (epilogue) {
      case EPI_BIAS:
        hipLaunchKernelGGL((gemm256_kernel<EPI_BIAS, false>), grid, block, smem, s, px, pw, pb, pr, po, rows, n_out, k_in, tiles_n, (int)nt, 0, 0, 0, 1, 0, 0, -1);
        break;
      case EPI_GELU:
        hipLaunchKernelGGL((gemm256_kernel<EPI_GELU, false>), grid, block, smem, s, px, pw, pb, pr, po, rows, n_out, k_in, tiles_n, (int)nt, 0, 0, 0, 1, 0, 0, -1);
        break;
      default:
        hipLaunchKernelGGL((gemm256_kernel<EPI_RESIDUAL, false>), grid, block, smem, s, px, pw, pb, pr, po, rows, n_out, k_in, tiles_n, (int)nt, 0, 0, 0, 1, 0, 0, -1);
        break;
    }
    return vc::check_launch();
  }
  const int tiles_m = (rows + BM - 1) / BM, tiles_n = n_out / BN;
  const long long nt = (long long)tiles_m * tiles_n;
  if (nt > 0x7fffffffLL) return VC_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)nt), block(256);
  switch (epilogue) {
    case EPI_BIAS:
      hipLaunchKernelGGL(gemm_kernel<EPI_BIAS>, grid, block, 0, s, px, pw, pb, pr, po, rows, n_out, k_in, tiles_n, (int)nt, 1);
      break;
    case EPI_GELU:
      hipLaunchKernelGGL(gemm_kernel<EPI_GELU>, grid, block, 0, s, px, pw, pb, pr, po, rows, n_out, k_in, tiles_n, (int)nt, 1);
      break;
    default:
      hipLaunchKernelGGL(gemm_kernel<EPI_RESIDUAL>, grid, block, 0, s, px, pw, pb, pr, po, rows, n_out, k_in, tiles_n, (int)nt, 1);
      break;
  }
  return vc::check_launch();
}


int vc_conv_taps_bf16(void* x, const void* weight, const void* bias, void* out, int batch, int height, int width, int c_in,
                      int n_out, int kh, int kw, int dy0, int dx0, int out_parity, int epilogue, vc_stream_t stream) {
  if (!x || !weight || !bias || !out || batch < 0 || height <= 0 || width <= 0 || c_in <= 0 || n_out <= 0) return VC_ERR_INVALID_ARG;
  if (epilogue != EPI_BIAS && epilogue != EPI_GELU) return VC_ERR_INVALID_ARG;
  if (kh <= 0 || kw <= 0 || kh * kw > 16 || dy0 < -8 || dy0 > 8 || dx0 < -8 || dx0 > 8) return VC_ERR_INVALID_ARG;
  if (out_parity < -1 || out_parity > 3) return VC_ERR_INVALID_ARG;
  if (n_out % G2N != 0 || c_in % BK != 0) return VC_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)weight) | ((uintptr_t)bias) | ((uintptr_t)out)) % 16 != 0) return VC_ERR_INVALID_ARG;
  const long long rows = (long long)batch * height * width;
  if (rows == 0) return VC_OK;
  // per-lane offsets are 32-bit: the image batch, its zero row and the largest tap shift must stay below 4 GiB
  if ((rows + 1 + (long long)(kh + 8) * width) * c_in * 2 >= (1LL << 32) || rows > 0x7fffffffLL) return VC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync((char*)x + (size_t)rows * c_in * 2, 0, (size_t)c_in * 2, s) != hipSuccess) return vc::fail(hipGetLastError());
  const int k_total = kh * kw * c_in;
  const int tiles_m = (int)((rows + G2M - 1) / G2M), tiles_n = n_out / G2N;
  const long long nt = (long long)tiles_m * tiles_n;
  if (nt > 0x7fffffffLL) return VC_ERR_UNSUPPORTED;
  static vc::PerDeviceOnce configured;
  if (int st = configured.run([] {
        hipError_t r = hipFuncSetAttribute((const void*)gemm256_kernel<EPI_BIAS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (r == hipSuccess) r = hipFuncSetAttribute((const void*)gemm256_kernel<EPI_GELU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return r;
      }))
    return st;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const dim3 grid((unsigned)(nt < cus ? nt : cus)), block(512);
  const __bf16 *px = (const __bf16*)x, *pw = (const __bf16*)weight, *pb = (const __bf16*)bias;
  __bf16* po = (__bf16*)out;
  if (epilogue == EPI_BIAS)
    hipLaunchKernelGGL((gemm256_kernel<EPI_BIAS, true>), grid, block, (size_t)G256_LDS, s, px, pw, pb, (const __bf16*)nullptr, po, (int)rows,
                       n_out, k_total, tiles_n, (int)nt, height, width, c_in, kw, dy0, dx0, out_parity);
  else
    hipLaunchKernelGGL((gemm256_kernel<EPI_GELU, true>), grid, block, (size_t)G256_LDS, s, px, pw, pb, (const __bf16*)nullptr, po, (int)rows,
                       n_out, k_total, tiles_n, (int)nt, height, width, c_in, kw, dy0, dx0, out_parity);
  return vc::check_launch();
}


size_t vc_linear_xs_weight_bytes(int n_out, int k_in) {
  if (n_out <= 0 || k_in != XK || n_out % 32 != 0) return 0;
  return (size_t)n_out * XK * 2;
}

int vc_linear_xs_prepare(const float* weight, const float* bias_or_null, const float* ln_gamma_or_null,
                         const float* ln_beta_or_null, int n_out, int k_in, void* weight_tiled, float* bias_folded,
                         vc_stream_t stream) {
  if (!weight || !weight_tiled || !bias_folded || n_out <= 0) return VC_ERR_INVALID_ARG;
  if (k_in != XK || n_out % 32 != 0 || n_out > XMaxN) return VC_ERR_UNSUPPORTED;
  if ((ln_gamma_or_null == nullptr) != (ln_beta_or_null == nullptr)) return VC_ERR_INVALID_ARG;
  hipLaunchKernelGGL(xs_prepare_kernel, dim3((n_out + 3) / 4), dim3(256), 0, (hipStream_t)stream, weight, bias_or_null,
                     ln_gamma_or_null, ln_beta_or_null, n_out, (__bf16*)weight_tiled, bias_folded);
  return vc::check_launch();
}

size_t vc_gelu_table_bytes(void) { return (size_t)kGtBytes; }

int vc_gelu_table_bf16(void* table, vc_stream_t stream) {
  if (!table || ((uintptr_t)table) % 16 != 0) return VC_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gelu_table_kernel, dim3((kGtN * 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint16_t*)table);
  return vc::check_launch();
}

int vc_linear_xs_bf16(const void* x, const void* weight_tiled, const float* bias_folded, const void* residual_or_null,
                      void* out, int rows, int n_out, int k_in, int epilogue, int fuse_layernorm, float ln_eps,
                      const void* gelu_table_or_null, vc_stream_t stream) {
  if (!x || !weight_tiled || !bias_folded || !out || rows < 0 || n_out <= 0) return VC_ERR_INVALID_ARG;
  if (epilogue < EPI_BIAS || epilogue > EPI_RESIDUAL) return VC_ERR_INVALID_ARG;
  if ((epilogue == EPI_RESIDUAL) != (residual_or_null != nullptr)) return VC_ERR_INVALID_ARG;
  if (k_in != XK || n_out % 32 != 0 || n_out > XMaxN) return VC_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)weight_tiled) | ((uintptr_t)bias_folded) | ((uintptr_t)residual_or_null) | ((uintptr_t)out)) % 16 != 0)
    return VC_ERR_INVALID_ARG;
  if (rows == 0) return VC_OK;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return vc::fail(hipErrorInvalidDevice);
  const int n_nb = n_out / 32;
  const long long stages = (long long)((rows + XRows - 1) / XRows) * n_nb;
  if (stages > 0x7fffffffLL) return VC_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(stages < cus ? stages : cus)), block(512);
  hipStream_t s = (hipStream_t)stream;
  const __bf16 *px = (const __bf16*)x, *pr = (const __bf16*)residual_or_null;
  const uint8_t* pw = (const uint8_t*)weight_tiled;
  __bf16* po = (__bf16*)out;
  const uint16_t* gt = (const uint16_t*)gelu_table_or_null;
  if (gt && (((uintptr_t)gt) % 16 != 0)) return VC_ERR_INVALID_ARG;
  // the table shares the 16 KiB staging area with the bias
  const bool use_gt = gt && epilogue == EPI_GELU && ((n_out * 4 + 15) & ~15) + kGtBytes <= XMaxN * 4;
#define VC_XS_LAUNCH(E, L, G) \
  hipLaunchKernelGGL((xs_kernel<E, L, G>), grid, block, 0, s, px, pw, bias_folded, pr, po, rows, n_out, n_nb, (int)stages, ln_eps, gt)
  const bool ln = fuse_layernorm != 0;
  if (epilogue == EPI_BIAS) { if (ln) VC_XS_LAUNCH(EPI_BIAS, true, false); else VC_XS_LAUNCH(EPI_BIAS, false, false); }
  else if (epilogue == EPI_GELU) {
    if (use_gt) { if (ln) VC_XS_LAUNCH(EPI_GELU, true, true); else VC_XS_LAUNCH(EPI_GELU, false, true); }
    else { if (ln) VC_XS_LAUNCH(EPI_GELU, true, false); else VC_XS_LAUNCH(EPI_GELU, false, false); }
  }
  else { if (ln) VC_XS_LAUNCH(EPI_RESIDUAL, true, false); else VC_XS_LAUNCH(EPI_RESIDUAL, false, false); }
#undef VC_XS_LAUNCH
  return vc::check_launch();
}


int vc_patch_embed_bf16(const void* patches, const void* weight, const void* bias, const void* pos_embed, void* out,
                        int n_images, int tokens, int n_out, int k_in, vc_stream_t stream) {
  if (!patches || !weight || !bias || !pos_embed || !out || n_images < 0 || tokens <= 0 || n_out <= 0 || k_in <= 0)
    return VC_ERR_INVALID_ARG;
  if (n_out % BN != 0 || k_in % BK != 0) return VC_ERR_UNSUPPORTED;
  if ((((uintptr_t)patches) | ((uintptr_t)weight) | ((uintptr_t)bias) | ((uintptr_t)pos_embed) | ((uintptr_t)out)) % 16 != 0)
    return VC_ERR_INVALID_ARG;
  if (n_images == 0) return VC_OK;
  const long long rows = (long long)n_images * tokens;
  const long long nt = ((rows + BM - 1) / BM) * (n_out / BN);
  if (rows > 0x7fffffffLL || nt > 0x7fffffffLL) return VC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gemm_kernel<EPI_PATCH>, dim3((unsigned)nt), dim3(256), 0, (hipStream_t)stream, (const __bf16*)patches,
                     (const __bf16*)weight, (const __bf16*)bias, (const __bf16*)pos_embed, (__bf16*)out, (int)rows, n_out,
                     k_in, n_out / BN, (int)nt, tokens);
  return vc::check_launch();
}


size_t vc_mlp_weight_bytes(int n_hidden, int dim) {
  if (dim != XK || n_hidden <= 0 || n_hidden % 32 != 0 || n_hidden > 2048) return 0;
  return (size_t)(n_hidden / 32) * MStage;
}

int vc_mlp_prepare(const float* w1, const float* b1_or_null, const float* ln_gamma_or_null, const float* ln_beta_or_null,
                   const float* w2, const float* b2_or_null, int n_hidden, int dim, void* weights_tiled, float* b1_folded,
                   float* b2_out, vc_stream_t stream) {
  if (!w1 || !w2 || !weights_tiled || !b1_folded || !b2_out) return VC_ERR_INVALID_ARG;
  if (vc_mlp_weight_bytes(n_hidden, dim) == 0) return VC_ERR_UNSUPPORTED;
  if ((ln_gamma_or_null == nullptr) != (ln_beta_or_null == nullptr)) return VC_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mlp_prepare_kernel, dim3((n_hidden + XK + 3) / 4), dim3(256), 0, (hipStream_t)stream, w1, b1_or_null,
                     ln_gamma_or_null, ln_beta_or_null, w2, b2_or_null, n_hidden, (__bf16*)weights_tiled, b1_folded, b2_out);
  return vc::check_launch();
}

int vc_mlp_bf16(void* x_inout, const void* weights_tiled, const float* b1_folded, const float* b2, const void* gelu_table,
                int rows, int n_hidden, int dim, float ln_eps, vc_stream_t stream) {
  if (!x_inout || !weights_tiled || !b1_folded || !b2 || !gelu_table || rows < 0) return VC_ERR_INVALID_ARG;
  if (vc_mlp_weight_bytes(n_hidden, dim) == 0) return VC_ERR_UNSUPPORTED;
  if ((((uintptr_t)x_inout) | ((uintptr_t)weights_tiled) | ((uintptr_t)b1_folded) | ((uintptr_t)b2) | ((uintptr_t)gelu_table)) % 16 != 0)
    return VC_ERR_INVALID_ARG;
  if (rows == 0) return VC_OK;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return vc::fail(hipErrorInvalidDevice);
  const int n_tiles = (rows + 127) / 128;
  const dim3 grid((unsigned)(n_tiles < cus ? n_tiles : cus));
  static const bool single = [] { const char* e = getenv("VITCOLMAP_MLP_SINGLE"); return e && atoi(e) != 0; }();   // developer A/B switch
  if (single || n_hidden > 1536)
    hipLaunchKernelGGL(mlp_kernel, grid, dim3(256), 0, (hipStream_t)stream, (__bf16*)x_inout, (const uint8_t*)weights_tiled, b1_folded,
                       b2, rows, n_hidden / 32, n_tiles, ln_eps, (const uint16_t*)gelu_table);
  else
    hipLaunchKernelGGL(mlp2_kernel, grid, dim3(512), 0, (hipStream_t)stream, (__bf16*)x_inout, (const uint8_t*)weights_tiled, b1_folded,
                       b2, rows, n_hidden / 32, n_tiles, ln_eps, (const uint16_t*)gelu_table);
  return vc::check_launch();
}

}  // extern "C"
