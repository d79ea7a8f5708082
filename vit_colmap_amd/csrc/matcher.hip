// Exhaustive descriptor matcher for gfx950 (MI355X): int8 MFMA similarity tiles with the
// per-row / per-column top-2 searches, the angle + ratio tests and the cross check fused
// into one kernel per image pair.
//
// Replaces the per-pair arithmetic of pycolmap.match_exhaustive (reference call site
// vit_colmap/pipeline/run_pipeline.py:351-363, options vit_colmap/utils/config.py:64-96).
// Specification: oracle/matcher_oracle.py (bit-exact target).
//
// Data layout
//   prepared image = n_tiles x KS fragments of 1 KiB + n_tiles*32 int32 row sums, where
//   fragment (tile, kk) holds, for lane l = 32*h + c, the 16 bytes [32*kk + 16*h, +16) of
//   descriptor row 32*tile + c, each byte XOR 0x80 (uint8 -> biased int8).  That is exactly
//   the A/B operand of v_mfma_i32_32x32x32_i8, so a fragment is one coalesced 1 KiB
//   global_load_lds (B side, shared by the workgroup through LDS) or one 16 B/lane register
//   load (A side, kept in VGPRs for a whole pass).
//   s(i,j) = sum (a-128)(b-128) + 128*(rowsum_a[i] + rowsum_b[j]) - 16384*D   (exact, int32);
//   the correction is folded into the accumulator's initial value.
//
// Top-2 without index registers: a similarity fits 26 bits (255^2 * 1024 < 2^26), so
//   key = (s << 6) | (63 - t)   with t = column-tile number (row search) or the lane's local
//   row number (column search) orders by (s desc, index asc) under one unsigned max, and
//   best' = max(best, key), second' = med3(best, second, key) is the whole update.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

#ifndef VC_WAVES
#define VC_WAVES 8
#endif
constexpr int kWaves = VC_WAVES;             // waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU)
constexpr int kThreads = kWaves * 64;
constexpr int kTile = 32;                    // MFMA tile edge
static_assert(VC_MAX_KEYPOINTS <= 64 * 32, "the row search packs the column-tile number into 6 bits");
constexpr int kFragBytes = 1024;             // 64 lanes x 16 B

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// 32-row tiles per prepared image, rounded up to a whole pass of the pair kernel (8 waves x 2
// row tiles) so that every tile index a workgroup touches exists; the extra tiles hold
// neutral rows (all-zero descriptors: similarity 0 with everything).
__host__ __device__ inline int tiles_of(int n_max) { return ceil_div(ceil_div(n_max, kTile), 16) * 16; }
__host__ __device__ inline int ksteps_of(int d) { return ceil_div(d, 32); }
// K steps in the "head" of a descriptor for the exact early-out of pair2_kernel (== ks: no early-out for that length)
#ifndef VC2_KH12
#define VC2_KH12 4   // head k-steps of a 384-byte descriptor (a multiple of the window below)
#endif
#ifndef VC2_NB12
#define VC2_NB12 4   // B window registers (fragments) at KS = 12
#endif
__host__ __device__ constexpr int head_steps_of(int ks) { return ks == 12 ? VC2_KH12 : (ks == 8 ? 4 : ks); }   // multiples of BWindow::NB
// prepared image = fragments | row sums int32 [n_pad] | packed head / tail row sums int32 [n_pad] | per-tile tail norm bounds int32 [n_tiles]
//   packed word   = (head << 16) | tail: sums of the row's bytes over the first head_steps_of(ks) * 32 dimensions and over the rest
//                   (field widths: packed_sums_fit() below, asserted for every length the early-out kernels are built for)
//   tail norm     = max over the tile's 32 rows of ceil(sqrt(sum of squared bytes over the remaining dimensions))
__host__ __device__ constexpr bool packed_sums_fit(int ks) {   // head < 2^15 (the word stays positive), tail < 2^16
  return head_steps_of(ks) == ks || (255 * 32 * head_steps_of(ks) < (1 << 15) && 255 * 32 * (ks - head_steps_of(ks)) < (1 << 16));
}
static_assert(packed_sums_fit(8) && packed_sums_fit(12), "head / tail row sums must fit the packed word (VC2_KH12 is a build parameter)");
__host__ __device__ inline int packed_head(int word) { return word >> 16; }
__host__ __device__ inline int packed_tail(int word) { return word & 0xffff; }
__host__ __device__ inline size_t image_bytes(int n_tiles, int ks) {
  return (size_t)n_tiles * ks * kFragBytes + (size_t)n_tiles * kTile * sizeof(int32_t) * 2 + (size_t)n_tiles * sizeof(int32_t);
}

// ---------------------------------------------------------------------------------------
// theta / accept — identical arithmetic to oracle/matcher_oracle.c
// ---------------------------------------------------------------------------------------
__device__ inline float theta_dev(int s) {
  float x = (float)s * (1.0f / (512.0f * 512.0f));
  x = x > 1.0f ? 1.0f : x;
  return (float)acos((double)x);
}

__device__ inline bool accept_dev(int best, int second, float max_ratio, float max_distance) {
  if (best <= 0) return false;
  const float tb = theta_dev(best);
  if (tb > max_distance) return false;
  const float ts = theta_dev(second);
  if (tb >= max_ratio * ts) return false;
  return true;
}

// theta(s) for every s in [0, 512^2], generated at build time by gen_theta_table.c (same expression as
// theta_dev; compared entry by entry on the device and against the oracle in tests/test_matcher_gpu.py).
// With it the acceptance tests of a row are two L2-resident loads, a multiply and two compares: no
// double-precision acos in the pair kernel, hence no register pressure from it around the pair loop.
__device__ inline unsigned int umin(unsigned int a, unsigned int b);
__device__ const uint32_t kThetaBits[512 * 512 + 1] = {
#include "theta_table.inc"
};
__device__ __forceinline__ float theta_tab(int s) {
  return __uint_as_float(kThetaBits[umin((u32)s, 512u * 512u)]);   // (unsigned clamp: no index is ever out of the table)
}
// accept_dev with table angles.  `s_low` (relevance_threshold): a best <= s_low is rejected whatever the
// runner-up is, so those rows never touch the table (on non-matching image pairs that is every row).
__device__ __forceinline__ bool accept_tab(int best, int second, float max_ratio, float max_distance, int s_low) {
  if (best <= 0 || best <= s_low) return false;
  const float tb = theta_tab(best);
  const float ts = theta_tab(second);
  if (tb > max_distance) return false;
  if (tb >= max_ratio * ts) return false;
  return true;
}

// ---------------------------------------------------------------------------------------
// cross-lane helpers (wave64; reductions run inside each 32-lane half)
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ inline u32 dpp_mov(u32 v) {
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ inline u32 umax(u32 a, u32 b) { return a > b ? a : b; }
__device__ inline u32 umin(u32 a, u32 b) { return a < b ? a : b; }
__device__ inline u32 umed3(u32 a, u32 b, u32 c) {
  // median of three: with a >= b it is the new runner-up after seeing c
  return umax(umin(a, b), umin(umax(a, b), c));
}
// max over the 32 lanes that share lane>>5; every lane of the half receives it
__device__ inline u32 half_max(u32 v) {
  v = umax(v, dpp_mov<0x128>(v));  // row_ror:8
  v = umax(v, dpp_mov<0x124>(v));  // row_ror:4
  v = umax(v, dpp_mov<0x122>(v));  // row_ror:2
  v = umax(v, dpp_mov<0x121>(v));  // row_ror:1
  v = umax(v, (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));  // lane ^ 16
  return v;
}

// ---------------------------------------------------------------------------------------
// prepare: uint8 [n_images][n_max][d] -> fragment-major biased int8 + row sums
// ---------------------------------------------------------------------------------------
__global__ void prepare_kernel(const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts,
                               int n_max, int d, int n_tiles, int ks, uint8_t* __restrict__ prepared) {
  const int img = blockIdx.y;
  const int tile = blockIdx.x;
  const int count = counts ? min(max(counts[img], 0), n_max) : n_max;
  const uint8_t* src = desc + (size_t)img * n_max * d;
  uint8_t* dst_img = prepared + (size_t)img * image_bytes(n_tiles, ks);
  uint8_t* dst = dst_img + (size_t)tile * ks * kFragBytes;
  int32_t* rowsum = (int32_t*)(dst_img + (size_t)n_tiles * ks * kFragBytes) + tile * kTile;
  const bool vec_ok = (d % 16 == 0) && (((uintptr_t)src) % 16 == 0);

  for (int chunk = threadIdx.x; chunk < ks * 64; chunk += blockDim.x) {
    const int kk = chunk >> 6, lane = chunk & 63;
    const int c = lane & 31, h = lane >> 5;
    const int row = tile * kTile + c;
    const int k0 = kk * 32 + 16 * h;
    uint4 out;
    if (row < count && vec_ok && k0 + 16 <= d) {
      uint4 v = *(const uint4*)(src + (size_t)row * d + k0);
      out = make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u);
    } else {
      uint8_t b[16];
      for (int t = 0; t < 16; ++t) {
        const int k = k0 + t;
        // padded k contributes 0 (int8 0); a padded ROW is uint8 0 = int8 -128 on real k
        b[t] = (k >= d) ? 0x00 : (row < count ? (uint8_t)(src[(size_t)row * d + k] ^ 0x80) : 0x80);
      }
      out = *(uint4*)b;
    }
    *(uint4*)(dst + (size_t)chunk * 16) = out;
  }
  int32_t* rowsum_head = (int32_t*)(dst_img + (size_t)n_tiles * ks * kFragBytes) + (size_t)n_tiles * kTile + tile * kTile;
  int32_t* tailnorm = (int32_t*)(dst_img + (size_t)n_tiles * ks * kFragBytes) + (size_t)2 * n_tiles * kTile + tile;
  __shared__ int s_tn;
  if (threadIdx.x == 0) s_tn = 0;
  __syncthreads();
  const int d_head = min(d, head_steps_of(ks) * 32);
  // Row sums: eight lanes per row, 16 bytes per load, byte sums and sums of squares by v_dot4 (one lane walking a row byte by
  // byte made this kernel 42 us per batch of 50 x 512 x 384: 12 k dependent byte loads per lane).  The vector path needs whole
  // 16-byte chunks on either side of the head / tail boundary; otherwise one lane of the group takes the byte loop.
  const bool sum_vec = vec_ok && (d_head % 16 == 0);
  for (int idx = threadIdx.x; idx < kTile * 8; idx += blockDim.x) {
    const int c = idx >> 3, t8 = idx & 7;
    const int row = tile * kTile + c;
    u32 sum = 0, head = 0, ss = 0;
    if (row < count) {
      if (sum_vec) {
        for (int j = t8; j < d / 16; j += 8) {
          const uint4 v = *(const uint4*)(src + (size_t)row * d + 16 * j);
          const u32 w[4] = {v.x, v.y, v.z, v.w};
          u32 cs = 0, cq = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            cs = __builtin_amdgcn_udot4(w[q], 0x01010101u, cs, false);
            cq = __builtin_amdgcn_udot4(w[q], w[q], cq, false);
          }
          sum += cs;
          if (16 * j < d_head) head += cs; else ss += cq;
        }
      } else if (t8 == 0) {
        for (int k = 0; k < d; ++k) {
          const u32 v = src[(size_t)row * d + k];
          sum += v;
          if (k < d_head) head += v; else ss += v * v;
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {          // the eight lanes of a row are neighbours in one wave
      sum += (u32)__shfl_xor((int)sum, o);
      head += (u32)__shfl_xor((int)head, o);
      ss += (u32)__shfl_xor((int)ss, o);
    }
    if (t8 == 0) {
      rowsum[c] = (int32_t)sum;
      // the early-out kernels read ONE word per row: head sum and tail sum packed (packed_sums_fit)
      rowsum_head[c] = head_steps_of(ks) < ks ? (int32_t)((head << 16) | (sum - head)) : (int32_t)head;
      int tn = (int)ceil(sqrt((double)ss));
      while ((long long)tn * tn < (long long)ss) ++tn;   // an upper bound of the Euclidean norm of the tail, whatever sqrt rounded to
      atomicMax(&s_tn, tn);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *tailnorm = s_tn;
}

// ---------------------------------------------------------------------------------------
// The pair kernel
// ---------------------------------------------------------------------------------------
// One workgroup (8 waves, 2 per SIMD) per image pair (a, b).
//   * wave w keeps RT 32-row tiles of image a as MFMA A-fragments in registers for a pass
//     (RT = 2: 8 x 64 = 512 rows per pass, so a 512-keypoint image needs ONE pass over b);
//   * image b streams through an LDS ring of NS tile slots (12 KiB per slot at D = 384),
//     one wave issuing the 1 KiB global_load_lds pieces of a whole tile, PF = NS-1 tiles
//     ahead; the only in-loop synchronisation is one raw s_barrier per tile and a
//     vmcnt(0) in the wave whose tile is due (no wave has more than one tile in flight);
//   * per-row top-2 lives in registers as packed keys, per-column top-2 is merged across
//     waves with two LDS atomics per lane and tile (order independent, see col_merge).
//
// Dynamic LDS: [ring NS x KS KiB][colbest u64 x n_pad][colsecond u32 x n_pad]
//              [cterm i32 x n_pad][m21 i32 x n_pad][rbest, rsecond, ridx i32 x n_pad]
//              [crow6 i32 x 8 waves x 64 rows][row-reduce scratch 8 waves x 4224 B][8 ints]
constexpr int kLdsBytes = 160 * 1024;
// end-of-pass row reduction: per wave 16 rows x (32 + 1 pad) entries of {best key, second key}
constexpr int kRowScratchEntries = 16 * 33;
constexpr int kRowScratchBytes = kRowScratchEntries * 8;
constexpr int kMaxSlots = 8;  // PF <= 7 keeps "one tile in flight per wave" true for 8 waves
constexpr int kPairWindow = 32;   // pair2_kernel: pairs-with-work described in LDS at a time (half a wave builds the list)

__host__ __device__ inline size_t lds_fixed_bytes(int n_pad) {
  return (size_t)n_pad * (8 + 4 + 4 + 4 + 12) + kWaves * 64 * 4 + kWaves * kRowScratchBytes + 64;
}
// pair2_kernel: column records of 16 bytes (best key, second, pad), column terms, m21 + row results, three row-term arrays per
// wave (full, head, full - head), row-reduce scratch, head column terms + tile bounds of b, the copy of the NEXT pair's
// column sums / tile bounds (LDS-DMA), the pair window
__host__ __device__ inline size_t lds_fixed_bytes2(int n_pad) {
  return (size_t)n_pad * (16 + 4 + 4 + 12) + 3 * (kWaves * 64 * 4) + kWaves * kRowScratchBytes + 64 +
         (size_t)n_pad * 4 + (size_t)(n_pad / kTile) * 4 + ((size_t)n_pad * 4 + (size_t)(n_pad / kTile) * 4) +
         kPairWindow * 16 + 16;
}
// number of ring slots for this problem size (0 = does not fit)
inline int plan_slots(int ks, int n_pad) {
  if (kWaves == 4) {
    // four-wave workgroups: stay within half the LDS when that still leaves a useful ring, so that two
    // workgroups (two image pairs) share a CU and one's prologue / finalisation runs under the other's tiles
    const long half = (long)kLdsBytes / 2 - (long)lds_fixed_bytes(n_pad);
    const long nh = half / ((long)ks * kFragBytes);
    if (nh >= 3) return (int)(nh > kMaxSlots ? kMaxSlots : nh);
  }
  const long avail = (long)kLdsBytes - (long)lds_fixed_bytes(n_pad);
  long ns = avail / ((long)ks * kFragBytes);
  if (ns > kMaxSlots) ns = kMaxSlots;
  return ns >= 2 ? (int)ns : 0;
}

#ifdef VC_EXP_STAMP
// diagnostic build only: shader-clock stamp (the wait keeps s_memtime ordered with LDS traffic)
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif

__device__ inline void wg_barrier() {
  // LDS writes/atomics of this wave are complete before it arrives; LDS-DMA stays in flight
#ifdef VC_EXP_NO_BARRIER
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// One 1 KiB LDS-DMA piece: 64 lanes x 16 B, global source per lane, LDS destination =
// wave-uniform byte address `lds_dst` + lane*16.  Issued from inline asm on purpose: hipcc
// treats the builtin form as an LDS write that may alias every later ds_read and inserts
// s_waitcnt vmcnt(0) before the next LDS read of the issuing wave, which drains the prefetch
// it has just started.  Hidden in asm, the copy is invisible to the compiler's counters; the
// kernel waits for it by hand (vmcnt(0) in the issuing wave, then the workgroup barrier).
__device__ __forceinline__ void glds16(const void* gsrc, u32 lds_dst_any) {
  // The destination is wave-uniform by construction, but under SGPR pressure the compiler may have computed it
  // with vector instructions; an explicit v_readfirstlane (which it cannot fold away) puts it where M0 can take it.
  // (The value is laundered through an empty asm so that the builtin is not folded away as "already uniform"; the
  // v_readfirstlane itself is the compiler's, which knows the wait state it needs after the VALU write of its
  // source — one written in asm right behind a v_mov read a stale register.)
  asm volatile("" : "+v"(lds_dst_any));
  const u32 lds_dst = (u32)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
  u32 keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

inline int plan_slots2(int ks, int n_pad) {
  long ns = ((long)kLdsBytes - (long)lds_fixed_bytes2(n_pad)) / ((long)ks * kFragBytes);
  if (ns > kMaxSlots) ns = kMaxSlots;
  return ns >= 3 ? (int)ns : 0;
}

// One 256-byte LDS-DMA piece (64 lanes x 4 B; inactive lanes copy nothing): LDS destination = `lds_dst` + lane*4.
__device__ __forceinline__ void glds4(const void* gsrc, u32 lds_dst_any) {
  asm volatile("" : "+v"(lds_dst_any));
  const u32 lds_dst = (u32)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
  u32 keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ u32 lds_addr(const void* p) {
  return (u32)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// Producer side of the ring.  The KS pieces of a tile are dealt to NP producer waves, M pieces
// each (KS = 12: waves 0..5 issue 2 pieces per tile), so no wave is held up for a whole tile's
// worth of LDS-DMA issue.  Every producer wave has the same number of pieces in flight per tile,
// which makes "my pieces of tile t have landed" a counted wait: vmcnt(M * tiles issued after t).
// PW: the waves that may produce (pair2_kernel: the four EARLY waves only — they wait at the barrier for the late half
// anyway, so neither the issue of the pieces nor the wait for them to land sits on the late waves' critical path).
template <int KS, int PW = kWaves>
struct Producer {
  static constexpr int M = (KS + PW - 1) / PW;
  static constexpr int NP = KS / M;
  static_assert(NP * M == KS && NP <= PW, "pieces must divide evenly over the producer waves");
};

template <int KS, int PW = kWaves>
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ tile_src, u32 slot_lds, int wave, int lane) {
  constexpr int M = Producer<KS, PW>::M;
#ifdef VC_EXP_NO_STAGE
  return;
#endif
  if (wave < Producer<KS, PW>::NP) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int kk = wave * M + m;
      glds16(tile_src + kk * kFragBytes + lane * 16, slot_lds + kk * kFragBytes);
    }
  }
}

// Wait until this wave's pieces of the oldest tile in flight have landed; `younger` tiles
// (0..7) were issued after it.  s_waitcnt takes an immediate, hence the switch.
template <int KS, int PW = kWaves>
__device__ __forceinline__ void wait_tile(int wave, int younger) {
  constexpr int M = Producer<KS, PW>::M;
#ifdef VC_EXP_NO_WAIT
  return;   // timing experiment only (results are wrong)
#endif
  if (wave < Producer<KS, PW>::NP) {
#define VC_W(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) * M) : "memory")
    // binary decision tree: three scalar branches per call instead of a chain of eight
    if (younger < 4) {
      if (younger < 2) { if (younger == 0) VC_W(0); else VC_W(1); }
      else             { if (younger == 2) VC_W(2); else VC_W(3); }
    } else {
      if (younger < 6) { if (younger == 4) VC_W(4); else VC_W(5); }
      else             { if (younger == 6) VC_W(6); else VC_W(7); }
    }
#undef VC_W
  }
}

// MFMA phase: similarity tiles of the wave's RT row tiles against one column tile.
// acc = sum (a-128)(b-128) over k (C operand 0); the bias terms are added in the epilogue.
// `mid()` runs after the first fragment group's MFMAs have been issued: the matrix pipe is busy for
// the next ~128 cycles, which hides the issue cost of the LDS-DMA pieces placed there (in-kernel
// stamps showed the producer waves' staging on the critical path when it sat right after the barrier).
template <int KS, int RT, typename Mid>
__device__ __forceinline__ void mfma_phase(const v4i (&afrag)[RT][KS], v16i (&acc)[RT], const uint8_t* slot,
                                           int lane, Mid mid) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0;
  // K loop in groups of G fragments, the next group's LDS reads issued ahead of this group's
  // MFMAs; sched_barrier keeps the compiler from hoisting every read to the top (which costs
  // 4*KS registers and spills).
  constexpr int G = (RT * KS >= 24) ? 2 : (KS < 4 ? KS : 4);  // smaller groups where registers are tight
  constexpr int NG = KS / G;
  static_assert(KS % G == 0, "KS must be a multiple of the fragment group");
  const uint8_t* src = slot + lane * 16;
#ifndef VC_NO_SETPRIO
  // the wave that feeds the matrix pipe wins issue arbitration against its SIMD partner's epilogue
  __builtin_amdgcn_s_setprio(1);
#endif
  v4i bf[2][G];
#pragma unroll
#ifdef VC_EXP_NO_LDSREAD
  for (int i = 0; i < G; ++i) { bf[0][i] = afrag[0][i]; bf[1][i] = afrag[0][(i + 1) % KS]; }
  asm volatile("" :: "v"(src));
#else
  for (int i = 0; i < G; ++i) bf[0][i] = *(const v4i*)(src + i * kFragBytes);
#endif
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#ifndef VC_EXP_NO_LDSREAD
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < G; ++i) bf[(g + 1) & 1][i] = *(const v4i*)(src + ((g + 1) * G + i) * kFragBytes);
    }
#endif
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#ifdef VC_EXP_NO_MFMA
        acc[rt][i] += afrag[rt][g * G + i][0] ^ bf[g & 1][i][1];
#else
        acc[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[rt][g * G + i], bf[g & 1][i], acc[rt], 0, 0, 0);
#endif
      }
    __builtin_amdgcn_sched_barrier(0);
    if (g == 0) mid();
  }
#ifndef VC_NO_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// Epilogue phase: top-2 updates for one column tile.
//
// Exact int32 arithmetic on biased bytes: with a' = a-128, b' = b-128,
//   s = sum a'b' + 128*ra + 128*rb - 16384*D = acc + rterm[i] + ct[j]          (one v_add3_u32)
//   ct[j] = 128*rb[j] + 32640*D (per lane),  rterm[i] = 128*ra[i] - 49024*D (per register, LDS).
//
// Irrelevant similarities.  The launch passes s_low such that a similarity s <= s_low can never
// influence the match list (see relevance_thresholds() for the proof sketch): it is too small to
// be an accepted best, and as a runner-up it can never fail the ratio test of an acceptable best.
// The searches may therefore ignore any such element.  Per 32x32 tile: one max over the lane's
// 16 similarities, and only if some lane of the wave holds a relevant one, the actual updates:
//   row search    key = s << 6 | (63 - column tile)          v_lshl_or, v_med3, v_max
//   column search key = s << 6 | (63 - local row code)       v_lshl_or, v_med3, v_max
// (s < 2^26: 255^2 * 1024 < 2^26.)  s_low = -1 disables the shortcut (every s >= 0 is relevant),
// which is what the one-way API uses because it reports best / second for every row.
//
// The lane's column result goes to LDS with two atomics (col_merge), valid in any order:
//   best64  = max over keys (s << 32 | ~row): highest s, lowest row on ties
//   second  = max over { every lane's second } U { every best that is not THE best }: a best that
//             loses its max contributes itself, one that wins contributes the value it displaced.
template <int RT, bool FUSED>
__device__ __forceinline__ void epilogue_phase(const v16i (&acc)[RT], u32 (&rbest)[RT][16], u32 (&rsec)[RT][16],
                                               const int* rterm_wave, const int* cterm,
                                               unsigned long long* colbest, u32* colsecond, int jt,
                                               int c, int h, u32 row_base, int s_low, bool (&dense)[RT]) {
#ifdef VC_EXP_NO_EPILOGUE
  {
    int keep = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; r += 4) keep ^= acc[rt][r];
    asm volatile("" :: "v"(keep));
    return;
  }
#endif
  const int ct = cterm[jt * kTile + c];
  const u32 jcode = 63u - (u32)jt;
  u32 cb = 0, cs2 = 0;
  bool any_hit = false;
  // Two regimes, chosen per row tile from what the previous column tile looked like (wave-uniform):
  //   sparse: pass 1 computes the lane's largest similarity only (24 VALU); if some lane is
  //           relevant, pass 2 recomputes the similarities and updates (and switches to dense);
  //   dense : one pass that updates and tracks the maximum (SIFT-like descriptors sit at
  //           cos ~0.64, right at the threshold: there almost every tile is relevant).
  // Pass 1 runs for all row tiles first, with every LDS read of the row terms issued up front:
  // measured with in-kernel stamps, the epilogue was LDS-latency-bound (890 cycles per tile for
  // ~70 instructions) when each row tile waited for its own reads.
  bool update[RT];
  bool all_dense = true;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { update[rt] = dense[rt]; all_dense = all_dense && dense[rt]; }
  if (!all_dense) {
    v4i cr[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) cr[rt][q] = *(const v4i*)(rterm_wave + rt * kTile + 8 * q + 4 * h);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int m = -1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int v0 = acc[rt][4 * q + 0] + cr[rt][q][0] + ct, v1 = acc[rt][4 * q + 1] + cr[rt][q][1] + ct;
        const int v2 = acc[rt][4 * q + 2] + cr[rt][q][2] + ct, v3 = acc[rt][4 * q + 3] + cr[rt][q][3] + ct;
        m = max(max(m, max(v0, v1)), max(v2, v3));
      }
      update[rt] = update[rt] || __any(m > s_low);
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (update[rt]) {
      any_hit = true;
      // the row terms are (re-)read through a pointer the optimiser cannot identify with pass 1's,
      // or it keeps all 16 live across the branch and spills
      const int* rterm2 = rterm_wave;
      asm volatile("" : "+v"(rterm2));
      int m = -1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4i cr = *(const v4i*)(rterm2 + rt * kTile + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          const int v = acc[rt][r] + cr[i] + ct;
          m = max(m, v);
          const u32 rk = ((u32)v << 6) | jcode;
          rsec[rt][r] = umed3(rbest[rt][r], rsec[rt][r], rk);
          rbest[rt][r] = umax(rbest[rt][r], rk);
          if (FUSED) {
            const u32 ck = ((u32)v << 6) | (u32)(63 - (rt * kTile + (r & 3) + 8 * (r >> 2)));
            cs2 = umed3(cb, cs2, ck);
            cb = umax(cb, ck);
          }
        }
      }
      dense[rt] = __any(m > s_low);
    }
  }
#ifndef VC_EXP_NO_COLATOMIC
  if (FUSED && any_hit) {
    if (cb != 0) {
      const int j = jt * kTile + c;
      const u32 sb = cb >> 6;
      const u32 grow = row_base + (63u - (cb & 63u)) + 4u * h;
      const unsigned long long key = ((unsigned long long)sb << 32) | (unsigned long long)(0xFFFFFFFFu - grow);
      const unsigned long long old = atomicMax(&colbest[j], key);
      u32 cand = key > old ? (u32)(old >> 32) : sb;
      cand = umax(cand, cs2 >> 6);
      atomicMax(&colsecond[j], cand);
    }
  }
#else
  asm volatile("" :: "v"(cb), "v"(cs2));
#endif
}

// ---------------------------------------------------------------------------------------
// Phases of the persistent kernel (pair2_kernel): accumulators that start from the row term
// ---------------------------------------------------------------------------------------
// MFMA phase with C = row term: acc[rt][4q + i] starts at rterm[32 rt + 8 q + 4 h + i] (the row that register
// holds in the 32x32 MFMA C layout), read from the wave's LDS slice straight into the accumulator registers,
// so the result is sum (a-128)(b-128) + 128 ra - 49024 D and the column term is all that is left to add.
// B operands come through a rolling window of NB fragment registers, D = NB - 1 k-steps ahead of the MFMAs that use them
// (stamps: with one two-fragment group of look-ahead a wave alone on the pipe ran its 24 MFMAs in ~1300 cycles instead
// of 768 — every group waited for an LDS read issued 128 pipe cycles earlier).
// Not kept (measured): letting the window run on into the NEXT tile of the ring (12 more registers live
// across the epilogue: spills, and any spill reload near the tile loop makes the compiler wait for vmcnt, which drains the
// LDS-DMA ring), and eight symmetric waves that issue the next tile's first reads in front of the barrier (12.3 M
// pairs/s against 14.1 M with the staggered halves).
template <int KS>
struct BWindow {
  static constexpr int NB = KS == 12 ? VC2_NB12 : (KS >= 4 ? 4 : KS);
  static constexpr int D = NB - 1;
  static_assert(KS % NB == 0 && head_steps_of(KS) % NB == 0, "window positions must line up at the head boundary");
};

// steps K0..K1-1 of a tile whose window is refilled up to fragment NF-1; `mid()` runs behind the MFMAs of step 0 (the
// matrix pipe is busy for the next 64 cycles, which hides part of the issue cost of the LDS-DMA pieces placed there)
template <int KS, int K0, int K1, int NF, typename Mid>
__device__ __forceinline__ void mfma_steps(const v4i (&afrag)[2][KS], v16i (&acc)[2], v4i (&bf)[BWindow<KS>::NB],
                                           const uint8_t* src, Mid mid) {
  constexpr int NB = BWindow<KS>::NB, D = BWindow<KS>::D;
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int f = k + D;
    if (NB > 1 && f < NF) bf[f % NB] = *(const v4i*)(src + f * kFragBytes);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      acc[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[rt][k], bf[k % NB], acc[rt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (NB == 1 && k + 1 < NF) bf[0] = *(const v4i*)(src + (k + 1) * kFragBytes);
    if (k == 0) mid();
  }
}

// MFMA phase of one column tile.  EARLY (descriptors of 129..384 bytes): the accumulators start from the HEAD row term
// and the first KH k-steps give, per element, s_head - (head column term).  The rest of the dot product is bounded by
// Cauchy-Schwarz on the bytes themselves: sum over the tail of a b <= |a_tail| |b_tail| <= TNa TNb (per-tile maxima of
// the rounded-up tail norms, from prepare_kernel), and the tail's share of the bias terms is known exactly — it is what
// the head row / column terms leave out.  If no lane holds  acc > s_low - cth - TNa TNb  the whole 32x32 tile stays at
// or below the relevance threshold whatever the tail holds: its remaining KS - KH MFMAs per row tile and its epilogue
// are skipped (nothing it could contribute would change the match list, see relevance_threshold()).  Otherwise the
// tile simply goes on: the tail MFMAs run on the same accumulators and the difference between the full and the head row
// term (`rdelta`, one LDS word per register) is added afterwards — a failed test costs the test and 32 additions, no
// MFMA is repeated.  Returns 3 if the wave's two row tiles were cut short (the epilogue skips them), else 0.
template <int KS, bool EARLY, typename Mid>
__device__ __forceinline__ int mfma_phase2(const v4i (&afrag)[2][KS], v16i (&acc)[2], const uint8_t* slot,
                                           const int* rterm_wave, const int* rterm_head_wave, const int* rdelta_wave,
                                           int lane, int h, int thr_early, Mid mid) {
  constexpr int RT = 2;
  constexpr int KH = head_steps_of(KS);
  constexpr int NB = BWindow<KS>::NB, D = BWindow<KS>::D;
  const uint8_t* src = slot + lane * 16;
#ifdef VC2_SETPRIO   // (raising the MFMA phase's priority paid on the round's earlier kernels; on the final one it costs the dense path 4 %)
  __builtin_amdgcn_s_setprio(1);
#endif
  v4i bf[NB];
#pragma unroll
  for (int i = 0; i < (NB > 1 ? D : 1); ++i) bf[i] = *(const v4i*)(src + i * kFragBytes);
  const int* rterm = EARLY ? rterm_head_wave : rterm_wave;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4i cr = *(const v4i*)(rterm + rt * kTile + 8 * q + 4 * h);
      acc[rt][4 * q + 0] = cr[0]; acc[rt][4 * q + 1] = cr[1]; acc[rt][4 * q + 2] = cr[2]; acc[rt][4 * q + 3] = cr[3];
    }
  int cut = 0;
  mfma_steps<KS, 0, KH, KS>(afrag, acc, bf, src, mid);
  auto nothing = []() {};
  if (EARLY) {
    // one test for both row tiles of the wave (threshold from the larger of their two tail bounds)
    int m = max(acc[0][0], acc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) m = max(m, max(acc[0][r], acc[1][r]));
    if (__any(m > thr_early)) {   // may matter after all: the tail, then the part of the row term the head left out
      mfma_steps<KS, KH, KS, KS>(afrag, acc, bf, src, nothing);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4i dr = *(const v4i*)(rdelta_wave + rt * kTile + 8 * q + 4 * h);
          acc[rt][4 * q + 0] += dr[0]; acc[rt][4 * q + 1] += dr[1]; acc[rt][4 * q + 2] += dr[2]; acc[rt][4 * q + 3] += dr[3];
        }
    } else {
      cut = 3;
    }
  } else if (KH < KS) {
    mfma_steps<KS, KH, KS, KS>(afrag, acc, bf, src, nothing);
  }
#ifdef VC2_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  return cut;
}

// per-column search state of pair2_kernel: one LDS address serves both atomics of the column merge
struct ColState {
  unsigned long long best;   // (s << 32) | ~row: highest s, lowest row on ties
  u32 second;
  u32 pad;
};

// Epilogue with the row term already inside the accumulators: s = acc + ct (ct: the lane's column term).
//   relevance: some lane holds acc > s_low - ct                       (8 v_max3 + 1 compare per 32x32 tile)
//   update   : key = (acc << 6) + ((ct << 6) | code)                  (one v_lshl_add per key)
// The relevance test is cheap enough to run on every tile, so there is no separate "dense" regime here.
// `ct` is read from LDS by the caller BEFORE the MFMA phase that precedes this call: read here it queued behind
// the other waves' fragment reads (stamps: ~450 of an epilogue's 660 cycles).  Column merge as in epilogue_phase.
__device__ __forceinline__ bool epilogue_phase2(const v16i (&acc)[2], u32 (&rbest)[2][16], u32 (&rsec)[2][16],
                                                int ct, ColState* col,
                                                int jt, int c, int h, u32 row_base, int s_low, int cut) {
  constexpr int RT = 2;
  const int thr = s_low == 0x7fffffff ? s_low : s_low - ct;   // acc > thr  <=>  acc + ct > s_low (|ct| < 2^27: no overflow)
  const u32 ctj = ((u32)ct << 6) + (63u - (u32)jt);           // key = (acc << 6) + ctj   (acc + ct >= 0)
  u32 cb = 0, cs2 = 0;
  bool hit = false;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if ((cut >> rt) & 1) continue;   // cut short by the early-out: proven irrelevant, accumulators incomplete
    int m = acc[rt][0];
#pragma unroll
    for (int r = 1; r < 16; r += 2) m = max(m, r + 1 < 16 ? max(acc[rt][r], acc[rt][r + 1]) : acc[rt][r]);
    if (__any(m > thr)) {
      hit = true;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const u32 rk = ((u32)acc[rt][r] << 6) + ctj;
        rsec[rt][r] = umed3(rbest[rt][r], rsec[rt][r], rk);
        rbest[rt][r] = umax(rbest[rt][r], rk);
        const u32 ck = (rk & ~63u) | (u32)(63 - (rt * kTile + (r & 3) + 8 * (r >> 2)));
        cs2 = umed3(cb, cs2, ck);
        cb = umax(cb, ck);
      }
      // Keep the updates INSIDE the branch: without these ties the optimiser sinks the 32 med3 / max below the
      // join with a zero key on the other path, i.e. runs them for every tile (seen in the ISA).
#pragma unroll
      for (int r = 0; r < 16; r += 4)
        asm volatile("" : "+v"(rbest[rt][r]), "+v"(rbest[rt][r + 1]), "+v"(rbest[rt][r + 2]), "+v"(rbest[rt][r + 3]),
                          "+v"(rsec[rt][r]), "+v"(rsec[rt][r + 1]), "+v"(rsec[rt][r + 2]), "+v"(rsec[rt][r + 3]));
    }
  }
  if (cb != 0) {
    const int j = jt * kTile + c;
    const u32 sb = cb >> 6;
    const u32 grow = row_base + (63u - (cb & 63u)) + 4u * h;
    const unsigned long long key = ((unsigned long long)sb << 32) | (unsigned long long)(0xFFFFFFFFu - grow);
    const unsigned long long old = atomicMax(&col[j].best, key);
    u32 cand = key > old ? (u32)(old >> 32) : sb;
    cand = umax(cand, cs2 >> 6);
    atomicMax(&col[j].second, cand);
  }
  return hit;
}

template <int KS, int RT, bool FUSED>
__global__ __launch_bounds__(kThreads, 2) void pair_kernel(
    const uint8_t* __restrict__ prepared, const int32_t* __restrict__ counts, int n_tiles_img, int d,
    const int32_t* __restrict__ pairs, float max_ratio, float max_distance, int cross_check,
    int n_max, int ns, int s_low, uint32_t* __restrict__ out_matches, int32_t* __restrict__ out_counts,
    // !FUSED: one-way outputs (rows of A against B)
    int32_t* __restrict__ o_idx, int32_t* __restrict__ o_best, int32_t* __restrict__ o_second) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int n_pad = n_tiles_img * kTile;
  uint8_t* ring = smem;
  unsigned long long* colbest = (unsigned long long*)(smem + (size_t)ns * KS * kFragBytes);
  u32* colsecond = (u32*)(colbest + n_pad);
  int* cterm = (int*)(colsecond + n_pad);
  int* m21 = cterm + n_pad;
  int* rbest_s = m21 + n_pad;
  int* rsecond_s = rbest_s + n_pad;
  int* ridx_s = rsecond_s + n_pad;
  int* crow6 = ridx_s + n_pad;              // [wave][RT*32]: row terms 128*ra - 49024*D
  int* wave_count = crow6 + kWaves * 64 + kWaves * kRowScratchBytes / 4;

#ifdef VC_EXP_STAMP
  const unsigned long long t_kernel_start = stamp();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  uint2* rscratch = (uint2*)(crow6 + kWaves * 64) + wave * kRowScratchEntries;  // this wave's slice

  const int p = blockIdx.x;
  const int img_a = pairs[2 * p], img_b = pairs[2 * p + 1];
  const int n1 = min(max(counts[img_a], 0), n_max);
  const int n2 = min(max(counts[img_b], 0), n_max);
  const size_t img_stride = image_bytes(n_tiles_img, KS);
  const uint8_t* a_frags = prepared + (size_t)img_a * img_stride;
  const uint8_t* b_frags = prepared + (size_t)img_b * img_stride;
  const int32_t* a_rowsum = (const int32_t*)(a_frags + (size_t)n_tiles_img * KS * kFragBytes);
  const int32_t* b_rowsum = (const int32_t*)(b_frags + (size_t)n_tiles_img * KS * kFragBytes);

  // Pass 0's A fragments and row sums depend only on the pair indices: issue them now so that
  // their latency overlaps the (dependent) count lookups, the LDS initialisation and the first
  // LDS-DMA pieces instead of following them.
  v4i afrag[RT][KS];
  {
    const int tile0 = wave * RT;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
        afrag[rt][kk] = *(const v4i*)(a_frags + ((size_t)(tile0 + rt) * KS + kk) * kFragBytes + lane * 16);
  }
  int rowsum0 = 0;
  if (lane < RT * kTile) rowsum0 = a_rowsum[wave * RT * kTile + lane];

  constexpr int kRowsPerPass = kWaves * RT * kTile;
  const int n_ct = ceil_div(n2, kTile);  // column tiles of b that hold valid rows
  const int n_pass = ceil_div(n1, kRowsPerPass);
  const int total = n_pass * n_ct;       // tiles consumed, in order (pass, jt)
  const int pf = ns - 1;

  if (total == 0) {  // an empty image: nothing can match (uniform exit)
    if (FUSED) { if (tid == 0) out_counts[p] = 0; }
    else for (int i = tid; i < n1; i += kThreads) { o_idx[i] = -1; o_best[i] = 0; o_second[i] = 0; }
    return;
  }

  // ---- per-pair LDS state ----------------------------------------------------------------
  for (int j = tid; j < n_ct * kTile; j += kThreads) {
    cterm[j] = 128 * b_rowsum[j] + 32640 * d;
    colbest[j] = 0ull;
    colsecond[j] = 0u;
  }
  if (FUSED)  // defaults for rows whose reduction round is skipped (ordered by the tile barriers)
    for (int i = tid; i < n1; i += kThreads) { rbest_s[i] = 0; rsecond_s[i] = 0; ridx_s[i] = -1; }
  const int rbias = -49024 * d;
  int* crow6_wave = crow6 + wave * 64;
  const u32 ring_lds = (u32)__builtin_amdgcn_readfirstlane((int)lds_addr(ring));

  // producer / consumer cursors over the tile sequence
  int prod_seq = 0, prod_jt = 0, prod_slot = 0;
  for (; prod_seq < pf && prod_seq < total; ++prod_seq) {
    stage_tile<KS>(b_frags + (size_t)prod_jt * KS * kFragBytes, ring_lds + (u32)prod_slot * (KS * kFragBytes), wave, lane);
    if (++prod_jt == n_ct) prod_jt = 0;
    if (++prod_slot == ns) prod_slot = 0;
  }
  int cons_seq = 0, cons_slot = 0;
  // The two waves of a SIMD (w and w+4) run half a tile apart: while waves 0-3 issue the MFMAs of
  // tile t, waves 4-7 run the VALU epilogue of tile t-1, and vice versa, so the matrix pipe and the
  // vector pipe of a SIMD work at the same time instead of being fought over in lockstep.
#ifndef VC_NO_STAGGER
  // (with four-wave workgroups the SIMD partner is a wave of the other workgroup on the CU: no stagger)
  const bool late = kWaves == 8 && wave >= kWaves / 2;
#else
  constexpr bool late = false;
#endif

  for (int pass = 0; pass < n_pass; ++pass) {
    const int tile0 = (pass * kWaves + wave) * RT;  // first 32-row tile of a owned by this wave
    // A fragments stay in registers for the whole pass (every tile index exists: tiles_of());
    // pass 0's were requested at kernel entry.
    u32 rbest[RT][16], rsec[RT][16];
    int rowsum = rowsum0;
    if (pass > 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          afrag[rt][kk] = *(const v4i*)(a_frags + ((size_t)(tile0 + rt) * KS + kk) * kFragBytes + lane * 16);
      if (lane < RT * kTile) rowsum = a_rowsum[tile0 * kTile + lane];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { rbest[rt][r] = 0; rsec[rt][r] = 0; }
    // row term of this lane's row (lane <-> row tile0*32 + lane of the wave's RT*32 rows)
    if (lane < RT * kTile) crow6_wave[lane] = 128 * rowsum + rbias;
    const u32 row_base = (u32)(tile0 * kTile);
    bool dense[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) dense[rt] = false;
    v16i acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0;

    // one iteration = one column tile: wait for it, barrier, then the two phases in the order of
    // this wave's half
#define VC_TILE_HEAD()                                                                              \
    wait_tile<KS>(wave, prod_seq - cons_seq - 1);                                                   \
    wg_barrier();                                                                                   \
    const uint8_t* slot = ring + (size_t)cons_slot * KS * kFragBytes;                               \
    if (++cons_slot == ns) cons_slot = 0;                                                           \
    ++cons_seq;
    // refill of the slot freed by the previous iteration; legal anywhere after the barrier
    auto produce = [&]() {
      if (prod_seq < total) {
        stage_tile<KS>(b_frags + (size_t)prod_jt * KS * kFragBytes,
                       ring_lds + (u32)prod_slot * (KS * kFragBytes), wave, lane);
        ++prod_seq;
        if (++prod_jt == n_ct) prod_jt = 0;
        if (++prod_slot == ns) prod_slot = 0;
      }
    };

    {
    // Staggered halves (late = waves 4-7): one loop, the epilogue shared, only the MFMA phase
    // placed before or after it.  The late half runs the epilogue of tile jt-1; for jt = 0 that
    // call is neutralised by an unreachable threshold (its accumulators are not defined yet).
#ifdef VC_EXP_STAMP
    unsigned long long sw = 0, sm = 0, se = 0, sm2 = 0;
    const unsigned long long t_begin = stamp();
    unsigned long long tp = t_begin;
#endif
    for (int jt = 0; jt < n_ct; ++jt) {
      VC_TILE_HEAD()
#ifdef VC_EXP_STAMP
      unsigned long long ta = stamp(); sw += ta - tp;
#endif
      if (!late) mfma_phase<KS, RT>(afrag, acc, slot, lane, produce);
#ifdef VC_EXP_STAMP
      unsigned long long tb = stamp(); sm += tb - ta;
#endif
      const int ejt = late ? (jt > 0 ? jt - 1 : 0) : jt;
      const int eth = (late && jt == 0) ? 0x7fffffff : s_low;
      epilogue_phase<RT, FUSED>(acc, rbest, rsec, crow6_wave, cterm, colbest, colsecond, ejt, c, h, row_base, eth, dense);
#ifdef VC_EXP_STAMP
      unsigned long long tc = stamp(); se += tc - tb;
#endif
      if (late) mfma_phase<KS, RT>(afrag, acc, slot, lane, produce);
#ifdef VC_EXP_STAMP
      tp = stamp(); sm2 += tp - tc;
#endif
    }
#ifdef VC_EXP_STAMP
    if (FUSED && lane == 0 && pass == 0) {
      uint32_t* dbg = out_matches + ((size_t)p * n_max + (n_max - 64)) * 2 + wave * 8;
      dbg[0] = (uint32_t)sw; dbg[1] = (uint32_t)(sm + sm2); dbg[2] = (uint32_t)se;
      dbg[3] = (uint32_t)(tp - t_begin); dbg[4] = (uint32_t)(t_begin - t_kernel_start);
    }
#endif
    if (late)
      epilogue_phase<RT, FUSED>(acc, rbest, rsec, crow6_wave, cterm, colbest, colsecond, n_ct - 1, c, h, row_base, s_low, dense);
    }
#undef VC_TILE_HEAD

    // ---- row results of this pass ------------------------------------------------------
    // Each row's candidates sit in 32 lanes (one per column residue c).  Transpose through a
    // wave-private LDS slice, 16 rows per round: lane (c, h) writes its {best, second} keys of
    // 8 registers, then lane L re-reads row L>>2, columns 8*(L&3) .. +7 in ascending order
    // (strict '>' keeps the lowest column on ties) and the four partial results of a row are
    // folded with two quad-permute steps.  No workgroup barrier: the slice belongs to the wave
    // and its LDS operations execute in order.
#ifdef VC_EXP_NO_ROWREDUCE
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(rbest[rt][r]), "v"(rsec[rt][r]));
    if (false)
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // a round whose 16 rows never saw a relevant similarity keeps the (0, 0, -1) defaults
        u32 seen = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) seen |= rbest[rt][8 * half + j];
        if (FUSED && !__any(seen != 0)) continue;  // (the one-way API reports every row)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * half + j;
          rscratch[(j + 8 * h) * 33 + c] = make_uint2(rbest[rt][r], rsec[rt][r]);
        }
        const int rl = lane >> 2, qd = lane & 3;       // row of the round, column quarter
        u32 rb = 0, rs = 0, rc = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint2 en = rscratch[rl * 33 + qd * 8 + e];
          const bool gt = en.x > rb;
          rs = gt ? umax(rb, en.y) : umax(rs, en.x);
          rc = gt ? (u32)(qd * 8 + e) : rc;
          rb = umax(rb, en.x);
        }
        // fold quarters: the lower quarter absorbs the higher one (strict '>': lower column wins)
#pragma unroll
        for (int step = 0; step < 2; ++step) {
          u32 pb, ps, pc;
          if (step == 0) { pb = dpp_mov<0xB1>(rb); ps = dpp_mov<0xB1>(rs); pc = dpp_mov<0xB1>(rc); }   // quad_perm [1,0,3,2]
          else           { pb = dpp_mov<0x4E>(rb); ps = dpp_mov<0x4E>(rs); pc = dpp_mov<0x4E>(rc); }   // quad_perm [2,3,0,1]
          const bool gt = pb > rb;
          rs = gt ? umax(rb, ps) : umax(rs, pb);
          rc = gt ? pc : rc;
          rb = umax(rb, pb);
        }
        // row of this lane's result inside the wave's RT*32 rows (see the MFMA C layout)
        const int jj = rl & 7, hh = rl >> 3;
        const int lrow = (jj & 3) + 16 * half + 8 * (jj >> 2) + 4 * hh;
        if (qd == 0) {
          const int row = (tile0 + rt) * kTile + lrow;
          if (row < n1) {
            const int sb = (int)(rb >> 6);
            const int idx = sb > 0 ? (63 - (int)(rb & 63)) * kTile + (int)rc : -1;
            const int s2v = sb > 0 ? (int)(rs >> 6) : 0;
            if (FUSED) { rbest_s[row] = sb; rsecond_s[row] = s2v; ridx_s[row] = idx; }
            else { o_idx[row] = idx; o_best[row] = sb; o_second[row] = s2v; }
          }
        }
      }
    }
  }  // passes

  if (!FUSED) return;
#ifdef VC_EXP_NO_FINALIZE
  if (tid == 0) out_counts[p] = 0;
  return;
#endif
  __syncthreads();

  // ---- angle + ratio tests, cross check, ordered compaction -----------------------------
  if (cross_check) {
    for (int j = tid; j < n2; j += kThreads) {
      const unsigned long long kb = colbest[j];
      const int sb = (int)(kb >> 32);
      const int row = (int)(0xFFFFFFFFu - (u32)kb);
      m21[j] = accept_dev(sb, (int)colsecond[j], max_ratio, max_distance) ? row : -1;
    }
  }
  __syncthreads();
  uint32_t* out = out_matches + (size_t)p * n_max * 2;
  int base = 0;
  for (int i0 = 0; i0 < n1; i0 += kThreads) {
    const int i = i0 + tid;
    bool ok = false;
    int j = -1;
    if (i < n1) {
      j = ridx_s[i];
      ok = accept_dev(rbest_s[i], rsecond_s[i], max_ratio, max_distance);
      if (ok && cross_check) ok = (m21[j] == i);
    }
    const unsigned long long mask = __ballot(ok);
    if (lane == 0) wave_count[wave] = __popcll(mask);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += wave_count[w];
    int chunk_total = 0;
    for (int w = 0; w < kWaves; ++w) chunk_total += wave_count[w];
    if (ok) {
      const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
      out[2 * pos] = (uint32_t)i;
      out[2 * pos + 1] = (uint32_t)j;
    }
    base += chunk_total;
    __syncthreads();
  }
  if (tid == 0) out_counts[p] = base;
#ifdef VC_EXP_STAMP
  if (lane == 0) {
    const unsigned long long t_end = stamp();
    uint32_t* dbg = out_matches + ((size_t)p * n_max + (n_max - 64)) * 2 + wave * 8;
    dbg[5] = (uint32_t)(t_end - t_kernel_start);
  }
#endif
}

// ---------------------------------------------------------------------------------------
// The persistent pair kernel (KS <= 12, two row tiles per wave): what vc_match_pairs_u8 launches.
// ---------------------------------------------------------------------------------------
// Same arithmetic, tile loop and LDS layout as pair_kernel<KS, 2, true>; what differs is everything AROUND
// the tile loop, which is where a workgroup of pair_kernel spent a third of its time with the matrix pipe idle
// (measured: a launch with epilogue, staging and finalisation compiled out still ran at 55 % of the int8 peak):
//   * one workgroup per CU walks a CONTIGUOUS range of the pair list.  The exhaustive list is ordered by
//     image a, so consecutive pairs share it: the A fragments (96 VGPRs per wave, 192 KiB per workgroup) are
//     re-fetched only when a (or the row pass) changes, not once per pair;
//   * the B ring never drains: the producer cursor runs over the flattened (pair, pass, column tile) sequence,
//     so the first tiles of the next pair are already in LDS while the current one is finalised;
//   * the next pair's column sums are fetched into registers before the current pair's tile loop;
//   * the angle / ratio tests read theta from a table (accept_tab) — no double-precision acos, so the
//     finalisation is short and its registers do not collide with the fragments that stay live across it;
//   * the accumulators start from the row term (read from LDS as the MFMA's C operand) instead of zero: the
//     relevance test of a 32x32 tile is then a maximum over the lane's 16 accumulators against a per-lane
//     threshold (9 VALU instead of 24), and an update key is one shift-add.
// Block -> range mapping: blocks b and b + 8 share an XCD (round robin), so ranges are dealt such that each
// XCD owns a contiguous eighth of the pair list (neighbouring pairs share image a in that XCD's L2).
struct PairInfo {     // wave-uniform description of one pair with work (n_ct * n_pass > 0)
  int p;              // index in the pair list (>= hi: none)
  int a, n1, n2, n_ct, n_pass;
  const uint8_t* b_frags;
};

// first pair at or after `p` that has tiles to process (scalar loads; runs once per pair, outside the tile loop)
__device__ __forceinline__ PairInfo next_pair_with_work(int p, int hi, const int32_t* __restrict__ pairs,
                                                        const int32_t* __restrict__ counts, int n_max,
                                                        const uint8_t* __restrict__ prepared, size_t img_stride,
                                                        int rows_per_pass = kWaves * 2 * kTile) {
  PairInfo r;
  r.a = 0; r.n1 = 0; r.n2 = 0; r.n_ct = 0; r.n_pass = 0; r.b_frags = prepared;
  for (; p < hi; ++p) {
    const int a = pairs[2 * p], b = pairs[2 * p + 1];
    const int n1 = min(max(counts[a], 0), n_max), n2 = min(max(counts[b], 0), n_max);
    if (n1 > 0 && n2 > 0) {
      r.a = a; r.n1 = n1; r.n2 = n2;
      r.n_ct = ceil_div(n2, kTile);
      r.n_pass = ceil_div(n1, rows_per_pass);
      r.b_frags = prepared + (size_t)b * img_stride;
      break;
    }
  }
  r.p = p;
  return r;
}

#ifdef VC2_ALL_PRODUCE
constexpr int kProd2 = kWaves;
#else
constexpr int kProd2 = kWaves / 2;   // LDS-DMA pieces are issued by the early half (waves 0..3)
#endif
#ifndef VC2_TILES_PER_BARRIER
#define VC2_TILES_PER_BARRIER 2
#endif
constexpr int kTilesPerBarrier = VC2_TILES_PER_BARRIER;   // 1 or 2
template <int KS>
__global__ __launch_bounds__(kThreads, 2) void pair2_kernel(
    const uint8_t* __restrict__ prepared, const int32_t* __restrict__ counts, int n_tiles_img, int d,
    const int32_t* __restrict__ pairs, int n_pairs, float max_ratio, float max_distance, int cross_check,
    int n_max, int ns, int s_low, uint32_t* __restrict__ out_matches, int32_t* __restrict__ out_counts) {
  static_assert(kWaves == 8, "the persistent kernel is written for eight-wave workgroups");
  constexpr int RT = 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int n_pad = n_tiles_img * kTile;
  uint8_t* ring = smem;
  ColState* col = (ColState*)(smem + (size_t)ns * KS * kFragBytes);
  int* cterm = (int*)(col + n_pad);   // (kept out of the 16-byte column records: a stride-16 read is 4-way bank conflicted, -3 %)
  int* m21 = cterm + n_pad;
  int* rbest_s = m21 + n_pad;
  int* rsecond_s = rbest_s + n_pad;
  int* ridx_s = rsecond_s + n_pad;
  int* crow6 = ridx_s + n_pad;              // [wave][RT*32]: row terms 128*ra - 49024*D (the HEAD part with the early-out)
  int* wave_count = crow6 + kWaves * 64 + kWaves * kRowScratchBytes / 4;
  // early-out state (head_steps_of(KS) < KS): head column terms, tail norm bounds of b's tiles, tail row terms per wave
  int* cterm_h = wave_count + 16;
  int* tnb = cterm_h + n_pad;
  int* crow6h = tnb + n_pad / kTile;        // [wave][RT*32]: HEAD row terms 128*ra_head - 49024*D_head
  // column sums (packed with the early-out) and tile bounds of the NEXT pair's image b, copied in by LDS-DMA while the
  // current pair runs (no global load, hence no compiler vmcnt wait, at a pair boundary)
  int* auxw = crow6h + kWaves * 64;         // [n_pad]
  int* auxt = auxw + n_pad;                 // [n_tiles_img]
  int4* winfo = (int4*)(auxt + n_tiles_img);       // [kPairWindow] {p, a, b, n1 | n2 << 16}: pairs with work, in order
  int* wmeta = (int*)(winfo + kPairWindow);        // [0] entries in the window
  int* crow6d = wmeta + 4;                         // [wave][RT*32]: full minus head row term (added when the early-out test fails)
  constexpr int KH = head_steps_of(KS);
#ifdef VC2_NO_EARLY
  constexpr bool kEarlyOut = false;
#else
  constexpr bool kEarlyOut = KH < KS;
#endif
  const int d_head = min(d, KH * 32);

#ifdef VC_EXP_STAMP
  // diagnostic build: per-wave cycle totals over the workgroup's whole range (tools/stamp_matcher.py)
  unsigned long long st_wait = 0, st_mfma = 0, st_epi = 0, st_init = 0, st_rowred = 0, st_final = 0;
  const unsigned long long st_t0 = stamp();
  unsigned long long st_tp = st_t0;
#define VC_ST(acc_) { const unsigned long long t_ = stamp(); acc_ += t_ - st_tp; st_tp = t_; }
#else
#define VC_ST(acc_)
#endif
#ifdef VC_EXP_TRACE
  // diagnostic build (with VC_EXP_STAMP, data without matches): per column tile and wave eight words
  // {barrier arrival, release, MFMA begin, MFMA end, epilogue end, cut, -, -} into the workgroup's own (empty) match blocks
  int trace_i = 0;
#define VC_TR(k_, v_) { if ((threadIdx.x & 63) == 0 && trace_i >= 16 && trace_i < trace_n) trace[((size_t)trace_i * 8 + wave) * 8 + (k_)] = (uint32_t)(v_); }
#else
#define VC_TR(k_, v_)
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  uint2* rscratch = (uint2*)(crow6 + kWaves * 64) + wave * kRowScratchEntries;  // this wave's slice
  int* crow6_wave = crow6 + wave * 64;
  int* crow6h_wave = crow6h + wave * 64;
  int* crow6d_wave = crow6d + wave * 64;
  const size_t img_stride = image_bytes(n_tiles_img, KS);
  const size_t frag_bytes_img = (size_t)n_tiles_img * KS * kFragBytes;
  const u32 ring_lds = (u32)__builtin_amdgcn_readfirstlane((int)lds_addr(ring));

  // this workgroup's range of the pair list
  const int G = gridDim.x;
  const int cid = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int lo = (int)(((long long)cid * n_pairs) / G);
  const int hi = (int)(((long long)(cid + 1) * n_pairs) / G);

  // ---- the pairs of the range that have work, kPairWindow at a time ------------------------------------------------
  // One wave reads 64 pairs and their counts at once, writes zero counts for those with an empty image and lists the
  // others in LDS; the per-pair description is then an LDS read.  (Read pair by pair with scalar loads, two dependent
  // round trips sat between every two pairs.)  fetch() is called by all waves at the same points, between pairs.
  int w_i = 0, w_n = 0, w_base = lo;   // wave-uniform
  auto fetch = [&]() -> PairInfo {
    PairInfo r;
    r.p = hi; r.a = 0; r.n1 = 0; r.n2 = 0; r.n_ct = 0; r.n_pass = 0; r.b_frags = prepared;
    while (w_i == w_n) {
      if (w_base >= hi) return r;
      __syncthreads();   // nobody is still reading the list this round replaces
      if (wave == kWaves - 1) {
        const int q = lane < kPairWindow ? w_base + lane : hi;
        int a = 0, b = 0, n1 = 0, n2 = 0;
        if (q < hi) {
          a = pairs[2 * q]; b = pairs[2 * q + 1];
          n1 = min(max(counts[a], 0), n_max); n2 = min(max(counts[b], 0), n_max);
        }
        const bool work = q < hi && n1 > 0 && n2 > 0;
        if (q < hi && !work) out_counts[q] = 0;   // an empty image: nothing can match
        const unsigned long long mask = __ballot(work);
        if (work) winfo[__popcll(mask & ((1ull << lane) - 1ull))] = make_int4(q, a, b, n1 | (n2 << 16));
        if (lane == 0) wmeta[0] = __popcll(mask);
      }
      __syncthreads();
      w_n = __builtin_amdgcn_readfirstlane(wmeta[0]);
      w_i = 0;
      w_base = min(w_base + kPairWindow, hi);
    }
    const int4 e = winfo[w_i++];
    r.p = __builtin_amdgcn_readfirstlane(e.x);
    r.a = __builtin_amdgcn_readfirstlane(e.y);
    const int b = __builtin_amdgcn_readfirstlane(e.z);
    const int nn = __builtin_amdgcn_readfirstlane(e.w);
    r.n1 = nn & 0xffff; r.n2 = nn >> 16;
    r.n_ct = ceil_div(r.n2, kTile);
    r.n_pass = ceil_div(r.n1, kWaves * 2 * kTile);
    r.b_frags = prepared + (size_t)b * img_stride;
    return r;
  };
  PairInfo cur = fetch();
  if (cur.p >= hi) return;
  PairInfo nxt = fetch();
#ifdef VC_EXP_TRACE
  uint32_t* trace = out_matches + (size_t)lo * n_max * 2;
  const int trace_n = (int)(((size_t)(hi - lo) * n_max * 2) / 64);
#endif

  // ---- producer: a stream of column tiles over (pair, pass, tile), PF tiles ahead of the consumer ----------
  // It sweeps the current pair's image b once per row pass, then moves on to the NEXT pair (whose description is
  // already in registers), so the ring stays full across the pair boundary.  It never runs more than one pair
  // ahead: a next pair shorter than the ring leaves it idle until the consumer gets there.
  const int tpb = ns >= 6 ? kTilesPerBarrier : 1;   // a short ring (large blocks) keeps its depth: one tile per barrier
  const int pf = ns - tpb;
  const uint8_t* p_src = cur.b_frags;      // next tile to stage
  const uint8_t* p_base = cur.b_frags;
  int p_left = cur.n_ct, p_nct = cur.n_ct, p_sweeps = cur.n_pass;
  bool p_on_next = false, p_active = true;
  int prod_seq = 0, prod_slot = 0, cons_seq = 0, cons_slot = 0;
  auto produce = [&]() {
    if (p_active) {
      stage_tile<KS, kProd2>(p_src, ring_lds + (u32)prod_slot * (KS * kFragBytes), wave, lane);
      ++prod_seq;
      if (++prod_slot == ns) prod_slot = 0;
      p_src += KS * kFragBytes;
      if (--p_left == 0) {
        if (--p_sweeps > 0) {
          p_src = p_base; p_left = p_nct;
        } else if (!p_on_next && nxt.p < hi) {
          p_on_next = true;
          p_src = p_base = nxt.b_frags; p_left = p_nct = nxt.n_ct; p_sweeps = nxt.n_pass;
        } else {
          p_active = false;
        }
      }
    }
  };
  for (int i = 0; i < pf; ++i) produce();

  int* pair_flag = wave_count + kWaves;   // "some tile of the current pair was relevant"
  // Cursor check: prod_seq / cons_seq exist once per wave (SGPRs) and must agree across the workgroup — `two`, the slot
  // indices and the counted waits are all derived from them.  Every wave folds its pair of cursors into a max and a min
  // word at the end of a pair's tile loop; if they differ the pair's count is written as -1 instead of a match count.
  int* cursor_chk = pair_flag + 1;        // [2]: max, min over the waves
  // with the early-out the per-row word is the packed (head << 16 | tail) sum, else the plain row sum
  const int rs_off = kEarlyOut ? n_pad : 0;
  // Image b's column sums for pair X travel by LDS-DMA into the aux buffer, issued by one LATE wave (which has no other
  // copy in flight, so its vmcnt(0) waits for exactly these) behind the first barrier of the pair before X — every
  // thread has read that pair's own sums by then — and are read a whole pair later, after that wave's wait and the
  // barrier that ends the pair in between.
  constexpr int kAuxWave = kWaves / 2;
  const u32 auxw_lds = (u32)__builtin_amdgcn_readfirstlane((int)lds_addr(auxw));
  const u32 auxt_lds = (u32)__builtin_amdgcn_readfirstlane((int)lds_addr(auxt));
  auto stage_aux = [&](const PairInfo& pi) {
    if (wave == kAuxWave && pi.p < hi) {
      int ln = lane;   // (laundered: lane-derived addresses are computed here, not carried across the tile loop)
      asm volatile("" : "+v"(ln));
      const uint8_t* sums = pi.b_frags + frag_bytes_img + (size_t)rs_off * 4;
      const int pieces = ceil_div(pi.n_ct * kTile, 256);            // 1 KiB = 256 column words per piece
      for (int i = 0; i < pieces; ++i)
        glds16(sums + (size_t)i * 1024 + ln * 16, auxw_lds + (u32)(i * 1024));
      if (kEarlyOut && ln < pi.n_ct)
        glds4(pi.b_frags + frag_bytes_img + (size_t)2 * n_pad * 4 + ln * 4, auxt_lds);
    }
  };
  stage_aux(cur);
  if (wave == kAuxWave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // The two waves of a SIMD (w and w + 4) run half a tile apart, see pair_kernel.  (On the final kernel: without it the
  // dense path loses 6 % and the sparse one gains 1.5-3 %; switching it per pair — on only behind a pair with relevant
  // similarities — was slower on both, 17.7 vs 18.1 M and 7.58 vs 7.71 M pairs/s: `late` is better a per-wave constant.)
#ifdef VC2_NO_STAGGER
  constexpr bool late = false;
#else
  const bool late = wave >= kWaves / 2;
#endif
  v4i afrag[RT][KS];
  int cur_a = -1, cur_tile0 = -1;
  int tna = 0;   // the larger tail norm bound of this wave's two row tiles (wave-uniform)
  int early_score = 0, early_probe = 0;   // gate of the early-out (wave-uniform, kept across pairs)
  bool pairs_dense = false;               // the previous pair held relevant similarities (workgroup-uniform)
  v16i acc[RT];   // (the late half's first epilogue of a pass looks at stale accumulators behind an unreachable threshold)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0;

  while (true) {
    const int p = cur.p;
    const int n1 = cur.n1, n2 = cur.n2, n_ct = cur.n_ct, n_pass = cur.n_pass;
    const uint8_t* a_frags = prepared + (size_t)cur.a * img_stride;
    const int32_t* a_rowsum = (const int32_t*)(a_frags + frag_bytes_img);

    // ---- per-pair LDS state (the previous pair's finalisation ended with a barrier) ---------------------------
    // (columns 0..511 come from the register fetched during the previous pair: the load's latency was 2 k cycles
    // of every pair when it sat here)
    auto init_column = [&](int j, int word) {
      const int head = kEarlyOut ? packed_head(word) : 0, total = kEarlyOut ? head + packed_tail(word) : word;
      if (kEarlyOut) cterm_h[j] = 128 * head + 32640 * d_head;
      cterm[j] = 128 * total + 32640 * d;
      col[j].best = 0ull;
      col[j].second = 0u;
    };
    // Per-pair code indexes through a laundered copy of the thread id: with `tid` itself the compiler keeps a dozen
    // LDS / global addresses alive across the tile loop, spills them, and every reload costs an s_waitcnt vmcnt that
    // drains the LDS-DMA ring (the compiler cannot see the copies in flight).  A few more VALU instructions per pair.
    int tidp = tid;
    asm volatile("" : "+v"(tidp));
    for (int j = tidp; j < n_ct * kTile; j += kThreads) init_column(j, auxw[j]);
    if (kEarlyOut && tidp < n_ct) tnb[tidp] = auxt[tidp];
    for (int i = tidp; i < n1; i += kThreads) { rbest_s[i] = 0; rsecond_s[i] = 0; ridx_s[i] = -1; }
    if (tidp == 0) { *pair_flag = 0; cursor_chk[0] = (int)0x80000000; cursor_chk[1] = 0x7fffffff; }
    bool pair_hit = false;   // some tile of this wave held a relevant similarity

    for (int pass = 0; pass < n_pass; ++pass) {
      const int tile0 = (pass * kWaves + wave) * RT;  // first 32-row tile of a owned by this wave
      if (cur.a != cur_a || tile0 != cur_tile0) {     // wave-uniform: a new image a or a new row pass
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int kk = 0; kk < KS; ++kk)
            afrag[rt][kk] = *(const v4i*)(a_frags + ((size_t)(tile0 + rt) * KS + kk) * kFragBytes + lane * 16);
        // row term of this lane's row (lane <-> row tile0*32 + lane of the wave's RT*32 rows); the slice is
        // wave-private and its LDS operations execute in order
        {
          const int word = a_rowsum[rs_off + tile0 * kTile + lane];
          const int head = kEarlyOut ? packed_head(word) : 0, total = kEarlyOut ? head + packed_tail(word) : word;
          crow6_wave[lane] = 128 * total - 49024 * d;
          if (kEarlyOut) {
            crow6h_wave[lane] = 128 * head - 49024 * d_head;
            crow6d_wave[lane] = (128 * total - 49024 * d) - (128 * head - 49024 * d_head);
            tna = __builtin_amdgcn_readfirstlane(max(a_rowsum[2 * n_pad + tile0], a_rowsum[2 * n_pad + tile0 + 1]));
          }
        }
        cur_a = cur.a;
        cur_tile0 = tile0;
        // The fragments must have landed before the tile loop: left pending, the compiler's own wait counts for
        // their first use land INSIDE the loop (vmcnt(0) after the last MFMA) and drain the LDS-DMA prefetch of
        // every tile.  vmcnt(0), other counters untouched (gfx9 encoding); a builtin, so the compiler sees it.
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      u32 rbest[RT][16], rsec[RT][16];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { rbest[rt][r] = 0; rsec[rt][r] = 0; }
      const u32 row_base = (u32)(tile0 * kTile);
      VC_ST(st_init)

      int ct = 0;   // column term of the tile whose epilogue comes next; always read ahead of an MFMA phase
      int cut = 0;  // row tiles of that tile the early-out cut short (wave-uniform)
      // thresholds of the early-out for column tile jt: acc_head > s_low - cth - TNa TNb  <=>  the tile may still matter
      // When to try it: a saturating score (+2 for a tile cut short, -1 for a failed test, which costs the test and the
      // row-term fix-up but no MFMA) gates it, and while the score is negative one tile in 32 probes whether the data has
      // changed.  It stays on while one test in three succeeds; data on which it never does pays < 1 % for the probes.
      auto mfma_tile = [&](int jt, const uint8_t* slot) -> int {
        if (kEarlyOut && s_low >= 0 && (early_score >= 0 || --early_probe <= 0)) {   // (s_low = -1: nothing may be skipped)
          const int thr = s_low - cterm_h[jt * kTile + c] - __mul24(tna, tnb[jt]);
          const int r = mfma_phase2<KS, true>(afrag, acc, slot, crow6_wave, crow6h_wave, crow6d_wave, lane, h, thr, produce);
          if (r != 0) early_score = min(early_score + 2, 8);
          else { early_score = max(early_score - 1, -8); early_probe = 32; }
          return r;
        }
        return mfma_phase2<KS, false>(afrag, acc, slot, crow6_wave, crow6h_wave, crow6d_wave, lane, h, 0, produce);
      };
      // One barrier per TWO column tiles (with two thirds of the MFMAs gone a tile is ~2400 cycles of which ~590 went into
      // the vmcnt wait, the barrier and the arrival skew).  The barrier publishes both tiles; the ring runs ns - 2 tiles
      // ahead so that the copy issued during the second tile never lands in a slot a slower wave is still reading.
      auto one_tile = [&](int jt, const uint8_t* slot) {
        if (!late) {
          VC_TR(2, st_tp)
          ct = cterm[jt * kTile + c];
          cut = mfma_tile(jt, slot);
          VC_ST(st_mfma)
          VC_TR(3, st_tp)
          VC_TR(5, cut)
        }
        const int ejt = late ? (jt > 0 ? jt - 1 : 0) : jt;
        const int eth = (late && jt == 0) ? 0x7fffffff : s_low;
        pair_hit |= epilogue_phase2(acc, rbest, rsec, ct, col, ejt, c, h, row_base, eth, cut);
        VC_ST(st_epi)
        VC_TR(4, st_tp)
        if (late) {
          VC_TR(2, st_tp)
          ct = cterm[jt * kTile + c];
          cut = mfma_tile(jt, slot);
          VC_ST(st_mfma)
          VC_TR(3, st_tp)
          VC_TR(5, cut)
        }
#ifdef VC_EXP_TRACE
        ++trace_i;
#endif
      };
      for (int jt = 0; jt < n_ct;) {
        VC_TR(0, st_tp)
        // (both tiles must be staged already: not so right after the producer idled — tiny images — or with a short ring)
        // Pairs after one that held relevant similarities run one tile per barrier: with the update path in every
        // epilogue the late half's deferred epilogue is the longer part and pairing the tiles loses 3 %.
        const bool two = tpb == 2 && !pairs_dense && jt + 1 < n_ct && prod_seq - cons_seq >= 2;
        // this wave's pieces of the tile(s) of this round have landed once only the younger tiles are pending
        wait_tile<KS, kProd2>(wave, max(prod_seq - cons_seq - (two ? 2 : 1), 0));
        wg_barrier();
        if (jt == 0 && pass == 0) stage_aux(nxt);   // (every thread is past this pair's initialisation)
        VC_ST(st_wait)
        VC_TR(1, st_tp)
        const int n_sub = two ? 2 : 1;
        {
          // (Straight-line on purpose.  As a second trip through ONE copy of this body hipcc 7.2.0 gave the SGPR pair that
          // carries produce()'s p_active across the back edge to the early-out gate as well: waves on the variant with the
          // test then saw the producer inactive, the waves of a workgroup disagreed on `two` and barriers paired up wrongly —
          // profiles/r03_matcher_looped_miscompile.md, tools/repro/looped_sub.patch.  The cursor check at the end of every pair
          // (below) turns that class of fault into an error code instead of a wrong match list.)
          const uint8_t* slot = ring + (size_t)cons_slot * KS * kFragBytes;
          if (++cons_slot == ns) cons_slot = 0;
          ++cons_seq;
          one_tile(jt, slot);
          if (two) {
            const uint8_t* slot2 = ring + (size_t)cons_slot * KS * kFragBytes;
            if (++cons_slot == ns) cons_slot = 0;
            ++cons_seq;
            one_tile(jt + 1, slot2);
          }
        }
        jt += n_sub;
      }
      if (late) pair_hit |= epilogue_phase2(acc, rbest, rsec, ct, col, n_ct - 1, c, h, row_base, s_low, cut);
      VC_ST(st_epi)

      // ---- row results of this pass (see pair_kernel) --------------------------------------------------------
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32 seen = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) seen |= rbest[rt][8 * half + j];
          if (!__any(seen != 0)) continue;   // the 16 rows keep the (0, 0, -1) defaults
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * half + j;
            rscratch[(j + 8 * h) * 33 + c] = make_uint2(rbest[rt][r], rsec[rt][r]);
          }
          const int rl = lane >> 2, qd = lane & 3;       // row of the round, column quarter
          u32 rb = 0, rs = 0, rc = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint2 en = rscratch[rl * 33 + qd * 8 + e];
            const bool gt = en.x > rb;
            rs = gt ? umax(rb, en.y) : umax(rs, en.x);
            rc = gt ? (u32)(qd * 8 + e) : rc;
            rb = umax(rb, en.x);
          }
#pragma unroll
          for (int step = 0; step < 2; ++step) {
            u32 pb, ps, pc;
            if (step == 0) { pb = dpp_mov<0xB1>(rb); ps = dpp_mov<0xB1>(rs); pc = dpp_mov<0xB1>(rc); }   // quad_perm [1,0,3,2]
            else           { pb = dpp_mov<0x4E>(rb); ps = dpp_mov<0x4E>(rs); pc = dpp_mov<0x4E>(rc); }   // quad_perm [2,3,0,1]
            const bool gt = pb > rb;
            rs = gt ? umax(rb, ps) : umax(rs, pb);
            rc = gt ? pc : rc;
            rb = umax(rb, pb);
          }
          const int jj = rl & 7, hh = rl >> 3;
          const int lrow = (jj & 3) + 16 * half + 8 * (jj >> 2) + 4 * hh;
          if (qd == 0) {
            const int row = (tile0 + rt) * kTile + lrow;
            if (row < n1) {
              const int sb = (int)(rb >> 6);
              rbest_s[row] = sb;
              rsecond_s[row] = sb > 0 ? (int)(rs >> 6) : 0;
              ridx_s[row] = sb > 0 ? (63 - (int)(rb & 63)) * kTile + (int)rc : -1;
            }
          }
        }
      }
      VC_ST(st_rowred)
    }  // passes

    int tidf = tid;   // (as tidp: the finalisation's addresses are computed here, not carried through the tile loop)
    asm volatile("" : "+v"(tidf));
    if (pair_hit && lane == 0) atomicOr(pair_flag, 1);
    if (lane == 0) {
      const int cursors = (cons_seq & 0xffff) | ((prod_seq & 0x7fff) << 16);
      atomicMax(&cursor_chk[0], cursors);
      atomicMin(&cursor_chk[1], cursors);
    }
    __syncthreads();
    const int pair_was_hit = __builtin_amdgcn_readfirstlane(*pair_flag);
    const bool cursors_agree = __builtin_amdgcn_readfirstlane(cursor_chk[0]) == __builtin_amdgcn_readfirstlane(cursor_chk[1]);
    pairs_dense = pair_was_hit != 0;   // (workgroup-uniform: every wave reads the same word behind the barrier)
    if (pair_was_hit == 0) {
      // No tile of the pair held a similarity above the relevance threshold: every row's best stays below what the
      // angle test accepts, the match list is empty and nothing of the finalisation has to run.
      if (tidf == 0) out_counts[p] = cursors_agree ? 0 : VC_COUNT_SELFCHECK_FAILED;
    } else {
    // ---- angle + ratio tests, cross check, ordered compaction ---------------------------------------------------
    if (cross_check) {
      for (int j = tidf; j < n2; j += kThreads) {
        const unsigned long long kb = col[j].best;
        const int row = (int)(0xFFFFFFFFu - (u32)kb);
        m21[j] = accept_tab((int)(kb >> 32), (int)col[j].second, max_ratio, max_distance, s_low) ? row : -1;
      }
      __syncthreads();
    }
    uint32_t* out = out_matches + (size_t)p * n_max * 2;
    int base = 0;
    for (int i0 = 0; i0 < n1; i0 += kThreads) {
      const int i = i0 + tidf;
      bool ok = false;
      int j = -1;
      if (i < n1) {
        j = ridx_s[i];
        ok = accept_tab(rbest_s[i], rsecond_s[i], max_ratio, max_distance, s_low);
        if (ok && cross_check) ok = (m21[j] == i);
      }
      const unsigned long long mask = __ballot(ok);
      if (lane == 0) wave_count[wave] = __popcll(mask);
      __syncthreads();
      int before = base, chunk_total = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const int wc = wave_count[w];
        before += w < wave ? wc : 0;
        chunk_total += wc;
      }
      if (ok) {
        const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
        out[2 * pos] = (uint32_t)i;
        out[2 * pos + 1] = (uint32_t)j;
      }
      base += chunk_total;
      __syncthreads();
    }
    if (tidf == 0) out_counts[p] = cursors_agree ? base : VC_COUNT_SELFCHECK_FAILED;
    }
    // the next pair's column sums (copied in during this pair) have landed before the barrier that lets it start
    if (wave == kAuxWave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the flag and the per-pair state are rewritten by the next pair's initialisation

    VC_ST(st_final)
    // ---- on to the next pair with work ---------------------------------------------------------------------------
    if (nxt.p >= hi) break;
    cur = nxt;
    nxt = fetch();
    if (p_on_next) {
      p_on_next = false;               // the producer's pair is the consumer's pair again
    }
    if (!p_active && nxt.p < hi) {     // the producer had run out of described work: resume on the new next pair
      p_active = true; p_on_next = true;
      p_src = p_base = nxt.b_frags; p_left = p_nct = nxt.n_ct; p_sweeps = nxt.n_pass;
    }
  }  // pairs
#ifdef VC_EXP_STAMP
  if (lane == 0) {
    uint32_t* dbg = out_matches + ((size_t)lo * n_max + (n_max - 64)) * 2 + wave * 8;
    dbg[0] = (uint32_t)st_wait; dbg[1] = (uint32_t)st_mfma; dbg[2] = (uint32_t)st_epi; dbg[3] = (uint32_t)st_init;
    dbg[4] = (uint32_t)st_rowred; dbg[5] = (uint32_t)st_final; dbg[6] = (uint32_t)(stamp() - st_t0); dbg[7] = (uint32_t)(hi - lo);
  }
#endif
#undef VC_ST
#undef VC_TR
}

// ---------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------
__global__ void mutual_ratio_kernel(const int32_t* idx12, const int32_t* best12, const int32_t* second12,
                                    int n1, const int32_t* idx21, const int32_t* best21,
                                    const int32_t* second21, int n2, float max_ratio,
                                    float max_distance, int cross_check, uint32_t* out_pairs,
                                    int32_t* out_count) {
  // single workgroup: ordered compaction over i
  __shared__ int wave_count[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  int base = 0;
  for (int i0 = 0; i0 < n1; i0 += blockDim.x) {
    const int i = i0 + tid;
    bool ok = false;
    int j = -1;
    if (i < n1) {
      j = idx12[i];
      ok = j >= 0 && j < n2 && accept_dev(best12[i], second12[i], max_ratio, max_distance);
      if (ok && cross_check)
        ok = idx21[j] == i && accept_dev(best21[j], second21[j], max_ratio, max_distance);
    }
    const unsigned long long mask = __ballot(ok);
    if (lane == 0) wave_count[wave] = __popcll(mask);
    __syncthreads();
    int before = base, tot = 0;
    for (int w = 0; w < nw; ++w) { if (w < wave) before += wave_count[w]; tot += wave_count[w]; }
    if (ok) {
      const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
      out_pairs[2 * pos] = (uint32_t)i;
      out_pairs[2 * pos + 1] = (uint32_t)j;
    }
    base += tot;
    __syncthreads();
  }
  if (tid == 0) *out_count = base;
}

__global__ void knn_setup_kernel(int32_t* meta, int n1, int n2) {
  meta[0] = n1; meta[1] = n2;   // counts
  meta[2] = 0;  meta[3] = 1;    // the pair
}

__global__ void theta_table_kernel(float* out, int n, int from_table) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) out[s] = from_table ? theta_tab(s) : theta_dev(s);
}

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// relevance threshold (host side launch parameter)
// ---------------------------------------------------------------------------------------
// theta on the host, same definition as theta_dev (tests/test_matcher_gpu.py compares the
// device table with the oracle's for every possible input).
inline float theta_host(int s) {
  float x = (float)s * (1.0f / (512.0f * 512.0f));
  x = x > 1.0f ? 1.0f : x;
  return (float)acos((double)x);
}

// Largest similarity that can be ignored by the searches without changing the match list.
//   S_T   = smallest s with theta(s) <= max_distance: a best below it is always rejected.
//   S_low = largest s (< S_T) with max_ratio * theta(s) > theta(S_T).
// Take a row (or column) with top-2 (b, s2) and drop any elements <= S_low from its scan:
//   b <  S_T : rejected before and after (the best can only shrink).
//   b >= S_T : b > S_low, so the best and its index are untouched.  If s2 > S_low it is untouched
//              too.  If s2 <= S_low, then max_ratio*theta(s2) >= max_ratio*theta(S_low) >
//              theta(S_T) >= theta(b): the ratio test passes, and it passes just the same with
//              whatever smaller runner-up is found instead.
// theta is non-increasing in s and multiplication by a non-negative float is monotone, so both
// thresholds are found by bisection.  A margin of 4 absorbs a last-bit difference between the
// host's and the device's acos (one ulp of theta moves a threshold by < 0.01).
// Returns -1 ("nothing can be ignored") for parameters outside the monotone regime.
inline int relevance_threshold(float max_ratio, float max_distance) {
  if (!(max_ratio >= 0.0f) || !(max_distance == max_distance)) return -1;
  constexpr int kSat = 512 * 512;
  if (theta_host(kSat) > max_distance) return kSat;  // nothing is ever accepted (max_distance < 0)
  int lo = 0, hi = kSat;  // smallest s with theta(s) <= max_distance
  while (lo < hi) {
    const int mid = (lo + hi) / 2;
    if (theta_host(mid) <= max_distance) hi = mid; else lo = mid + 1;
  }
  const int s_t = lo;
  const float t_t = theta_host(s_t);
  if (!(max_ratio * theta_host(0) > t_t)) return -1;
  lo = 0; hi = s_t > 0 ? s_t - 1 : 0;  // largest s with max_ratio*theta(s) > t_t
  while (lo < hi) {
    const int mid = (lo + hi + 1) / 2;
    if (max_ratio * theta_host(mid) > t_t) lo = mid; else hi = mid - 1;
  }
  int s_low = lo < s_t - 1 ? lo : s_t - 1;
  s_low -= 4;
  return s_low < -1 ? -1 : s_low;
}

// K-step counts with a compiled kernel; a descriptor is zero-padded up to the next one.
constexpr int kKsList[] = {2, 4, 8, 12, 16, 24, 32};

inline int pick_ks(int d) {
  const int need = ksteps_of(d);
  for (int ks : kKsList)
    if (ks >= need) return ks;
  return -1;
}

template <int KS, int RT, bool FUSED>
int launch_pair(const void* prepared, const int32_t* counts, int n_tiles, int d, const int32_t* pairs,
                int n_pairs, float max_ratio, float max_distance, int cross_check, int n_max,
                uint32_t* out_matches, int32_t* out_counts, int32_t* o_idx, int32_t* o_best,
                int32_t* o_second, hipStream_t stream) {
  // the one-way API reports every row's best / second, so nothing may be skipped there
  const int s_low = FUSED ? relevance_threshold(max_ratio, max_distance) : -1;
  const int n_pad = n_tiles * kTile;
  const int ns = plan_slots(KS, n_pad);
  if (ns == 0 || (RT == 2 && ns < 3)) return VC_ERR_UNSUPPORTED;
  const size_t smem = (size_t)ns * KS * kFragBytes + lds_fixed_bytes(n_pad);
  static vc::PerDeviceOnce configured;  // per instantiation and device
  if (int st = configured.run([] {
        return hipFuncSetAttribute((const void*)pair_kernel<KS, RT, FUSED>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
      }))
    return st;
  hipLaunchKernelGGL((pair_kernel<KS, RT, FUSED>), dim3(n_pairs), dim3(kThreads), smem, stream,
                     (const uint8_t*)prepared, counts, n_tiles, d, pairs, max_ratio, max_distance,
                     cross_check, n_max, ns, s_low, out_matches, out_counts, o_idx, o_best, o_second);
  return vc::check_launch();
}

// The persistent kernel: one workgroup per CU, each walking a contiguous range of the pair list.
template <int KS>
int launch_pair2(const void* prepared, const int32_t* counts, int n_tiles, int d, const int32_t* pairs,
                 int n_pairs, float max_ratio, float max_distance, int cross_check, int n_max,
                 uint32_t* out_matches, int32_t* out_counts, hipStream_t stream) {
  const int s_low = relevance_threshold(max_ratio, max_distance);
  const int n_pad = n_tiles * kTile;
  const int ns = plan_slots2(KS, n_pad);
  if (ns < 3) return VC_ERR_UNSUPPORTED;
  const size_t smem = (size_t)ns * KS * kFragBytes + lds_fixed_bytes2(n_pad);
  static vc::PerDeviceOnce configured;
  if (int st = configured.run([] {
        return hipFuncSetAttribute((const void*)pair2_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
      }))
    return st;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const int grid = n_pairs < cus ? n_pairs : cus;
  hipLaunchKernelGGL((pair2_kernel<KS>), dim3(grid), dim3(kThreads), smem, stream, (const uint8_t*)prepared, counts,
                     n_tiles, d, pairs, n_pairs, max_ratio, max_distance, cross_check, n_max, ns, s_low, out_matches,
                     out_counts);
  return vc::check_launch();
}

// RT = 2 (512 rows per pass) while its register budget holds (A fragments: 2*KS*4 VGPRs);
// long descriptors fall back to one row tile per wave.
template <bool FUSED>
int dispatch_pair(int ks, const void* prepared, const int32_t* counts, int n_tiles, int d,
                  const int32_t* pairs, int n_pairs, float max_ratio, float max_distance,
                  int cross_check, int n_max, uint32_t* out_matches, int32_t* out_counts,
                  int32_t* o_idx, int32_t* o_best, int32_t* o_second, hipStream_t stream) {
#define VC_CASE(K, R)                                                                           \
  case K:                                                                                       \
    return launch_pair<K, R, FUSED>(prepared, counts, n_tiles, d, pairs, n_pairs, max_ratio,    \
                                    max_distance, cross_check, n_max, out_matches, out_counts,  \
                                    o_idx, o_best, o_second, stream);
  switch (ks) {
    VC_CASE(2, 2) VC_CASE(4, 2) VC_CASE(8, 2) VC_CASE(12, 2) VC_CASE(16, 1) VC_CASE(24, 1) VC_CASE(32, 1)
    default: return VC_ERR_UNSUPPORTED;
  }
#undef VC_CASE
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

size_t vc_prepared_bytes(int n_images, int n_max, int d) {
  if (n_images <= 0 || n_max <= 0 || d <= 0) return 0;
  const int ks = pick_ks(d);
  if (ks < 0) return 0;
  return (size_t)n_images * image_bytes(tiles_of(n_max), ks);
}

int vc_prepare_descriptors(const uint8_t* desc, const int32_t* counts, int n_images, int n_max, int d,
                           void* prepared, vc_stream_t stream) {
  if (!desc || !prepared || n_images < 0 || n_max <= 0 || d <= 0) return VC_ERR_INVALID_ARG;
  if (n_max > VC_MAX_KEYPOINTS || d > VC_MAX_DESC_DIM) return VC_ERR_UNSUPPORTED;
  if (((uintptr_t)prepared) % 16 != 0) return VC_ERR_INVALID_ARG;
  if (n_images == 0) return VC_OK;
  const int ks = pick_ks(d);
  const int n_tiles = tiles_of(n_max);
  hipLaunchKernelGGL(prepare_kernel, dim3(n_tiles, n_images), dim3(256), 0, (hipStream_t)stream, desc,
                     counts, n_max, d, n_tiles, ks, (uint8_t*)prepared);
  return vc::check_launch();
}

int vc_match_pairs_u8(const void* prepared, const int32_t* counts, int n_images, int n_max, int d,
                      const int32_t* pairs, int n_pairs, float max_ratio, float max_distance,
                      int cross_check, uint32_t* out_matches, int32_t* out_counts, vc_stream_t stream) {
  if (!prepared || !counts || !pairs || !out_matches || !out_counts) return VC_ERR_INVALID_ARG;
  if (n_images <= 0 || n_max <= 0 || d <= 0 || n_pairs < 0) return VC_ERR_INVALID_ARG;
  if (n_max > VC_MAX_KEYPOINTS || d > VC_MAX_DESC_DIM) return VC_ERR_UNSUPPORTED;
  if (n_pairs == 0) return VC_OK;
#if VC_WAVES == 8 && !defined(VC_OLD_PAIR_KERNEL)
  int st2 = VC_ERR_UNSUPPORTED;
  switch (pick_ks(d)) {   // descriptors up to 384 bytes: the persistent kernel (two row tiles per wave)
#define VC_CASE2(K)                                                                                      \
  case K:                                                                                                \
    st2 = launch_pair2<K>(prepared, counts, tiles_of(n_max), d, pairs, n_pairs, max_ratio, max_distance, \
                          cross_check, n_max, out_matches, out_counts, (hipStream_t)stream);             \
    break;
    VC_CASE2(2) VC_CASE2(4) VC_CASE2(8) VC_CASE2(12)
#undef VC_CASE2
    default: break;
  }
  if (st2 != VC_ERR_UNSUPPORTED) return st2;   // (blocks too large for its LDS plan: the kernel with one workgroup per pair)
#endif
  return dispatch_pair<true>(pick_ks(d), prepared, counts, tiles_of(n_max), d, pairs, n_pairs, max_ratio,
                             max_distance, cross_check, n_max, out_matches, out_counts, nullptr,
                             nullptr, nullptr, (hipStream_t)stream);
}

// workspace: prepared copies of d1 and d2 (each as a 1-image set, same n_max), a 2-entry counts
// array and one (0,1) pair record.
size_t vc_knn_workspace_bytes(int n1, int n2, int d) {
  const int n_max = n1 > n2 ? n1 : n2;
  if (n_max <= 0 || d <= 0 || pick_ks(d) < 0) return 0;
  return 2 * image_bytes(tiles_of(n_max), pick_ks(d)) + 64;
}

int vc_knn_top2_u8(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int d, int32_t* out_idx,
                   int32_t* out_best, int32_t* out_second, void* workspace, size_t workspace_bytes,
                   vc_stream_t stream) {
  if (n1 < 0 || n2 < 0 || d <= 0) return VC_ERR_INVALID_ARG;
  if (n1 == 0) return VC_OK;
  if (!d1 || !out_idx || !out_best || !out_second || !workspace) return VC_ERR_INVALID_ARG;
  if (n2 > 0 && !d2) return VC_ERR_INVALID_ARG;
  const int n_max = n1 > n2 ? n1 : n2;
  if (n_max > VC_MAX_KEYPOINTS || d > VC_MAX_DESC_DIM) return VC_ERR_UNSUPPORTED;
  if (((uintptr_t)workspace) % 16 != 0) return VC_ERR_INVALID_ARG;
  if (workspace_bytes < vc_knn_workspace_bytes(n1, n2, d)) return VC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int ks = pick_ks(d), n_tiles = tiles_of(n_max);
  const size_t img = image_bytes(n_tiles, ks);
  uint8_t* ws = (uint8_t*)workspace;
  int32_t* meta = (int32_t*)(ws + 2 * img);
  hipLaunchKernelGGL(knn_setup_kernel, dim3(1), dim3(1), 0, s, meta, n1, n2);
  // each set is prepared as image 0 of its own 1-image block, with its own row count
  hipLaunchKernelGGL(prepare_kernel, dim3(n_tiles, 1), dim3(256), 0, s, d1, meta + 0, n1, d, n_tiles, ks, ws);
  if (n2 > 0)
    hipLaunchKernelGGL(prepare_kernel, dim3(n_tiles, 1), dim3(256), 0, s, d2, meta + 1, n2, d, n_tiles, ks,
                       ws + img);
  int st = vc::check_launch();
  if (st != VC_OK) return st;
  return dispatch_pair<false>(ks, ws, meta, n_tiles, d, meta + 2, 1, 0.f, 0.f, 0, n_max, nullptr, nullptr,
                              out_idx, out_best, out_second, s);
}

int vc_mutual_ratio(const int32_t* idx12, const int32_t* best12, const int32_t* second12, int n1,
                    const int32_t* idx21, const int32_t* best21, const int32_t* second21, int n2,
                    float max_ratio, float max_distance, int cross_check, uint32_t* out_pairs,
                    int32_t* out_count, vc_stream_t stream) {
  if (n1 < 0 || n2 < 0 || !out_count) return VC_ERR_INVALID_ARG;
  if (n1 > 0 && (!idx12 || !best12 || !second12 || !out_pairs)) return VC_ERR_INVALID_ARG;
  if (cross_check && n1 > 0 && n2 > 0 && (!idx21 || !best21 || !second21)) return VC_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mutual_ratio_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, idx12, best12,
                     second12, n1, idx21, best21, second21, n2, max_ratio, max_distance, cross_check,
                     out_pairs, out_count);
  return vc::check_launch();
}

int vc_theta_table(float* out, int n, vc_stream_t stream) {
  if (!out || n < 0) return VC_ERR_INVALID_ARG;
  if (n == 0) return VC_OK;
  hipLaunchKernelGGL(theta_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, n, 1);
  return vc::check_launch();
}

int vc_theta_eval(float* out, int n, vc_stream_t stream) {
  if (!out || n < 0) return VC_ERR_INVALID_ARG;
  if (n == 0) return VC_OK;
  hipLaunchKernelGGL(theta_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, n, 0);
  return vc::check_launch();
}

}  // extern "C"
