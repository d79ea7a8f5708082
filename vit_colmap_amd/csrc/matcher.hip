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

constexpr int kWaves = 8;                    // waves per workgroup (2 per SIMD)
constexpr int kThreads = kWaves * 64;        // 512
constexpr int kTile = 32;                    // MFMA tile edge
constexpr int kPassRows = kWaves * kTile;    // rows of image A covered per pass: 256
constexpr int kMaxN = VC_MAX_KEYPOINTS;      // 2048 -> at most 64 column tiles (6-bit code)
constexpr int kFragBytes = 1024;             // 64 lanes x 16 B

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int tiles_of(int n_max) { return ceil_div(n_max, kTile); }
__host__ __device__ inline int ksteps_of(int d) { return ceil_div(d, 32); }
__host__ __device__ inline size_t image_bytes(int n_tiles, int ks) {
  return (size_t)n_tiles * ks * kFragBytes + (size_t)n_tiles * kTile * sizeof(int32_t);
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
  for (int c = threadIdx.x; c < kTile; c += blockDim.x) {
    const int row = tile * kTile + c;
    int32_t sum = 0;
    if (row < count)
      for (int k = 0; k < d; ++k) sum += src[(size_t)row * d + k];
    rowsum[c] = sum;
  }
}

// ---------------------------------------------------------------------------------------
// The pair kernel
// ---------------------------------------------------------------------------------------
struct ColState {  // per column of image B, in LDS
  int best, second, idx;
};

template <int KS>
struct Smem {
  alignas(16) uint8_t ring[2][KS * kFragBytes];  // B tiles, fragment-major (glds destination)
  uint2 part[2][kWaves][kTile];                  // per-wave column partials of one tile
  int cterm[kMaxN];                              // 128 * rowsum_b[j]
  ColState col[kMaxN];
  int rbest[kMaxN], rsecond[kMaxN], ridx[kMaxN];  // row results (image A)
  int m21[kMaxN];
  int wave_count[kWaves];
};

// Issue the global->LDS copy of B tile `jt` into ring slot `slot`; fragments are dealt
// round-robin to the waves.  The destination is wave-uniform base + lane*16 (LDS-DMA rule).
template <int KS>
__device__ inline void stage_tile(const uint8_t* __restrict__ b_frags, int jt, uint8_t* slot,
                                  int wave, int lane) {
  const uint8_t* src = b_frags + (size_t)jt * KS * kFragBytes;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    if ((kk % kWaves) == wave) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + kk * kFragBytes + lane * 16),
          (__attribute__((address_space(3))) void*)(slot + kk * kFragBytes), 16, 0, 0);
    }
  }
}

template <int KS, bool FUSED>
__global__ __launch_bounds__(kThreads, 2) void pair_kernel(
    const uint8_t* __restrict__ prepared, const int32_t* __restrict__ counts, int n_tiles_img, int d,
    const int32_t* __restrict__ pairs, float max_ratio, float max_distance, int cross_check,
    int n_max, uint32_t* __restrict__ out_matches, int32_t* __restrict__ out_counts,
    // !FUSED: one-way outputs (rows of A against B)
    int32_t* __restrict__ o_idx, int32_t* __restrict__ o_best, int32_t* __restrict__ o_second) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  Smem<KS>& sm = *reinterpret_cast<Smem<KS>*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;

  const int p = blockIdx.x;
  const int img_a = pairs[2 * p], img_b = pairs[2 * p + 1];
  const int n1 = min(max(counts[img_a], 0), n_max);
  const int n2 = min(max(counts[img_b], 0), n_max);
  const size_t img_stride = image_bytes(n_tiles_img, KS);
  const uint8_t* a_frags = prepared + (size_t)img_a * img_stride;
  const uint8_t* b_frags = prepared + (size_t)img_b * img_stride;
  const int32_t* a_rowsum = (const int32_t*)(a_frags + (size_t)n_tiles_img * KS * kFragBytes);
  const int32_t* b_rowsum = (const int32_t*)(b_frags + (size_t)n_tiles_img * KS * kFragBytes);

  const int n_ct = ceil_div(n2, kTile);          // column tiles of B that hold valid rows
  const int n_pass = ceil_div(n1, kPassRows);

  // ---- per-pair LDS state ----------------------------------------------------------------
  for (int j = tid; j < n_ct * kTile; j += kThreads) {
    sm.cterm[j] = 128 * b_rowsum[j];
    sm.col[j].best = 0;
    sm.col[j].second = 0;
    sm.col[j].idx = -1;
  }
  if (n_ct > 0) stage_tile<KS>(b_frags, 0, sm.ring[0], wave, lane);
  const int bias = -16384 * d;

  for (int pass = 0; pass < n_pass; ++pass) {
    const int row_tile = pass * kWaves + wave;           // 32-row tile of A owned by this wave
    const bool tile_valid = row_tile * kTile < n1;       // wave-uniform
    // A fragments stay in registers for the whole pass
    v4i afrag[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      afrag[kk] = tile_valid ? *(const v4i*)(a_frags + ((size_t)row_tile * KS + kk) * kFragBytes + lane * 16)
                             : v4i{0, 0, 0, 0};
    }
    // accumulator start values: 128*rowsum_a[row] - 16384*D for the 16 rows of this lane
    int rinit[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
      rinit[r] = tile_valid ? 128 * a_rowsum[row_tile * kTile + lrow] + bias : 0;
    }
    u32 rbest[16], rsec[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rbest[r] = 0; rsec[r] = 0; }

    for (int jt = 0; jt < n_ct; ++jt) {
      __syncthreads();  // tile jt landed (vmcnt(0) + barrier); everyone is done with tile jt-1
      // next tile (wraps to tile 0 for the next pass)
      {
        const int nxt = (jt + 1 < n_ct) ? jt + 1 : 0;
        const bool more = (jt + 1 < n_ct) || (pass + 1 < n_pass);
        const int seq = pass * n_ct + jt + 1;
        if (more) stage_tile<KS>(b_frags, nxt, sm.ring[seq & 1], wave, lane);
      }
      const int seq = pass * n_ct + jt;
      // column partials of the previous tile are complete: one wave folds them into sm.col
      if (FUSED && seq > 0 && wave == ((seq - 1) % kWaves) && lane < kTile) {
        const int pseq = seq - 1;
        const int pjt = pseq % n_ct, ppass = pseq / n_ct;
        ColState cs = sm.col[pjt * kTile + lane];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          const uint2 pr = sm.part[pseq & 1][w][lane];
          const int sb = (int)(pr.x >> 6), ss = (int)(pr.y >> 6);
          const int row = ppass * kPassRows + w * kTile + (63 - (int)(pr.x & 63));
          if (sb > cs.best) { cs.second = max(cs.best, ss); cs.best = sb; cs.idx = row; }
          else { cs.second = max(cs.second, sb); }
        }
        sm.col[pjt * kTile + lane] = cs;
      }

      const uint8_t* slot = sm.ring[seq & 1];
      const int ct = sm.cterm[jt * kTile + c];
      v16i acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = rinit[r] + ct;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const v4i bfrag = *(const v4i*)(slot + kk * kFragBytes + lane * 16);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[kk], bfrag, acc, 0, 0, 0);
      }
      // top-2 updates
      const u32 jcode = 63u - (u32)jt;
      u32 cb = 0, cs2 = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const u32 t = tile_valid ? (u32)acc[r] : 0u;
        const u32 rk = (t << 6) | jcode;
        rsec[r] = umed3(rbest[r], rsec[r], rk);
        rbest[r] = umax(rbest[r], rk);
        if (FUSED) {
          const u32 ck = (t << 6) | (u32)(63 - ((r & 3) + 8 * (r >> 2)));
          cs2 = umed3(cb, cs2, ck);
          cb = umax(cb, ck);
        }
      }
      if (FUSED) {
        // rows of the upper half-wave are 4 further down: make the codes comparable
        cb -= 4u * h;
        cs2 -= 4u * h;
        const u32 ob = (u32)__shfl_xor((int)cb, 32), os = (u32)__shfl_xor((int)cs2, 32);
        const u32 nb = umax(cb, ob);
        const u32 ns = umax(umin(cb, ob), umax(cs2, os));
        if (h == 0) sm.part[seq & 1][wave][c] = make_uint2(nb, ns);
      }
    }  // column tiles

    // ---- row results of this pass: reduce over the 32 lanes that share a row ------------
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const u32 m = half_max(rbest[r]);
      const unsigned long long eq = __ballot(rbest[r] == m);
      const int win = h ? __builtin_ctz((u32)(eq >> 32)) : __builtin_ctz((u32)eq);  // lowest column wins
      const u32 x = (c == win) ? rsec[r] : rbest[r];
      const u32 s2 = half_max(x);
      if (c == 0) {
        const int row = row_tile * kTile + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < n1) {
          const int sb = (int)(m >> 6);
          const int idx = sb > 0 ? (63 - (int)(m & 63)) * kTile + win : -1;
          const int s2v = sb > 0 ? (int)(s2 >> 6) : 0;
          if (FUSED) { sm.rbest[row] = sb; sm.rsecond[row] = s2v; sm.ridx[row] = idx; }
          else { o_idx[row] = idx; o_best[row] = sb; o_second[row] = s2v; }
        }
      }
    }
  }  // passes

  if (!FUSED) return;

  // ---- fold the last tile's column partials ---------------------------------------------
  __syncthreads();
  const int total = n_pass * n_ct;
  if (total > 0 && wave == 0 && lane < kTile) {
    const int pseq = total - 1;
    const int pjt = pseq % n_ct, ppass = pseq / n_ct;
    ColState cs = sm.col[pjt * kTile + lane];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint2 pr = sm.part[pseq & 1][w][lane];
      const int sb = (int)(pr.x >> 6), ss = (int)(pr.y >> 6);
      const int row = ppass * kPassRows + w * kTile + (63 - (int)(pr.x & 63));
      if (sb > cs.best) { cs.second = max(cs.best, ss); cs.best = sb; cs.idx = row; }
      else { cs.second = max(cs.second, sb); }
    }
    sm.col[pjt * kTile + lane] = cs;
  }
  __syncthreads();

  // ---- angle + ratio tests, cross check, ordered compaction -----------------------------
  if (cross_check) {
    for (int j = tid; j < n2; j += kThreads) {
      const ColState cs = sm.col[j];
      sm.m21[j] = accept_dev(cs.best, cs.second, max_ratio, max_distance) ? cs.idx : -1;
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
      j = sm.ridx[i];
      ok = accept_dev(sm.rbest[i], sm.rsecond[i], max_ratio, max_distance);
      if (ok && cross_check) ok = (sm.m21[j] == i);
    }
    const unsigned long long mask = __ballot(ok);
    if (lane == 0) sm.wave_count[wave] = __popcll(mask);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += sm.wave_count[w];
    int chunk_total = 0;
    for (int w = 0; w < kWaves; ++w) chunk_total += sm.wave_count[w];
    if (ok) {
      const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
      out[2 * pos] = (uint32_t)i;
      out[2 * pos + 1] = (uint32_t)j;
    }
    base += chunk_total;
    __syncthreads();
  }
  if (tid == 0) out_counts[p] = base;
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

__global__ void theta_table_kernel(float* out, int n) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) out[s] = theta_dev(s);
}

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------
// K-step counts with a compiled kernel; a descriptor is zero-padded up to the next one.
constexpr int kKsList[] = {2, 4, 8, 12, 16, 24, 32};

inline int pick_ks(int d) {
  const int need = ksteps_of(d);
  for (int ks : kKsList)
    if (ks >= need) return ks;
  return -1;
}

template <int KS, bool FUSED>
int launch_pair(const void* prepared, const int32_t* counts, int n_tiles, int d, const int32_t* pairs,
                int n_pairs, float max_ratio, float max_distance, int cross_check, int n_max,
                uint32_t* out_matches, int32_t* out_counts, int32_t* o_idx, int32_t* o_best,
                int32_t* o_second, hipStream_t stream) {
  const size_t smem = sizeof(Smem<KS>);
  static thread_local bool configured = false;  // per instantiation
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)pair_kernel<KS, FUSED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return vc::fail(e);
    configured = true;
  }
  hipLaunchKernelGGL((pair_kernel<KS, FUSED>), dim3(n_pairs), dim3(kThreads), smem, stream,
                     (const uint8_t*)prepared, counts, n_tiles, d, pairs, max_ratio, max_distance,
                     cross_check, n_max, out_matches, out_counts, o_idx, o_best, o_second);
  return vc::check_launch();
}

template <bool FUSED>
int dispatch_pair(int ks, const void* prepared, const int32_t* counts, int n_tiles, int d,
                  const int32_t* pairs, int n_pairs, float max_ratio, float max_distance,
                  int cross_check, int n_max, uint32_t* out_matches, int32_t* out_counts,
                  int32_t* o_idx, int32_t* o_best, int32_t* o_second, hipStream_t stream) {
#define VC_CASE(K)                                                                              \
  case K:                                                                                       \
    return launch_pair<K, FUSED>(prepared, counts, n_tiles, d, pairs, n_pairs, max_ratio,       \
                                 max_distance, cross_check, n_max, out_matches, out_counts,     \
                                 o_idx, o_best, o_second, stream);
  switch (ks) {
    VC_CASE(2) VC_CASE(4) VC_CASE(8) VC_CASE(12) VC_CASE(16) VC_CASE(24) VC_CASE(32)
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
  hipLaunchKernelGGL(theta_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, n);
  return vc::check_launch();
}

}  // extern "C"
