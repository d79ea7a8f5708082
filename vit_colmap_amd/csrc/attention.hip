// Flash-style multi-head self-attention forward for the DINOv2 blocks (head_dim 64, bf16), written
// for 64-wide wavefronts and the gfx950 MFMA / LDS model.  Replaces the
// `scaled_dot_product_attention` call inside the model behind reference
// vit_colmap/features/vit_extractor.py:135-146 (torch.hub DINOv2 `forward_features`).
//
// Input  qkv [B][N][3][H][64] bf16 — exactly what the fused qkv GEMM writes, no permute/copy;
// output out [B][N][H*64] bf16   — exactly what the projection GEMM reads.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries.
//   * swapped product: S^T = K Q^T (v_mfma_f32_32x32x16_bf16, A = K fragment from LDS, B = Q
//     fragment kept in registers), so a lane holds 32 scores of ONE query (its partner lane,
//     l ^ 32, the other 32 of the 64-key block): the row max / row sum are lane-local plus one
//     cross-half exchange, no LDS round trip;
//   * the f32 S^T accumulators, converted pairwise to bf16, ARE the B operand of the second
//     product O^T = V^T P^T (accumulator tile as the next MFMA's operand: registers 8s..8s+7 are
//     k-step s, whose k index maps to key 16s + 8(j>>2) + 4h + (j&3));
//   * V stays row-major in LDS and is read with ds_read_b64_tr_b16 (hardware transpose): two reads
//     give a lane the 8 keys of its k-step for its d column — no V^T copy anywhere;
//   * K / V tiles of 64 keys are staged global -> registers -> LDS one block ahead (the loads are
//     issued before the block's MFMAs, the LDS writes after them), double buffered, one barrier
//     per block; both tiles are XOR-swizzled against bank conflicts (K: 16-byte slot ^ (key & 7)
//     for the ds_read_b128 column slices; V: byte bit 6 ^ bit 1 of the key for the transposed reads);
//   * online softmax in exp2 domain with the rescale skipped when no row maximum moved.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kHD = 64;        // head dim
constexpr int kWavesA = 4;     // waves per workgroup
constexpr int kQW = 32;        // queries per wave
constexpr int kQB = kWavesA * kQW;  // 128 queries per workgroup
constexpr int kKV = 64;        // keys per block
constexpr int kTileBytes = kKV * kHD * 2;  // 8 KiB

__device__ inline uint32_t k_off(int key, int slot) { return (uint32_t)key * 128u + (uint32_t)((slot ^ (key & 7)) << 4); }
__device__ inline uint32_t v_off(int key, int byte_in_row) {
  return (uint32_t)key * 128u + ((uint32_t)byte_in_row ^ (uint32_t)(((key >> 1) & 1) << 6));
}

// QT = 32-query tiles per wave.  QT = 2 reads every K / V fragment from LDS once for two MFMAs and
// stages every K / V tile once for 256 instead of 128 query rows (at 2 waves per SIMD instead of 3).
// LAZY: the queries arrive already multiplied by scale * log2(e) (folded into the qkv projection), and from the
// second key block on the running maximum is subtracted INSIDE the matrix product — it is the accumulator's
// initial value — so a probability is exp2(accumulator) with no v_fma_f32 per score, and the per-block maximum
// is only an integer test "some score exceeds the running maximum by more than kLazyTh" (float bit patterns of
// positive numbers order as integers).  Any reference value gives the same softmax after normalisation; the
// exact maximum is re-established (standard rescale) when the test fires, so probabilities stay <= 2^kLazyTh.
// The softmax over-subscribes the SIMD's issue port (~11 slots per MFMA gap against the ~6 that hide,
// profiles/r02_overlap_probe.md): removing one of the five VALU issue slots per score is a direct saving.
constexpr float kLazyTh = 6.0f;
#ifdef VC_ATTN_STAMP
// diagnostic build only (tools/stamp_attn.py): shader-clock totals per wave and phase, written over the wave's first output row
__device__ __forceinline__ unsigned long long stamp_a() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define VC_AST(acc_) { const unsigned long long t_ = stamp_a(); acc_ += t_ - st_tp; st_tp = t_; }
#else
#define VC_AST(acc_)
#endif
template <int QT, bool LAZY>
__global__ __launch_bounds__(256, (QT == 1 ? 3 : 2)) void attention_kernel(const __bf16* __restrict__ qkv,
                                                                           __bf16* __restrict__ out, int N, int H,
                                                                           float scale_log2e, int unit0, int nqb) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[3][2][kTileBytes];  // ring slot x [K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  // work unit = (batch * head, block of 128 QT query rows), numbered bh-major so that neighbouring workgroups share K / V
  const int unit = unit0 + (int)blockIdx.x;
  const int bh = unit / nqb, b = bh / H, h = bh - b * H;
  const int q_base = (unit - bh * nqb) * (kQB * QT);
  if (q_base >= N) return;   // (the second half of a ragged last block when the tail runs as 128-row units)
  const int q_row0 = q_base + wave * (kQW * QT) + r;   // + 32*qt
  const size_t tok_stride = (size_t)3 * H * kHD;  // elements between consecutive tokens
  const __bf16* base = qkv + (size_t)b * N * tok_stride + (size_t)h * kHD;

  // Q fragments (B operand): lane (query r, half hh) holds Q[q][16ks + 8hh .. +8]
  v8bf qf[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const __bf16* qp = base + (size_t)min(q_row0 + 32 * qt, N - 1) * tok_stride;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *(const v8bf*)(qp + 16 * ks + 8 * hh);
  }
  // The Q loads have landed before the first LDS-DMA copy is issued.  Left pending, the compiler puts its waits for them at
  // their first use — INSIDE the key-block loop, where they stay: s_waitcnt vmcnt(7) ... vmcnt(0) in front of the QK^T MFMAs
  // of EVERY block, and since it cannot see the copies issued from inline asm, that vmcnt(0) also waited for the K / V copy
  // issued a few instructions earlier for two blocks ahead (seen in the ISA with tools/isa_waits.sh; the trap of DESIGN.md
  // §4.1).  Here the copy was an L2 hit that the second wave of the SIMD covered — 237 vs 239 us per layer — but the ring
  // now really runs two blocks ahead.  (Not kept: K / V fragments through an explicit three-register window with
  // sched_barriers, 241 us: the compiler's own [read, wait, 2 MFMAs] chain interleaves better with the other wave.)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), other counters untouched (gfx9 encoding)

  // K / V tiles (64 keys) go global -> LDS by LDS-DMA, two blocks ahead, into a ring of three
  // slots.  A block is 16 pieces of 1 KiB (8 keys x 128 B); wave w issues pieces 4w..4w+3.  The
  // LDS destination of a piece is lane-linear, so the bank swizzles are applied to the per-lane
  // SOURCE address: LDS chunk c' of key k receives source chunk c' ^ (k & 7) (K) or
  // c' ^ (((k >> 1) & 1) << 2) (V).  Issued from inline asm and waited for with a counted vmcnt
  // (hipcc would drain them before every LDS read, and serialised the register-staged variant).
  const int n_blk = (N + kKV - 1) / kKV;
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0][0][0]);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // Per-lane byte offsets of this wave's four pieces inside a block of 64 keys, computed ONCE: a block's source is then
  // a wave-uniform base (SGPR pair, advanced by scalar instructions) plus these.  Computing the full 64-bit address per
  // lane, piece and block cost ~250 issue cycles per block (v_mul_lo_u32 / v_mad_u64_u32 are quarter-rate).
  uint32_t voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pce = wave_u * 4 + i;
    const int which = pce >> 3, kg = pce & 7;
    const int krow = lane >> 3, cdst = lane & 7;
    const int key = kg * 8 + krow;
    const int csrc = which == 0 ? (cdst ^ krow) : (cdst ^ (((krow >> 1) & 1) << 2));
    voff[i] = (uint32_t)(((size_t)key * tok_stride + (size_t)(1 + which) * H * kHD + csrc * 8) * sizeof(__bf16));
  }
  const size_t blk_bytes = (size_t)kKV * tok_stride * sizeof(__bf16);
  auto issue_block = [&](int blk, int slot) {
    blk = min(blk, n_blk - 1);   // past the end: re-stage the last block into a slot nobody reads
    if ((blk + 1) * kKV <= N) {  // every key of the block exists: uniform base + precomputed offsets
      const char* sbase = (const char*)base + (size_t)blk * blk_bytes;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pce = wave_u * 4 + i;
        const uint32_t dst = lds0 + (uint32_t)slot * (2 * kTileBytes) + (uint32_t)(pce >> 3) * kTileBytes + (uint32_t)(pce & 7) * 1024u;
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[i]), "s"(sbase), "s"(dst)
            : "memory");
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // the ragged last block: keys past the end are clamped to the last token
      const int pce = wave_u * 4 + i;
      const int which = pce >> 3, kg = pce & 7;
      const int krow = lane >> 3, cdst = lane & 7;
      const int key = kg * 8 + krow;
      const int csrc = which == 0 ? (cdst ^ krow) : (cdst ^ (((krow >> 1) & 1) << 2));
      const int tok = min(blk * kKV + key, N - 1);
      const __bf16* src = base + (size_t)tok * tok_stride + (size_t)(1 + which) * H * kHD + csrc * 8;
      const uint32_t dst = lds0 + (uint32_t)slot * (2 * kTileBytes) + (uint32_t)which * kTileBytes + (uint32_t)kg * 1024u;
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
  issue_block(0, 0);
  issue_block(1, 1);

  v16f acc_o[QT][2];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -1e30f;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_o[qt][dt][i] = 0.f;
  }

  // transposed-read addressing: 16-lane group g, lane i of the group supplies row q = i>>2,
  // columns 4p..4p+3 (p = i&3) of a 4-key x 16-d block
  const int grp = lane >> 4, gi = lane & 15, tq = gi >> 2, tp = gi & 3;

#ifdef VC_ATTN_STAMP
  unsigned long long st_wait = 0, st_dma = 0, st_qk = 0, st_sm = 0, st_pv = 0;
  const unsigned long long st_t0 = stamp_a();
  unsigned long long st_tp = st_t0;
#endif
  int slot = 0;
  for (int blk = 0; blk < n_blk; ++blk) {
    // this wave's 4 pieces of block blk have landed when at most the 4 of block blk+1 are pending
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    VC_AST(st_wait)
    // the slot read in the previous iteration is free now: refill it two blocks ahead
    issue_block(blk + 2, slot == 0 ? 2 : slot - 1);
    VC_AST(st_dma)
    const uint8_t* kt = lds[slot][0];
    const uint8_t* vt = lds[slot][1];

    // ---- S^T = K Q^T : two 32-key tiles (each K fragment feeds QT MFMAs) -----------------------
    v16f acc_s[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      // LAZY: the running maximum of the lane's query (log2 units) is subtracted by the MFMA itself
      float c0 = (LAZY && blk > 0) ? -m_run[qt] : 0.f;
      asm volatile("" : "+v"(c0));   // one select, then plain moves (left alone the compiler emits a v_cndmask per register)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_s[qt][t][i] = c0;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = t * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const v8bf kf = *(const v8bf*)(kt + k_off(key, 2 * ks + hh));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          acc_s[qt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qt][ks], acc_s[qt][t], 0, 0, 0);
      }
    }
    // keys past the end of the sequence (last block only)
    if (blk == n_blk - 1 && (N & (kKV - 1)) != 0) {
      const int kv0 = blk * kKV;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kv0 + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (key >= N) acc_s[qt][t][i] = -1e30f;
          }
    }

    VC_AST(st_qk)
    // ---- online softmax (exp2 domain); a lane and its partner (l ^ 32) share one query ------
    v8bf pf[QT][4];  // P^T as B operand: k-step s <- registers 8(s&1)..+7 of tile s>>1
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (LAZY && blk > 0) {
        // scores are relative to the running maximum already; integer test for "one of them is too large"
        int imax = __float_as_int(acc_s[qt][0][0]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) imax = max(imax, __float_as_int(acc_s[qt][t][i]));
        if (!__any(imax > __float_as_int(kLazyTh))) {
#ifdef VC_ATTN_PKSUM
          typedef float v2f __attribute__((ext_vector_type(2)));
          v2f lsum2 = {0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int t = s >> 1, r0 = 8 * (s & 1);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const v2f pj = {__builtin_amdgcn_exp2f(acc_s[qt][t][r0 + j]), __builtin_amdgcn_exp2f(acc_s[qt][t][r0 + j + 1])};
              lsum2 += pj;
              pf[qt][s][j] = (__bf16)pj[0];
              pf[qt][s][j + 1] = (__bf16)pj[1];
            }
          }
          l_run[qt] += lsum2[0] + lsum2[1];
#else
          float lsum[4] = {0.f, 0.f, 0.f, 0.f};   // four independent chains: an add never waits for its predecessor
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int t = s >> 1, r0 = 8 * (s & 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float pj = __builtin_amdgcn_exp2f(acc_s[qt][t][r0 + j]);
              lsum[j & 3] += pj;
              pf[qt][s][j] = (__bf16)pj;
            }
          }
          l_run[qt] += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
#endif
          continue;
        }
        // rare: back to absolute scores, then the standard update below
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_s[qt][t][i] += m_run[qt];
      }
      float mloc = acc_s[qt][0][0];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, acc_s[qt][t][i]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
      const float m_new = fmaxf(m_run[qt], mloc);
      if (!__all(m_new == m_run[qt])) {  // some row maximum moved: rescale the running state
        const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * scale_log2e);
        l_run[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[qt][dt][i] *= alpha;
        m_run[qt] = m_new;
      }
      const float mb = m_run[qt] * scale_log2e;
      float lsum = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int t = s >> 1, r0 = 8 * (s & 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // one v_fma_f32 + one v_exp_f32 (exp2f() would add range scaling: 5 instructions)
          const float pj = __builtin_amdgcn_exp2f(__builtin_fmaf(acc_s[qt][t][r0 + j], scale_log2e, -mb));
          lsum += pj;
          pf[qt][s][j] = (__bf16)pj;
        }
      }
      l_run[qt] += lsum;
    }

    VC_AST(st_sm)
    // ---- O^T += V^T P^T : two 32-d tiles x four 16-key k-steps (each V fragment feeds QT MFMAs) --
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dbase = 16 * (grp & 1) + 32 * dt;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kbase = (s >> 1) * 32 + 16 * (s & 1) + 4 * hh;   // keys of elements j = 0..3
        const int key0 = kbase + tq, key1 = kbase + 8 + tq;        // ... and of j = 4..7
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(vt + v_off(key0, (dbase + 4 * tp) * 2)));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(vt + v_off(key1, (dbase + 4 * tp) * 2)));
        v8bf vf;
        *(v4s*)&vf = lo;
        *((v4s*)&vf + 1) = hi;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qt][s], acc_o[qt][dt], 0, 0, 0);
      }
    }
    slot = slot == 2 ? 0 : slot + 1;
    VC_AST(st_pv)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the two clamped refills before exiting
#ifdef VC_ATTN_STAMP
  const unsigned long long st_total = stamp_a() - st_t0;
#endif

  // ---- normalise and store: lane (query r, half hh) holds d = (i&3) + 8(i>>2) + 4hh + 32dt -------
  // Stored from that layout (16 stores of 8 bytes per lane, 32 rows per instruction) the tail of a unit is bound by store
  // issue.  Instead v_permlane32_swap gives every lane 16 consecutive d of its query (as in the GEMM epilogues), the wave
  // turns its QT x 32 queries x 64 d through 4 KiB x QT of the K/V ring — idle once every wave is past its last key block —
  // and writes whole rows: 8 queries x 128 contiguous bytes (one head's 64 d) per store instruction, 4 QT per lane.
  typedef uint32_t v4u32a __attribute__((ext_vector_type(4)));
  __syncthreads();
  uint8_t* const stg = (uint8_t*)lds + wave * (QT * 4096);
  auto stg_at = [&](int row, int slot16) { return stg + row * 128 + ((slot16 ^ ((row >> 1) & 7)) << 4); };
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      uint32_t ep[8];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const v2bf lo = {(__bf16)(acc_o[qt][dt][4 * g4] * inv), (__bf16)(acc_o[qt][dt][4 * g4 + 1] * inv)};
        const v2bf hi = {(__bf16)(acc_o[qt][dt][4 * g4 + 2] * inv), (__bf16)(acc_o[qt][dt][4 * g4 + 3] * inv)};
        ep[2 * g4] = *(const uint32_t*)&lo;
        ep[2 * g4 + 1] = *(const uint32_t*)&hi;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {       // groups (0, 2) and (1, 3) change hands: hh = 0 ends with d 0..15 of the block, hh = 1 with 16..31
        const auto sw = __builtin_amdgcn_permlane32_swap(ep[k], ep[k + 4], false, false);
        ep[k] = sw[0];
        ep[k + 4] = sw[1];
      }
      const int row = 32 * qt + r;
      *(v4u32a*)stg_at(row, 4 * dt + 2 * hh) = (v4u32a){ep[0], ep[1], ep[4], ep[5]};
      *(v4u32a*)stg_at(row, 4 * dt + 2 * hh + 1) = (v4u32a){ep[2], ep[3], ep[6], ep[7]};
    }
  }
  {
    const int lrow = lane >> 3, lslot = lane & 7;
    const int q_wave0 = q_base + wave * (kQW * QT);
#pragma unroll
    for (int i = 0; i < 4 * QT; ++i) {
      const int row = 8 * i + lrow;
      const v4u32a o = *(const v4u32a*)stg_at(row, lslot);
      if (q_wave0 + row < N)
        *(v4u32a*)(out + ((size_t)b * N + q_wave0 + row) * (size_t)H * kHD + (size_t)h * kHD + lslot * 8) = o;
    }
  }
#ifdef VC_ATTN_STAMP
  __syncthreads();   // after every wave's real stores: the wave's first output row now carries its stamps instead
  if (lane == 0) {
    uint32_t* dbg = (uint32_t*)(out + ((size_t)b * N + min(q_base + wave * (kQW * QT), N - 1)) * (size_t)H * kHD + (size_t)h * kHD);
    dbg[0] = (uint32_t)st_wait; dbg[1] = (uint32_t)st_dma; dbg[2] = (uint32_t)st_qk; dbg[3] = (uint32_t)st_sm;
    dbg[4] = (uint32_t)st_pv; dbg[5] = (uint32_t)st_total; dbg[6] = (uint32_t)n_blk; dbg[7] = 0x5354414du;
  }
#endif
}

}  // namespace

extern "C" {

int vc_attention_bf16(const void* qkv, int batch, int n_tokens, int n_heads, int head_dim, int q_prescaled, void* out,
                      vc_stream_t stream) {
  if (!qkv || !out || batch < 0 || n_tokens <= 0 || n_heads <= 0) return VC_ERR_INVALID_ARG;
  if (head_dim != kHD) return VC_ERR_UNSUPPORTED;
  if ((((uintptr_t)qkv) | ((uintptr_t)out)) % 16 != 0) return VC_ERR_INVALID_ARG;
  if (batch == 0) return VC_OK;
  // q_prescaled: the q rows already carry 1/sqrt(64) * log2(e) (folded into the qkv projection)
  const float scale_log2e = q_prescaled ? 1.0f : 0.125f * 1.4426950408889634f;
  // two query tiles per wave once the sequence is long enough to fill the chip with 256-row blocks
  int qt = n_tokens >= 512 ? 2 : 1;
  if (const char* e = getenv("VITCOLMAP_ATTN_QT")) qt = atoi(e) == 1 ? 1 : 2;   // developer A/B switch
  const __bf16* pq = (const __bf16*)qkv;
  __bf16* po = (__bf16*)out;
  hipStream_t st = (hipStream_t)stream;
  const int bh = batch * n_heads;
#define VC_ATT(Q, L, UNITS, UNIT0, NQB) \
  hipLaunchKernelGGL((attention_kernel<Q, L>), dim3(UNITS), dim3(256), 0, st, pq, po, n_tokens, n_heads, scale_log2e, UNIT0, NQB)
  if (qt == 1) {
    const int nqb = (n_tokens + kQB - 1) / kQB;
    if (q_prescaled) VC_ATT(1, true, nqb * bh, 0, nqb); else VC_ATT(1, false, nqb * bh, 0, nqb);
    return vc::check_launch();
  }
  // (Running the units of the last, partial "round" — 1800 units over 512 resident workgroups — as half-size QT = 1 units was
  // measured: 241.5 vs 239.0 us, no gain; workgroups do not advance in lockstep rounds.  One launch.)
  const int nqb2 = (n_tokens + 2 * kQB - 1) / (2 * kQB);
  if (q_prescaled) VC_ATT(2, true, nqb2 * bh, 0, nqb2); else VC_ATT(2, false, nqb2 * bh, 0, nqb2);
#undef VC_ATT
  return vc::check_launch();
}

}  // extern "C"
