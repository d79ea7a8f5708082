// Developer tool: what limits the L2 -> LDS weight stream of the x-stationary kernels (csrc/gemm.hip)?
// Every CU runs one 8-wave workgroup that streams a 2.25 MiB buffer (L2 resident, the size of one block's MLP weights)
// into an LDS ring by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces, PIECES per wave and stage, 8*PIECES KiB per stage),
// DEPTH stages in flight, one counted vmcnt + one workgroup barrier per stage — the loop skeleton of xs_kernel / mlp2_kernel
// without the MFMAs.  Optionally every wave also reads the stage back with ds_read_b128 (READS per wave), as the MFMA
// operand fetch does.  Output: bytes per clock and CU as a function of the bytes in flight.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_probe.hip -o tools/exp/dma_probe && tools/exp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// MF: 0 = no MFMA; 1 = one v_mfma_f32_32x32x16_bf16 per fragment read, all into ONE accumulator (xs_kernel's chain);
// 4 = as 1 with the MFMA phases of waves 0-3 and 4-7 (the two waves of every SIMD) separated by a second barrier;
// 2 = alternating between two accumulators; 3 = one accumulator, 24 distinct B operands; RD = fragments read ahead of the MFMA that consumes them
template <int DEPTH, int PIECES, int READS, bool BARRIER, int MF = 0, int RD = 8>
__global__ __launch_bounds__(512, 1) void k(const uint8_t* __restrict__ w, size_t w_bytes, int stages, uint32_t* out) {
  constexpr int StageBytes = 8 * PIECES * 1024;
  constexpr int NS = DEPTH + 1;
  __shared__ __attribute__((aligned(1024))) uint8_t lds[NS * StageBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)&lds[0]);
  const size_t n_stage_src = w_bytes / StageBytes;
  size_t src_stage = (blockIdx.x * 7) % n_stage_src;      // CUs start at different places, as row tiles do
  int islot = 0;
  auto issue = [&]() {
    const uint8_t* src = w + src_stage * StageBytes + (size_t)(wave * PIECES) * 1024 + lane * 16;
    const uint32_t dst = lds0 + (uint32_t)islot * StageBytes + (uint32_t)(wave * PIECES) * 1024u;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + i * 1024), "s"(dst + (uint32_t)i * 1024u) : "memory");
    }
    src_stage = src_stage + 1 == n_stage_src ? 0 : src_stage + 1;
    islot = islot + 1 == NS ? 0 : islot + 1;
  };
  for (int j = 0; j < DEPTH; ++j) issue();
  v4u sink = {0, 0, 0, 0};
  v16f acc0 = {0}, acc1 = {0};
  v8bf xb;
#pragma unroll
  for (int j = 0; j < 8; ++j) xb[j] = (__bf16)(0.01f * (float)(j + (threadIdx.x & 7)));
  v8bf xbv[24];                                         // MF == 3: 24 distinct B operands (96 VGPRs), as the x rows of xs_kernel
#pragma unroll
  for (int q = 0; q < 24; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) xbv[q][j] = (__bf16)(0.001f * (float)(q * 8 + j + (threadIdx.x & 15)) + (float)stages * 1e-6f);
  int slot = 0;
  uint64_t tp = __builtin_amdgcn_s_memtime();
  uint32_t seg[3] = {0, 0, 0};
#define ST(k_) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); seg[k_] += (uint32_t)(t_ - tp); tp = t_; }
  for (int i = 0; i < stages; ++i) {
    // the oldest stage has landed when only the (DEPTH - 1) younger ones are pending
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES) : "memory");
    if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ST(0)
    issue();                                            // refills the slot freed in the previous iteration
    ST(1)
    const uint8_t* st = lds + slot * StageBytes + lane * 16;
    if (MF == 4 && wave >= 4) asm volatile("s_barrier" ::: "memory");   // late half: MFMA phase after the early half's
    if (MF == 0) {
#pragma unroll
      for (int q = 0; q < READS; ++q) {
        const v4u v = *(const v4u*)(st + (q % (8 * PIECES)) * 1024);
        sink[0] ^= v[0]; sink[1] ^= v[1]; sink[2] ^= v[2]; sink[3] ^= v[3];
      }
    } else {
      v8bf wf[READS > 0 ? READS : 1];
#pragma unroll
      for (int q = 0; q < RD && q < READS; ++q) wf[q] = *(const v8bf*)(st + (q % (8 * PIECES)) * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < READS; ++q) {
        if (MF == 2 && (q & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q], xb, acc1, 0, 0, 0);
        else if (MF == 3) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q], xbv[q % 24], acc0, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q], xb, acc0, 0, 0, 0);
        if (q + RD < READS) wf[q + RD] = *(const v8bf*)(st + ((q + RD) % (8 * PIECES)) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MF) asm volatile("s_nop 0" :: "v"(acc0), "v"(acc1));
    if (MF == 4 && wave < 4) asm volatile("s_barrier" ::: "memory");    // early half: done, release the late half
    ST(2)
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && blockIdx.x < 256) {                  // per-wave segment totals: wait+barrier, issue, reads+MFMAs
    uint32_t* o = out + 256 * 512 + (blockIdx.x * 8 + wave) * 4;
    o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2];
  }
  if (MF) { sink[0] ^= __float_as_uint(acc0[3] + acc1[5]); }
  // a store the compiler cannot prove dead (and that practically never happens)
  if ((sink[0] ^ sink[1]) == (uint32_t)stages * 0x9e3779b9u + 12345u) out[blockIdx.x * 512 + threadIdx.x] = sink[2] ^ sink[3];
}

template <int DEPTH, int PIECES, int READS, bool BARRIER, int MF = 0, int RD = 8>
void run(const uint8_t* w, size_t w_bytes, uint32_t* out, int cus) {
  const int stages = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<DEPTH, PIECES, READS, BARRIER, MF, RD>), dim3(cus), dim3(512), 0, 0, w, w_bytes, stages, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_cu = (double)stages * 8 * PIECES * 1024;
  const double ghz = 2.2;   // typical clock under this load (GRBM_GUI_ACTIVE in the PMC runs)
  static uint32_t hseg[256 * 8 * 4];
  hipMemcpy(hseg, out + 256 * 512, sizeof(hseg), hipMemcpyDeviceToHost);
  double sg[3] = {0, 0, 0};
  for (int b = 0; b < cus; ++b) for (int w_ = 0; w_ < 8; ++w_) for (int k_ = 0; k_ < 3; ++k_) sg[k_] += hseg[(b * 8 + w_) * 4 + k_];
  for (int k_ = 0; k_ < 3; ++k_) sg[k_] /= (double)cus * 8 * stages;
  printf("CUs %3d  stage %2d KiB  in flight %3d KiB  reads/wave %2d  barrier %d  mfma %d rd %d : %7.3f ms  %6.2f us/stage  %6.1f B/clk/CU  %6.2f TB/s aggregate | per stage and wave (s_memtime ticks): wait+barrier %5.0f  issue %5.0f  reads+mfma %5.0f\n",
         cus, 8 * PIECES, DEPTH * 8 * PIECES, READS, (int)BARRIER, MF, RD, ms, ms * 1e3 / stages, bytes_cu / (ms * 1e-3) / (ghz * 1e9),
         bytes_cu * cus / (ms * 1e-3) / 1e12, sg[0], sg[1], sg[2]);
}

int main(int argc, char** argv) {
  const size_t w_bytes = 2304 * 1024;
  uint8_t* w; uint32_t* out;
  hipMalloc(&w, w_bytes); hipMemset(w, 1, w_bytes);
  hipMalloc(&out, 256 * 512 * 4 + 256 * 8 * 4 * 4);
  const bool hot = argc > 1 && argv[1][0] == 'h';       // "hot": the skeleton on weight buffers of 288 KiB .. 2.25 MiB (proj .. MLP size)
  if (hot) {
    for (size_t kb : {288, 864, 1152, 2304}) {
      printf("weight buffer %zu KiB: ", kb);
      run<2, 3, 24, true, 1, 8>(w, kb * 1024, out, 256);
    }
    return 0;
  }
  const bool wide = argc > 1;                           // any other argument: the 48 KiB-stage variants only
  for (int cus : {256, 64}) {
    if (wide) {
      if (cus != 256) continue;
      run<2, 3, 24, true, 1, 8>(w, w_bytes, out, cus);    // reference: 24 KiB stages
      run<1, 6, 48, true, 1, 8>(w, w_bytes, out, cus);    // 48 KiB stages (two feature blocks per barrier), one in flight
      run<2, 6, 48, true, 1, 8>(w, w_bytes, out, cus);    // ... two in flight (144 KiB of ring)
      run<1, 6, 48, true, 2, 8>(w, w_bytes, out, cus);    // ... two accumulators
      run<2, 3, 24, true, 4, 8>(w, w_bytes, out, cus);    // 24 KiB stages, the two waves of a SIMD strictly one after the other
      run<2, 3, 24, true, 4, 12>(w, w_bytes, out, cus);
      continue;
    }
    run<1, 3, 0, true>(w, w_bytes, out, cus);
    run<2, 3, 0, true>(w, w_bytes, out, cus);
    run<4, 3, 0, true>(w, w_bytes, out, cus);
    run<2, 6, 0, true>(w, w_bytes, out, cus);
    run<2, 3, 0, false>(w, w_bytes, out, cus);
    run<2, 3, 24, true>(w, w_bytes, out, cus);          // + the operand reads of 8 waves
    run<2, 3, 24, true, 1, 8>(w, w_bytes, out, cus);    // + the MFMAs: xs_kernel's loop skeleton
    run<2, 3, 24, true, 2, 8>(w, w_bytes, out, cus);    // two accumulators
    run<2, 3, 24, true, 3, 8>(w, w_bytes, out, cus);    // 24 distinct B operands
    run<2, 3, 24, true, 1, 4>(w, w_bytes, out, cus);
    run<2, 3, 24, true, 1, 12>(w, w_bytes, out, cus);
    run<2, 3, 24, false, 1, 8>(w, w_bytes, out, cus);   // no workgroup barrier (not a legal kernel: the ring is unprotected)
    run<4, 3, 24, true, 1, 8>(w, w_bytes, out, cus);
    run<2, 3, 0, true, 1, 8>(w, w_bytes, out, cus);     // DMA + barrier only with MF set: no reads, no MFMAs (control)
  }
  return 0;
}
