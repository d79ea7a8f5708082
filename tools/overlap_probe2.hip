// Developer tool: how many single-issue VALU instructions hide in the gap of a v_mfma_f32_32x32x16_bf16?
// Re-run of tools/overlap_probe.hip the way VERDICT r01 (weak #8) asks: INDEPENDENT accumulators (16 per wave, every
// filler touches a different one than its neighbours: no dependency chain shorter than a whole MFMA gap), fillers written
// in asm (nothing is merged, reordered or packed by the compiler), 1 and 2 waves per SIMD, 0 / 2 / 4 / 5 / 6 / 8 fillers
// per gap, kinds: integer add, float fma, and the "four float + one v_exp_f32" mix of an online-softmax loop.
// Time is the shader clock (s_memtime) inside the kernel: cycles per MFMA and SIMD, nothing host-side.
//   expectation (MI355X_MICROARCH.md, rows "vector-instruction ISSUE cost" and "HIDDEN per gap"): an MFMA holds the
//   issue port for 8 of its 32 cycles, an ordinary VALU instruction for 4, v_exp_f32 for 8; the gap costs
//   max(32, 8 + sum) per wave's MFMA, so <= 5 fillers (one of them an exp) should be free at one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int NM = 32;   // MFMAs per wave and iteration

template <int FILL, int KIND, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(unsigned long long* cyc, float* sink, int iters, float seed) {
  v8bf a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (float)(j + (threadIdx.x & 7)) * seed); b[j] = (__bf16)(0.02f * (float)(j + 1)); }
  v16f acc0 = {0}, acc1 = {0};
  float f[16];
  int v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { f[j] = 1.0f + 1e-3f * (float)(j + (int)threadIdx.x) * seed; v[j] = j + (int)threadIdx.x; }
  const float c = 1.0f - 1e-6f * seed, d = 1e-3f * seed;
  const int vi = (int)seed + 3;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NM; ++u) {
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < FILL; ++j) {
        const int r = (u * FILL + j) % 16;            // 16 independent chains: the same register again after 16 / FILL gaps
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[r]) : "v"(vi));
        else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r]) : "v"(c), "v"(d));
        else if (KIND == 2) {                          // softmax-like mix: one transcendental in five
          if (j % 5 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f[r]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r]) : "v"(c), "v"(d));
        } else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[r]) : "v"(c));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += f[j] + (float)v[j] + acc0[j] + acc1[j];
  sink[blockIdx.x * THREADS + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int FILL, int KIND, int THREADS>
double run(unsigned long long* cyc, float* sink) {
  const int iters = 400, waves = 256 * THREADS / 64;
  k<FILL, KIND, THREADS><<<256, THREADS>>>(cyc, sink, 20, 1.0f);
  hipDeviceSynchronize();
  k<FILL, KIND, THREADS><<<256, THREADS>>>(cyc, sink, iters, 1.5f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(waves);
  hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per_wave_mfma = (double)h[waves / 2] / ((double)iters * NM);   // median wave: cycles per own MFMA
  return per_wave_mfma / (THREADS / 256);                                      // per MFMA of the SIMD (2 waves share it)
}

template <int KIND, int THREADS>
void row(const char* name, unsigned long long* cyc, float* sink) {
  printf("| %-28s | %d | %6.1f | %6.1f | %6.1f | %6.1f | %6.1f | %6.1f |\n", name, THREADS / 256,
         run<0, KIND, THREADS>(cyc, sink), run<2, KIND, THREADS>(cyc, sink), run<4, KIND, THREADS>(cyc, sink),
         run<5, KIND, THREADS>(cyc, sink), run<6, KIND, THREADS>(cyc, sink), run<8, KIND, THREADS>(cyc, sink));
}

int main() {
  unsigned long long* cyc; float* sink;
  hipMalloc(&cyc, 8 * 4096); hipMalloc(&sink, 4 * 512 * 256);
  printf("cycles per v_mfma_f32_32x32x16_bf16 AND SIMD (median wave, s_memtime; 32.0 = the matrix pipe is the only limit)\n");
  printf("| filler kind (per MFMA gap and wave) | waves/SIMD | 0 | 2 | 4 | 5 | 6 | 8 |\n|---|---|---|---|---|---|---|---|\n");
  row<0, 256>("v_add_u32", cyc, sink);              row<0, 512>("v_add_u32", cyc, sink);
  row<1, 256>("v_fma_f32", cyc, sink);              row<1, 512>("v_fma_f32", cyc, sink);
  row<2, 256>("4 v_fma_f32 : 1 v_exp_f32", cyc, sink); row<2, 512>("4 v_fma_f32 : 1 v_exp_f32", cyc, sink);
  row<3, 256>("v_cvt_pk_bf16_f32", cyc, sink);      row<3, 512>("v_cvt_pk_bf16_f32", cyc, sink);
  return 0;
}
