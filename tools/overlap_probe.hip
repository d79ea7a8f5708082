// Developer tool: do the matrix pipe and the vector ALU of one SIMD overlap across its two waves?
//   mode 0: every wave runs NM back-to-back int8 MFMAs per iteration
//   mode 1: every wave runs NV dependent-free v_add3/v_max VALU instructions per iteration
//   mode 2: waves 0-3 run the MFMA body, waves 4-7 the VALU body (partner waves of each SIMD)
//   mode 3: every wave runs both bodies back to back (MFMA cluster, then VALU cluster)
//   mode 4: every wave interleaves: 1 MFMA, NV/NM VALU, 1 MFMA, ...
// If the pipes overlap across waves, t(2) ~ max(t(0), t(1)); if a SIMD runs one wave's cluster at a time, t(2) ~ t(0)+t(1).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int NM = 24;
template <int NV, int MODE, int KIND>   // KIND 0: int add3/max, 1: f32 fma, 2: f32 fma + exp (1 in 4), 3: bf16 MFMA + f32 fma
__global__ __launch_bounds__(512, 1) void k(int* out, int iters, int seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v4i a = {seed + (int)threadIdx.x * 0x01010101, seed * 3 + 7, (int)threadIdx.x * 0x00030507, seed ^ 0x55aa55aa};
  v4i b = {seed * 5 + 1, (int)threadIdx.x * 0x01020304, seed + 99, (int)blockIdx.x * 0x07070707 + 1};
  v16i acc = {0};
  int v[8];
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = seed + j + threadIdx.x; f[j] = 1.0f + 1e-3f * (float)(seed + j + (int)threadIdx.x); }
  const float fc = 1.0f - 1e-6f * (float)seed;
  auto valu = [&](int j) {
    if (KIND == 0) v[j & 7] = max(v[j & 7] + seed + j, v[(j + 1) & 7]);
    else if (KIND == 1 || KIND == 3) { f[j & 7] = __builtin_fmaf(f[j & 7], fc, f[(j + 1) & 7]); f[(j + 3) & 7] = __builtin_fmaf(f[(j + 3) & 7], fc, 0.5f); }
    else { if ((j & 3) == 0) f[j & 7] = __builtin_amdgcn_exp2f(f[j & 7] * 1e-3f); else f[j & 7] = __builtin_fmaf(f[j & 7], fc, f[(j + 1) & 7]); f[(j + 3) & 7] = __builtin_fmaf(f[(j + 3) & 7], fc, 0.5f); }
  };
  typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
  typedef float v16f __attribute__((ext_vector_type(16)));
  v8bf ab, bb;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(0.01f * (float)(j + (threadIdx.x & 7))); bb[j] = (__bf16)(0.02f * (float)(j + 1)); }
  v16f accf = {0};
  auto mfma = [&]() {
    if (KIND == 3) accf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, accf, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
  };
  const bool do_m = MODE == 0 || MODE == 3 || MODE == 4 || (MODE == 2 && wave < 4);
  const bool do_v = MODE == 1 || MODE == 3 || MODE == 4 || (MODE == 2 && wave >= 4);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 4) {
      constexpr int PER = NV / NM;
#pragma unroll
      for (int u = 0; u < NM; ++u) {
        mfma();
#pragma unroll
        for (int j = 0; j < PER; ++j) valu(u * PER + j);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int u = 0; u < NM; ++u) mfma();
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_v) {
#pragma unroll
        for (int j = 0; j < NV; ++j) valu(j);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  int s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int j = 0; j < 8; ++j) s ^= v[j] ^ __float_as_int(f[j]);
  for (int r = 0; r < 16; ++r) s ^= __float_as_int(accf[r]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int MODE, int KIND>
float run(int* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  k<NV, MODE, KIND><<<256, 512>>>(out, 50, 1); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 3; ++rep) k<NV, MODE, KIND><<<256, 512>>>(out, iters, rep + 3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 3 / iters * 1e6f;   // ns per iteration
}
template <int NV, int KIND>
void sweep(int* out) {
  printf("kind %d  NM=%d MFMA, NV=%d VALU statements per iteration: ns/iter  mfma-only %.0f  valu-only %.0f  split-waves %.0f  clustered-in-wave %.0f  interleaved-in-wave %.0f\n",
         KIND, NM, NV, run<NV, 0, KIND>(out), run<NV, 1, KIND>(out), run<NV, 2, KIND>(out), run<NV, 3, KIND>(out), run<NV, 4, KIND>(out));
}
int main() {
  int* out; hipMalloc(&out, 4 * 512 * 4096);
  sweep<48, 0>(out); sweep<96, 0>(out); sweep<192, 0>(out);
  sweep<48, 1>(out); sweep<96, 1>(out); sweep<192, 1>(out);
  sweep<48, 2>(out); sweep<96, 2>(out); sweep<192, 2>(out);
  sweep<48, 3>(out); sweep<96, 3>(out); sweep<192, 3>(out);
  return 0;
}
