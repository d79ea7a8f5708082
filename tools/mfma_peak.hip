// Developer tool: sustained int8 MFMA rate (registers only) -> practical ceiling for the matcher.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void k(int* out, int iters, int seed) {
  v4i a = {seed + (int)threadIdx.x * 0x01010101, seed * 3 + 7, (int)threadIdx.x * 0x00030507, seed ^ 0x55aa55aa};
  v4i b = {seed * 5 + 1, (int)threadIdx.x * 0x01020304, seed + 99, (int)blockIdx.x * 0x07070707 + 1};
  v16i acc0 = {0}, acc1 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
    }
  }
  int s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] ^ acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  int* out; hipMalloc(&out, 4 * 512 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 2048}) {
    const int iters = 4000;
    k<<<blocks, 512>>>(out, 100, 1); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) k<<<blocks, 512>>>(out, iters, rep + 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = 5.0 * blocks * 8.0 /*waves*/ * iters * 16.0 * 2.0 * 32 * 32 * 32;
    printf("blocks=%d: %.3f ms, %.1f Tops int8 (%.1f%% of 5033 nominal) -> effective clock %.2f GHz if the pipe is full\n",
           blocks, ms, ops / ms / 1e9, ops / ms / 1e9 / 5033 * 100, ops / ms / 1e9 / 5033 * 2.4);
  }
  return 0;
}
