// Issue rate of v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 from one wave per SIMD, operands in registers (the inner loops of
// csrc/diffnet_h2.hip without their loads): cycles per MFMA by accumulator count.  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC, int NA>
__global__ __launch_bounds__(256, 1) void k16(float* out, unsigned long long* clk, int iters) {
  f32x4 acc[NACC];
  f16x8 a[NA], b[NA];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < NA; ++i) {
    for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(threadIdx.x * 0.001f + i + j); b[i][j] = (_Float16)(threadIdx.x * 0.002f - i + j); }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) % NA], b[(i * 3 + r) % NA], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int NACC, int NA>
__global__ __launch_bounds__(256, 1) void k32(float* out, unsigned long long* clk, int iters) {
  f32x16 acc[NACC];
  f16x8 a[NA], b[NA];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int i = 0; i < NA; ++i) {
    for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(threadIdx.x * 0.001f + i + j); b[i][j] = (_Float16)(threadIdx.x * 0.002f - i + j); }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) % NA], b[(i * 3 + r) % NA], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
// the same with the accumulator pinned (destination = source C) in VGPRs (ACCV) or AGPRs, and the A operand in VGPRs or AGPRs (AA)
template <int NACC, bool ACCV, bool AA>
__global__ __launch_bounds__(256, 1) void k16asm(float* out, unsigned long long* clk, int iters) {
  f32x4 acc[NACC];
  f16x8 a[4], b[4];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(threadIdx.x * 0.001f + i + j); b[i][j] = (_Float16)(threadIdx.x * 0.002f - i + j); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (ACCV && !AA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 3]), "v"(b[(i * 3 + r) & 3]));
        else if constexpr (!ACCV && !AA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i + r) & 3]), "v"(b[(i * 3 + r) & 3]));
        else if constexpr (ACCV && AA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a[(i + r) & 3]), "v"(b[(i * 3 + r) & 3]));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(a[(i + r) & 3]), "v"(b[(i * 3 + r) & 3]));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <typename K>
static void run(const char* name, K kern, int nacc, float* out, unsigned long long* clk, int wgs) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, out, clk, iters); hipDeviceSynchronize(); }
  unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-28s %3d workgroups: %6.1f cycles per MFMA\n", name, wgs, (double)c / (iters * 3.0 * nacc));
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 64);
  run("16x16x32 asm acc V, A V, 4", k16asm<4, true, false>, 4, out, clk, 128);
  run("16x16x32 asm acc A, A V, 4", k16asm<4, false, false>, 4, out, clk, 128);
  run("16x16x32 asm acc V, A A, 4", k16asm<4, true, true>, 4, out, clk, 128);
  run("16x16x32 asm acc A, A A, 4", k16asm<4, false, true>, 4, out, clk, 128);
  run("16x16x32 asm acc A, A V, 8", k16asm<8, false, false>, 8, out, clk, 128);
  run("16x16x32 asm acc V, A V, 1", k16asm<1, true, false>, 1, out, clk, 128);
  for (int wgs : {128}) {
    run("16x16x32 f16, 4 acc", k16<4, 4>, 4, out, clk, wgs);
    run("16x16x32 f16, 8 acc", k16<8, 8>, 8, out, clk, wgs);
    run("32x32x16 f16, 2 acc", k32<2, 4>, 2, out, clk, wgs);
    run("32x32x16 f16, 4 acc", k32<4, 4>, 4, out, clk, wgs);
  }
  return 0;
}
