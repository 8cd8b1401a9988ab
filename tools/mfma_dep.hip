// Issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 as a function of how many independent accumulators a wave rotates
// through (1 = every MFMA waits for the previous one's result).  One wave per SIMD, s_memtime around 4096 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 4096 / 16; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// dependent chain whose destination is NOT its srcC (what the register allocator produces when it renames an accumulator)
__global__ __launch_bounds__(256) void k16pp(float* out, unsigned long long* cyc, float a, float b) {
  f32x4 c = f32x4{0, 0, 0, 0}, d;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 4096 / 16; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(c) : "v"(a), "v"(b), "v"(d));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = c[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 4096 / 16; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename K>
static void run(const char* name, K kern, float* out, unsigned long long* cyc) {
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, cyc, 1.0f, 2.0f);
    hipDeviceSynchronize();
  }
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s %6.1f cycles per MFMA\n", name, (double)c / 4096.0);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run("16x16x4 f32, 1 accumulator", k16<1>, out, cyc);
  run("16x16x4 f32, 2 accumulators", k16<2>, out, cyc);
  run("16x16x4 f32, 4 accumulators", k16<4>, out, cyc);
  run("16x16x4 f32, dst != srcC chain", k16pp, out, cyc);
  run("32x32x2 f32, 1 accumulator", k32<1>, out, cyc);
  run("32x32x2 f32, 2 accumulators", k32<2>, out, cyc);
  return 0;
}
