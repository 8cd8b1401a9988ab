#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int SHAPE>
__global__ __launch_bounds__(512, 4) void k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  if (SHAPE == 16) {
    f32x4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m1, 0, 0, 0);
        m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m2, 0, 0, 0);
        m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m3, 0, 0, 0);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = m0[0] + m1[1] + m2[2] + m3[3];
  } else {
    f32x16 m0, m1;
    for (int r = 0; r < 16; ++r) m0[r] = m1[r] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m1, 0, 0, 0);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = m0[0] + m1[1];
  }
}
int main() {
  float* out; hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {256, 512}) for (int shape : {16, 32}) {
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(wgs), dim3(512), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<32>, dim3(wgs), dim3(512), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = shape == 16 ? 16.0 * iters * 16 * 16 * 4 * 2 : 8.0 * iters * 32 * 32 * 2 * 2;
    const double flops = mf * wgs * 8;
    printf("wgs=%d shape=%dx%d: %.3f ms  %.1f TFLOP/s\n", wgs, shape, shape, ms, flops / ms / 1e9);
  }
  return 0;
}
