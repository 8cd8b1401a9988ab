// Back-to-back dependent launches of an (almost) empty kernel with the residual layer's launch shape: what a launch costs on
// this device when nothing is computed (hipcc --offload-arch=gfx950 -O3 tools/launch_gap.hip; run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(float* p, int n) {
  extern __shared__ float lds[];
  if (n < 0) { lds[threadIdx.x] = p[threadIdx.x]; p[blockIdx.x] = lds[0]; }
}
int main() {
  float* p; hipMalloc(&p, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {64, 256, 512, 1024}) for (int lds : {0, 48 * 1024}) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, p, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, p, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid %4d x 512 threads, %2d KB LDS: %.2f us per launch\n", wgs, lds / 1024, ms * 1e3 / 2000);
  }
  return 0;
}
