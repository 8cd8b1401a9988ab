// Which XCD does workgroup i of a launch run on?  residual_part_h2_kernel (csrc/diffnet_h2.hip) places the parts of a tile on workgroups
// i, i + 8, i + 16, ... and exchanges z / the image among them with PLAIN stores, which are only visible inside one XCD's L2; it checks the
// placement at run time by comparing HW_REG_XCC_ID of the parts.  This probe prints that register for the first workgroups of a launch of
// one workgroup per CU:   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcc_probe tools/xcc_probe.hip && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 1) void probe(unsigned* out) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 1000) lds[0] = 1;
}
int main() {
  unsigned* out; hipMalloc(&out, 1024 * 4);
  const size_t lds = 90 * 1024;   // more than half of a CU's LDS: one workgroup per CU, like the part launches
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int wgs : {128, 256}) {
    hipMemset(out, 0xff, 1024 * 4);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), lds, 0, out);
    hipDeviceSynchronize();
    unsigned h[1024]; hipMemcpy(h, out, wgs * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < wgs; ++i) bad += h[i] != h[i & 7];
    printf("%d workgroups: XCC id of workgroups 0..15:", wgs);
    for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
    printf("   workgroups i and i mod 8 on different XCDs: %d\n", bad);
  }
  return 0;
}
