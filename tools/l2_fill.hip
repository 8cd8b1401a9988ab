// How fast can every CU of the chip stream the SAME few MB out of its XCD's L2 into registers?  That is the weight stream of the
// one-workgroup-per-CU stack launches (csrc/diffnet_h2.hip: 2.1 MB of fp16 fragments per layer and CU, csrc/diffnet_bf16.hip: 1.05 MB),
// and its ceiling decides which forms of GEMM1 can be matrix-bound at all ("What bounds the stack launch").
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_fill tools/l2_fill.hip && /tmp/l2_fill
// One workgroup of 8 waves per CU (the whole LDS is requested so that no second workgroup joins it), every lane 16-byte buffer loads in
// the kernels' fragment order (a wave reads 1 KB contiguous per load; the 8 waves of a workgroup read the 8 row tiles of a k-step), `DEPTH`
// loads in flight per lane, the buffer (`MB` megabytes, shared by all workgroups) streamed `passes` times.  Prints B/clk/CU at the shader
// clock measured in the kernel (s_memtime / s_memrealtime), GB/s per CU and TB/s over the chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using rsrc_t = __amdgpu_buffer_rsrc_t;

template <int DEPTH>
__global__ __launch_bounds__(512, 2) void stream_kernel(const unsigned* __restrict__ buf, unsigned bytes, int passes, unsigned* out,
                                                        unsigned long long* clk, int skew) {
  extern __shared__ char lds[];
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, bytes, 0x00020000);
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_amdgcn_s_memtime(); clk[1] = __builtin_amdgcn_s_memrealtime(); }
  u32x4 acc = {0, 0, 0, 0};
  const int voff = tid * 16;                    // 512 lanes x 16 B = 8 KB per "k-step"
  const int steps = bytes / 8192;
  // skew: every workgroup starts at another place of the buffer (the CUs of an XCD then ask for DIFFERENT lines at any moment, like tiles that
  // drift apart), instead of all of them streaming the same lines together
  const int start = skew ? (int)(((long long)blockIdx.x * steps / gridDim.x) / DEPTH * DEPTH) : 0;
  for (int p = 0; p < passes; ++p) {
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += DEPTH) {
      const int s = (s0 + start) % steps;
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (s + d) * 8192, 0);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
  }
  if (blockIdx.x == 0 && tid == 0) { clk[2] = __builtin_amdgcn_s_memtime(); clk[3] = __builtin_amdgcn_s_memrealtime(); }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x * 512 + tid] = acc[0];   // keeps the loads alive
  if (tid == 0 && passes < 0) lds[0] = 1;
}

template <int DEPTH>
static void run(const unsigned* buf, unsigned bytes, int wgs, int passes, unsigned* out, unsigned long long* clk, int skew = 0) {
  const size_t lds = 150 * 1024;   // one workgroup per CU, like the stack launches
  hipFuncSetAttribute((const void*)stream_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(wgs), dim3(512), lds, 0, buf, bytes, passes, out, clk, skew);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c[4];
  hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
  const double us = (double)(c[3] - c[1]) / 100.0, mhz = (double)(c[2] - c[0]) / us;
  const double per_cu = (double)bytes * passes;   // bytes one workgroup pulled in
  printf("%sbuffer %5.2f MB  workgroups %3d  loads in flight/lane %2d: %8.1f us  shader %4.0f MHz  %6.1f GB/s per CU = %5.1f B/clk/CU  chip %5.2f TB/s\n",
         skew ? "skewed starts  " : "", bytes / 1048576.0, wgs, DEPTH, ms * 1e3, mhz, per_cu / (ms * 1e-3) / 1e9, per_cu / (ms * 1e-3) / (mhz * 1e6), per_cu * wgs / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned* buf; unsigned* out; unsigned long long* clk;
  const unsigned maxb = 8u << 20;
  hipMalloc(&buf, maxb); hipMemset(buf, 1, maxb);
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 64);
  for (unsigned mb : {1u, 2u, 4u, 8u})                       // 1 / 2 MB: the bf16 / split-fp16 weight set of a layer; 8 MB: beyond one XCD's 4-MB L2
    for (int wgs : {32, 256}) {                              // 32: one utterance (4 CUs per XCD busy); 256: the full chip
      run<4>(buf, mb << 20, wgs, 64, out, clk);
      run<8>(buf, mb << 20, wgs, 64, out, clk);
      run<16>(buf, mb << 20, wgs, 64, out, clk);
    }
  for (int wgs : {128, 256}) {                               // the same with every workgroup at another place of a 2-MB buffer
    run<8>(buf, 2u << 20, wgs, 64, out, clk, 1);
    run<16>(buf, 2u << 20, wgs, 64, out, clk, 1);
  }
  return 0;
}
