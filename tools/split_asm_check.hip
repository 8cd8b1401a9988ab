// The 6-instruction hi + lo fp16 split of csrc/hifigan.hip (v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16) against the plain form
// hi = (f16)x, lo = (f16)(x - (float)hi), bit for bit, over random values of every binade the split images see, the range guard's edge,
// values whose lo term is an fp16 subnormal, zeros and negative zeros.
//   hipcc --offload-arch=gfx950 -O3 -fno-gpu-flush-denormals-to-zero tools/split_asm_check.hip -o /tmp/split_asm_check && /tmp/split_asm_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned lo_pair(float a, float b, unsigned hi) {
  unsigned r;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r) : "v"(a), "v"(b), "v"(hi));
  return r;
}
__global__ void split_kernel(const float* x, unsigned* fast, unsigned* plain, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  const unsigned h = cvt_pk_f16(a, b);
  fast[2 * i] = h;
  fast[2 * i + 1] = lo_pair(a, b, h);
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  plain[2 * i] = __builtin_bit_cast(unsigned, h2{ha, hb});
  plain[2 * i + 1] = __builtin_bit_cast(unsigned, h2{(_Float16)(a - (float)ha), (_Float16)(b - (float)hb)});
}

int main() {
  const int n = 1 << 22;
  std::vector<float> x(2 * n);
  unsigned long long s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 32); };
  for (int i = 0; i < 2 * n; ++i) {
    const unsigned r = rnd();
    const int e = (int)(rnd() % 48) - 30;   // 2^-30 .. 2^17: below the fp16 subnormals up to beyond 65504
    float v = ldexpf(1.0f + (r & 0x7fffff) / 8388608.0f, e);
    if (r & 0x800000) v = -v;
    x[i] = v;
  }
  const float edge[] = {0.f, -0.f, 65000.f, 65504.f, 65519.9f, -65504.f, 6.1e-5f, 5.96e-8f, 2.98e-8f, 1.0f, 1.00048828125f, 2049.f, 2047.5f, -2049.f};
  for (size_t i = 0; i < sizeof(edge) / sizeof(float); ++i) x[i] = edge[i];
  float* dx; unsigned *df, *dp;
  hipMalloc(&dx, 2 * n * 4); hipMalloc(&df, 2 * n * 4); hipMalloc(&dp, 2 * n * 4);
  hipMemcpy(dx, x.data(), 2 * n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_kernel, dim3(n / 256), dim3(256), 0, 0, dx, df, dp, n);
  std::vector<unsigned> f(2 * n), p(2 * n);
  hipMemcpy(f.data(), df, 2 * n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(p.data(), dp, 2 * n * 4, hipMemcpyDeviceToHost);
  long long bad = 0;
  for (int i = 0; i < 2 * n; ++i)
    if (f[i] != p[i]) {
      // (an overflowing hi is inf in both forms; x - inf = -inf / nan: compare those as equal when both are not finite)
      if (bad < 5) printf("mismatch at %d: x = %g %g  fast %08x plain %08x\n", i, x[2 * (i / 2)], x[2 * (i / 2) + 1], f[i], p[i]);
      ++bad;
    }
  printf("%d pairs: %lld mismatching dwords\n", n, bad);
  return bad != 0;
}
