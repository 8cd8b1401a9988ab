// Shared host/device helpers for libbisinger_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bisinger_hip.h"

namespace bsg {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define BSG_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::bsg::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
      return _e == hipErrorOutOfMemory ? BSG_ENOMEM : BSG_EHIP;                               \
    }                                                                                         \
  } while (0)

#define BSG_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::bsg::set_error(__VA_ARGS__);  \
      return BSG_EINVAL;              \
    }                                 \
  } while (0)

#define BSG_LAUNCH_CHECK() BSG_HIP(hipGetLastError())

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Row of register r (0..15) of a 32x32 MFMA accumulator for lane half h (lane >> 5); the column is
// lane & 31 (cdna_hip_programming.md §3, C/D layout — dtype independent on gfx950).
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Activation codes shared by the GEMM epilogue.
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_MISH = 3 };

// Generic fp32 MFMA GEMM (gemm.hip):  C[b] = epi( sum_tap A[b][rows + shift(tap)] * B_tap )
struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  long long sA, sB, sC;  // batch strides (elements) of the outer batch index z / batch2
  int batch2;            // inner batch count (0/1 = none): z = zo*batch2 + zi, offset = zo*s + zi*s2
  long long sA2, sB2, sC2;
  int trans_b;           // 1: B is [N,K] row-major; 0: B is [K,N] row-major
  int taps;              // conv-as-GEMM over A rows: K-segments; A row = i + tap_shift0 + tap
  int tap_shift0;
  long long sTapB;       // B offset per tap
  const float* bias_m;   // per output row
  const float* bias_n;   // per output column
  long long sBiasN;      // batch stride of bias_n (0: shared)
  float alpha;           // v = (acc + bias) * alpha
  int alpha_ncols;       // 0: every column; n > 0: only columns < n are scaled (q part of a fused QKV projection)
  int act;
  const float* post_scale_n;  // after the activation: v = v * post_scale_n[j] + post_shift_n[j] (eval-mode BatchNorm)
  const float* post_shift_n;
  const float* R;        // residual added after the activation: v += R[b][i][j]
  int ldr;
  long long sR;
  const float* rowscale; // v *= rowscale[b][i] (the reference's nonpadding masks), after the residual
  long long sRS;
  int batch;
};
int launch_gemm(const GemmArgs& g, hipStream_t st);
// the split-fp16 forms (gemm.hip, fs2.hip flash attention): true while products are formed on the 16-bit matrix pipe (BSG_GEMM_SPLIT,
// bsg_gemm_set_split); an operand that cannot be split is counted in the range-event counter (bsg_gemm_range_events)
bool gemm_split_enabled();
unsigned* gemm_range_counter();   // device address of the range-event counter (null on error): kernels of other translation units add to it

}  // namespace bsg
