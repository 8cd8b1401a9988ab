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
  unsigned* range_events;   // set by launch_gemm: the counter of the handle in whose call the product runs (gemm_range_counter())
};
int launch_gemm(const GemmArgs& g, hipStream_t st);
// the split-fp16 forms (gemm.hip, fs2.hip flash attention): true while products are formed on the 16-bit matrix pipe (BSG_GEMM_SPLIT,
// bsg_gemm_set_split); an operand that cannot be split is counted in the range-event counter (bsg_gemm_range_events)
bool gemm_split_enabled();

// Range guard state of ONE handle (bsg_diffnet, bsg_fs2midi, bsg_hifigan, bsg_pitchext, bsg_fftden each own one; include/bisinger_hip.h
// "distinct handles are independent"): the device word its split-fp16 kernels count out-of-range operands into, and its switch between
// the split-fp16 GEMMs (1) and the fp32 matrix pipe (0).  Every compute entry of a handle opens a GuardScope on its guard: inside it
// gemm_range_counter() / gemm_split_enabled() answer for THAT handle (the scope is thread-local, so two host threads driving two
// handles do not see each other's state).  Outside any scope (bsg_gemm_f32, bsg_gemm_presplit_f32, tools) the process-wide counter and
// bsg_gemm_set_split apply — a test hook, also AND-ed into every handle's switch.
struct Guard {
  unsigned* counter = nullptr;   // device word, zero at create
  int split = 1;
};
int guard_init(Guard* g, hipStream_t st);
void guard_free(Guard* g);
int guard_events(Guard* g, int32_t* events, int reset, hipStream_t st);       // waits for `st`
int guard_events_async(Guard* g, int32_t* host_word, hipStream_t st);        // pinned host word, valid once `st` has passed
struct GuardScope {
  explicit GuardScope(Guard* g);
  ~GuardScope();
  Guard* prev;
};

// Split-fp16 GEMM with PRE-SPLIT operands (gemm_h2w.hip): weights packed once as hi / lo fp16 MFMA fragments in execution order, the
// activation as two fp16 planes [rows][K] of 16 x value written by its producer.
struct H2wWeights {
  unsigned short* pack = nullptr;   // [slice][tap][kk][plane][Wn / 32][64 lanes][8] fp16 of 16 x w (rows padded with zeros to a multiple of 128)
  int Wn = 0, K = 0, taps = 0;      // Wn: padded row count
  long long halfs = 0;
  bool ok = false;                  // packed, and every |16 w| inside the fp16 range
};
struct H2wArgs {
  const unsigned short* act;   // hi plane [rows][lda] fp16 of 16 x activation; the lo plane `act_plane` halfs behind it
  long long act_plane;
  int lda;
  long long sAct;              // batch stride (halfs)
  const unsigned short* wpack; // H2wWeights::pack
  long long sW;                // stride between the packed weights of successive weight batches (halfs)
  int zdiv;                    // batch index z: activation batch z % zdiv, weight batch z / zdiv (0: one weight set, zdiv = batch)
  int rows, K, Wn, taps, tap_shift0;   // activation rows per batch item; output (i, n) = sum_tap sum_k act[i + tap_shift0 + tap][k] w[tap][n][k]
  int act_is_a;                // 1: C[i][n] (tokens x features, ldc = row stride); 0: C[n][i] (features x frames: the [B, C, T] layout)
  float* C;
  int ldc;
  long long sC;                // per batch index z
  float* Cq;                   // optional (act_is_a = 0, direct store only): the same output once more in CHANNEL-QUAD order [Wn / 4][rows][4] (+ z x sC)
  unsigned short* Ch;          // optional (direct store only): the output rounded to bf16 in channel-quad order [Wn / 4][rows][4] (+ z x sC elements) — what the
                               // bf16-operand stack launch loads; with Ch the fp32 output C may be null (round 5: the term went through HBM as fp32 first)
  unsigned short* out;         // optional (act_is_a only): the result as hi / lo planes [rows][ldo] for the next GEMM, lo `out_plane` halfs behind
  long long out_plane;
  int ldo;
  long long sO;
  const float* bias;           // per weight row (+ weight batch x sBias)
  long long sBias;
  float alpha;                 // v = (acc + bias) * alpha for weight rows < alpha_ncols (0: all)
  int alpha_ncols;
  int act_fn;
  const float* R;              // residual at the output position (ldr, + z x sR)
  int ldr;
  long long sR;
  const float* rowscale;       // per activation row (+ activation batch x sRS)
  long long sRS;
  int batch;
  unsigned* range_events;      // set by launch_gemm_h2w
  // polyphase output (act_is_a = 0 only; up_u = 0: off) — a transposed convolution with kernel 2 u, stride u as ONE 2-tap product: weight row
  // n = co * up_u + r, activation row q; C[co * ldc + q * up_u + r - up_p] for positions inside [0, up_lout); bias per co
  int up_u, up_p, up_lout;
  // FS2 QKV projection feeding flash_attn_planes_kernel (act_is_a = 1 only; qkv_T = 0: off): weight rows 0 .. 2 qkv_H - 1 (Q | K) go to the planes
  // `out` as usual (ldo = 2 qkv_H), weight rows >= 2 qkv_H (V) are written TRANSPOSED as planes vt[(b * qkv_H + d) * qkv_Tp + key'] (lo plane
  // vt_plane halfs behind), b = row / qkv_T, keys of a group of 16 in MFMA-fragment order (0-3, 8-11, 4-7, 12-15), keys qkv_T .. qkv_Tp - 1 zero
  int qkv_T, qkv_Tp, qkv_H;
  unsigned short* vt;
  long long vt_plane;
  int no_direct;               // set by launch_gemm_h2w (BSG_H2W_DIRECT=0): the [feature][frame] output through the LDS image even where the accumulators could be stored directly
};
bool h2w_supports(int rows, int Wn, int K, int taps, int lda);
// W(tap, n, k) at src[tap * ts + n * rs + k * ks]; allocates w->pack on first use; |16 w| >= 65000 counted into *bad_dev (device word)
int h2w_pack(H2wWeights* w, const float* src, int Wn, int K, int taps, long long ts, long long rs, long long ks, unsigned* bad_dev, hipStream_t st);
int h2w_pack_into(unsigned short* dst, const float* src, int Wn, int K, int taps, long long ts, long long rs, long long ks, unsigned* bad_dev,
                  hipStream_t st);   // Wn % 128 == 0: 2 * Wn * K * taps halfs at dst
void h2w_free(H2wWeights* w);
int h2w_split_rows(const float* src, unsigned short* hi, unsigned short* lo, long long rows, int K, long long ld, hipStream_t st);
int h2w_split_transposed(const float* src, unsigned short* hi, unsigned short* lo, int B, int K, int T, hipStream_t st);   // [B][K][T] -> [B][T][K]
// the same with LeakyReLU(slope) applied first, `rows_per_b` >= T rows per batch item and Kp >= K columns per row in the planes (the padding is zero)
int h2w_split_transposed_lrelu(const float* src, unsigned short* hi, unsigned short* lo, int B, int K, int Kp, int T, int rows_per_b, float slope,
                               hipStream_t st);
int launch_gemm_h2w(const H2wArgs& g, hipStream_t st);
unsigned* gemm_range_counter();   // device address of the range-event counter (null on error): kernels of other translation units add to it

}  // namespace bsg
