// HiFi-GAN generator forward (mel -> waveform) on gfx950.
//
// Reference semantics: /root/reference/train_bisinger/modules/hifigan/hifigan.py
//   ResBlock1.forward :54-61, HifiGanGenerator.__init__ :105-142, forward :144-173, remove_weight_norm :175-182.
//
// The generator is 72 small-channel convolutions (128 -> 8 channels while the time axis grows 256x):
// arithmetic intensity ~10 FLOP/B, i.e. HBM/LDS-bound elementwise-like work, not matrix work — and fp32
// MFMA has no rate advantage over fp32 VALU on gfx950 anyway (both 256 FLOP/clk/CU).  So the convs are
// direct VALU kernels on the native [B,C,T] layout:
//   * a wave's 64 lanes own 64 consecutive samples (x TT strided repeats): every HBM/LDS access of a
//     wave is one contiguous 256-B row segment;
//   * input channels are staged through LDS in chunks ([CI_CHUNK][tile + halo], input LeakyReLU applied
//     while staging), shared by the workgroup's waves;
//   * the CO_BLK output channels a workgroup produces are wave-uniform, so the weights are scalar loads
//     (SGPR operands of v_fma): the inner loop is one ds_read_b32 per CO_BLK FMAs;
//   * bias, residual add, the MRF sum / num_kernels and tanh are fused epilogues.
#include <math.h>

#include <vector>

#include "bsg_common.h"

namespace bsg {
namespace {
using f32x2 = __attribute__((ext_vector_type(2))) float;

// samples per lane TT (stride 256) is a template parameter: 4 when the launch fills the chip anyway, 2 or 1 for short
// inputs (B = 1), where 1024-sample tiles leave most CUs without a workgroup
constexpr int CI_CHUNK = 8;

struct ConvArgs {
  const float* x;      // [B][Cin][L]
  const float* w;      // packed [ceil(Cout/CO_BLK)][Cin][K][CO_BLK] (zero padded): the CO_BLK weights of one (ci, k) are one wide scalar load
  const float* bias;   // [Cout]
  float* y;            // [B][Cout][L]
  const float* res;    // optional [B][Cout][L]: y = conv + res
  const float* acc_in; // optional [B][Cout][L]: y = acc_in + y   (MRF running sum, hifigan.py:161-166)
  float out_div;       // y /= out_div (num_kernels on the last resblock, :167)
  float in_slope;      // LeakyReLU slope applied to the input (1 = identity)
  int out_tanh;
  int Cin, Cout, L, dil, pad;
  // polyphase ConvTranspose1d (n_phase = stride u > 0): blockIdx.z = b*u + phase; position q of phase r is output sample
  // q*u + r + ph_off of a row of Lout samples; the phase's 2-tap weights start at w + r*w_phase_stride
  int n_phase, ph_off, Lq, Lout;
  long long w_phase_stride;
};

template <int K, int CO_BLK, int TT>
__global__ __launch_bounds__(256) void conv1d_kernel(ConvArgs a) {
  constexpr int TILE = 256 * TT;  // samples per workgroup
  extern __shared__ float xs[];   // [CI_CHUNK][TILE + (K-1)*dil]
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * TILE;
  const int co0 = blockIdx.y * CO_BLK;
  const int ph = a.n_phase > 0 ? (int)blockIdx.z % a.n_phase : 0;
  const int b = a.n_phase > 0 ? (int)blockIdx.z / a.n_phase : (int)blockIdx.z;
  const int span = TILE + (K - 1) * a.dil;
  const float* __restrict__ xb = a.x + (long long)b * a.Cin * a.L;
  const float* __restrict__ wbase = a.w + (long long)ph * a.w_phase_stride;

  float acc[CO_BLK][TT];
#pragma unroll
  for (int c = 0; c < CO_BLK; ++c) {
    const float bv = (co0 + c < a.Cout) ? a.bias[co0 + c] : 0.f;
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[c][j] = bv;
  }

  for (int ci0 = 0; ci0 < a.Cin; ci0 += CI_CHUNK) {
    __syncthreads();
    for (int idx0 = tid; idx0 < CI_CHUNK * span; idx0 += 4 * 256) {   // 4 loads in flight per thread (the rolled load -> store loop paid a round trip per item)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = idx0 + 256 * u;
        const int ci = idx / span, j = idx - ci * span;
        const int t = t0 - a.pad + j;
        v[u] = (idx < CI_CHUNK * span && ci0 + ci < a.Cin && t >= 0 && t < a.L) ? xb[(long long)(ci0 + ci) * a.L + t] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (idx0 + 256 * u < CI_CHUNK * span) xs[idx0 + 256 * u] = v[u] > 0.f ? v[u] : v[u] * a.in_slope;
    }
    __syncthreads();
    const int nci = (a.Cin - ci0) < CI_CHUNK ? (a.Cin - ci0) : CI_CHUNK;
#pragma unroll 1
    for (int ci = 0; ci < nci; ++ci) {
      // wave-uniform -> scalar loads, CO_BLK consecutive floats per tap
      const float* __restrict__ wp = wbase + ((long long)blockIdx.y * a.Cin + (ci0 + ci)) * (K * CO_BLK);
      const float* __restrict__ xr = xs + ci * span + tid;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float xv[TT];
#pragma unroll
        for (int j = 0; j < TT; ++j) xv[j] = xr[k * a.dil + 256 * j];
#pragma unroll
        for (int c = 0; c < CO_BLK; ++c) {
          const float wv = wp[k * CO_BLK + c];
#pragma unroll
          for (int j = 0; j < TT; ++j) acc[c][j] = fmaf(wv, xv[j], acc[c][j]);
        }
      }
    }
  }

  if (a.n_phase > 0) {   // polyphase transposed conv: strided store of this phase's samples
#pragma unroll
    for (int c = 0; c < CO_BLK; ++c) {
      const long long rowo = ((long long)b * a.Cout + co0 + c) * a.Lout;
#pragma unroll
      for (int j = 0; j < TT; ++j) {
        const int q = t0 + tid + 256 * j;
        const int to = q * a.n_phase + ph + a.ph_off;
        if (q < a.Lq && to >= 0 && to < a.Lout && co0 + c < a.Cout) a.y[rowo + to] = acc[c][j];
      }
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < CO_BLK; ++c) {
    const long long rowo = ((long long)b * a.Cout + co0 + c) * a.L;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int t = t0 + tid + 256 * j;
      if (t >= a.L || co0 + c >= a.Cout) continue;
      float v = acc[c][j];
      if (a.res) v += a.res[rowo + t];
      if (a.acc_in) v = a.acc_in[rowo + t] + v;
      if (a.out_div != 1.0f) v = v / a.out_div;
      if (a.out_tanh) v = tanhf(v);
      a.y[rowo + t] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// One (dilated conv, conv) pair of a ResBlock1, fused (hifigan.py:54-61):
//     y = x + conv2_{K,1}( lrelu( conv1_{K,d}( lrelu(x) ) ) )        [+ running MRF sum] [/ num_kernels]
// Un-fused, the pair moves its [B,C,L] tensor through HBM five times (conv1: read + write; conv2: read + residual + write); here
// the intermediate never leaves the CU: one read of x, one write of y (the residual read is an L2 hit on the lines just staged).
// Workgroup = 256 threads = P = 256 TT consecutive positions of one utterance, ALL C channels (C TT = 64 accumulators per thread):
//   A  conv1 at positions [t0 - h2, t0 - h2 + P), h2 = (K-1)/2: raw x staged through LDS in chunks of CI input channels
//      (span P + (K-1) d, LeakyReLU applied when a lane reads its operand: lrelu(x) = max(x, slope x)); per (ci, tap) a lane
//      reads TT operands and does C TT FMAs whose weights are wave-uniform (scalar loads of C consecutive floats);
//   B  t1 = lrelu(conv1 + b1), zero outside [0, L) (conv2 pads ITS input with zeros) -> LDS [C][P] (aliases the x chunks);
//   C  conv2 (dilation 1) over the LDS image at the P - (K-1) central positions: again C TT FMAs per LDS read;
//   D  + b2 + x (+ acc_in) (/ out_div) -> y.
// The (K-1) halo positions of t1 are computed twice (by neighbouring workgroups): 2 % at P = 512.
// ------------------------------------------------------------------------------------------------
struct PairArgs {
  const float* x;       // [B][C][L]
  const float* w1;      // conv1 packed [C ci][K][C co]
  const float* b1;
  const float* w2;      // conv2 packed [C ci][K][C co]
  const float* b2;
  float* y;             // [B][C][L]
  const float* acc_in;  // optional [B][C][L]: y = acc_in + y   (MRF running sum, hifigan.py:161-166)
  float out_div;        // y /= out_div (num_kernels after the last resblock, :167)
  float slope;
  int L, dil;
  unsigned* range_events;   // split-fp16 form: counter of operands beyond the fp16 range (gemm_range_counter())
};

template <int K, int C, int TT>
__global__ __launch_bounds__(256) void resblock_pair_kernel(PairArgs a) {
  constexpr int P = 256 * TT, H2 = (K - 1) / 2, POUT = P - (K - 1);
  constexpr int CI = C == 16 ? 8 : 4;   // input channels per staged chunk (span <= P + 50 floats each): stays inside the [C][P] image
  extern __shared__ float lds[];        // [C][P] (+ K-1 floats so that the last row's taps stay inside); phase A uses its first CI * span floats
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * POUT;     // first output position of this workgroup
  const int b = blockIdx.y;
  const int h1 = H2 * a.dil, span = P + 2 * h1;
  const float* __restrict__ xb = a.x + (long long)b * C * a.L;
  const float slope = a.slope;

  float acc[C][TT];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float bv = a.b1[c];
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[c][j] = bv;
  }
  // ---- A: conv1 (dilation d) at u = t0 - H2 + tid + 256 j ---------------------------------------------------------------
  for (int ci0 = 0; ci0 < C; ci0 += CI) {
    __syncthreads();
    for (int idx0 = tid; idx0 < CI * span; idx0 += 4 * 256) {   // (4 loads in flight per thread: the trip count is not a constant, the loop stays rolled)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = idx0 + 256 * u;
        const int ci = idx / span, jx = idx - ci * span;
        const int t = t0 - H2 - h1 + jx;
        v[u] = (idx < CI * span && t >= 0 && t < a.L) ? xb[(long long)(ci0 + ci) * a.L + t] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (idx0 + 256 * u < CI * span) lds[idx0 + 256 * u] = v[u];
    }
    __syncthreads();
#pragma unroll 1
    for (int ci = 0; ci < CI; ++ci) {
      const float* __restrict__ wp = a.w1 + (long long)(ci0 + ci) * (K * C);
      const float* __restrict__ xr = lds + ci * span + tid;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float xv[TT];
#pragma unroll
        for (int j = 0; j < TT; ++j) {
          const float v = xr[k * a.dil + 256 * j];
          xv[j] = fmaxf(v, v * slope);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float wv = wp[k * C + c];
#pragma unroll
          for (int j = 0; j < TT; ++j) acc[c][j] = fmaf(wv, xv[j], acc[c][j]);
        }
      }
    }
  }
  __syncthreads();   // every wave is done with the x chunks
  // ---- B: t1 = lrelu(conv1), zero outside [0, L) -> LDS [C][P] --------------------------------------------------------
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    const int u = t0 - H2 + tid + 256 * j;
    const bool in = u >= 0 && u < a.L;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float v = acc[c][j];
      lds[c * P + tid + 256 * j] = in ? fmaxf(v, v * slope) : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float bv = a.b2[c];
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[c][j] = bv;
  }
  __syncthreads();
  // ---- C: conv2 (dilation 1) at t = t0 + tid + 256 j (positions >= POUT are not outputs of this tile: masked below) ----
#pragma unroll 1
  for (int ci = 0; ci < C; ++ci) {
    const float* __restrict__ wp = a.w2 + (long long)ci * (K * C);
    const float* __restrict__ tr = lds + ci * P + tid;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float tv[TT];
#pragma unroll
      for (int j = 0; j < TT; ++j) tv[j] = tr[256 * j + k];   // the tile's last K-1 positions read into the next row: never stored
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float wv = wp[k * C + c];
#pragma unroll
        for (int j = 0; j < TT; ++j) acc[c][j] = fmaf(wv, tv[j], acc[c][j]);
      }
    }
  }
  // ---- D: residual, MRF sum, store (all loads of a position batch in flight before the first use) ---------------------------
  const bool has_acc = a.acc_in != nullptr, has_div = a.out_div != 1.0f;
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    const int o = tid + 256 * j, t = t0 + o;
    if (o >= POUT || t >= a.L) continue;
    const long long i0 = (long long)b * C * a.L + t;
    float xr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xr[c] = a.x[i0 + (long long)c * a.L];
    if (has_acc) {
      float ar[C];
#pragma unroll
      for (int c = 0; c < C; ++c) ar[c] = a.acc_in[i0 + (long long)c * a.L];
#pragma unroll
      for (int c = 0; c < C; ++c) xr[c] = ar[c] + (acc[c][j] + xr[c]);
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) xr[c] = acc[c][j] + xr[c];
    }
    if (has_div) {
#pragma unroll
      for (int c = 0; c < C; ++c) xr[c] = xr[c] / a.out_div;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) a.y[i0 + (long long)c * a.L] = xr[c];
  }
}

// ------------------------------------------------------------------------------------------------
// The same fused pair on the matrix pipe, for C = 32 and C = 64 channels (stages 1 and 0: two thirds of the generator's FLOPs).
// fp32 MFMA has the VALU's peak rate, but one v_mfma_f32_32x32x2_f32 replaces 32 wave-wide FMA instructions, so the issue
// slots the VALU form spends on FMAs (and its scalar weight loads) are free: each conv is an implicit GEMM
//     out[co][p] = sum_{tap k} sum_{ci} W[co][ci][k] * in[ci][p + k d]          M = C, N = positions, K = C per tap
// with A = weights pre-packed in fragment order (one 16-byte load per lane = 4 k-steps = 8 input channels of one tap) and
// B = one conflict-free ds_read_b32 per MFMA (lanes = 32 consecutive positions, the two lane halves = two input channels).
// Workgroup = 4 waves = PT = 256 positions of the intermediate t1 (POUT = 256 - (K-1) outputs), all C channels: wave w owns
// positions [64 w, 64 w + 64) = 2 column tiles, and C / 32 row tiles.  LDS: lrelu(x) [C][256 + (K-1) d], then (aliased, after a
// barrier) t1 [C][272]: 40 KB at C = 32 (4 workgroups per CU), 79 KB at C = 64 (2 per CU).
// ------------------------------------------------------------------------------------------------
// out[(((rt*K + k)*(C/8) + g)*64 + lane)*4 + j] = W[co = 32 rt + (lane & 31)][ci = 8 g + 2 j + (lane >> 5)][k],  W = [C][C][K]
__global__ void pack_conv_mfma_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * C * K) return;
  const int j = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
  const int G = C / 8;
  const int g = rest % G, k = (rest / G) % K, rt = rest / (G * K);
  out[i] = w[((long long)(32 * rt + (lane & 31)) * C + 8 * g + 2 * j + (lane >> 5)) * K + k];
}

// acc[rt][nb] += W_tap-major * img: img = LDS [C][stride], this wave's positions start at column n0; tap k reads column p + k * dil
template <int K, int C, int NB>
__device__ __forceinline__ void conv_mfma(f32x16 (&acc)[C / 32][NB], const float* __restrict__ wpk, const float* img, int stride, int n0,
                                          int dil, int lane) {
  constexpr int RT = C / 32, G = C / 8;
  const int l31 = lane & 31, lh = lane >> 5;
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(wpk) + lane;
  const float* bp = img + lh * stride + n0 + l31;
  f32x4 A[2][RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) A[0][rt] = wp[(rt * K * G) * 64];
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const float* bk = bp + k * dil;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // weights of the next group (the first group of the next tap after the last of this one; a harmless repeat at the very end)
      const int kn = g + 1 < G ? k : (k + 1 < K ? k + 1 : k), gn = g + 1 < G ? g + 1 : (k + 1 < K ? 0 : g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) A[(g + 1) & 1][rt] = wp[((rt * K + kn) * G + gn) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = bk[(8 * g + 2 * j) * stride + 32 * nb];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[rt][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g & 1][rt][j], bv[nb], acc[rt][nb], 0, 0, 0);
      }
    }
  }
}

template <int K, int C, int NB>   // NB column tiles of 32 positions per wave: 2 (256 positions per workgroup), or 1 for short inputs (more workgroups)
__global__ __launch_bounds__(256) void resblock_pair_mfma_kernel(PairArgs a) {
  constexpr int PT = 128 * NB, H2 = (K - 1) / 2, POUT = PT - (K - 1), RT = C / 32, TS = PT + 16;
  static_assert(C % 32 == 0 && (C / 8) % 2 == 0, "channel count");
  extern __shared__ float lds[];   // lrelu(x) [C][XS], then t1 [C][TS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int t0 = blockIdx.x * POUT, b = blockIdx.y;
  const int h1 = H2 * a.dil, span = PT + 2 * h1, XS = (span + 3) & ~3;
  const float* __restrict__ xb = a.x + (long long)b * C * a.L;
  const float slope = a.slope;

  // ---- stage lrelu(x) over [t0 - H2 - h1, + span), zero outside [0, L) ---------------------------------------------------
  for (int idx = tid; idx < C * span; idx += 256) {
    const int ci = idx / span, jx = idx - ci * span;
    const int t = t0 - H2 - h1 + jx;
    float v = (t >= 0 && t < a.L) ? xb[(long long)ci * a.L + t] : 0.f;
    lds[ci * XS + jx] = fmaxf(v, v * slope);
  }
  const int n0 = 32 * NB * wave;
  f32x16 acc[RT][NB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = a.b1[32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb][r] = bv;
    }
  __syncthreads();
  // ---- conv1 (dilation d) at t1 positions u = t0 - H2 + p: reads x column p + k d ------------------------------------------------
  conv_mfma<K, C, NB>(acc, a.w1, lds, XS, n0, a.dil, lane);
  __syncthreads();   // every wave is done reading x
  // ---- t1 = lrelu(conv1), zero outside [0, L) (conv2 pads its input) -> LDS [C][TS] ------------------------------------------
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int pcol = n0 + 32 * nb + l31, u = t0 - H2 + pcol;
    const bool in = u >= 0 && u < a.L;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[rt][nb][r];
        lds[(32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh) * TS + pcol] = in ? fmaxf(v, v * slope) : 0.f;
      }
  }
  if (tid < C) {   // columns PT .. TS-1 are read by the (masked) last K-1 positions of the tile: keep them finite
#pragma unroll
    for (int j = PT; j < TS; ++j) lds[tid * TS + j] = 0.f;
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = a.b2[32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb][r] = bv;
    }
  __syncthreads();
  // ---- conv2 (dilation 1) at output positions t = t0 + p: reads t1 column p + k ---------------------------------------------
  conv_mfma<K, C, NB>(acc, a.w2, lds, TS, n0, 1, lane);
  // ---- residual, MRF sum, store --------------------------------------------------------------------------------------------
  const bool has_acc = a.acc_in != nullptr, has_div = a.out_div != 1.0f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int pcol = n0 + 32 * nb + l31, t = t0 + pcol;
    if (pcol >= POUT || t >= a.L) continue;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const long long i0 = ((long long)b * C + 32 * rt + 4 * lh) * a.L + t;
      float xr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xr[r] = a.x[i0 + (long long)((r & 3) + 8 * (r >> 2)) * a.L];
      if (has_acc) {
        float ar[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ar[r] = a.acc_in[i0 + (long long)((r & 3) + 8 * (r >> 2)) * a.L];
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = ar[r] + (acc[rt][nb][r] + xr[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = acc[rt][nb][r] + xr[r];
      }
      if (has_div) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = xr[r] / a.out_div;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) a.y[i0 + (long long)((r & 3) + 8 * (r >> 2)) * a.L] = xr[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Polyphase ConvTranspose1d (K = 2u, stride u, padding u/2) with coalesced stores: a lane owns ONE input position q and produces
// all u output phases of CO output channels, i.e. u consecutive output samples per channel (32 B at u = 8): a wave's store
// instructions cover contiguous 2-KB runs.  (The launch-per-phase form of conv1d_kernel<2,..> stores one float per lane every
// u floats: the same bytes as 8x the store transactions; it ran at 0.6 TB/s.)
//   y[co][q u + r - p] = b[co] + sum_ci ( lrelu(x[ci][q-1]) w[ci][co][r + u] + lrelu(x[ci][q]) w[ci][co][r] )
// weights: the per-phase 2-tap packing of pack_convT_w_kernel with CO-blocks of 8: [r][co-block][ci][2][8].
// ------------------------------------------------------------------------------------------------
struct UpArgs {
  const float* x;     // [B][Cin][Lin]
  const float* w;     // [u][ceil(Cout/8)][Cin][2][8]
  const float* bias;
  float* y;           // [B][Cout][Lin*u]
  float slope;
  int Cin, Cout, Lin, p;
};
template <int U>
__global__ __launch_bounds__(256) void upsample_kernel(UpArgs a) {
  constexpr int CO = 8, CIC = 16;
  __shared__ float xs[CIC][260];
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * 256, q = q0 + tid;     // q in [0, Lin]: position q contributes outputs [q U - p, q U - p + U)
  const int cb = blockIdx.y, b = blockIdx.z;
  const int nblk = (a.Cout + CO - 1) / CO;
  const float* __restrict__ xb = a.x + (long long)b * a.Cin * a.Lin;
  float acc[CO][U];
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float bv = cb * CO + c < a.Cout ? a.bias[cb * CO + c] : 0.f;
#pragma unroll
    for (int r = 0; r < U; ++r) acc[c][r] = bv;
  }
  for (int ci0 = 0; ci0 < a.Cin; ci0 += CIC) {
    __syncthreads();
    for (int idx = tid; idx < CIC * 257; idx += 256) {
      const int ci = idx / 257, j = idx - ci * 257;
      const int t = q0 - 1 + j;
      float v = (ci0 + ci < a.Cin && t >= 0 && t < a.Lin) ? xb[(long long)(ci0 + ci) * a.Lin + t] : 0.f;
      xs[ci][j] = fmaxf(v, v * a.slope);
    }
    __syncthreads();
    const int nci = a.Cin - ci0 < CIC ? a.Cin - ci0 : CIC;
#pragma unroll 1
    for (int ci = 0; ci < nci; ++ci) {
      const float xm = xs[ci][tid], x0 = xs[ci][tid + 1];   // x[q-1], x[q]
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const float* __restrict__ wp = a.w + (((long long)r * nblk + cb) * a.Cin + ci0 + ci) * (2 * CO);
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c][r] = fmaf(wp[CO + c], x0, fmaf(wp[c], xm, acc[c][r]));
      }
    }
  }
  if (q > a.Lin) return;
  const int Lout = a.Lin * U;
  const int o0 = q * U - a.p;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    if (cb * CO + c >= a.Cout) continue;
    float* __restrict__ yr = a.y + ((long long)b * a.Cout + cb * CO + c) * Lout;
    if (o0 >= 0 && o0 + U <= Lout && (U % 4) == 0 && ((o0 & 3) == 0)) {
#pragma unroll
      for (int r = 0; r < U; r += 4) *reinterpret_cast<f32x4*>(yr + o0 + r) = f32x4{acc[c][r], acc[c][r + 1], acc[c][r + 2], acc[c][r + 3]};
    } else {
#pragma unroll
      for (int r = 0; r < U; ++r)
        if (o0 + r >= 0 && o0 + r < Lout) yr[o0 + r] = acc[c][r];
    }
  }
}

// ConvTranspose1d(K = 4, stride 2, padding 1) — the last upsampling stages, 2 x (read + write) of the largest tensors: a lane produces 4
// CONSECUTIVE output samples of CO channels (16-byte stores, a wave covers 1 KB per channel) from x[2m-1 .. 2m+2]:
//   y[4m]   = x[2m-1] w3 + x[2m]   w1      y[4m+1] = x[2m]   w2 + x[2m+1] w0
//   y[4m+2] = x[2m]   w3 + x[2m+1] w1      y[4m+3] = x[2m+1] w2 + x[2m+2] w0          (w_k = w[ci][co][k], x = lrelu(input), zero outside)
// The weights are read in their own layout [Cin][Cout][4] (wave-uniform: scalar loads).  upsample_kernel<2> wrote 2 samples per lane with
// 4-byte stores at odd offsets and staged x once per block of 8 output channels: 117 us per stage at B = 16 against 52 us of HBM time; this
// form: 100 / 90 us (16 / 8 output channels) — the staging of a 16-channel chunk and its FMAs still alternate.
template <int CO>
__global__ __launch_bounds__(256) void upsample2_kernel(UpArgs a) {
  constexpr int CIC = 16, XS = 516;      // xs[ci][j] = lrelu(x[2 m0 - 2 + j]): the lane's x[2m-1] sits at the EVEN index 2 tid + 1... see below
  __shared__ __attribute__((aligned(16))) float xs[CIC][XS];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * 256, m = m0 + tid;   // outputs [4m, 4m + 4)
  const int cb = blockIdx.y, b = blockIdx.z;
  const float* __restrict__ xb = a.x + (long long)b * a.Cin * a.Lin;
  float acc[CO][4];
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float bv = a.bias[cb * CO + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[c][r] = bv;
  }
  for (int ci0 = 0; ci0 < a.Cin; ci0 += CIC) {
    __syncthreads();
    // xs[ci][j] = x[2 m0 - 1 + j], j in [0, 514): lane tid reads j = 2 tid .. 2 tid + 3 (two aligned 8-byte reads)
    for (int idx0 = tid; idx0 < CIC * 514; idx0 += 8 * 256) {   // 8 loads in flight per thread (a rolled load -> store loop pays one round trip per item)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + 256 * u;
        const int ci = idx / 514, j = idx - ci * 514;
        const int t = 2 * m0 - 1 + j;
        v[u] = (idx < CIC * 514 && ci0 + ci < a.Cin && t >= 0 && t < a.Lin) ? xb[(long long)(ci0 + ci) * a.Lin + t] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + 256 * u;
        if (idx < CIC * 514) xs[idx / 514][idx % 514] = fmaxf(v[u], v[u] * a.slope);
      }
    }
    __syncthreads();
    const int nci = a.Cin - ci0 < CIC ? a.Cin - ci0 : CIC;
#pragma unroll 1
    for (int ci = 0; ci < nci; ++ci) {
      const f32x2 xa = *reinterpret_cast<const f32x2*>(&xs[ci][2 * tid]), xb2 = *reinterpret_cast<const f32x2*>(&xs[ci][2 * tid + 2]);
      const float xm = xa[0], x0 = xa[1], x1 = xb2[0], x2 = xb2[1];   // x[2m-1], x[2m], x[2m+1], x[2m+2]
      const float* __restrict__ wp = a.w + ((long long)(ci0 + ci) * a.Cout + cb * CO) * 4;
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const float w0 = wp[4 * c], w1 = wp[4 * c + 1], w2 = wp[4 * c + 2], w3 = wp[4 * c + 3];
        acc[c][0] = fmaf(x0, w1, fmaf(xm, w3, acc[c][0]));
        acc[c][1] = fmaf(x1, w0, fmaf(x0, w2, acc[c][1]));
        acc[c][2] = fmaf(x1, w1, fmaf(x0, w3, acc[c][2]));
        acc[c][3] = fmaf(x2, w0, fmaf(x1, w2, acc[c][3]));
      }
    }
  }
  const int Lout = a.Lin * 2, o0 = 4 * m;
  if (o0 >= Lout) return;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    float* __restrict__ yr = a.y + ((long long)b * a.Cout + cb * CO + c) * Lout;
    if (o0 + 3 < Lout && (Lout & 3) == 0) {
      *reinterpret_cast<f32x4*>(yr + o0) = f32x4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (o0 + r < Lout) yr[o0 + r] = acc[c][r];
    }
  }
}

// conv_post (hifigan.py:169-171): Conv1d(C -> 1, 7, padding 3) on LeakyReLU(x, 0.01), then tanh.  One output channel: the generic conv1d_kernel
// spends 7 of its 8 output-channel lanes' FMAs on nothing.  A lane produces 4 consecutive samples from three 16-byte loads per input channel
// (x[4m-4 .. 4m+7]; neighbours' loads overlap in L1), no LDS; weights wave-uniform.  98 -> 33 us at B = 16 (147 MB: 4.4 TB/s).
template <int C>
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, int L, float slope) {
  const int m = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const int t0 = 4 * m;
  if (t0 >= L) return;
  const float* __restrict__ xb = x + (long long)b * C * L;
  float acc[4] = {bias[0], bias[0], bias[0], bias[0]};
  const bool inner = t0 >= 4 && t0 + 8 <= L && (L & 3) == 0;
#pragma unroll
  for (int ci = 0; ci < C; ++ci) {
    float v[12];   // x[t0 - 4 + j]
    const float* __restrict__ xr = xb + (long long)ci * L;
    if (inner) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(xr + t0 - 4 + 4 * g);
        v[4 * g] = q[0]; v[4 * g + 1] = q[1]; v[4 * g + 2] = q[2]; v[4 * g + 3] = q[3];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int t = t0 - 4 + j;
        v[j] = (t >= 0 && t < L) ? xr[t] : 0.f;
      }
    }
#pragma unroll
    for (int j = 1; j < 11; ++j) v[j] = fmaxf(v[j], v[j] * slope);   // (slope in [0, 1])
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float wk = w[ci * 7 + k];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(wk, v[r + k + 1], acc[r]);   // y[t0 + r] += w[k] x[t0 + r + k - 3]
    }
  }
  float* __restrict__ yr = y + (long long)b * L + t0;
  if (t0 + 3 < L && (L & 3) == 0) {
    *reinterpret_cast<f32x4*>(yr) = f32x4{tanhf(acc[0]), tanhf(acc[1]), tanhf(acc[2]), tanhf(acc[3])};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (t0 + r < L) yr[r] = tanhf(acc[r]);
  }
}

// ConvTranspose1d(Cin -> Cout, K, stride u, padding p) on LeakyReLU(x): y[co][t'] = b[co] + sum over (ci, k, i) with
// t' = i*u - p + k.  ~3 % of the generator's FLOPs: one thread per output sample, CO_BLK channels, taps gathered.
struct ConvTArgs {
  const float* x;     // [B][Cin][Lin]
  const float* w;     // [Cin][Cout][K]
  const float* bias;
  float* y;           // [B][Cout][Lin*u]
  float in_slope;
  int Cin, Cout, Lin, K, u, p;
};
template <int CO_BLK>
__global__ __launch_bounds__(256) void conv_transpose1d_kernel(ConvTArgs a) {
  const int Lout = a.Lin * a.u;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * CO_BLK;
  const int b = blockIdx.z;
  if (t >= Lout) return;
  float acc[CO_BLK];
#pragma unroll
  for (int c = 0; c < CO_BLK; ++c) acc[c] = (co0 + c < a.Cout) ? a.bias[co0 + c] : 0.f;
  const int tp = t + a.p;
  const int r = tp % a.u, q = tp / a.u;
  const float* __restrict__ xb = a.x + (long long)b * a.Cin * a.Lin;
  for (int k = r; k < a.K; k += a.u) {
    const int i = q - (k - r) / a.u;
    if (i < 0 || i >= a.Lin) continue;
    for (int ci = 0; ci < a.Cin; ++ci) {
      float xv = xb[(long long)ci * a.Lin + i];
      xv = xv > 0.f ? xv : xv * a.in_slope;
      const float* __restrict__ wp = a.w + ((long long)ci * a.Cout + co0) * a.K + k;
#pragma unroll
      for (int c = 0; c < CO_BLK; ++c)
        if (co0 + c < a.Cout) acc[c] = fmaf(wp[(long long)c * a.K], xv, acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < CO_BLK; ++c)
    if (co0 + c < a.Cout) a.y[((long long)b * a.Cout + co0 + c) * Lout + t] = acc[c];
}

// weight = g * v / ||v||, norm over every dim but 0  (torch.nn.utils.weight_norm, dim = 0).  One wave per slice.
__global__ void weight_norm_fold_kernel(const float* __restrict__ g, const float* __restrict__ v, float* __restrict__ w,
                                        int dim0, int inner) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= dim0) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < inner; i += 64) {
    const float x = v[(long long)row * inner + i];
    s += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float scale = g[row] / sqrtf(s);
  for (int i = lane; i < inner; i += 64) w[(long long)row * inner + i] = v[(long long)row * inner + i] * scale;
}

// [Cout][Cin][K] -> [ceil(Cout/CO)][Cin][K][CO], zero padded
__global__ void pack_conv_w_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int K, int CO) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nblk = (Cout + CO - 1) / CO;
  if (i >= nblk * Cin * K * CO) return;
  const int c = i % CO, k = (i / CO) % K, ci = (i / (CO * K)) % Cin, blk = i / (CO * K * Cin);
  const int co = blk * CO + c;
  out[i] = co < Cout ? w[((long long)co * Cin + ci) * K + k] : 0.f;
}

// ConvTranspose1d weight [Cin][Cout][K], K = 2u  ->  per output phase r a 2-tap Conv1d over the input positions:
//   y[co][q*u + r - p] = b + sum_ci ( x[ci][q-1] * w[ci][co][r+u] + x[ci][q] * w[ci][co][r] )
// packed like pack_conv_w_kernel: out[r][co-block][ci][k'][CO], k' = 0 -> tap x[q-1], k' = 1 -> tap x[q]
__global__ void pack_convT_w_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int u, int CO) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nblk = (Cout + CO - 1) / CO;
  const int per_phase = nblk * Cin * 2 * CO;
  if (i >= u * per_phase) return;
  const int r = i / per_phase, rem = i - r * per_phase;
  const int c = rem % CO, kk = (rem / CO) % 2, ci = (rem / (2 * CO)) % Cin, blk = rem / (2 * CO * Cin);
  const int co = blk * CO + c;
  out[i] = co < Cout ? w[((long long)ci * Cout + co) * (2 * u) + r + (1 - kk) * u] : 0.f;
}

template <int K, int TT>
int launch_conv_kt(const ConvArgs& a, int B, hipStream_t st) {
  constexpr int TILE = 256 * TT;
  const size_t lds = (size_t)CI_CHUNK * (TILE + (K - 1) * a.dil) * sizeof(float);
  if (a.Cout >= 16) {
    dim3 grid(cdiv(a.L, TILE), cdiv(a.Cout, 16), B);
    hipLaunchKernelGGL((conv1d_kernel<K, 16, TT>), grid, dim3(256), lds, st, a);
  } else {
    dim3 grid(cdiv(a.L, TILE), cdiv(a.Cout, 8), B);
    hipLaunchKernelGGL((conv1d_kernel<K, 8, TT>), grid, dim3(256), lds, st, a);
  }
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

template <int K>
int launch_conv_k(const ConvArgs& a, int B, hipStream_t st) {
  // workgroups at 1024 samples per workgroup; below ~2 per CU (256 CUs) use shorter tiles
  const long long wgs4 = (long long)cdiv(a.L, 1024) * cdiv(a.Cout, a.Cout >= 16 ? 16 : 8) * B;
  if (wgs4 >= 512) return launch_conv_kt<K, 4>(a, B, st);
  if (wgs4 >= 256) return launch_conv_kt<K, 2>(a, B, st);
  return launch_conv_kt<K, 1>(a, B, st);
}

// polyphase ConvTranspose1d (K = 2u): one launch, blockIdx.z = (batch row, phase), 2-tap convs with strided stores
template <int TT>
int launch_convT_t(const ConvArgs& a, int B, hipStream_t st) {
  constexpr int TILE = 256 * TT;
  const size_t lds = (size_t)CI_CHUNK * (TILE + 1) * sizeof(float);
  if (a.Cout >= 16) hipLaunchKernelGGL((conv1d_kernel<2, 16, TT>), dim3(cdiv(a.Lq, TILE), cdiv(a.Cout, 16), B * a.n_phase), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((conv1d_kernel<2, 8, TT>), dim3(cdiv(a.Lq, TILE), cdiv(a.Cout, 8), B * a.n_phase), dim3(256), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
int launch_convT(const ConvArgs& a, int B, hipStream_t st) {
  const long long wgs4 = (long long)cdiv(a.Lq, 1024) * cdiv(a.Cout, a.Cout >= 16 ? 16 : 8) * B * a.n_phase;
  if (wgs4 >= 512) return launch_convT_t<4>(a, B, st);
  if (wgs4 >= 256) return launch_convT_t<2>(a, B, st);
  return launch_convT_t<1>(a, B, st);
}

template <int K, int C, int TT>
int launch_pair_t(const PairArgs& a, int B, hipStream_t st) {
  constexpr int P = 256 * TT, POUT = P - (K - 1);
  const size_t lds = ((size_t)C * P + 16) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    BSG_HIP(hipFuncSetAttribute((const void*)resblock_pair_kernel<K, C, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  hipLaunchKernelGGL((resblock_pair_kernel<K, C, TT>), dim3(cdiv(a.L, POUT), B), dim3(256), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
// positions per thread: as many as the 64-accumulator budget allows (C TT <= 64), fewer while the launch would leave CUs idle
template <int K, int C>
int launch_pair_kc(const PairArgs& a, int B, hipStream_t st) {
  auto wgs = [&](int tt) { return (long long)cdiv(a.L, 256 * tt - (K - 1)) * B; };
  static int cap = -1;
  if (cap < 0) { const char* e = getenv("BSG_HG_ACC"); cap = e ? atoi(e) : 16; }   // accumulators per thread (C TT): 16 measured best (13.6 vs 16.9 ms per forward at 64: occupancy)
  if constexpr (C * 8 <= 64) { if (C * 8 <= cap && wgs(8) >= 512) return launch_pair_t<K, C, 8>(a, B, st); }
  if constexpr (C * 4 <= 64) { if (C * 4 <= cap && wgs(4) >= 512) return launch_pair_t<K, C, 4>(a, B, st); }
  if constexpr (C * 2 <= 64) { if (C * 2 <= cap && wgs(2) >= 512) return launch_pair_t<K, C, 2>(a, B, st); }
  return launch_pair_t<K, C, 1>(a, B, st);
}
template <int K>
int launch_pair_k(const PairArgs& a, int C, int B, hipStream_t st) {
  switch (C) {
    case 8: return launch_pair_kc<K, 8>(a, B, st);
    case 16: return launch_pair_kc<K, 16>(a, B, st);
    case 32: return launch_pair_kc<K, 32>(a, B, st);
    case 64: return launch_pair_kc<K, 64>(a, B, st);
    default: return BSG_EINVAL;
  }
}
// ------------------------------------------------------------------------------------------------
// The same fused pair with every product on the 16-bit matrix pipe (split-fp16: gemm.hip / diffnet_h2.hip have the argument).  Weights
// x 2^8 and activations x 2^4 are split exactly into hi + lo fp16 terms — the weights once at create (pack_conv_h2_kernel), lrelu(x) while
// it is staged, the intermediate t1 in registers — and each fp32 product is hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16 with fp32
// accumulation (the accumulators carry 2^12): per tap and 16 input channels 3 MFMAs of 32 matrix cycles instead of 8 of 64.  The LDS
// images are channels-last [position][C fp16 + 16 B] planes (an MFMA B fragment = 8 consecutive input channels of one position = one
// ds_read_b128; the row stride of 36 / 20 dwords keeps the 16 rows of a lane group on different bank quads); the staging transposes on
// the fly (lanes = consecutive positions: coalesced rows of [C][L]; a thread packs 4 channels into one 8-byte write per plane).
// ------------------------------------------------------------------------------------------------
using hf16x8 = __attribute__((ext_vector_type(8))) _Float16;
using hf16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
constexpr float HG_WSC = 256.0f, HG_ASC = 16.0f;

// out[((((rt*K + k)*(C/16) + ks)*2 + plane)*64 + lane)*8 + j] = hi / lo of 2^8 W[co = 32 rt + (lane & 31)][ci = 16 ks + 8 (lane >> 5) + j][k]
// `bad` counts weights whose scaled value leaves the fp16 range (or is not finite): the handle then keeps the fp32-MFMA form
__global__ void pack_conv_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int C, int K, unsigned* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * C * K) return;
  const int j = i & 7, lane = (i >> 3) & 63, rest = i >> 9;
  const int KS = C / 16;
  const int ks = rest % KS, k = (rest / KS) % K, rt = rest / (KS * K);
  const float v = w[((long long)(32 * rt + (lane & 31)) * C + 16 * ks + 8 * (lane >> 5) + j) * K + k] * HG_WSC;
  if (!(fabsf(v) < 60000.0f)) atomicAdd(bad, 1u);
  const _Float16 hi = (_Float16)v;
  const long long base = ((((long long)(rt * K + k) * KS + ks) * 2) * 64 + lane) * 8 + j;
  out[base] = hi;
  out[base + 512] = (_Float16)(v - (float)hi);
}

// acc[rt][nb] += W * img for one conv: img = LDS planes [rows][rowb] (lo plane at + plane), this wave's positions start at row n0; tap k
// reads row p + k * dil.
// The weight fragments stream from L2 through a RING of D k-steps (round 4).  Until round 3 they were requested ONE k-step ahead: a k-step is
// 12 (C = 64) or 6 (C = 32) MFMAs = 384 / 192 matrix clocks, an L2 round trip is ~1 000, so every k-step ended waiting for its weights (the
// pairs ran at 13-32 % of the matrix pipe).  D = 4 k-steps for 64 channels (1 536 clocks ahead), 8 for 32; the loop over taps and channel slices
// is fully unrolled so that the ring slots are compile-time registers.
template <int K, int C, int NB>
__device__ __forceinline__ void conv_h2(f32x16 (&acc)[C / 32][NB], const _Float16* __restrict__ wpk, const char* img, int rowb, int plane,
                                        int n0, int dil, int lane) {
  constexpr int RT = C / 32, KS = C / 16, NS = K * KS;
  constexpr int D0 = RT == 2 ? 4 : 8, D = D0 < NS ? D0 : NS;
  const int l31 = lane & 31, lh = lane >> 5;
  const hf16x8* __restrict__ wp = reinterpret_cast<const hf16x8*>(wpk) + lane;
  const char* bp = img + (n0 + l31) * rowb + lh * 16;
  hf16x8 A[D][RT][2];
  auto lda = [&](int i, int slot) {   // k-step i = tap i / KS, channel slice i % KS
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      A[slot][rt][0] = wp[((rt * K + i / KS) * KS + i % KS) * 128];
      A[slot][rt][1] = wp[((rt * K + i / KS) * KS + i % KS) * 128 + 64];
    }
  };
#pragma unroll
  for (int i = 0; i < D; ++i) lda(i, i);
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int k = i / KS, ks = i % KS;
    const char* bk = bp + k * dil * rowb;
    hf16x8 bh[NB], bl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bh[nb] = *reinterpret_cast<const hf16x8*>(bk + 32 * nb * rowb + ks * 32);
      bl[nb] = *reinterpret_cast<const hf16x8*>(bk + 32 * nb * rowb + ks * 32 + plane);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i % D][rt][0], bh[nb], acc[rt][nb], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i % D][rt][0], bl[nb], acc[rt][nb], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i % D][rt][1], bh[nb], acc[rt][nb], 0, 0, 0);
    if (i + D < NS) lda(i + D, i % D);   // the slot is free: refill it D k-steps ahead
    // pin the issue order (the scheduler otherwise sinks the refills towards their use to save registers, which is the round-3 form again):
    // the B fragments' LDS reads, then the MFMAs with the ring's refills among the first of them
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NB, 0);
#pragma unroll
    for (int q = 0; q < 2 * RT; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * RT * NB - 2 * RT, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int K, int C, int NB>
__global__ __launch_bounds__(256) void resblock_pair_h2_kernel(PairArgs a) {
  constexpr int PT = 128 * NB, H2 = (K - 1) / 2, POUT = PT - (K - 1), RT = C / 32, TS = PT + 16, ROWB = 2 * C + 16;
  constexpr float ACC_SC = HG_WSC * HG_ASC, ACC_INV = 1.0f / (HG_WSC * HG_ASC);
  static_assert(C % 32 == 0, "channel count");
  extern __shared__ __attribute__((aligned(16))) char hlds[];   // lrelu(x) planes [span][ROWB], then t1 planes [TS][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int t0 = blockIdx.x * POUT, b = blockIdx.y;
  const int h1 = H2 * a.dil, span = PT + 2 * h1;
  const int plane = (span > TS ? span : TS) * ROWB;
  const float* __restrict__ xb = a.x + (long long)b * C * a.L;
  const float slope = a.slope;
  bool bad = false;
  auto split4 = [&](const float (&v)[4], hf16x4& hi, hf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e] * HG_ASC;
      bad |= !(fabsf(x) < 65000.0f);
      hi[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)hi[e]);
    }
  };
  // ---- stage lrelu(x) over [t0 - H2 - h1, + span), zero outside [0, L): item = (4 channels, position), lanes = consecutive positions ----
  // the residual's x values in accumulator layout, requested first (see resblock_pair_h16_kernel): the staging loop then finds their lines in
  // L2, and the epilogue does not fetch the tile's x from HBM a second time.  Only where the 32 registers do not cost a resident wave: C = 32
  // with K = 3 (117 -> 93 us); with the 8-slot weight ring of K = 7 / 11 the kernel drops from 3 to 2 waves per SIMD and runs 8 % SLOWER
  // (191 / 150 against 177 / 137 us), and C = 64 would need 64 registers
  constexpr bool XRES = C == 32 && K == 3;
  float xres[XRES ? NB : 1][XRES ? RT : 1][16];
  if constexpr (XRES) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int pcol = 32 * NB * wave + 32 * nb + l31, t = t0 + pcol;
      const bool ok = pcol < POUT && t < a.L;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) xres[nb][rt][r] = ok ? xb[(long long)(32 * rt + 4 * lh + (r & 3) + 8 * (r >> 2)) * a.L + t] : 0.f;
    }
  }
  // (4 items' loads are requested before the first of them is split and written: the trip count depends on the dilation, so the compiler
  // leaves the loop rolled, and a rolled loop pays one memory round trip per item — ~10 per thread)
  for (int idx0 = tid; idx0 < (C / 4) * span; idx0 += 4 * 256) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + 256 * u;
      const int cq = idx / span, jx = idx - cq * span;
      const int t = t0 - H2 - h1 + jx;
      const bool ok = idx < (C / 4) * span && t >= 0 && t < a.L;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = ok ? xb[(long long)(4 * cq + e) * a.L + t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + 256 * u;
      if (idx >= (C / 4) * span) break;
      const int cq = idx / span, jx = idx - cq * span;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = fmaxf(v[u][e], v[u][e] * slope);
      hf16x4 hi, lo;
      split4(v[u], hi, lo);
      *reinterpret_cast<hf16x4*>(hlds + jx * ROWB + cq * 8) = hi;
      *reinterpret_cast<hf16x4*>(hlds + plane + jx * ROWB + cq * 8) = lo;
    }
  }
  const int n0 = 32 * NB * wave;
  f32x16 acc[RT][NB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = a.b1[32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh] * ACC_SC;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb][r] = bv;
    }
  __syncthreads();
  // ---- conv1 (dilation d) at t1 positions u = t0 - H2 + p: reads x row p + k d ------------------------------------------------------
  conv_h2<K, C, NB>(acc, reinterpret_cast<const _Float16*>(a.w1), hlds, ROWB, plane, n0, a.dil, lane);
  __syncthreads();   // every wave is done reading x
  // ---- t1 = lrelu(conv1), zero outside [0, L) (conv2 pads its input) -> LDS planes [TS][ROWB] -----------------------------------------
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int pcol = n0 + 32 * nb + l31, u = t0 - H2 + pcol;
    const bool in = u >= 0 && u < a.L;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = acc[rt][nb][4 * g + e] * ACC_INV;
          v[e] = in ? fmaxf(y, y * slope) : 0.f;
        }
        hf16x4 hi, lo;
        split4(v, hi, lo);
        char* dst = hlds + pcol * ROWB + (32 * rt + 8 * g + 4 * lh) * 2;
        *reinterpret_cast<hf16x4*>(dst) = hi;
        *reinterpret_cast<hf16x4*>(dst + plane) = lo;
      }
  }
  // rows PT .. TS-1 are read by the (masked) last K-1 positions of the tile: keep them finite (both planes)
  for (int idx = tid; idx < (TS - PT) * (ROWB / 4); idx += 256) {
    const int row = PT + idx / (ROWB / 4), c = idx % (ROWB / 4);
    *reinterpret_cast<unsigned*>(hlds + row * ROWB + c * 4) = 0u;
    *reinterpret_cast<unsigned*>(hlds + plane + row * ROWB + c * 4) = 0u;
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = a.b2[32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh] * ACC_SC;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb][r] = bv;
    }
  __syncthreads();
  // ---- conv2 (dilation 1) at output positions t = t0 + p: reads t1 row p + k ----------------------------------------------------------
  conv_h2<K, C, NB>(acc, reinterpret_cast<const _Float16*>(a.w2), hlds, ROWB, plane, n0, 1, lane);
  if (a.range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(a.range_events, 1u);
  // ---- residual, MRF sum, store ----------------------------------------------------------------------------------------------------------
  const bool has_acc = a.acc_in != nullptr, has_div = a.out_div != 1.0f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int pcol = n0 + 32 * nb + l31, t = t0 + pcol;
    if (pcol >= POUT || t >= a.L) continue;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const long long i0 = ((long long)b * C + 32 * rt + 4 * lh) * a.L + t;
      float xr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (XRES) xr[r] = xres[nb][rt][r];
        else xr[r] = a.x[i0 + (long long)((r & 3) + 8 * (r >> 2)) * a.L];
      }
      if (has_acc) {
        float ar[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ar[r] = a.acc_in[i0 + (long long)((r & 3) + 8 * (r >> 2)) * a.L];
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = ar[r] + (acc[rt][nb][r] * ACC_INV + xr[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = acc[rt][nb][r] * ACC_INV + xr[r];
      }
      if (has_div) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = xr[r] / a.out_div;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) a.y[i0 + (long long)((r & 3) + 8 * (r >> 2)) * a.L] = xr[r];
    }
  }
}

template <int K, int C, int NB>
int launch_pair_h2_t(const PairArgs& a, int B, hipStream_t st) {
  constexpr int PT = 128 * NB, POUT = PT - (K - 1), ROWB = 2 * C + 16;
  const int span = PT + 2 * ((K - 1) / 2 * a.dil);
  const size_t lds = (size_t)2 * (span > PT + 16 ? span : PT + 16) * ROWB;
  static size_t attr = 0;
  if (lds > attr) {
    BSG_HIP(hipFuncSetAttribute((const void*)resblock_pair_h2_kernel<K, C, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  hipLaunchKernelGGL((resblock_pair_h2_kernel<K, C, NB>), dim3(cdiv(a.L, POUT), B), dim3(256), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
template <int K, int C>
int launch_pair_h2_kc(const PairArgs& a, int B, hipStream_t st) {
  if ((long long)cdiv(a.L, 256 - (K - 1)) * B >= 256) return launch_pair_h2_t<K, C, 2>(a, B, st);
  return launch_pair_h2_t<K, C, 1>(a, B, st);
}
int launch_pair_h2(const PairArgs& a, int K, int C, int B, hipStream_t st) {
  if (C == 32) {
    if (K == 3) return launch_pair_h2_kc<3, 32>(a, B, st);
    if (K == 7) return launch_pair_h2_kc<7, 32>(a, B, st);
    if (K == 11) return launch_pair_h2_kc<11, 32>(a, B, st);
  } else if (C == 64) {
    if (K == 3) return launch_pair_h2_kc<3, 64>(a, B, st);
    if (K == 7) return launch_pair_h2_kc<7, 64>(a, B, st);
    if (K == 11) return launch_pair_h2_kc<11, 64>(a, B, st);
  }
  set_error("hifigan: no split-fp16 pair kernel for K=%d C=%d", K, C);
  return BSG_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// The same fused pair for 16 and 8 channels on the 16-bit matrix pipe (round 3): v_mfma_f32_16x16x32_f16, M = 16 output channels (8 used for
// C = 8), N = 16 positions, K = 32 = TPK taps x C input channels (TPK = 2 for C = 16, 4 for C = 8; the tap count is padded to a multiple
// of TPK with zero weights, and the padded taps read the last real tap's row so that no garbage meets a zero).  Lane l holds A[row l & 15]
// [k = 8 (l >> 4) + j] and B[k][column l & 15] (cdna_hip_programming.md, "A/B operand lane maps"): a B fragment = 8 consecutive channels of
// ONE position = one ds_read_b128 of the channels-last image [position][C fp16 + pad] (48-byte rows: the 16 rows of a lane group fall on
// 16 different bank quads); C/D: column l & 15, rows 4 (l >> 4) + r.  Split-fp16 products as above (weights x 2^8 at create, activations
// x 2^4 while staged, 3 MFMAs per product).  These stages move 96 / 48 B per position and pair against 33 / 17 kFLOP: the VALU kernels
// they replace ran at 1.2-2 TB/s, these are bound by the memory side.
// ------------------------------------------------------------------------------------------------
using f32x4h = __attribute__((ext_vector_type(4))) float;

// out[((ks*2 + plane)*64 + lane)*8 + j] = hi / lo of 2^8 W[co = lane & 15][ci][tap] with kb = lane >> 4,
//   C = 16: tap = 2 ks + (kb >> 1), ci = 8 (kb & 1) + j;   C = 8: tap = 4 ks + kb, ci = j;   zero for co >= C or tap >= K
__global__ void pack_conv_h16_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int C, int K, unsigned* __restrict__ bad) {
  const int TPK = 32 / C, KS = (K + TPK - 1) / TPK;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KS * 512) return;
  const int j = i & 7, lane = (i >> 3) & 63, ks = i >> 9;
  const int co = lane & 15, kb = lane >> 4;
  const int tap = C == 16 ? 2 * ks + (kb >> 1) : 4 * ks + kb;
  const int ci = C == 16 ? 8 * (kb & 1) + j : j;
  float v = 0.f;
  if (co < C && tap < K) v = w[((long long)co * C + ci) * K + tap] * HG_WSC;
  if (!(fabsf(v) < 60000.0f)) atomicAdd(bad, 1u);
  const _Float16 hi = (_Float16)v;
  const long long base = ((long long)(ks * 2) * 64 + lane) * 8 + j;
  out[base] = hi;
  out[base + 512] = (_Float16)(v - (float)hi);
}

// LDS image of the 16-row forms (round 5): channels-last rows of 2 C bytes, NO padding — 32 B for 16 channels, 16 B for 8 — per plane.  A B
// fragment is one ds_read_b128 of 8 channels of one position; ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
// (MI355X_MICROARCH.md, LDS): 8 lanes of one k-block and 8 of the next.  With 48-byte rows (rounds 3-4) every group had five 2-way conflicts — PMC: half
// of SQ_LDS_IDX_ACTIVE was SQ_LDS_BANK_CONFLICT.  With 32-byte rows the two k-blocks of a tap (channel halves 0 / 1) land on even / odd 16-byte
// slots and a group covers all 16 slots of the 256-byte bank row once — provided the halves of rows 4..7 (mod 8) are swapped, which also halves the
// conflicts of the 8-byte stores (rows r and r + 4 of one channel quad no longer share a bank).  8 channels: one slot per row, rows of a group are
// consecutive or identical (broadcast): conflict-free except where the taps of two k-blocks are 16 rows apart (d = 5: one pair per group).
template <int C>
__device__ __forceinline__ int h16_off(int row, int byte) {   // byte offset of (row, byte within the logical row) in a plane
  if (C == 16) return row * 32 + (byte ^ ((row & 4) << 2));
  return row * 16 + byte;
}

// Exact hi + lo fp16 split of four values that are ALREADY scaled (x 2^4), 6 instructions: the compiler's form of `hi = (f16)x; lo = (f16)(x -
// (float)hi)` is convert, convert back, subtract, convert, pack — 4.5 per value, and the 8- / 16-channel launches are bound by exactly this
// vector work (PMC: 23 VALU instructions per MFMA in the 8-channel chain).  v_fma_mix{lo,hi}_f16 takes the f16 hi straight from the packed
// pair, forms x - hi in fp32 (exact) and rounds it to f16 into one half of the destination: the same two roundings, bit for bit
// (tools/split_asm_check.hip).  `worst` collects max |x| as integer bits (NaN included) for the range guard.
struct Split4 { unsigned h0, h1, l0, l1; };
#ifndef BSG_HG_ASM_SPLIT
#define BSG_HG_ASM_SPLIT 1
#endif
__device__ __forceinline__ Split4 split4_scaled(float x0, float x1, float x2, float x3, unsigned& worst) {
  worst = max(max(worst, max(__builtin_bit_cast(unsigned, x0) & 0x7fffffffu, __builtin_bit_cast(unsigned, x1) & 0x7fffffffu)),
              max(__builtin_bit_cast(unsigned, x2) & 0x7fffffffu, __builtin_bit_cast(unsigned, x3) & 0x7fffffffu));
  Split4 r;
#if BSG_HG_ASM_SPLIT && defined(__HIP_DEVICE_COMPILE__)   // (the host pass of hipcc parses the function too)
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.h0) : "v"(x0), "v"(x1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.h1) : "v"(x2), "v"(x3));
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r.l0) : "v"(x0), "v"(x1), "v"(r.h0));
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r.l1) : "v"(x2), "v"(x3), "v"(r.h1));
#else
  const _Float16 a = (_Float16)x0, b = (_Float16)x1, c = (_Float16)x2, d = (_Float16)x3;
  using h2 = __attribute__((ext_vector_type(2))) _Float16;
  r.h0 = __builtin_bit_cast(unsigned, h2{a, b});
  r.h1 = __builtin_bit_cast(unsigned, h2{c, d});
  r.l0 = __builtin_bit_cast(unsigned, h2{(_Float16)(x0 - (float)a), (_Float16)(x1 - (float)b)});
  r.l1 = __builtin_bit_cast(unsigned, h2{(_Float16)(x2 - (float)c), (_Float16)(x3 - (float)d)});
#endif
  return r;
}
constexpr unsigned HG_RANGE_BITS = 0x477DE800u;   // 65000.0f: a scaled value at or beyond it (or not finite) counts as a range event

// the whole conv's weights (KS x 2 KB of fragments) into registers: requested a phase BEFORE the conv that uses them (under the staging loop, the
// t1 / image writes and the barrier), so that no conv starts with an L2 round trip
template <int K, int C>
__device__ __forceinline__ void conv_h16_weights(hf16x8 (&A)[(K + 32 / C - 1) / (32 / C)][2], const float* __restrict__ wpk, int lane) {
  constexpr int TPK = 32 / C, KS = (K + TPK - 1) / TPK;
  const hf16x8* __restrict__ wp = reinterpret_cast<const hf16x8*>(wpk) + lane;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    A[ks][0] = wp[ks * 128];
    A[ks][1] = wp[ks * 128 + 64];
  }
}

#ifndef BSG_HG_PF
#define BSG_HG_PF 2
#endif
// acc[ct] += W * img over the wave's NC column tiles of 16 positions: tap k of position p reads image row row0 + p + k dil (row0 = the wave's
// first row for tap 0).  BSG_HG_PF (build-time): column tiles per prefetch group — the B fragments of a step (a k-step x a group) are read from LDS
// while the previous step's 3 G MFMAs run, and the MFMAs of a step are ordered product-major so that consecutive ones write different accumulators;
// 0 = the first form (2 fragments, wait, 3 MFMAs on one accumulator).  Per accumulator the order of the products is the same in all forms.
template <int K, int C, int NC>   // NC column tiles of 16 positions per wave
__device__ __forceinline__ void conv_h16(f32x4h (&acc)[NC], const hf16x8 (&A)[(K + 32 / C - 1) / (32 / C)][2], const char* img, int plane, int row0,
                                         int dil, int lane) {
  constexpr int TPK = 32 / C, KS = (K + TPK - 1) / TPK, ROWB = 2 * C;
  const int l15 = lane & 15, kb = lane >> 4;
  auto frag = [&](int ks) {   // this lane's fragment address of k-step ks, column tile 0
    int tap = C == 16 ? 2 * ks + (kb >> 1) : 4 * ks + kb;
    if (tap > K - 1) tap = K - 1;   // padded tap (zero weights): a row that exists
    return img + h16_off<C>(row0 + l15 + tap * dil, C == 16 ? (kb & 1) * 16 : 0);
  };
#if BSG_HG_PF == 0
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const char* bk = frag(ks);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const hf16x8 bh1 = *reinterpret_cast<const hf16x8*>(bk + 16 * ct * ROWB);
      const hf16x8 bl1 = *reinterpret_cast<const hf16x8*>(bk + 16 * ct * ROWB + plane);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks][0], bh1, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks][0], bl1, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks][1], bh1, acc[ct], 0, 0, 0);
    }
  }
#else
  constexpr int G = NC < BSG_HG_PF ? NC : BSG_HG_PF, NG = NC / G, NSTEP = KS * NG;
  static_assert(NC % G == 0, "column tiles per wave");
  hf16x8 bh[2][G], bl[2][G];
  auto ldb = [&](int step, int buf) {
    const char* bk = frag(step / NG) + 16 * G * (step % NG) * ROWB;   // (16 rows further: the swizzle bit of the row is unchanged)
#pragma unroll
    for (int j = 0; j < G; ++j) {
      bh[buf][j] = *reinterpret_cast<const hf16x8*>(bk + 16 * j * ROWB);
      bl[buf][j] = *reinterpret_cast<const hf16x8*>(bk + 16 * j * ROWB + plane);
    }
  };
  ldb(0, 0);
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int ks = step / NG, g = step % NG, buf = step & 1;
    if (step + 1 < NSTEP) ldb(step + 1, buf ^ 1);
#pragma unroll
    for (int j = 0; j < G; ++j) acc[G * g + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks][0], bh[buf][j], acc[G * g + j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < G; ++j) acc[G * g + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks][0], bl[buf][j], acc[G * g + j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < G; ++j) acc[G * g + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks][1], bh[buf][j], acc[G * g + j], 0, 0, 0);
    if (step + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 2 * G, 0);   // the next step's LDS reads are issued first ...
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * G, 0);                          // ... and land under this step's MFMAs
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

template <int K, int C, int NB>
__global__ __launch_bounds__(256) void resblock_pair_h16_kernel(PairArgs a) {
  constexpr int PT = 128 * NB, H2 = (K - 1) / 2, POUT = PT - (K - 1), TS = PT + 16, ROWB = 2 * C, NC = 2 * NB;
  constexpr float ACC_SC = HG_WSC * HG_ASC, ACC_INV = 1.0f / (HG_WSC * HG_ASC);
  static_assert(C == 16 || C == 8, "channel count");
  extern __shared__ __attribute__((aligned(16))) char hlds[];   // lrelu(x) planes [span][ROWB], then t1 planes [TS][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kb = lane >> 4;
  const int t0 = blockIdx.x * POUT, b = blockIdx.y;
  const int h1 = H2 * a.dil, span = PT + 2 * h1;
  const int plane = (span > TS ? span : TS) * ROWB;
  const float* __restrict__ xb = a.x + (long long)b * C * a.L;
  const float slope = a.slope;
  bool bad = false;
  auto split4 = [&](const float (&v)[4], hf16x4& hi, hf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e] * HG_ASC;
      bad |= !(fabsf(x) < 65000.0f);
      hi[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)hi[e]);
    }
  };
  // ---- stage lrelu(x) over [t0 - H2 - h1, + span), zero outside [0, L): item = (4 channels, position), lanes = consecutive positions ----
  // the residual's x values in ACCUMULATOR layout (16 registers), requested first: the staging loop below then finds their lines in L2 instead
  // of the epilogue fetching the tile's x from HBM a second time, ~100 us after it was staged (PMC: 2.7 x the tensor fetched per pair)
  hf16x8 Aw[(K + 32 / C - 1) / (32 / C)][2];
  conv_h16_weights<K, C>(Aw, a.w1, lane);
  float xres[2 * NB][4];
  {
    const bool rows_ok_ = 4 * (lane >> 4) < C;
#pragma unroll
    for (int ct = 0; ct < 2 * NB; ++ct) {
      const int pcol = 32 * NB * wave + 16 * ct + (lane & 15), t = t0 + pcol;
      const bool ok = rows_ok_ && pcol < POUT && t < a.L;
#pragma unroll
      for (int r = 0; r < 4; ++r) xres[ct][r] = ok ? xb[(long long)(4 * (lane >> 4) + r) * a.L + t] : 0.f;
    }
  }
  // (the loads of 4 items are requested before the first of them is split and written: see resblock_pair_h2_kernel)
  for (int idx0 = tid; idx0 < (C / 4) * span; idx0 += 4 * 256) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + 256 * u;
      const int cq = idx / span, jx = idx - cq * span;
      const int t = t0 - H2 - h1 + jx;
      const bool ok = idx < (C / 4) * span && t >= 0 && t < a.L;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = ok ? xb[(long long)(4 * cq + e) * a.L + t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + 256 * u;
      if (idx >= (C / 4) * span) break;
      const int cq = idx / span, jx = idx - cq * span;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = fmaxf(v[u][e], v[u][e] * slope);
      hf16x4 hi, lo;
      split4(v[u], hi, lo);
      *reinterpret_cast<hf16x4*>(hlds + h16_off<C>(jx, cq * 8)) = hi;
      *reinterpret_cast<hf16x4*>(hlds + plane + h16_off<C>(jx, cq * 8)) = lo;
    }
  }
  const int n0 = 32 * NB * wave;
  const bool rows_ok = 4 * kb < C;   // C = 8: accumulator rows 8..15 do not exist
  f32x4h acc[NC];
  auto init_acc = [&](const float* bias) {
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = rows_ok ? bias[4 * kb + r] * ACC_SC : 0.f;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = f32x4h{bv[0], bv[1], bv[2], bv[3]};
  };
  init_acc(a.b1);
  __syncthreads();
  // ---- conv1 (dilation d) at t1 positions u = t0 - H2 + p: reads x row p + k d ------------------------------------------------------
  conv_h16<K, C, NC>(acc, Aw, hlds, plane, n0, a.dil, lane);
  conv_h16_weights<K, C>(Aw, a.w2, lane);
  __syncthreads();   // every wave is done reading x
  // ---- t1 = lrelu(conv1), zero outside [0, L) (conv2 pads its input) -> LDS planes [TS][ROWB] -----------------------------------------
  if (rows_ok) {
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const int pcol = n0 + 16 * ct + l15, u = t0 - H2 + pcol;
      const bool in = u >= 0 && u < a.L;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = acc[ct][e] * ACC_INV;
        v[e] = in ? fmaxf(y, y * slope) : 0.f;
      }
      hf16x4 hi, lo;
      split4(v, hi, lo);
      char* dst = hlds + h16_off<C>(pcol, (4 * kb) * 2);
      *reinterpret_cast<hf16x4*>(dst) = hi;
      *reinterpret_cast<hf16x4*>(dst + plane) = lo;
    }
  }
  // rows PT .. TS-1 are read by the (masked) last K-1 positions of the tile: keep them finite (both planes)
  for (int idx = tid; idx < (TS - PT) * (ROWB / 4); idx += 256) {
    const int row = PT + idx / (ROWB / 4), c = idx % (ROWB / 4);
    *reinterpret_cast<unsigned*>(hlds + row * ROWB + c * 4) = 0u;
    *reinterpret_cast<unsigned*>(hlds + plane + row * ROWB + c * 4) = 0u;
  }
  init_acc(a.b2);
  __syncthreads();
  // ---- conv2 (dilation 1) at output positions t = t0 + p: reads t1 row p + k ----------------------------------------------------------
  conv_h16<K, C, NC>(acc, Aw, hlds, plane, n0, 1, lane);
  if (a.range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(a.range_events, 1u);
  // ---- residual, MRF sum, store: lanes = 16 consecutive positions of 4 channels -------------------------------------------------------------
  if (!rows_ok) return;
  const bool has_acc = a.acc_in != nullptr, has_div = a.out_div != 1.0f;
#pragma unroll
  for (int ct = 0; ct < NC; ++ct) {
    const int pcol = n0 + 16 * ct + l15, t = t0 + pcol;
    if (pcol >= POUT || t >= a.L) continue;
    const long long i0 = ((long long)b * C + 4 * kb) * a.L + t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[ct][r] * ACC_INV + xres[ct][r];
      if (has_acc) v = a.acc_in[i0 + (long long)r * a.L] + v;
      if (has_div) v = v / a.out_div;
      a.y[i0 + (long long)r * a.L] = v;
    }
  }
}

template <int K, int C, int NB>
int launch_pair_h16_t(const PairArgs& a, int B, hipStream_t st) {
  constexpr int PT = 128 * NB, POUT = PT - (K - 1), ROWB = 2 * C;
  const int span = PT + 2 * ((K - 1) / 2 * a.dil);
  const size_t lds = (size_t)2 * (span > PT + 16 ? span : PT + 16) * ROWB;
  static size_t attr = 0;
  if (lds > attr) {
    BSG_HIP(hipFuncSetAttribute((const void*)resblock_pair_h16_kernel<K, C, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  hipLaunchKernelGGL((resblock_pair_h16_kernel<K, C, NB>), dim3(cdiv(a.L, POUT), B), dim3(256), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
template <int K, int C>
int launch_pair_h16_kc(const PairArgs& a, int B, hipStream_t st) {
  if ((long long)cdiv(a.L, 256 - (K - 1)) * B >= 256) return launch_pair_h16_t<K, C, 2>(a, B, st);
  return launch_pair_h16_t<K, C, 1>(a, B, st);
}
int launch_pair_h16(const PairArgs& a, int K, int C, int B, hipStream_t st) {
  if (C == 16) {
    if (K == 3) return launch_pair_h16_kc<3, 16>(a, B, st);
    if (K == 7) return launch_pair_h16_kc<7, 16>(a, B, st);
    if (K == 11) return launch_pair_h16_kc<11, 16>(a, B, st);
  }
  if (C == 8) {
    if (K == 3) return launch_pair_h16_kc<3, 8>(a, B, st);
    if (K == 7) return launch_pair_h16_kc<7, 8>(a, B, st);
    if (K == 11) return launch_pair_h16_kc<11, 8>(a, B, st);
  }
  set_error("hifigan: no 16-row split-fp16 pair kernel for K=%d, C=%d", K, C);
  return BSG_EINVAL;
}
// (C = 8 — 4 taps x 8 channels per k-step, half of the MFMA's rows unused — measured SLOWER in round 3 than the VALU pairs it would replace: 220 / 208
// / 150 us against 199 / 153 / 96 us for K = 11 / 7 / 3 at B=16, T=1000; for C = 16: 180 / 161 / 118 against 355 / 223 / 126 us.  With round 4's
// staging and residual prefetch: 176 / 160 / 120 us — ahead for K = 11 only, which is what the default takes.)
// BSG_HG_H16_C8: 8-channel pairs on the 16-row matrix form — default 11 = only K = 11 (176 against 195 us; K = 7 / 3 are faster on the vector
// pipe: 146 / 93 against 160 / 120 us), 0 = never, 1 = every K
static int h16_c8() { static int v = -1; if (v < 0) { const char* e = getenv("BSG_HG_H16_C8"); v = e ? atoi(e) : 11; } return v; }
bool pair_h16_supported(int K, int C) {
  return (K == 3 || K == 7 || K == 11) && (C == 16 || (C == 8 && (h16_c8() == 1 || (h16_c8() == 11 && K == 11))));
}


// ------------------------------------------------------------------------------------------------
// A whole ResBlock1 of 8 / 16 channels in ONE launch (round 5): its n_dil (dilated conv, conv) pairs run back to back on a
// tile that carries the block's receptive-field halo, HT = (K - 1) / 2 x sum(d + 1) positions per side (60 for K = 11 and d = 1, 3, 5), so the
// stage tensor is read ONCE and written once per ResBlock instead of once per pair (launched pair by pair a ResBlock moved the tensor through
// HBM 6-7 times).  The residual stream stays in registers in accumulator layout (fp32, as the pairs store it), the image of lrelu(x) / t1 in
// ONE LDS buffer of split-fp16 planes with zero pad rows on both sides — the convolutions are the pairs' conv_h16, the arithmetic is the
// pairs' in the same order, so the valid positions are bit-identical to the pair launches.  All PT positions of the tile are computed in every
// convolution; the outer HT see pad rows instead of their true neighbours and are never stored (at PT = 512: 23 % of the work for K = 11, 14 %
// for K = 7, 5 % for K = 3).  Positions outside [0, L) are written as zeros at every step (each Conv1d pads its own input, hifigan.py:54-61).
// ------------------------------------------------------------------------------------------------
struct ChainArgs {
  const float* x;        // [B][C][L]
  const float* w1[3];    // conv1 of pair m: hi / lo fp16 fragments (pack_conv_h16_kernel)
  const float* b1[3];
  const float* w2[3];
  const float* b2[3];
  float* y;              // [B][C][L]
  const float* acc_in;   // optional MRF running sum (may alias y: every element is read and written by the same lane)
  float out_div, slope;
  int L, n_pairs;
  int dil[3];
  unsigned* range_events;
};

template <int K, int C, int NC>   // NC column tiles of 16 positions per wave: a tile of PT = 64 NC positions
__global__ __launch_bounds__(256) void resblock_chain_h16_kernel(ChainArgs a) {
  constexpr int PT = 64 * NC, H2 = (K - 1) / 2, ROWB = 2 * C;
  constexpr float ACC_SC = HG_WSC * HG_ASC, ACC_INV = 1.0f / (HG_WSC * HG_ASC);
  static_assert(C == 16 || C == 8, "channel count");
  extern __shared__ __attribute__((aligned(16))) char hlds[];   // planes [pa + PT + pa][ROWB]: lrelu(x_m), then t1 of pair m, then lrelu(x_m+1) ...
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kb = lane >> 4;
  int ht = 0, dmax = 1;
  for (int m = 0; m < a.n_pairs; ++m) {
    ht += H2 * (a.dil[m] + 1);
    dmax = a.dil[m] > dmax ? a.dil[m] : dmax;
  }
  const int pa = H2 * dmax;              // pad rows on each side of the image (always zero)
  const int plane = (PT + 2 * pa) * ROWB;
  float* btab = reinterpret_cast<float*>(hlds + 2 * plane);   // [2 n_pairs][16]: the biases x 2^12 (a global load in front of every conv was an L2 round trip)
  if (tid < 32 * a.n_pairs) {
    const int cv = tid >> 4, r = tid & 15;
    btab[tid] = r < C ? ((cv & 1) ? a.b2[cv >> 1] : a.b1[cv >> 1])[r] * ACC_SC : 0.f;
  }
  const int pout = PT - 2 * ht;
  const int t0 = blockIdx.x * pout - ht, b = blockIdx.y;   // tile row p <-> position t0 + p
  const float* __restrict__ xb = a.x + (long long)b * C * a.L;
  const float slope = a.slope;
  const int n0 = 16 * NC * wave;
  const bool rows_ok = 4 * kb < C;       // C = 8: accumulator rows 8..15 do not exist
  unsigned worst = 0;   // max |scaled value| written to the image, as bits
  // ---- the residual stream x of the tile, accumulator layout (row 4 kb + r, column n0 + 16 ct + l15), zero outside [0, L) ----------------
  float xres[NC][4];
  unsigned in_mask = 0;
#pragma unroll
  for (int ct = 0; ct < NC; ++ct) {
    const int t = t0 + n0 + 16 * ct + l15;
    const bool in = t >= 0 && t < a.L;
    in_mask |= in ? 1u << ct : 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) xres[ct][r] = (in && rows_ok) ? xb[(long long)(4 * kb + r) * a.L + t] : 0.f;
  }
  for (int idx = tid; idx < 2 * pa * (ROWB / 4); idx += 256) {   // the pad rows, both planes, once
    const int r = idx / (ROWB / 4), c = idx % (ROWB / 4);
    const int row = r < pa ? r : PT + r;
    *reinterpret_cast<unsigned*>(hlds + row * ROWB + c * 4) = 0u;
    *reinterpret_cast<unsigned*>(hlds + plane + row * ROWB + c * 4) = 0u;
  }
  // rows pa + p <- split(lrelu(sc val(ct, r)) 2^4), sc a power of two: this lane's 4 channels of its NC positions.  lrelu commutes with the
  // power-of-two scales, so they are one product; a tile that lies inside [0, L) (all but the two at the ends of an utterance) skips the masks
  const bool all_in = __builtin_amdgcn_ballot_w64(in_mask != (1u << NC) - 1u) == 0ull;
  auto write_img = [&](auto&& val, float sc) {
    if (!rows_ok) return;
    const float sc_s = sc * slope;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = val(ct, e);
        v[e] = fmaxf(y * sc, y * sc_s);
      }
      Split4 q = split4_scaled(v[0], v[1], v[2], v[3], worst);
      if (!all_in && !(in_mask >> ct & 1u)) q = Split4{0u, 0u, 0u, 0u};
      char* dst = hlds + h16_off<C>(pa + n0 + 16 * ct + l15, (4 * kb) * 2);
      *reinterpret_cast<u32x2*>(dst) = u32x2{q.h0, q.h1};
      *reinterpret_cast<u32x2*>(dst + plane) = u32x2{q.l0, q.l1};
    }
  };
  f32x4h acc[NC];
  auto init_acc = [&](int cv) {   // behind a barrier that follows the table's writes
    const f32x4h bv = *reinterpret_cast<const f32x4h*>(btab + 16 * cv + 4 * kb);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = bv;
  };
  hf16x8 Aw[(K + 32 / C - 1) / (32 / C)][2];
  conv_h16_weights<K, C>(Aw, a.w1[0], lane);
  write_img([&](int ct, int e) { return xres[ct][e]; }, HG_ASC);
#pragma unroll 1
  for (int m = 0; m < a.n_pairs; ++m) {
    const int d = a.dil[m];
    __syncthreads();   // the image of lrelu(x_m) is complete
    init_acc(2 * m);
    // conv1 (dilation d): t1[p] reads image rows pa + p + (k - H2) d
    conv_h16<K, C, NC>(acc, Aw, hlds, plane, pa - H2 * d + n0, d, lane);
    conv_h16_weights<K, C>(Aw, a.w2[m], lane);
    __syncthreads();   // every wave is done reading x_m's image
    write_img([&](int ct, int e) { return acc[ct][e]; }, ACC_INV * HG_ASC);   // t1 = lrelu(conv1), zero outside [0, L)
    __syncthreads();
    init_acc(2 * m + 1);
    // conv2 (dilation 1): reads t1 rows pa + p + k - H2
    conv_h16<K, C, NC>(acc, Aw, hlds, plane, pa - H2 + n0, 1, lane);
    if (m + 1 < a.n_pairs) conv_h16_weights<K, C>(Aw, a.w1[m + 1], lane);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) xres[ct][r] = (in_mask >> ct & 1u) ? acc[ct][r] * ACC_INV + xres[ct][r] : 0.f;
    if (m + 1 < a.n_pairs) {
      __syncthreads();   // every wave is done reading t1
      write_img([&](int ct, int e) { return xres[ct][e]; }, HG_ASC);
    }
  }
  if (a.range_events && __builtin_amdgcn_ballot_w64(worst >= HG_RANGE_BITS) != 0ull && lane == 0) atomicAdd(a.range_events, 1u);
  // ---- MRF sum, store: the inner pout positions --------------------------------------------------------------------------------------------
  if (!rows_ok) return;
  const bool has_acc = a.acc_in != nullptr, has_div = a.out_div != 1.0f;
#pragma unroll
  for (int ct = 0; ct < NC; ++ct) {
    const int pcol = n0 + 16 * ct + l15, t = t0 + pcol;
    if (pcol < ht || pcol >= PT - ht || t >= a.L) continue;
    const long long i0 = ((long long)b * C + 4 * kb) * a.L + t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = xres[ct][r];
      if (has_acc) v = a.acc_in[i0 + (long long)r * a.L] + v;
      if (has_div) v = v / a.out_div;
      a.y[i0 + (long long)r * a.L] = v;
    }
  }
}

template <int K, int C, int NC>
int launch_chain_h16_t(const ChainArgs& a, int B, hipStream_t st) {
  constexpr int PT = 64 * NC, H2 = (K - 1) / 2, ROWB = 2 * C;
  int ht = 0, dmax = 1;
  for (int m = 0; m < a.n_pairs; ++m) { ht += H2 * (a.dil[m] + 1); dmax = a.dil[m] > dmax ? a.dil[m] : dmax; }
  const int pout = PT - 2 * ht;
  const size_t lds = (size_t)2 * (PT + 2 * H2 * dmax) * ROWB + 6 * 16 * sizeof(float);
  static size_t attr = 0;
  if (lds > attr) {
    BSG_HIP(hipFuncSetAttribute((const void*)resblock_chain_h16_kernel<K, C, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  hipLaunchKernelGGL((resblock_chain_h16_kernel<K, C, NC>), dim3(cdiv(a.L, pout), B), dim3(256), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
// the tile width by the number of tiles: 512 positions while that leaves >= 2 workgroups per CU, else 256 (more, smaller workgroups)
bool chain_h16_supported(int K, int C, int n_pairs, const int* dil) {
  if (!((K == 3 || K == 7 || K == 11) && (C == 16 || C == 8)) || n_pairs < 1 || n_pairs > 3) return false;
  int ht = 0, dmax = 1;
  for (int m = 0; m < n_pairs; ++m) { ht += (K - 1) / 2 * (dil[m] + 1); dmax = dil[m] > dmax ? dil[m] : dmax; }
  return 2 * ht <= 128 && (size_t)2 * (512 + (K - 1) * dmax) * 2 * C <= 80 * 1024;   // >= half of a 256-position tile is output; two workgroups per CU
}
template <int K, int C>
int launch_chain_h16_kc(const ChainArgs& a, int B, hipStream_t st) {
  int ht = 0;
  for (int m = 0; m < a.n_pairs; ++m) ht += (K - 1) / 2 * (a.dil[m] + 1);
  static int nc_env = -1;   // BSG_HG_CHAIN_NC = 4 / 8: force the tile width
  if (nc_env < 0) { const char* e = getenv("BSG_HG_CHAIN_NC"); nc_env = e ? atoi(e) : 0; }
  const bool wide = nc_env ? nc_env == 8 : (long long)cdiv(a.L, 512 - 2 * ht) * B >= 512;
  if (wide) return launch_chain_h16_t<K, C, 8>(a, B, st);
  return launch_chain_h16_t<K, C, 4>(a, B, st);
}
int launch_chain_h16(const ChainArgs& a, int K, int C, int B, hipStream_t st) {
  if (C == 16) {
    if (K == 3) return launch_chain_h16_kc<3, 16>(a, B, st);
    if (K == 7) return launch_chain_h16_kc<7, 16>(a, B, st);
    if (K == 11) return launch_chain_h16_kc<11, 16>(a, B, st);
  }
  if (C == 8) {
    if (K == 3) return launch_chain_h16_kc<3, 8>(a, B, st);
    if (K == 7) return launch_chain_h16_kc<7, 8>(a, B, st);
    if (K == 11) return launch_chain_h16_kc<11, 8>(a, B, st);
  }
  set_error("hifigan: no ResBlock chain kernel for K=%d, C=%d", K, C);
  return BSG_EINVAL;
}

template <int K, int C, int NB>
int launch_pair_mfma_t(const PairArgs& a, int B, hipStream_t st) {
  constexpr int PT = 128 * NB, POUT = PT - (K - 1);
  const int h1 = (K - 1) / 2 * a.dil;
  const int XS = (PT + 2 * h1 + 3) & ~3;
  const size_t lds = (size_t)C * (XS > PT + 16 ? XS : PT + 16) * sizeof(float);
  static size_t attr = 0;
  if (lds > attr) {
    BSG_HIP(hipFuncSetAttribute((const void*)resblock_pair_mfma_kernel<K, C, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  hipLaunchKernelGGL((resblock_pair_mfma_kernel<K, C, NB>), dim3(cdiv(a.L, POUT), B), dim3(256), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
template <int K, int C>
int launch_pair_mfma_kc(const PairArgs& a, int B, hipStream_t st) {
  // 256 positions per workgroup unless that leaves CUs without one (single utterances): then 128
  if ((long long)cdiv(a.L, 256 - (K - 1)) * B >= 256) return launch_pair_mfma_t<K, C, 2>(a, B, st);
  return launch_pair_mfma_t<K, C, 1>(a, B, st);
}
int launch_pair_mfma(const PairArgs& a, int K, int C, int B, hipStream_t st) {
  if (C == 32) {
    if (K == 3) return launch_pair_mfma_kc<3, 32>(a, B, st);
    if (K == 7) return launch_pair_mfma_kc<7, 32>(a, B, st);
    if (K == 11) return launch_pair_mfma_kc<11, 32>(a, B, st);
  } else if (C == 64) {
    if (K == 3) return launch_pair_mfma_kc<3, 64>(a, B, st);
    if (K == 7) return launch_pair_mfma_kc<7, 64>(a, B, st);
    if (K == 11) return launch_pair_mfma_kc<11, 64>(a, B, st);
  }
  return BSG_EINVAL;
}
bool pair_mfma_supported(int K, int C) { return (K == 3 || K == 7 || K == 11) && (C == 32 || C == 64); }
bool pair_supported(int K, int C) { return (K == 3 || K == 7 || K == 11) && (C == 8 || C == 16 || C == 32 || C == 64); }
int launch_pair(const PairArgs& a, int K, int C, int B, hipStream_t st) {
  switch (K) {
    case 3: return launch_pair_k<3>(a, C, B, st);
    case 7: return launch_pair_k<7>(a, C, B, st);
    case 11: return launch_pair_k<11>(a, C, B, st);
    default: return BSG_EINVAL;
  }
}

int launch_conv(const ConvArgs& a, int K, int B, hipStream_t st) {
  switch (K) {
    case 3: return launch_conv_k<3>(a, B, st);
    case 5: return launch_conv_k<5>(a, B, st);      // ResBlock2 configurations (HiFi-GAN V3: kernels 3, 5, 7)
    case 7: return launch_conv_k<7>(a, B, st);
    case 11: return launch_conv_k<11>(a, B, st);
    default: set_error("hifigan: conv kernel size %d not built (3, 5, 7, 11)", K); return BSG_EINVAL;
  }
}

}  // namespace
}  // namespace bsg

namespace bsg {
int nsf_launch_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w, const float* lin_b,
                      float* sw_tmp, float* har, int B, int T, int hop, int NH, float sr, hipStream_t st);
int nsf_launch_source_add(float* x, const float* har, const float* w, const float* bias, int B, int Cc, int Lx, long long Lh, int k,
                          int stride, int pad, hipStream_t st);
}
using namespace bsg;

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != BSG_OK) return _rc; \
  } while (0)

struct ConvW {
  float* w = nullptr;
  float* wpk = nullptr;   // Conv1d only: repacked for conv1d_kernel (CO_BLK = 16 if cout >= 16 else 8)
  float* wpc = nullptr;   // ResBlock convs: [cin][k][cout] for resblock_pair_kernel
  float* wpm = nullptr;   // ResBlock convs with 32 / 64 channels: MFMA fragment order for resblock_pair_mfma_kernel
  float* wps = nullptr;   // the same as hi / lo fp16 fragments (x 2^8) for resblock_pair_h2_kernel (m floats = 2 planes of m halves)
  float* wp16 = nullptr;  // 8 / 16 channels: hi / lo fp16 fragments of v_mfma_f32_16x16x32_f16 for resblock_pair_h16_kernel [ks][plane][64][8]
  float* wp16c = nullptr; // the same fragments for resblock_chain_h16_kernel: packed for every K of an 8- / 16-channel ResBlock1 (wp16 where that exists)
  float* wpu = nullptr;   // ConvTranspose1d with K = 2u: per-phase 2-tap weights in CO-blocks of 8 for upsample_kernel
  float* b = nullptr;
  int cout = 0, cin = 0, k = 0;
  // round 4: the split-fp16 GEMM with pre-split operands (gemm_h2w.hip) for the two dense convolutions outside the ResBlocks —
  // ConvTranspose1d(k = 16, u = 8) as ONE 2-tap product over (co, phase) weight rows with a polyphase store, conv_pre as a 7-tap product
  H2wWeights h2w;
  int h2w_kp = 0;         // input channels padded to a multiple of 64 in the activation planes
};

struct bsg_hifigan {
  bsg_hifigan_cfg cfg;
  Guard guard;   // this handle's range-event word and split-fp16 switch
  std::vector<float*> owned;
  std::vector<float*> create_tmp;   // reorder temporaries of the weight packing: freed behind create's stream synchronize
  ConvW pre, post;
  std::vector<ConvW> ups;                 // weight [Cin][Cout][K]
  std::vector<ConvW> rb1, rb2;            // [n_ups * n_kernels * n_dil]
  size_t cap = 0;                         // elements of one stage buffer
  float* buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // NSF (use_pitch_embed): source merge Linear(harmonics+1 -> 1) and one noise conv per upsampling stage
  float *src_w = nullptr, *src_b = nullptr;
  std::vector<ConvW> noise_convs;
  size_t cap_src = 0;
  float *sw_tmp = nullptr, *har = nullptr;
  // split-fp16 pairs: weights are packed as hi + lo fp16 of 2^8 w at create; w_range_bad (device) counts weights beyond that range, and
  // h2_ok says whether the packed form may be used (else the fp32-MFMA pairs: never a clipped weight)
  unsigned* w_range_bad = nullptr;
  bool h2_ok = true;
  unsigned short* planes = nullptr;       // activation planes of the gemm_h2w products (hi, then lo)
  size_t planes_cap = 0;                  // halfs
};

extern "C" void bsg_hifigan_destroy(bsg_hifigan* h) {
  if (!h) return;
  guard_free(&h->guard);
  for (float* p : h->owned) (void)hipFree(p);
  for (float* p : h->buf)
    if (p) (void)hipFree(p);
  if (h->sw_tmp) (void)hipFree(h->sw_tmp);
  if (h->har) (void)hipFree(h->har);
  if (h->planes) (void)hipFree(h->planes);
  h2w_free(&h->pre.h2w);
  for (ConvW& c : h->ups) h2w_free(&c.h2w);
  for (float* t : h->create_tmp) (void)hipFree(t);
  delete h;
}

static int hg_alloc(bsg_hifigan* h, float** p, size_t n) {
  BSG_HIP(hipMalloc((void**)p, n * sizeof(float)));
  h->owned.push_back(*p);
  return BSG_OK;
}

// consumes (bias, weight) or (bias, weight_g, weight_v) from the state_dict-ordered pointer list
static int take_conv(bsg_hifigan* h, ConvW& c, const void* const*& w, int dim0, int inner, hipStream_t st) {
  const size_t n = (size_t)dim0 * inner;
  TRY(hg_alloc(h, &c.w, n));
  const void* bias = *w++;
  if (h->cfg.weight_norm) {
    const float* g = (const float*)*w++;
    const float* v = (const float*)*w++;
    hipLaunchKernelGGL(weight_norm_fold_kernel, dim3(cdiv(dim0, 4)), dim3(256), 0, st, g, v, c.w, dim0, inner);
    BSG_LAUNCH_CHECK();
  } else {
    BSG_HIP(hipMemcpyAsync(c.w, *w++, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  TRY(hg_alloc(h, &c.b, c.cout));
  BSG_HIP(hipMemcpyAsync(c.b, bias, c.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
  return BSG_OK;
}

static int pack_conv(bsg_hifigan* h, ConvW& c, hipStream_t st, bool pair = false) {
  const int CO = c.cout >= 16 ? 16 : 8;
  const int n = cdiv(c.cout, CO) * c.cin * c.k * CO;
  TRY(hg_alloc(h, &c.wpk, n));
  hipLaunchKernelGGL(pack_conv_w_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)c.w, c.wpk, c.cout, c.cin, c.k, CO);
  BSG_LAUNCH_CHECK();
  if (pair && c.cin == c.cout && pair_supported(c.k, c.cout)) {   // one block of all output channels: [cin][k][cout]
    const int m = c.cin * c.k * c.cout;
    TRY(hg_alloc(h, &c.wpc, m));
    hipLaunchKernelGGL(pack_conv_w_kernel, dim3(cdiv(m, 256)), dim3(256), 0, st, (const float*)c.w, c.wpc, c.cout, c.cin, c.k, c.cout);
    BSG_LAUNCH_CHECK();
    if (pair_h16_supported(c.k, c.cout)) {
      const int tpk = 32 / c.cout, ks = (c.k + tpk - 1) / tpk;
      TRY(hg_alloc(h, &c.wp16, (size_t)ks * 512));   // ks x 2 planes x 64 lanes x 8 halves = ks x 512 floats
      hipLaunchKernelGGL(pack_conv_h16_kernel, dim3(cdiv(ks * 512, 256)), dim3(256), 0, st, (const float*)c.w, reinterpret_cast<_Float16*>(c.wp16), c.cout,
                         c.k, h->w_range_bad);
      BSG_LAUNCH_CHECK();
      c.wp16c = c.wp16;
    } else if ((c.cout == 8 || c.cout == 16) && (c.k == 3 || c.k == 7 || c.k == 11)) {   // (8 channels, K = 3 / 7: pairs on the vector pipe, chain on the matrix pipe)
      const int tpk = 32 / c.cout, ks = (c.k + tpk - 1) / tpk;
      TRY(hg_alloc(h, &c.wp16c, (size_t)ks * 512));
      hipLaunchKernelGGL(pack_conv_h16_kernel, dim3(cdiv(ks * 512, 256)), dim3(256), 0, st, (const float*)c.w, reinterpret_cast<_Float16*>(c.wp16c), c.cout,
                         c.k, h->w_range_bad);
      BSG_LAUNCH_CHECK();
    }
    if (pair_mfma_supported(c.k, c.cout)) {
      TRY(hg_alloc(h, &c.wpm, m));
      hipLaunchKernelGGL(pack_conv_mfma_kernel, dim3(cdiv(m, 256)), dim3(256), 0, st, (const float*)c.w, c.wpm, c.cout, c.k);
      BSG_LAUNCH_CHECK();
      TRY(hg_alloc(h, &c.wps, m));
      hipLaunchKernelGGL(pack_conv_h2_kernel, dim3(cdiv(m, 256)), dim3(256), 0, st, (const float*)c.w, reinterpret_cast<_Float16*>(c.wps), c.cout, c.k,
                         h->w_range_bad);
      BSG_LAUNCH_CHECK();
    }
  }
  return BSG_OK;
}

// ConvTranspose1d with K == 2*u: per-phase 2-tap weights for the polyphase launch
// tmp[tap][n = co u + r][ci] = w[ci][co][r + u (1 - tap)] (ConvTranspose1d weight [Cin][Cout][2u]): tap 0 meets x[q - 1], tap 1 x[q]
__global__ void up_h2w_reorder_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int u) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int Wn = Cout * u;
  if (i >= 2 * Wn * Cin) return;
  const int ci = i % Cin, n = (i / Cin) % Wn, tap = i / (Cin * Wn);
  const int co = n / u, r = n - co * u;
  out[i] = w[((long long)ci * Cout + co) * (2 * u) + r + u * (1 - tap)];
}
// tmp[tap][co][ci < CinP] = w[co][ci][tap] (Conv1d weight [Cout][Cin][K]), zero for ci >= Cin
__global__ void conv_h2w_reorder_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int CinP, int Cout, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * Cout * CinP) return;
  const int ci = i % CinP, co = (i / CinP) % Cout, tap = i / (CinP * Cout);
  out[i] = ci < Cin ? w[((long long)co * Cin + ci) * K + tap] : 0.f;
}
static int pack_up_h2w(bsg_hifigan* h, ConvW& c, int u, hipStream_t st) {
  if (u != 8 || c.k != 16 || c.cin % 64 || (c.cout * u) % 128) return BSG_OK;
  float* tmp = nullptr;
  const int n = 2 * c.cout * u * c.cin;
  BSG_HIP(hipMalloc((void**)&tmp, (size_t)n * sizeof(float)));
  h->create_tmp.push_back(tmp);   // (2-8 MB each; they used to stay until the handle was destroyed)
  hipLaunchKernelGGL(up_h2w_reorder_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)c.w, tmp, c.cin, c.cout, u);
  BSG_LAUNCH_CHECK();
  TRY(h2w_pack(&c.h2w, tmp, c.cout * u, c.cin, 2, (long long)c.cout * u * c.cin, c.cin, 1, h->w_range_bad, st));
  c.h2w.ok = true;   // (subject to bsg_hifigan::h2_ok: the range counter is read once, at the end of create)
  c.h2w_kp = c.cin;
  return BSG_OK;
}
static int pack_pre_h2w(bsg_hifigan* h, ConvW& c, hipStream_t st) {
  const int kp = (c.cin + 63) / 64 * 64;
  if (c.cout % 128 || c.k > 17) return BSG_OK;
  float* tmp = nullptr;
  const int n = c.k * c.cout * kp;
  BSG_HIP(hipMalloc((void**)&tmp, (size_t)n * sizeof(float)));
  h->create_tmp.push_back(tmp);
  hipLaunchKernelGGL(conv_h2w_reorder_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)c.w, tmp, c.cin, kp, c.cout, c.k);
  BSG_LAUNCH_CHECK();
  TRY(h2w_pack(&c.h2w, tmp, c.cout, kp, c.k, (long long)c.cout * kp, kp, 1, h->w_range_bad, st));
  c.h2w.ok = true;
  c.h2w_kp = kp;
  return BSG_OK;
}
static int ensure_planes(bsg_hifigan* h, size_t halfs, hipStream_t st) {
  if (halfs > h->planes_cap) {
    BSG_HIP(hipStreamSynchronize(st));
    if (h->planes) (void)hipFree(h->planes);
    h->planes = nullptr; h->planes_cap = 0;
    BSG_HIP(hipMalloc((void**)&h->planes, halfs * sizeof(unsigned short)));
    h->planes_cap = halfs;
  }
  return BSG_OK;
}

static int pack_convT(bsg_hifigan* h, ConvW& c, int u, hipStream_t st) {
  if (c.k != 2 * u) return BSG_OK;   // other shapes keep the gather kernel
  const int CO = c.cout >= 16 ? 16 : 8;
  const int n = u * cdiv(c.cout, CO) * c.cin * 2 * CO;
  TRY(hg_alloc(h, &c.wpk, n));
  hipLaunchKernelGGL(pack_convT_w_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const float*)c.w, c.wpk, c.cin, c.cout, u, CO);
  BSG_LAUNCH_CHECK();
  if (u == 8 || u == 2 || u == 4) {
    const int n8 = u * cdiv(c.cout, 8) * c.cin * 2 * 8;
    TRY(hg_alloc(h, &c.wpu, n8));
    hipLaunchKernelGGL(pack_convT_w_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, st, (const float*)c.w, c.wpu, c.cin, c.cout, u, 8);
    BSG_LAUNCH_CHECK();
  }
  return BSG_OK;
}

extern "C" int bsg_hifigan_n_weights(const bsg_hifigan_cfg* c) {
  const int convs = 2 + c->n_ups + (c->resblock == 2 ? 1 : 2) * c->n_ups * c->n_kernels * c->n_dil;
  return convs * (c->weight_norm ? 3 : 2) + (c->use_nsf ? 2 + 2 * c->n_ups : 0);
}

extern "C" int bsg_weight_norm_fold(const float* g, const float* v, float* w, int32_t dim0, int32_t inner, void* stream) {
  BSG_REQUIRE(g && v && w && dim0 > 0 && inner > 0, "weight_norm_fold: bad argument");
  hipLaunchKernelGGL(weight_norm_fold_kernel, dim3(cdiv(dim0, 4)), dim3(256), 0, (hipStream_t)stream, g, v, w, dim0, inner);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" int bsg_hifigan_create(bsg_hifigan** out, const bsg_hifigan_cfg* cfg, const void* const* dev_weights,
                                  int32_t n_weights, void* stream) {
  BSG_REQUIRE(out && cfg && dev_weights, "hifigan_create: null argument");
  BSG_REQUIRE(cfg->n_ups > 0 && cfg->n_ups <= 8 && cfg->n_kernels > 0 && cfg->n_kernels <= 8 && cfg->n_dil > 0 && cfg->n_dil <= 4,
              "hifigan_create: bad stage counts");
  BSG_REQUIRE(cfg->n_mel > 0 && cfg->upsample_initial_channel >= (1 << cfg->n_ups), "hifigan_create: bad channel config");
  BSG_REQUIRE(cfg->resblock >= 0 && cfg->resblock <= 2, "hifigan_create: resblock=%d (1 or 2)", cfg->resblock);
  BSG_REQUIRE(!cfg->use_nsf || (cfg->harmonic_num >= 0 && cfg->harmonic_num <= 31 && cfg->sample_rate > 0), "hifigan_create: bad NSF config");
  BSG_REQUIRE(n_weights == bsg_hifigan_n_weights(cfg), "hifigan_create: expected %d weight tensors, got %d",
              bsg_hifigan_n_weights(cfg), n_weights);
  for (int i = 0; i < cfg->n_ups; ++i) {
    const int k = cfg->upsample_kernel_sizes[i], u = cfg->upsample_rates[i];
    BSG_REQUIRE(u > 0 && k >= u && (k - u) % 2 == 0, "hifigan_create: upsample stage %d (k=%d, u=%d) unsupported", i, k, u);
  }
  for (int j = 0; j < cfg->n_kernels; ++j) {
    const int k = cfg->resblock_kernel_sizes[j];
    BSG_REQUIRE(k == 3 || k == 7 || k == 11 || (k == 5 && cfg->resblock == 2), "hifigan_create: resblock kernel %d not built (3, 7, 11; 5 with ResBlock2)", k);
    for (int m = 0; m < cfg->n_dil; ++m)
      BSG_REQUIRE(cfg->resblock_dilations[j][m] >= 1 && cfg->resblock_dilations[j][m] <= 16, "hifigan_create: dilation out of range");
  }
  for (int i = 0; i < n_weights; ++i) BSG_REQUIRE(dev_weights[i] != nullptr, "hifigan_create: weight %d is null", i);
  hipStream_t st = (hipStream_t)stream;
  bsg_hifigan* h = new bsg_hifigan();
  h->cfg = *cfg;
  const void* const* w = dev_weights;
  int rc = BSG_OK;
  auto fail = [&](int code) { bsg_hifigan_destroy(h); return code; };
  if ((rc = guard_init(&h->guard, st)) != BSG_OK) return fail(rc);
  const int C0 = cfg->upsample_initial_channel;
  {
    float* cnt = nullptr;
    if ((rc = hg_alloc(h, &cnt, 1)) != BSG_OK) return fail(rc);
    h->w_range_bad = reinterpret_cast<unsigned*>(cnt);
    if (hipMemsetAsync(h->w_range_bad, 0, sizeof(unsigned), st) != hipSuccess) { set_error("hifigan_create: memset failed"); return fail(BSG_EHIP); }
  }
  if (cfg->use_nsf) {
    // m_source.l_linear.{weight [1,NH], bias [1]}, noise_convs.i.{weight [c,1,k], bias [c]} precede conv_pre (hifigan.py:111-132)
    const int NH = cfg->harmonic_num + 1;
    auto cp = [&](float** dst, const void* src, size_t n) -> int {
      int r = hg_alloc(h, dst, n);
      if (r != BSG_OK) return r;
      if (hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("hifigan_create: copy failed"); return BSG_EHIP; }
      return BSG_OK;
    };
    if ((rc = cp(&h->src_w, *w++, NH)) != BSG_OK) return fail(rc);
    if ((rc = cp(&h->src_b, *w++, 1)) != BSG_OK) return fail(rc);
    h->noise_convs.resize(cfg->n_ups);
    for (int i = 0; i < cfg->n_ups; ++i) {
      ConvW& c = h->noise_convs[i];
      int stride = 1;
      for (int j = i + 1; j < cfg->n_ups; ++j) stride *= cfg->upsample_rates[j];
      c.cout = C0 >> (i + 1); c.cin = 1; c.k = i + 1 < cfg->n_ups ? 2 * stride : 1;
      if ((rc = cp(&c.w, *w++, (size_t)c.cout * c.k)) != BSG_OK) return fail(rc);
      if ((rc = cp(&c.b, *w++, c.cout)) != BSG_OK) return fail(rc);
    }
  }
  h->pre.cout = C0; h->pre.cin = cfg->n_mel; h->pre.k = 7;
  if ((rc = take_conv(h, h->pre, w, C0, cfg->n_mel * 7, st)) != BSG_OK) return fail(rc);
  if ((rc = pack_conv(h, h->pre, st)) != BSG_OK) return fail(rc);
  if ((rc = pack_pre_h2w(h, h->pre, st)) != BSG_OK) return fail(rc);
  h->ups.resize(cfg->n_ups);
  for (int i = 0; i < cfg->n_ups; ++i) {
    ConvW& c = h->ups[i];
    c.cin = C0 >> i; c.cout = C0 >> (i + 1); c.k = cfg->upsample_kernel_sizes[i];
    if ((rc = take_conv(h, c, w, c.cin, c.cout * c.k, st)) != BSG_OK) return fail(rc);   // weight [Cin][Cout][K], g over dim 0 = Cin
    if ((rc = pack_convT(h, c, cfg->upsample_rates[i], st)) != BSG_OK) return fail(rc);
    if ((rc = pack_up_h2w(h, c, cfg->upsample_rates[i], st)) != BSG_OK) return fail(rc);
  }
  const int nrb = cfg->n_ups * cfg->n_kernels;
  h->rb1.resize((size_t)nrb * cfg->n_dil);
  if (cfg->resblock != 2) h->rb2.resize((size_t)nrb * cfg->n_dil);
  for (int r = 0; r < nrb; ++r) {
    const int ch = C0 >> (r / cfg->n_kernels + 1);
    const int k = cfg->resblock_kernel_sizes[r % cfg->n_kernels];
    for (int pass = 0; pass < (cfg->resblock == 2 ? 1 : 2); ++pass)
      for (int m = 0; m < cfg->n_dil; ++m) {
        ConvW& c = (pass == 0 ? h->rb1 : h->rb2)[(size_t)r * cfg->n_dil + m];
        c.cin = c.cout = ch; c.k = k;
        if ((rc = take_conv(h, c, w, ch, ch * k, st)) != BSG_OK) return fail(rc);
        if ((rc = pack_conv(h, c, st, cfg->resblock != 2)) != BSG_OK) return fail(rc);   // (the fused pair forms are ResBlock1's)
      }
  }
  h->post.cout = 1; h->post.cin = C0 >> cfg->n_ups; h->post.k = 7;
  if ((rc = take_conv(h, h->post, w, 1, h->post.cin * 7, st)) != BSG_OK) return fail(rc);
  if ((rc = pack_conv(h, h->post, st)) != BSG_OK) return fail(rc);
  if (hipStreamSynchronize(st) != hipSuccess) { set_error("hifigan_create: stream sync failed"); return fail(BSG_EHIP); }
  for (float* t : h->create_tmp) (void)hipFree(t);
  h->create_tmp.clear();
  {
    unsigned nbad = 0;
    if (hipMemcpy(&nbad, h->w_range_bad, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) { set_error("hifigan_create: copy failed"); return fail(BSG_EHIP); }
    h->h2_ok = nbad == 0;   // a ResBlock weight with |w| * 2^8 >= 60000 (or not finite): the pairs stay on the fp32 matrix pipe
  }
  *out = h;
  return BSG_OK;
}

static int run_conv(const ConvW& c, const float* x, float* y, int B, int L, int dil, float in_slope, const float* res,
                    const float* acc_in, float out_div, int out_tanh, hipStream_t st) {
  ConvArgs a{};
  a.x = x; a.w = c.wpk; a.bias = c.b; a.y = y; a.res = res; a.acc_in = acc_in; a.out_div = out_div; a.in_slope = in_slope;
  a.out_tanh = out_tanh; a.Cin = c.cin; a.Cout = c.cout; a.L = L; a.dil = dil; a.pad = (c.k * dil - dil) / 2;
  return launch_conv(a, c.k, B, st);
}

static int hifigan_run(bsg_hifigan* h, const float* mel, float* wav, int32_t B, int32_t T, void* stream, const float* har, long long Lh) {
  BSG_REQUIRE(h && mel && wav && B > 0 && T > 0, "hifigan_forward: bad argument");
  BSG_REQUIRE(B <= 65535, "hifigan_forward: B=%d > 65535", B);
  hipStream_t st = (hipStream_t)stream;
  const bsg_hifigan_cfg& c = h->cfg;
  // largest stage tensor: channels halve while length grows by u_i
  size_t need = (size_t)B * c.upsample_initial_channel * T;
  {
    long long L = T;
    for (int i = 0; i < c.n_ups; ++i) {
      L *= c.upsample_rates[i];
      const size_t n = (size_t)B * (c.upsample_initial_channel >> (i + 1)) * L;
      if (n > need) need = n;
    }
    BSG_REQUIRE(L < (1LL << 31) / 4, "hifigan_forward: output length %lld too large", L);
  }
  if (need > h->cap) {
    BSG_HIP(hipStreamSynchronize(st));
    for (float*& p : h->buf) { if (p) (void)hipFree(p); p = nullptr; }
    h->cap = 0;
    for (float*& p : h->buf) BSG_HIP(hipMalloc((void**)&p, need * sizeof(float)));
    h->cap = need;
  }
  float *x = h->buf[0], *ya = h->buf[1], *yb = h->buf[2], *tmp = h->buf[3], *xs = h->buf[4];
  const float slope = 0.1f;   // LRELU_SLOPE, hifigan.py:11
  static int h2w_env = -1;   // BSG_HG_H2W=0: conv_pre and the u = 8 transposed convolutions on the vector pipe (conv1d_kernel / upsample_kernel)
  if (h2w_env < 0) { const char* e = getenv("BSG_HG_H2W"); h2w_env = e ? atoi(e) : 1; }
  static int hg_split_env = -1;   // BSG_HG_SPLIT=0 (the fp32-MFMA / vector forms of the ResBlock pairs) takes these two off the 16-bit pipe as well: the switch
  if (hg_split_env < 0) { const char* e = getenv("BSG_HG_SPLIT"); hg_split_env = e ? atoi(e) : 1; }   // means the WHOLE vocoder on fp32 products
  const bool h2w_on = h2w_env && hg_split_env && h->h2_ok && gemm_split_enabled();
  if (h2w_on && h->pre.h2w.ok && h2w_supports(T, h->pre.cout, h->pre.h2w_kp, h->pre.k, h->pre.h2w_kp)) {
    // conv_pre :150 as a 7-tap product of the mel planes [T][n_mel padded to 128] with the pre-split weights; output [C0][T]
    const ConvW& p = h->pre;
    const long long n = (long long)B * T * p.h2w_kp;
    TRY(ensure_planes(h, (size_t)2 * n, st));
    TRY(h2w_split_transposed_lrelu(mel, h->planes, h->planes + n, B, p.cin, p.h2w_kp, T, T, 1.0f, st));
    H2wArgs g{};
    g.act = h->planes; g.act_plane = n; g.lda = p.h2w_kp; g.sAct = (long long)T * p.h2w_kp; g.wpack = p.h2w.pack; g.rows = T; g.K = p.h2w_kp;
    g.Wn = p.cout; g.taps = p.k; g.tap_shift0 = -(p.k / 2); g.act_is_a = 0; g.C = x; g.ldc = T; g.sC = (long long)p.cout * T; g.bias = p.b;
    g.alpha = 1.f; g.act_fn = ACT_NONE; g.batch = B;
    TRY(launch_gemm_h2w(g, st));
  } else {
    TRY(run_conv(h->pre, mel, x, B, T, 1, 1.0f, nullptr, nullptr, 1.0f, 0, st));            // conv_pre :150
  }
  int L = T;
  float* cur = x;
  for (int i = 0; i < c.n_ups; ++i) {
    const ConvW& up = h->ups[i];
    ConvTArgs t{};
    t.x = cur; t.w = up.w; t.bias = up.b; t.y = (cur == x ? xs : x); t.in_slope = slope; t.Cin = up.cin; t.Cout = up.cout;
    t.Lin = L; t.K = up.k; t.u = c.upsample_rates[i]; t.p = (up.k - c.upsample_rates[i]) / 2;
    const int Lout = L * t.u;
    static int up_env = -1;
    if (up_env < 0) { const char* e = getenv("BSG_HG_UP"); up_env = e ? atoi(e) : 1; }
    if (h2w_on && up.h2w.ok && t.u == 8 && t.p == 4 && h2w_supports(L + 1, up.cout * 8, up.cin, 2, up.cin)) {
      // the transposed convolution on the matrix pipe: lrelu(x) as planes [L + 1][Cin] (row L zero), ONE 2-tap product with the (co, phase) weight
      // rows, stored phase-interleaved (H2wArgs::up_u)
      const long long n = (long long)B * (L + 1) * up.cin;
      TRY(ensure_planes(h, (size_t)2 * n, st));
      TRY(h2w_split_transposed_lrelu(cur, h->planes, h->planes + n, B, up.cin, up.cin, L, L + 1, slope, st));
      H2wArgs g{};
      g.act = h->planes; g.act_plane = n; g.lda = up.cin; g.sAct = (long long)(L + 1) * up.cin; g.wpack = up.h2w.pack; g.rows = L + 1; g.K = up.cin;
      g.Wn = up.cout * 8; g.taps = 2; g.tap_shift0 = -1; g.act_is_a = 0; g.C = t.y; g.ldc = Lout; g.sC = (long long)up.cout * Lout; g.bias = up.b;
      g.alpha = 1.f; g.act_fn = ACT_NONE; g.batch = B; g.up_u = 8; g.up_p = t.p; g.up_lout = Lout;
      TRY(launch_gemm_h2w(g, st));
    } else if (up_env && t.u == 2 && up.k == 4 && t.p == 1 && (up.cout == 16 || up.cout == 8) && (long long)cdiv(Lout, 1024) * B >= 512 &&
               !getenv("BSG_NO_UP2")) {   // (single utterances: 125 workgroups of this form are slower than the phase-per-block form, 38 against 28 us)
      UpArgs ua{};
      ua.x = cur; ua.w = up.w; ua.bias = up.b; ua.y = t.y; ua.slope = slope; ua.Cin = up.cin; ua.Cout = up.cout; ua.Lin = L; ua.p = t.p;
      const dim3 grid(cdiv(Lout, 1024), 1, B);
      if (up.cout == 16) hipLaunchKernelGGL(upsample2_kernel<16>, grid, dim3(256), 0, st, ua);
      else hipLaunchKernelGGL(upsample2_kernel<8>, grid, dim3(256), 0, st, ua);
      BSG_LAUNCH_CHECK();
    } else if (up.wpu && up_env && t.p * 2 == t.u && (long long)cdiv(L + 1, 256) * cdiv(up.cout, 8) * B >= 512) {   // short inputs: the phase-per-block form has u x the workgroups
      // all phases of a position in one lane: contiguous stores (upsample_kernel)
      UpArgs ua{};
      ua.x = cur; ua.w = up.wpu; ua.bias = up.b; ua.y = t.y; ua.slope = slope; ua.Cin = up.cin; ua.Cout = up.cout; ua.Lin = L; ua.p = t.p;
      const dim3 grid(cdiv(L + 1, 256), cdiv(up.cout, 8), B);
      if (t.u == 8) hipLaunchKernelGGL(upsample_kernel<8>, grid, dim3(256), 0, st, ua);
      else if (t.u == 4) hipLaunchKernelGGL(upsample_kernel<4>, grid, dim3(256), 0, st, ua);
      else hipLaunchKernelGGL(upsample_kernel<2>, grid, dim3(256), 0, st, ua);
      BSG_LAUNCH_CHECK();
    } else if (up.wpk && !getenv("BSG_NO_POLYPHASE")) {
      // polyphase form: u interleaved 2-tap convolutions over the input positions (weights wave-uniform -> scalar loads)
      ConvArgs a{};
      a.x = cur; a.w = up.wpk; a.bias = up.b; a.y = t.y; a.out_div = 1.0f; a.in_slope = slope;
      a.Cin = up.cin; a.Cout = up.cout; a.L = L; a.dil = 1; a.pad = 1;
      a.n_phase = t.u; a.ph_off = -t.p; a.Lq = L + 1; a.Lout = Lout;
      a.w_phase_stride = (long long)cdiv(up.cout, up.cout >= 16 ? 16 : 8) * up.cin * 2 * (up.cout >= 16 ? 16 : 8);
      TRY(launch_convT(a, B, st));
    } else {
      if (up.cout >= 16) hipLaunchKernelGGL(conv_transpose1d_kernel<16>, dim3(cdiv(Lout, 256), cdiv(up.cout, 16), B), dim3(256), 0, st, t);
      else hipLaunchKernelGGL(conv_transpose1d_kernel<8>, dim3(cdiv(Lout, 256), cdiv(up.cout, 8), B), dim3(256), 0, st, t);
      BSG_LAUNCH_CHECK();
    }
    if (har) {   // x = x + LayerNorm_C(relu(noise_conv_i(har_source)))                       (hifigan.py:154-160)
      const ConvW& nc = h->noise_convs[i];
      const int stride = i + 1 < c.n_ups ? nc.k / 2 : 1;
      TRY(nsf_launch_source_add(t.y, har, nc.w, nc.b, B, nc.cout, Lout, Lh, nc.k, stride, i + 1 < c.n_ups ? stride / 2 : 0, st));
    }
    float* xin = t.y;                       // stage input (after upsampling)
    float* sum = (xin == x ? xs : x);       // MRF running sum / stage output
    L = Lout;
    for (int j = 0; j < c.n_kernels; ++j) {
      const int r = i * c.n_kernels + j;
      const float* y = xin;
      {
        // a whole ResBlock1 of 8 / 16 channels in one launch (resblock_chain_h16_kernel; BSG_HG_CHAIN=0: pair by pair)
        static int chain_env = -1, split_env = -1, mfma_env2 = -1, h16_env2 = -1;
        if (chain_env < 0) { const char* e = getenv("BSG_HG_CHAIN"); chain_env = e ? atoi(e) : 1; }
        if (split_env < 0) { const char* e = getenv("BSG_HG_SPLIT"); split_env = e ? atoi(e) : 1; }
        if (mfma_env2 < 0) { const char* e = getenv("BSG_HG_MFMA"); mfma_env2 = e ? atoi(e) : 1; }
        if (h16_env2 < 0) { const char* e = getenv("BSG_HG_H16"); h16_env2 = e ? atoi(e) : 1; }
        const ConvW& f1 = h->rb1[(size_t)r * c.n_dil];
        bool chain = chain_env && split_env && mfma_env2 && h16_env2 && c.resblock != 2 && h->h2_ok && gemm_split_enabled() && c.n_dil <= 3 &&
                     chain_h16_supported(f1.k, f1.cout, c.n_dil, c.resblock_dilations[j]);
        for (int m = 0; chain && m < c.n_dil; ++m)
          chain = h->rb1[(size_t)r * c.n_dil + m].wp16c && h->rb2[(size_t)r * c.n_dil + m].wp16c && h->rb1[(size_t)r * c.n_dil + m].k == f1.k &&
                  h->rb2[(size_t)r * c.n_dil + m].k == f1.k;
        if (chain) {
          ChainArgs ca{};
          ca.x = xin; ca.y = sum; ca.acc_in = j > 0 ? sum : nullptr; ca.out_div = j == c.n_kernels - 1 ? (float)c.n_kernels : 1.0f;
          ca.slope = slope; ca.L = L; ca.n_pairs = c.n_dil; ca.range_events = gemm_range_counter();
          for (int m = 0; m < c.n_dil; ++m) {
            const ConvW& c1 = h->rb1[(size_t)r * c.n_dil + m];
            const ConvW& c2 = h->rb2[(size_t)r * c.n_dil + m];
            ca.w1[m] = c1.wp16c; ca.b1[m] = c1.b; ca.w2[m] = c2.wp16c; ca.b2[m] = c2.b; ca.dil[m] = c.resblock_dilations[j][m];
          }
          TRY(launch_chain_h16(ca, f1.k, f1.cout, B, st));
          continue;
        }
      }
      for (int m = 0; m < c.n_dil; ++m) {
        const bool last = m == c.n_dil - 1;
        const ConvW& c1 = h->rb1[(size_t)r * c.n_dil + m];
        float* dst = last ? sum : (y == ya ? yb : ya);
        if (c.resblock == 2) {
          // ResBlock2 (hifigan.py:70-91): x = conv_d(lrelu(x)) + x per dilation; the MRF sum / mean in the last one's epilogue (:161-167)
          TRY(run_conv(c1, y, dst, B, L, c.resblock_dilations[j][m], slope, y, (last && j > 0) ? sum : nullptr,
                       (last && j == c.n_kernels - 1) ? (float)c.n_kernels : 1.0f, 0, st));
          y = dst;
          continue;
        }
        const ConvW& c2 = h->rb2[(size_t)r * c.n_dil + m];
        static int fused_env = -1;
        if (fused_env < 0) { const char* e = getenv("BSG_HG_FUSED"); fused_env = e ? atoi(e) : 1; }
        // fused pair (one HBM round trip instead of two and a half) unless the launch would leave most CUs without a workgroup
        const long long pair_wgs = (long long)cdiv(L, 256 - (c1.k - 1)) * B;
        static int mfma_env = -1;
        if (mfma_env < 0) { const char* e = getenv("BSG_HG_MFMA"); mfma_env = e ? atoi(e) : 1; }
        const bool use_mfma = mfma_env && c1.wpm && c2.wpm;
        if (fused_env && c1.wpc && c2.wpc && (use_mfma || pair_wgs >= 128 || fused_env == 2)) {
          PairArgs pa{};
          pa.x = y; pa.w1 = c1.wpc; pa.b1 = c1.b; pa.w2 = c2.wpc; pa.b2 = c2.b; pa.y = dst;
          pa.acc_in = (last && j > 0) ? sum : nullptr;
          pa.out_div = (last && j == c.n_kernels - 1) ? (float)c.n_kernels : 1.0f;
          pa.slope = slope; pa.L = L; pa.dil = c.resblock_dilations[j][m];
          static int h2_env = -1;   // BSG_HG_SPLIT=0: the fp32-MFMA form even while the GEMMs run split-fp16
          if (h2_env < 0) { const char* e = getenv("BSG_HG_SPLIT"); h2_env = e ? atoi(e) : 1; }
          static int h16_env = -1;  // BSG_HG_H16=0: the VALU pairs for 8 / 16 channels
          if (h16_env < 0) { const char* e = getenv("BSG_HG_H16"); h16_env = e ? atoi(e) : 1; }
          if (mfma_env && h16_env && h2_env && h->h2_ok && c1.wp16 && c2.wp16 && gemm_split_enabled()) {
            pa.w1 = c1.wp16; pa.w2 = c2.wp16; pa.range_events = gemm_range_counter();
            TRY(launch_pair_h16(pa, c1.k, c1.cout, B, st));
          } else if (use_mfma && h2_env && h->h2_ok && c1.wps && c2.wps && gemm_split_enabled()) {
            pa.w1 = c1.wps; pa.w2 = c2.wps; pa.range_events = gemm_range_counter();
            TRY(launch_pair_h2(pa, c1.k, c1.cout, B, st));
          } else if (use_mfma) {
            pa.w1 = c1.wpm; pa.w2 = c2.wpm;
            TRY(launch_pair_mfma(pa, c1.k, c1.cout, B, st));
          } else {
            TRY(launch_pair(pa, c1.k, c1.cout, B, st));
          }
          y = dst;
          continue;
        }
        TRY(run_conv(c1, y, tmp, B, L, c.resblock_dilations[j][m], slope, nullptr, nullptr, 1.0f, 0, st));     // :56-57
        TRY(run_conv(c2, tmp, dst, B, L, 1, slope, y, (last && j > 0) ? sum : nullptr,
                     (last && j == c.n_kernels - 1) ? (float)c.n_kernels : 1.0f, 0, st));                     // :58-60, :161-167
        y = dst;
      }
    }
    cur = sum;
  }
  if (h->post.cin == 8 && h->post.k == 7 && !getenv("BSG_NO_CONV_POST")) {
    hipLaunchKernelGGL(conv_post_kernel<8>, dim3(cdiv(L, 1024), B), dim3(256), 0, st, (const float*)cur, (const float*)h->post.w, (const float*)h->post.b, wav, L, 0.01f);
    BSG_LAUNCH_CHECK();
  } else {
    TRY(run_conv(h->post, cur, wav, B, L, 1, 0.01f, nullptr, nullptr, 1.0f, 1, st));          // :169-171 (default slope 0.01)
  }
  return BSG_OK;
}

extern "C" int bsg_hifigan_forward(bsg_hifigan* h, const float* mel, float* wav, int32_t B, int32_t T, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  BSG_REQUIRE(h && !h->cfg.use_nsf, "hifigan_forward: this generator was created with the NSF source; call bsg_hifigan_forward_nsf");
  return hifigan_run(h, mel, wav, B, T, stream, nullptr, 0);
}

extern "C" int bsg_hifigan_forward_nsf(bsg_hifigan* h, const float* mel, const float* f0, const float* rand_ini, const float* noise,
                                       float* wav, int32_t B, int32_t T, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  BSG_REQUIRE(h && h->cfg.use_nsf, "hifigan_forward_nsf: generator created without the NSF source");
  BSG_REQUIRE(mel && f0 && rand_ini && noise && wav && B > 0 && T > 0, "hifigan_forward_nsf: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int hop = 1;
  for (int i = 0; i < h->cfg.n_ups; ++i) hop *= h->cfg.upsample_rates[i];
  const long long L = (long long)T * hop;
  const int NH = h->cfg.harmonic_num + 1;
  const size_t need = (size_t)B * L;
  if (need > h->cap_src) {
    BSG_HIP(hipStreamSynchronize(st));
    if (h->sw_tmp) (void)hipFree(h->sw_tmp);
    if (h->har) (void)hipFree(h->har);
    h->sw_tmp = h->har = nullptr;
    h->cap_src = 0;
    BSG_HIP(hipMalloc((void**)&h->sw_tmp, need * NH * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->har, need * sizeof(float)));
    h->cap_src = need;
  }
  TRY(nsf_launch_source(f0, rand_ini, noise, h->src_w, h->src_b, h->sw_tmp, h->har, B, T, hop, NH, (float)h->cfg.sample_rate, st));
  return hifigan_run(h, mel, wav, B, T, stream, h->har, L);
}

namespace bsg {
Guard* guard_of_hifigan(void* h) { return &static_cast<bsg_hifigan*>(h)->guard; }
}
