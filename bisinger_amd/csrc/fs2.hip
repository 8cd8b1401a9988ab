// FastSpeech2-MIDI encoder / decoder (the condition generator of the diffusion decoder) on gfx950.
//
// Reference semantics (paths relative to /root/reference/train_bisinger):
//   modules/diffsinger_midi/fs2.py:14-65 (FastspeechMIDIEncoder), :94-197 (FastSpeech2MIDI.forward)
//   modules/fastspeech/tts_modules.py:39-58 (LayerNorm eps 1e-12), :61-153 (DurationPredictor),
//     :156-191 (LengthRegulator), :253-309 (FFTBlocks)
//   modules/commons/common_layers.py:106-179 (SinusoidalPositionalEmbedding), :282-346 (MultiheadAttention ->
//     F.multi_head_attention_forward), :598-644 (TransformerFFNLayer), :664-730 (EncSALayer), :832-860 (ESM)
//   modules/commons/espnet_positional_embedding.py:90-114 (RelPositionalEncoding)
//
// Activations live as [rows = b*T + t][C] (C contiguous), so every Linear / attention product / Conv1d-as-
// K-segmented-GEMM goes through the fp32 MFMA GEMM of gemm.hip with a fused epilogue (bias, k^-1/2 scale,
// GELU, residual add, non-padding mask).  The row-wise pieces (LayerNorm, masked softmax) are one-wave-
// per-row kernels with wavefront shuffles; gathers are coalesced 1-KB row copies.  FS2 is ~1.3 % of the
// path's FLOPs (27 MFLOP per frame vs 2.1 GFLOP for 100 diffusion steps), so it is built for exactness
// and simplicity; the unfused attention materialises the [B*H,T,T] score matrix in HBM.
#include <math.h>

#include <type_traits>
#include <vector>

#include "bsg_common.h"
#include "diffnet_res.h"

namespace bsg {
namespace {

constexpr int H = 256;   // hidden size (checked at create)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- LayerNorm over C = 256: one wave per row, 4 contiguous floats per lane ---------------------
// y = (x - mean) * rsqrt(var + eps) * w + b, then * rowscale[row] (the reference's "* nonpadding").
using ln_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
// yh / yl (optional, instead of y): the result as hi / lo fp16 planes of 16 x value — the activation operand of gemm_h2w_kernel
__global__ void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                 float* __restrict__ y, const float* __restrict__ rowscale, long long rows, float eps,
                                 _Float16* __restrict__ yh, _Float16* __restrict__ yl, unsigned* __restrict__ range_events) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const f32x4 v = reinterpret_cast<const f32x4*>(x + row * H)[lane];
  const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / H);
  const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
  const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / H);
  const float rstd = 1.0f / sqrtf(var + eps);
  const f32x4 wv = reinterpret_cast<const f32x4*>(w)[lane], bv = reinterpret_cast<const f32x4*>(b)[lane];
  const float rs = rowscale ? rowscale[row] : 1.0f;
  f32x4 o = {(d0 * rstd * wv[0] + bv[0]) * rs, (d1 * rstd * wv[1] + bv[1]) * rs, (d2 * rstd * wv[2] + bv[2]) * rs,
             (d3 * rstd * wv[3] + bv[3]) * rs};
  if (y) reinterpret_cast<f32x4*>(y + row * H)[lane] = o;
  if (yh) {
    ln_f16x4 hv, lv;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = o[e] * 16.0f;
      bad |= !(fabsf(s) < 65000.0f);
      hv[e] = (_Float16)s;
      lv[e] = (_Float16)(s - (float)hv[e]);
    }
    reinterpret_cast<ln_f16x4*>(yh + row * H)[lane] = hv;
    reinterpret_cast<ln_f16x4*>(yl + row * H)[lane] = lv;
    if (range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(range_events, 1u);
  }
}

// ---- masked softmax over keys: one 256-thread workgroup per (batch*head, query) row --------------
__global__ void masked_softmax_kernel(float* __restrict__ S, const float* __restrict__ keep, int Tq, int Tk, int heads) {
  __shared__ float red[4];
  const long long row = blockIdx.x;                 // (b*heads + h)*Tq + q
  const int b = (int)(row / ((long long)heads * Tq));
  float* __restrict__ s = S + row * Tk;
  const float* __restrict__ kp = keep + (long long)b * Tk;   // 1 = real key, 0 = padded key (-> -inf)
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int k = tid; k < Tk; k += 256) {
    const float v = kp[k] != 0.f ? s[k] : -INFINITY;
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int k = tid; k < Tk; k += 256) {
    const float e = kp[k] != 0.f ? expf(s[k] - m) : 0.f;
    s[k] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  for (int k = tid; k < Tk; k += 256) s[k] = s[k] / sum;
}

// ---- fused self-attention (EncSALayer's MultiheadAttention, common_layers.py:282-370, head dim 128) -------------------
// softmax(Q K^T + key-padding mask) V without materialising the [T,T] scores, fp32 MFMA throughout.  One wave = 32 queries;
// a workgroup of NW waves shares the K / V tiles of a 32-key block in LDS.  Per block:
//   S^T[key, q] = K[32 x 128] Q^T           A = K rows from LDS (stride 129: conflict-free), B = Q from registers (64 per lane)
//   online softmax over keys: a lane owns ONE query (its accumulator column), its keys sit in its 16 registers and in the
//   partner lane (lane ^ 32): max / sum are 16 in-register steps + one cross-half shuffle; the running output is rescaled
//   in registers (exp(m_old - m_new) is per lane)
//   O^T[d, q] += V^T[d, key] P[key, q]      the S accumulator IS the B operand: MFMA step j takes key row (j&3)+8(j>>2) (+4 for
//   the upper lane half) = register j of the lane, the A operand V[key][d] is read from LDS with that same key order.
// q is pre-scaled by head_dim^-1/2 (fused into the QKV projection); masked keys (keep == 0, or beyond T) get -inf as in the
// reference's masked_fill; rows are [b*T + t][3H] with Q | K | V column blocks.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ keep,
                                                                float* __restrict__ out, int T, int heads, int ld, int ldo) {
  constexpr int D = 128, BK = 32, LDK = D + 1;
  __shared__ float Ks[BK * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[BK * D];
  __shared__ float kp[BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y / heads, hh = blockIdx.y - b * heads;
  const int q_row = (blockIdx.x * NW + wave) * 32 + l31;
  const bool q_ok = q_row < T;
  const float* __restrict__ base = qkv + (long long)b * T * ld + hh * D;
  float qreg[64];
  {
    const float* __restrict__ qp = base + (long long)(q_ok ? q_row : T - 1) * ld;
#pragma unroll
    for (int s = 0; s < 64; ++s) qreg[s] = qp[2 * s + lh];
  }
  f32x16 O[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int H3 = ld / 3;
  for (int k0 = 0; k0 < T; k0 += BK) {
    __syncthreads();   // the previous block's tiles are consumed
#pragma unroll
    for (int j = 0; j < 1024 / (64 * NW); ++j) {
      const int idx = tid + 64 * NW * j;
      const int key = idx >> 5, c4 = (idx & 31) << 2;
      const int kt = k0 + key;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (kt < T) {
        const float* __restrict__ rp = base + (long long)kt * ld + c4;
        kv = *reinterpret_cast<const f32x4*>(rp + H3);
        vv = *reinterpret_cast<const f32x4*>(rp + 2 * H3);
      }
      Ks[key * LDK + c4] = kv[0]; Ks[key * LDK + c4 + 1] = kv[1]; Ks[key * LDK + c4 + 2] = kv[2]; Ks[key * LDK + c4 + 3] = kv[3];
      *reinterpret_cast<f32x4*>(&Vs[key * D + c4]) = vv;
    }
    if (tid < BK) kp[tid] = (k0 + tid < T) ? keep[(long long)b * T + k0 + tid] : 0.f;
    __syncthreads();
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    {
      const float* __restrict__ kr = Ks + l31 * LDK + lh;
#pragma unroll
      for (int s = 0; s < 64; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * s], qreg[s], S, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (kp[acc_row(r, lh)] == 0.f) S[r] = -INFINITY;
      mx = fmaxf(mx, S[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float ms = m_new == -INFINITY ? 0.f : m_new;     // a block of masked keys only must not produce inf - inf
    const float scale = expf(m - ms);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = expf(S[r] - ms);
      ps += S[r];
    }
    ps += __shfl_xor(ps, 32);
    l = l * scale + ps;
    m = m_new;
    const bool rescale = __any(scale != 1.0f);   // wave-uniform: once the running maxima have settled nothing is rescaled
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      if (rescale) {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] *= scale;
      }
      const float* __restrict__ vr = Vs + 4 * lh * D + 32 * dt + l31;
#pragma unroll
      for (int j = 0; j < 16; ++j) O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[((j & 3) + 8 * (j >> 2)) * D], S[j], O[dt], 0, 0, 0);
    }
  }
  if (q_ok) {
    const float inv = 1.0f / l;
    float* __restrict__ op = out + ((long long)b * T + q_row) * ldo + hh * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) op[32 * dt + acc_row(r, lh)] = O[dt][r] * inv;
  }
}

// ---- the same fused attention with every product on the 16-bit matrix pipe (split-fp16, see gemm.hip / diffnet_h2.hip) --------------
// Q, K, V are scaled by 2^4 and split exactly into hi + lo fp16 terms (Q once, into registers; K and V while their 32-key block is staged:
// K as [key][128 d] planes, V TRANSPOSED as [d][32 keys] planes), the probabilities P = exp(S - m) in [0, 1] by 2^10 and split in
// registers; every fp32 product is hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: 48 MFMAs of 32 matrix cycles
// per key block and wave instead of 128 of 64.  The softmax runs in fp32 on the un-scaled scores exactly as above.
//   S^T[key, q]: A = K fragment (lane = key, 8 consecutive d per lane half), B = Q fragment of the lane's query.
//   O^T[d, q] += V^T[d, key] P[key, q]: the lane's 16 probabilities are keys (r&3) + 8 (r>>2) + 4 lh; MFMA step t takes its registers
//   8t .. 8t+7 as the 8 k-values of the lane half, so the A fragment of V^T is read with that key order: two 8-byte pieces of a [d] row.
// Operands beyond the fp16 range (|v| >= 4062) are counted in the GEMMs' range-event counter (the host repeats the call on the
// fp32 pipe).
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x2_t = __attribute__((ext_vector_type(2))) unsigned;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_attn_split_kernel(const float* __restrict__ qkv, const float* __restrict__ keep,
                                                                      float* __restrict__ out, int T, int heads, int ld, int ldo,
                                                                      unsigned* __restrict__ range_events, _Float16* __restrict__ out_h,
                                                                      long long out_plane) {
  constexpr int D = 128, BK = 32, KROW = 2 * D + 16, VROW = 2 * BK + 16;   // bytes per LDS row: 272 (68 dwords = 4 mod 64), 80
  constexpr float SIN = 16.0f, PSC = 1024.0f;
  __shared__ __attribute__((aligned(16))) char Kp[2 * BK * KROW];    // hi plane, lo plane
  __shared__ __attribute__((aligned(16))) char Vp[2 * D * VROW];
  __shared__ float kp[BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y / heads, hh = blockIdx.y - b * heads;
  const int q_row = (blockIdx.x * NW + wave) * 32 + l31;
  const bool q_ok = q_row < T;
  const float* __restrict__ base = qkv + (long long)b * T * ld + hh * D;
  bool bad = false;
  auto split8 = [&](const f32x4 a, const f32x4 c, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (e < 4 ? a[e] : c[e - 4]) * SIN;
      bad |= !(fabsf(x) < 65000.0f);
      hi[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)hi[e]);
    }
  };
  f16x8 qh[8], ql[8];   // d = 16 s + 8 lh + 0..7
  {
    const float* __restrict__ qp = base + (long long)(q_ok ? q_row : T - 1) * ld + 8 * lh;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      split8(*reinterpret_cast<const f32x4*>(qp + 16 * s), *reinterpret_cast<const f32x4*>(qp + 16 * s + 4), qh[s], ql[s]);
  }
  f32x16 O[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int H3 = ld / 3;
  for (int k0 = 0; k0 < T; k0 += BK) {
    __syncthreads();   // the previous block's tiles are consumed
#pragma unroll 2
    for (int j = 0; j < 1024 / (64 * NW); ++j) {
      const int idx = tid + 64 * NW * j;
      const int key = idx >> 5, c4 = (idx & 31) << 2;
      const int kt = k0 + key;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (kt < T) {
        const float* __restrict__ rp = base + (long long)kt * ld + c4;
        kv = *reinterpret_cast<const f32x4*>(rp + H3);
        vv = *reinterpret_cast<const f32x4*>(rp + 2 * H3);
      }
      f16x4 kh4, kl4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = kv[e] * SIN, y = vv[e] * SIN;
        bad |= !(fabsf(x) < 65000.0f) || !(fabsf(y) < 65000.0f);
        kh4[e] = (_Float16)x;
        kl4[e] = (_Float16)(x - (float)kh4[e]);
        const _Float16 vh = (_Float16)y, vl = (_Float16)(y - (float)vh);
        *reinterpret_cast<_Float16*>(Vp + (c4 + e) * VROW + key * 2) = vh;
        *reinterpret_cast<_Float16*>(Vp + D * VROW + (c4 + e) * VROW + key * 2) = vl;
      }
      *reinterpret_cast<f16x4*>(Kp + key * KROW + c4 * 2) = kh4;
      *reinterpret_cast<f16x4*>(Kp + BK * KROW + key * KROW + c4 * 2) = kl4;
    }
    if (tid < BK) kp[tid] = (k0 + tid < T) ? keep[(long long)b * T + k0 + tid] : 0.f;
    __syncthreads();
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    {
      const char* kr = Kp + l31 * KROW + 16 * lh;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(kr + 32 * s);
        const f16x8 al = *reinterpret_cast<const f16x8*>(kr + 32 * s + BK * KROW);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], S, 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] *= 1.0f / (SIN * SIN);
      if (kp[acc_row(r, lh)] == 0.f) S[r] = -INFINITY;
      mx = fmaxf(mx, S[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float ms = m_new == -INFINITY ? 0.f : m_new;     // a block of masked keys only must not produce inf - inf
    const float scale = expf(m - ms);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = expf(S[r] - ms);
      ps += S[r];
    }
    ps += __shfl_xor(ps, 32);
    l = l * scale + ps;
    m = m_new;
    f16x8 ph[2], pl[2];   // step t: registers 8t .. 8t+7 = keys 16 t + 4 lh + (j & 3) + 8 (j >> 2)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = S[8 * t + j] * PSC;
        ph[t][j] = (_Float16)x;
        pl[t][j] = (_Float16)(x - (float)ph[t][j]);
      }
    const bool rescale = __any(scale != 1.0f);   // wave-uniform: once the running maxima have settled nothing is rescaled
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      if (rescale) {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] *= scale;
      }
      const char* vr = Vp + (32 * dt + l31) * VROW + 8 * lh;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u32x2_t h0 = *reinterpret_cast<const u32x2_t*>(vr + 32 * t), h1 = *reinterpret_cast<const u32x2_t*>(vr + 32 * t + 16);
        const u32x2_t l0 = *reinterpret_cast<const u32x2_t*>(vr + 32 * t + D * VROW), l1 = *reinterpret_cast<const u32x2_t*>(vr + 32 * t + 16 + D * VROW);
        const f16x8 ah = __builtin_bit_cast(f16x8, u32x4_t{h0[0], h0[1], h1[0], h1[1]});
        const f16x8 al = __builtin_bit_cast(f16x8, u32x4_t{l0[0], l0[1], l1[0], l1[1]});
        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ph[t], O[dt], 0, 0, 0);
        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pl[t], O[dt], 0, 0, 0);
        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ph[t], O[dt], 0, 0, 0);
      }
    }
  }
  if (q_ok && out_h) {
    // the attention output is read by the output projection only: written as the hi / lo fp16 planes of 16 x value that gemm_h2w_kernel
    // stages (registers 4 g .. 4 g + 3 of a lane are 4 consecutive d: one 8-byte store per plane)
    const float inv = 1.0f / (l * SIN * PSC);
    _Float16* __restrict__ oph = out_h + ((long long)b * T + q_row) * ldo + hh * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f16x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = O[dt][4 * gq + e] * inv * SIN;
          bad |= !(fabsf(x) < 65000.0f);
          hv[e] = (_Float16)x;
          lv[e] = (_Float16)(x - (float)hv[e]);
        }
        *reinterpret_cast<f16x4*>(oph + 32 * dt + 8 * gq + 4 * lh) = hv;
        *reinterpret_cast<f16x4*>(oph + out_plane + 32 * dt + 8 * gq + 4 * lh) = lv;
      }
  }
  if (range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(range_events, 1u);
  if (q_ok && !out_h) {
    const float inv = 1.0f / (l * SIN * PSC);
    float* __restrict__ op = out + ((long long)b * T + q_row) * ldo + hh * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) op[32 * dt + acc_row(r, lh)] = O[dt][r] * inv;
  }
}

// ---- token embeddings ----------------------------------------------------------------------------
constexpr int FLP_LDS = 2 * (2 * 32 * 272 + 2 * 128 * 80);   // flash_attn_planes_kernel: two K and two V^T buffers = 75 776 B
// ---- round 4: the fused attention on PRE-SPLIT operands -------------------------------------------------------------------------------
// flash_attn_split_kernel splits K and V while it stages every 32-key block — and transposes V with 2-byte LDS writes — once per 64 or 128
// queries: at T = 1000 a K / V element is split 8-16 times, between two barriers and with its global loads exposed.  Here
// qkv_split_kernel (BSG_QKV_FUSED=0; by default the QKV product's own epilogue, H2wArgs::qkv_T — one launch less per layer) splits the QKV
// projection's output ONCE into fp16 planes — Q and K as [row][256] (hi, lo), V TRANSPOSED as
// [b][256 (head, d)][Tp keys] (Tp = T rounded up to 32, the padding zeroed; keys in fragment order within groups of 16) — plus one mask word per (utterance, 32-key block), and
// flash_attn_planes_kernel stages a key block as plain 16-byte copies while the MFMAs of the blocks before it run: K two blocks ahead, V^T one,
// one barrier per block, no vector arithmetic in the staging, the softmax in the log2 domain (v_exp_f32, the 2^10 of the P operand folded into the
// exponent), its two cross-half reductions as v_permlane32_swap, blocks whose mask word is all ones skip the masking.  Same products, fragments
// and key order as flash_attn_split_kernel (its comment above).
// Measured (T = 1000, decoder layer): 118 -> 88 us at B = 16, 99 -> 63 us at B = 1 against the first planes form (the split-in-kernel form: 270 /
// 200 us).  PMC of this kernel (profiles/r04_flash_planes_pmc.txt): matrix pipe busy 25 % of the wave cycles, vector issue 29 %, LDS 8 %,
// s_waitcnt 18 % — the grid is B x heads x T / 32 = 1024 waves at B = 16, ONE wave per SIMD, so nothing hides the in-order dependences of a wave.
// Batches that leave SIMDs empty split the KEYS of a query tile over gridDim.z = 2 or 4 workgroups (each a shorter serial chain of key blocks);
// the last to arrive combines the un-normalised partials in the order of z: a single utterance 63 -> 27 us per decoder layer, the 8 utterances of
// a configs[3] rank 66 -> 50 us.  (With agent-scope fences around the count the same split was SLOWER than no split — 100 against 66 us: every
// workgroup's release is an L2 write-back; write-through stores + sc1 loads, the residual launches' hand-off form, cost nothing measurable.)
// mask words of flash_attn_planes_kernel when the QKV product writes the planes itself (H2wArgs::qkv_T): one wave per (utterance, 32-key block)
__global__ __launch_bounds__(256) void key_mask_kernel(const float* __restrict__ keep, unsigned* __restrict__ kmask, int B, int T, int Tp) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, nblk = Tp / 32;
  if (w >= B * nblk) return;
  const int b = w / nblk, t = (w - b * nblk) * 32 + lane;
  const unsigned long long mk = __builtin_amdgcn_ballot_w64(lane < 32 && t < T && keep[(long long)b * T + t] != 0.f);
  if (lane == 0) kmask[w] = (unsigned)mk;
}

__global__ __launch_bounds__(256) void qkv_split_kernel(const float* __restrict__ qkv, _Float16* __restrict__ qk, long long qk_plane,
                                                        _Float16* __restrict__ vt, long long vt_plane, int T, int Tp,
                                                        const float* __restrict__ keep, unsigned* __restrict__ kmask,
                                                        unsigned* __restrict__ range_events) {
  __shared__ __attribute__((aligned(16))) _Float16 vh[32][H + 8], vl[32][H + 8];   // the tile's V rows for the transpose
  const int b = blockIdx.y, t0 = blockIdx.x * 32, tid = threadIdx.x;
  bool bad = false;
  if (tid < 64) {   // bit j of the tile's mask word: key t0 + j takes part in the softmax (keep != 0, j < T)
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(tid < 32 && t0 + tid < T && keep[(long long)b * T + t0 + tid] != 0.f);
    if (tid == 0) kmask[b * (Tp / 32) + blockIdx.x] = (unsigned)mk;
  }
  // Q | K: 32 rows x 512 columns, 8 values per item -> [row][512] planes (Q = columns 0..255, K = 256..511)
  for (int it = tid; it < 32 * 96; it += 256) {
    const int r = it / 96, c8 = (it - r * 96) * 8, t = t0 + r;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
    if (t < T) {
      const float* p = qkv + ((long long)b * T + t) * (3 * H) + c8;
      a = *reinterpret_cast<const f32x4*>(p);
      c = *reinterpret_cast<const f32x4*>(p + 4);
    }
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (e < 4 ? a[e] : c[e - 4]) * 16.0f;
      bad |= !(fabsf(x) < 65000.0f);
      h[e] = (_Float16)x;
      l[e] = (_Float16)(x - (float)h[e]);
    }
    if (c8 < 2 * H) {
      if (t < T) {
        _Float16* d = qk + ((long long)b * T + t) * (2 * H) + c8;
        *reinterpret_cast<f16x8*>(d) = h;
        *reinterpret_cast<f16x8*>(d + qk_plane) = l;
      }
    } else {
      *reinterpret_cast<f16x8*>(&vh[r][c8 - 2 * H]) = h;      // rows t >= T: zeros
      *reinterpret_cast<f16x8*>(&vl[r][c8 - 2 * H]) = l;
    }
  }
  __syncthreads();
  // V^T: thread = one (head, d) row, 32 keys = 64 bytes per plane
  {
    const int d = tid;
    _Float16 th[32], tl[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) { th[r] = vh[r][d]; tl[r] = vl[r][d]; }
    _Float16* dst = vt + ((long long)b * H + d) * Tp + t0;
#pragma unroll
    // within a group of 16 keys the stored order is 0-3, 8-11, 4-7, 12-15: the 8 keys one lane feeds to a P V MFMA (16 t + 4 lh + (j & 3) + 8 (j >> 2))
    // are then 16 contiguous bytes
    for (int g = 0; g < 4; ++g) {
      const int k = 16 * (g >> 1) + 4 * (g & 1);
      *reinterpret_cast<f16x8*>(dst + 8 * g) = f16x8{th[k], th[k + 1], th[k + 2], th[k + 3], th[k + 8], th[k + 9], th[k + 10], th[k + 11]};
      *reinterpret_cast<f16x8*>(dst + vt_plane + 8 * g) = f16x8{tl[k], tl[k + 1], tl[k + 2], tl[k + 3], tl[k + 8], tl[k + 9], tl[k + 10], tl[k + 11]};
    }
  }
  if (range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && (tid & 63) == 0) atomicAdd(range_events, 1u);
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void flash_attn_planes_kernel(const _Float16* __restrict__ qk, long long qk_plane,
                                                                       const _Float16* __restrict__ vt, long long vt_plane,
                                                                       const unsigned* __restrict__ kmask, int T, int Tp, int heads,
                                                                       _Float16* __restrict__ out_h, long long out_plane, int ldo,
                                                                       unsigned* __restrict__ range_events, float* __restrict__ part,
                                                                       unsigned* __restrict__ cnt) {
  constexpr int D = 128, BK = 32, KROW = 2 * D + 16, VROW = 2 * BK + 16;   // bytes per LDS row: 272 (68 dwords = 4 mod 64), 80
  constexpr int KB = 2 * BK * KROW, VB = 2 * D * VROW;                     // K hi | K lo, V^T hi | V^T lo of one key block
  constexpr int NTH = 64 * NW, NPC = 1024 / NTH;                           // 16-byte pieces of a block's K (or V^T) per thread
  constexpr float SIN = 16.0f;
  constexpr float C2 = 1.4426950408889634f / (SIN * SIN);                  // scores -> log2 domain (operands carry SIN each)
  extern __shared__ __attribute__((aligned(16))) char fl[];                // K buffers [2][KB] | V^T buffers [2][VB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y / heads, hh = blockIdx.y - b * heads;
  const int q_row = (blockIdx.x * NW + wave) * 32 + l31;
  const bool q_ok = q_row < T;
  const _Float16* __restrict__ qkb = qk + (long long)b * T * (2 * H) + hh * D;          // Q of this head; K at + H
  const _Float16* __restrict__ vtb = vt + ((long long)b * H + hh * D) * Tp;
  const unsigned* __restrict__ kmb = kmask + b * (Tp / 32);
  f16x8 qh[8], ql[8];   // d = 16 s + 8 lh + 0..7
  {
    const _Float16* qp = qkb + (long long)(q_ok ? q_row : T - 1) * (2 * H) + 8 * lh;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      qh[s] = *reinterpret_cast<const f16x8*>(qp + 16 * s);
      ql[s] = *reinterpret_cast<const f16x8*>(qp + 16 * s + qk_plane);
    }
  }
  // staging: K of a block = 1024 16-byte pieces (plane p >> 9, key (p >> 4) & 31, chunk p & 15), V^T likewise (plane p >> 9, row d = (p >> 2) & 127,
  // chunk p & 3).  One register set serves both.  The loop is software-pipelined: iteration i forms the scores of block i + 1 (matrix pipe)
  // while the softmax of block i runs (vector pipe), so K is staged TWO blocks ahead and V^T one:
  //   iteration i reads K(i+1) from K buffer (i+1)&1 and V^T(i) from V buffer i&1, writes K(i+2) to K buffer i&1 (last read in iteration i-1) and
  //   V^T(i+1) to V buffer (i+1)&1 (last read in iteration i-1); one barrier per iteration.
  u32x4_t st[NPC], sv[NPC];
  // piece j of thread tid: K key (tid >> 4) + KPP j' (j' = j mod NPC/2), chunk tid & 15, plane j / (NPC/2); V^T row (tid >> 2) + VPP j', chunk
  // tid & 3.  Per thread ONE 32-bit offset each for K, V^T and the two LDS images; everything that depends on j or on the block is wave-uniform.
  constexpr int KPP = NTH / 16, VPP = NTH / 4, HP = NPC / 2;
  const unsigned koff = ((tid >> 4) * (2 * H) + (tid & 15) * 8) * 2, kls = (tid >> 4) * KROW + (tid & 15) * 16;
  const unsigned voff = ((tid >> 2) * Tp + (tid & 3) * 8) * 2, vls = (tid >> 2) * VROW + (tid & 3) * 16;
  const char* __restrict__ kbase = reinterpret_cast<const char*>(qkb + H);
  const char* __restrict__ vbase = reinterpret_cast<const char*>(vtb);
  auto load_k = [&](int k0) {   // rows past T (k0 is clamped to the padded length): the next utterance's keys or stale workspace — their scores are masked
    k0 = min(k0, Tp - BK);
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const char* bj = kbase + ((long long)(k0 + KPP * (j % HP)) * (2 * H) + (j / HP ? qk_plane : 0)) * 2;
      st[j] = *reinterpret_cast<const u32x4_t*>(bj + koff);
    }
  };
  auto store_k = [&](int buf) {
    char* sb = fl + buf * KB + kls;
#pragma unroll
    for (int j = 0; j < NPC; ++j) *reinterpret_cast<u32x4_t*>(sb + (j / HP) * BK * KROW + KPP * (j % HP) * KROW) = st[j];
  };
  auto load_v = [&](int k0) {   // k0 + 32 <= Tp: the planes are padded to Tp keys and the padding is zero
    k0 = min(k0, Tp - BK);
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const char* bj = vbase + ((long long)(VPP * (j % HP)) * Tp + k0 + (j / HP ? vt_plane : 0)) * 2;
      sv[j] = *reinterpret_cast<const u32x4_t*>(bj + voff);
    }
  };
  auto store_v = [&](int buf) {
    char* sb = fl + 2 * KB + buf * VB + vls;
#pragma unroll
    for (int j = 0; j < NPC; ++j) *reinterpret_cast<u32x4_t*>(sb + (j / HP) * D * VROW + VPP * (j % HP) * VROW) = sv[j];
  };
  auto scores = [&](int buf, f32x16& S) {   // S[key][query] = sum_d K Q, 3 fp16 products per fp32 product
    const char* kr = fl + buf * KB + l31 * KROW + 16 * lh;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(kr + 32 * s);
      const f16x8 al = *reinterpret_cast<const f16x8*>(kr + 32 * s + BK * KROW);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], s == 0 ? zero : S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], S, 0, 0, 0);
    }
  };
  f32x16 O[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;   // running maximum (log2 domain), running sum (carries the 2^10 of the P operand)
  // One key block: S = this block's scores (formed in the previous iteration), Sn <- the next block's.  Straight-line code — the staging is
  // unconditional (past the end it moves clamped rows nobody reads) — so that the scheduler can run the Q K^T MFMAs of block i + 1 under the softmax
  // of block i and the staging under the P V MFMAs.
  auto block = [&](auto masked, auto curc, int k0, f32x16& S, f32x16& Sn, unsigned km) {
    constexpr bool MASKED = decltype(masked)::value;
    constexpr int cur = decltype(curc)::value;
    store_k(cur);            // K(i+2), requested at the start of the previous iteration
    load_k(k0 + 3 * BK);
    load_v(k0 + BK);
    scores(cur ^ 1, Sn);
    if constexpr (MASKED) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (!((km >> acc_row(r, lh)) & 1u)) S[r] = -INFINITY;
    }
    float mx = S[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);   // the other 16 keys of this query
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float m_new = fmaxf(m, mx * C2);
    const float ms = m_new == -INFINITY ? 0.f : m_new;     // a block of masked keys only must not produce inf - inf
    const float scale = __builtin_amdgcn_exp2f(m - ms);
    const float off = 10.0f - ms;                          // P carries 2^10: the fp16 hi / lo split of values <= 1024
    float ps = 0.f;
    f16x8 ph[2], pl[2];   // step t: registers 8t .. 8t+7 = keys 16 t + 4 lh + (j & 3) + 8 (j >> 2)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = __builtin_amdgcn_exp2f(fmaf(S[8 * t + j], C2, off));
        ps += x;
        ph[t][j] = (_Float16)x;
        pl[t][j] = (_Float16)(x - (float)ph[t][j]);
      }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps), __float_as_uint(ps), false, false);
      ps = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    l = l * scale + ps;
    m = m_new;
    if (__any(scale != 1.0f)) {   // wave-uniform: once the running maxima have settled nothing is rescaled
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] *= scale;
    }
    const char* Vp = fl + 2 * KB + cur * VB;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const char* vr = Vp + (32 * dt + l31) * VROW + 16 * lh;   // the keys of a lane are contiguous (qkv_split_kernel's order): one 16-byte read, conflict-free
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(vr + 32 * t);
        const f16x8 al = *reinterpret_cast<const f16x8*>(vr + 32 * t + D * VROW);
        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ph[t], O[dt], 0, 0, 0);
        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pl[t], O[dt], 0, 0, 0);
        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ph[t], O[dt], 0, 0, 0);
      }
    }
    store_v(cur ^ 1);        // V^T(i+1)
    __syncthreads();         // this block is consumed; K(i+2) and V^T(i+1) are complete
  };
  using std::integral_constant;
  // key split (gridDim.z = KS > 1, small batches): this workgroup takes the key blocks [kbeg, kend) of its query tile and leaves an un-normalised
  // partial (O, m, l); the last of the KS workgroups to arrive combines them in the order of z (below)
  const int KS = (int)gridDim.z;
  const int nbz = (Tp / BK + KS - 1) / KS;
  const int kbeg = (int)blockIdx.z * nbz * BK, kend = min(T, kbeg + nbz * BK);
  load_k(kbeg);
  store_k(0);
  load_k(kbeg + BK);
  store_k(1);
  load_v(kbeg);
  store_v(0);
  load_k(kbeg + 2 * BK);
  __syncthreads();
  f32x16 S0, S1;
  scores(0, S0);
  __syncthreads();   // K buffer 0 is free for K(2)
  for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
    const unsigned km0 = kmb[k0 >> 5];   // wave-uniform
    if (km0 == 0xffffffffu) block(integral_constant<bool, false>{}, integral_constant<int, 0>{}, k0, S0, S1, km0);
    else block(integral_constant<bool, true>{}, integral_constant<int, 0>{}, k0, S0, S1, km0);
    if (k0 + BK >= kend) break;
    const unsigned km1 = kmb[(k0 >> 5) + 1];
    if (km1 == 0xffffffffu) block(integral_constant<bool, false>{}, integral_constant<int, 1>{}, k0 + BK, S1, S0, km1);
    else block(integral_constant<bool, true>{}, integral_constant<int, 1>{}, k0 + BK, S1, S0, km1);
  }
  if (KS > 1) {
    // partial of (unit, z): O [NW][16 register quads][64 lanes][4], m [NW][64], l [NW][64] (16 bytes per lane: coalesced).  The hand-off form of
    // the residual launches (diffnet_h2.hip): write-through (sc1) stores, drained, a barrier, then the count; the combining workgroup reads with sc1
    // loads — no agent-scope fence (an L2 write-back per workgroup: measured 45 us per launch of 512 workgroups)
    constexpr int PS = NW * 64 * 66;
    const int unit = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    const rsrc_t rp = mk_rsrc(part + (long long)unit * KS * PS, (unsigned)(KS * PS * 4));
    const int zoff = (int)blockIdx.z * PS * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = {O[dt][4 * g4], O[dt][4 * g4 + 1], O[dt][4 * g4 + 2], O[dt][4 * g4 + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp, ((wave * 16 + dt * 4 + g4) * 64 + lane) * 16, zoff, 16);
      }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), rp, (NW * 4096 + wave * 64 + lane) * 4, zoff, 16);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, l), rp, (NW * 4096 + NW * 64 + wave * 64 + lane) * 4, zoff, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* arrived = reinterpret_cast<unsigned*>(fl);   // (the key buffers are dead)
    if (tid == 0) *arrived = __hip_atomic_fetch_add(cnt + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*arrived != (unsigned)(KS - 1)) return;
    if (tid == 0) __hip_atomic_store(cnt + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    float M = -INFINITY;
    for (int z = 0; z < KS; ++z)
      M = fmaxf(M, __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (NW * 4096 + wave * 64 + lane) * 4, z * PS * 4, 16)));
    const float Ms = M == -INFINITY ? 0.f : M;
    l = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
    for (int z = 0; z < KS; ++z) {   // fixed order: the result does not depend on which workgroup combines
      const float mz = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (NW * 4096 + wave * 64 + lane) * 4, z * PS * 4, 16));
      const float lz = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (NW * 4096 + NW * 64 + wave * 64 + lane) * 4, z * PS * 4, 16));
      const float wz = __builtin_amdgcn_exp2f(mz - Ms);
      l = fmaf(wz, lz, l);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, ((wave * 16 + dt * 4 + g4) * 64 + lane) * 16, z * PS * 4, 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) O[dt][4 * g4 + e] = fmaf(wz, v[e], O[dt][4 * g4 + e]);
        }
    }
  }
  bool bad = false;
  if (q_ok) {
    // the attention output is read by the output projection only: hi / lo fp16 planes of 16 x value (registers 4 g .. 4 g + 3 of a lane are 4
    // consecutive d: one 8-byte store per plane)
    const float inv = 1.0f / (l * SIN);   // l carries the 2^10 of P
    _Float16* __restrict__ oph = out_h + ((long long)b * T + q_row) * ldo + hh * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f16x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = O[dt][4 * gq + e] * inv * SIN;
          bad |= !(fabsf(x) < 65000.0f);
          hv[e] = (_Float16)x;
          lv[e] = (_Float16)(x - (float)hv[e]);
        }
        *reinterpret_cast<f16x4*>(oph + 32 * dt + 8 * gq + 4 * lh) = hv;
        *reinterpret_cast<f16x4*>(oph + out_plane + 32 * dt + 8 * gq + 4 * lh) = lv;
      }
  }
  if (range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(range_events, 1u);
}

// x0 = sqrt(H) * E_tok[txt]; lang_e = E_lang[lang]                      (diffsinger_midi/fs2.py:28,122)
__global__ void embed_tokens_kernel(const long long* __restrict__ txt, const long long* __restrict__ lang,
                                    const float* __restrict__ Etok, const float* __restrict__ Elang, float* __restrict__ x0,
                                    float* __restrict__ lang_e, long long rows, float scale, _Float16* __restrict__ x0h,
                                    _Float16* __restrict__ x0l, unsigned* __restrict__ range_events) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  f32x4 e = reinterpret_cast<const f32x4*>(Etok + txt[row] * H)[lane];
  e *= scale;
  reinterpret_cast<f32x4*>(x0 + row * H)[lane] = e;
  if (x0h) {   // x0 as the operand of the ESM's query projection on the pre-split GEMM: hi / lo fp16 planes of 16 x0
    ln_f16x4 hv, lv;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = e[c] * 16.0f;
      bad |= !(fabsf(x) < 65000.0f);
      hv[c] = (_Float16)x;
      lv[c] = (_Float16)(x - (float)hv[c]);
    }
    reinterpret_cast<ln_f16x4*>(x0h + row * H)[lane] = hv;
    reinterpret_cast<ln_f16x4*>(x0l + row * H)[lane] = lv;
    if (range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(range_events, 1u);
  }
  reinterpret_cast<f32x4*>(lang_e + row * H)[lane] = reinterpret_cast<const f32x4*>(Elang + lang[row] * H)[lane];
}

// x = ((((x0 + midi) + mdur) + slur) + dyn) * sqrt(H) + pe_rev[j], then * keep      (fs2.py:30-34, FFTBlocks :297)
__global__ void embed_finish_kernel(const float* __restrict__ x0, const float* __restrict__ dyn,
                                    const long long* __restrict__ txt, const long long* __restrict__ pitch_midi,
                                    const float* __restrict__ midi_dur, const long long* __restrict__ is_slur,
                                    const float* __restrict__ Emidi, const float* __restrict__ Wdur,
                                    const float* __restrict__ bdur, const float* __restrict__ Eslur,
                                    const float* __restrict__ pe, float* __restrict__ x, float* __restrict__ keep,
                                    long long rows, int Tt, float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int j = (int)(row % Tt);
  const float kp = txt[row] != 0 ? 1.f : 0.f;
  const f32x4 a = reinterpret_cast<const f32x4*>(x0 + row * H)[lane];
  const f32x4 mi = reinterpret_cast<const f32x4*>(Emidi + pitch_midi[row] * H)[lane];
  const f32x4 wd = reinterpret_cast<const f32x4*>(Wdur)[lane], bd = reinterpret_cast<const f32x4*>(bdur)[lane];
  const f32x4 sl = reinterpret_cast<const f32x4*>(Eslur + is_slur[row] * H)[lane];
  const f32x4 dy = reinterpret_cast<const f32x4*>(dyn + row * H)[lane];
  const f32x4 pv = reinterpret_cast<const f32x4*>(pe + (long long)j * H)[lane];
  const float md = midi_dur[row];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = __fadd_rn(a[e], mi[e]);
    v = __fadd_rn(v, __fadd_rn(__fmul_rn(md, wd[e]), bd[e]));
    v = __fadd_rn(v, sl[e]);
    v = __fadd_rn(v, dy[e]);
    o[e] = __fadd_rn(__fmul_rn(v, scale), pv[e]) * kp;
  }
  reinterpret_cast<f32x4*>(x + row * H)[lane] = o;
  if (lane == 0) keep[row] = kp;
}

// ---- ESM attention over the BATCH axis (reference quirk, common_layers.py:853) --------------------
// q,k,v: [L=B][N=Tt][H]; for every (n, head) softmax over the L keys.  One thread per (l, n, head).
// q / o hold the Lq QUERY utterances only (all L of them, or a rank's rows: bsg_fs2midi_encode_rows); k / v all L.
template <int HD>
__global__ void esm_attention_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                     float* __restrict__ o, int L, int Lq, int N, int heads, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Lq * N * heads;
  if (idx >= total) return;
  const int h = (int)(idx % heads);
  const int n = (int)((idx / heads) % N);
  const int l = (int)(idx / ((long long)heads * N));
  float qv[HD];
  const float* qp = q + ((long long)l * N + n) * H + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) qv[d] = qp[d] * scale;
  float m = -INFINITY;
  for (int j = 0; j < L; ++j) {
    const float* kp = k + ((long long)j * N + n) * H + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qv[d], kp[d], s);
    m = fmaxf(m, s);
  }
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  float sum = 0.f;
  for (int j = 0; j < L; ++j) {
    const float* kp = k + ((long long)j * N + n) * H + h * HD;
    const float* vp = v + ((long long)j * N + n) * H + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qv[d], kp[d], s);
    const float e = expf(s - m);
    sum += e;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = fmaf(e, vp[d], acc[d]);
  }
  float* op = o + ((long long)l * N + n) * H + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) op[d] = acc[d] / sum;
}

// The same for L <= 64 utterances with ONE WAVE per (position n, head): lane = query utterance l; the L keys and values of (n, head) are staged in
// LDS once (128-byte rows, coalesced) and read back as broadcasts, instead of every thread streaming its own copy of them through L1 with
// 64 different cache lines per load instruction (186 us at the 64-row token front of configs[3]; this form: see DESIGN.md).  Two passes as
// above (max, then exp and sum) in the same order of operations.  ld = row stride of q / k / v; out fp32 [.][H] and / or hi / lo planes of 16 x out.
__global__ __launch_bounds__(256) void esm_attention_wave_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, int ldq, int ldkv, float* __restrict__ o,
                                                                 _Float16* __restrict__ oh, long long o_plane, int L, int Lq, int N, int heads,
                                                                 float scale, unsigned* __restrict__ range_events) {
  constexpr int HD = 32;
  __shared__ __attribute__((aligned(16))) float kv[4][2][64][HD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 4 + wave;          // (n, head)
  const bool live = unit < N * heads;              // (no early return: the barrier below is the workgroup's)
  const int n = live ? unit / heads : 0, hh = live ? unit - n * heads : 0;
  for (int it = 0; it < (L * 8 + 63) / 64; ++it) {
    const int j = it * 8 + (lane >> 3), d4 = (lane & 7) * 4;
    if (live && j < L) {
      const long long row = ((long long)j * N + n) * ldkv + hh * HD + d4;
      *reinterpret_cast<f32x4*>(&kv[wave][0][j][d4]) = *reinterpret_cast<const f32x4*>(k + row);
      *reinterpret_cast<f32x4*>(&kv[wave][1][j][d4]) = *reinterpret_cast<const f32x4*>(v + row);
    }
  }
  __syncthreads();
  if (!live || lane >= Lq) return;
  const int l = lane;   // LOCAL query row: q, o and the planes hold the Lq query utterances only; the keys are all L
  float qv[HD];
  {
    const float* qp = q + ((long long)l * N + n) * ldq + hh * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
      qv[d] = t[0] * scale; qv[d + 1] = t[1] * scale; qv[d + 2] = t[2] * scale; qv[d + 3] = t[3] * scale;
    }
  }
  float m = -INFINITY;
  for (int j = 0; j < L; ++j) {
    const float* kp = kv[wave][0][j];
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qv[d], kp[d], s);
    m = fmaxf(m, s);
  }
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  float sum = 0.f;
  for (int j = 0; j < L; ++j) {
    const float* kp = kv[wave][0][j];
    const float* vp = kv[wave][1][j];
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qv[d], kp[d], s);
    const float e = expf(s - m);
    sum += e;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = fmaf(e, vp[d], acc[d]);
  }
  const long long orow = ((long long)l * N + n) * H + hh * HD;
  bool bad = false;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    f32x4 t;
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = acc[d + c] / sum;
    if (o) *reinterpret_cast<f32x4*>(o + orow + d) = t;
    if (oh) {
      ln_f16x4 hv, lv;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float x = t[c] * 16.0f;
        bad |= !(fabsf(x) < 65000.0f);
        hv[c] = (_Float16)x;
        lv[c] = (_Float16)(x - (float)hv[c]);
      }
      *reinterpret_cast<ln_f16x4*>(oh + orow + d) = hv;
      *reinterpret_cast<ln_f16x4*>(oh + o_plane + orow + d) = lv;
    }
  }
  if (range_events && bad) atomicAdd(range_events, 1u);
}

// ---- encoder -> frames ---------------------------------------------------------------------------
// dur_inp = (enc + spk) * src_keep                                                    (fs2.py:164)
__global__ void add_spk_kernel(const float* __restrict__ enc, const long long* __restrict__ spk_id,
                               const float* __restrict__ Espk, const float* __restrict__ keep, float* __restrict__ out,
                               long long rows, int Tt) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(row / Tt);
  f32x4 e = reinterpret_cast<const f32x4*>(enc + row * H)[lane];
  const f32x4 s = reinterpret_cast<const f32x4*>(Espk + spk_id[b] * H)[lane];
  e = (e + s) * keep[row];
  reinterpret_cast<f32x4*>(out + row * H)[lane] = e;
}

// decoder_inp[b,t] = (pad(enc)[b, mel2ph[b,t]] + spk[b] + style[b]) * (mel2ph > 0)          (fs2.py:168-189)
__global__ void gather_frames_kernel(const float* __restrict__ enc, const long long* __restrict__ mel2ph,
                                     const long long* __restrict__ spk_id, const long long* __restrict__ style_id,
                                     const float* __restrict__ Espk, const float* __restrict__ Estyle,
                                     float* __restrict__ out, long long rows, int T, int Tt) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(row / T);
  const long long ph = mel2ph[row];
  f32x4 e = {0.f, 0.f, 0.f, 0.f};
  if (ph > 0 && ph <= Tt) e = reinterpret_cast<const f32x4*>(enc + ((long long)b * Tt + (ph - 1)) * H)[lane];
  const f32x4 s = reinterpret_cast<const f32x4*>(Espk + spk_id[b] * H)[lane];
  const f32x4 st = reinterpret_cast<const f32x4*>(Estyle + style_id[b] * H)[lane];
  const float kp = ph > 0 ? 1.f : 0.f;
  e = ((e + s) + st) * kp;
  reinterpret_cast<f32x4*>(out + row * H)[lane] = e;
}

// FFTBlocks entry for the decoder (tts_modules.py:289-297): padding = (sum |x| == 0), positions =
// cumsum(x[...,0] != 0) * (x[...,0] != 0) (utils/__init__.py:146-158), x = (x + alpha * table[pos]) * keep.
// One wave per utterance scans T in 64-frame chunks; then every lane copies rows.
__global__ void decoder_positions_kernel(const float* __restrict__ x, int* __restrict__ pos, float* __restrict__ keep, int T) {
  const int b = blockIdx.x, lane = threadIdx.x;   // 64 threads
  int carry = 0;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    bool nz = false;
    if (t < T) nz = x[((long long)b * T + t) * H] != 0.f;
    const unsigned long long bal = __ballot(nz);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull)) + (nz ? 1 : 0);
    if (t < T) pos[(long long)b * T + t] = nz ? carry + pre : 0;
    carry += __popcll(bal);
  }
  (void)keep;
}
__global__ void decoder_entry_kernel(float* __restrict__ x, const int* __restrict__ pos, const float* __restrict__ table,
                                     const float* __restrict__ alpha, float* __restrict__ keep, long long rows, int n_pos) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  f32x4 v = reinterpret_cast<const f32x4*>(x + row * H)[lane];
  const float asum = wave_sum(fabsf(v[0]) + fabsf(v[1]) + fabsf(v[2]) + fabsf(v[3]));
  const float kp = asum == 0.f ? 0.f : 1.f;
  int p = pos[row];
  p = p < n_pos ? p : n_pos - 1;
  const f32x4 pe = reinterpret_cast<const f32x4*>(table + (long long)p * H)[lane];
  const float a = alpha[0];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], __fmul_rn(a, pe[e])) * kp;
  reinterpret_cast<f32x4*>(x + row * H)[lane] = v;
  if (lane == 0) keep[row] = kp;
}

// ---- duration predictor tail + length regulator --------------------------------------------------
// dur = clamp(round(exp(xs) - 1), min=0) with xs already masked                       (tts_modules.py:124-129)
__global__ void dur_from_log_kernel(const float* __restrict__ xs, long long* __restrict__ dur, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = rintf(expf(xs[i]) - 1.0f);
  dur[i] = d > 0.f ? (long long)d : 0;
}
// mel2ph[b,j] = sum_i i * [cum_{i-1} <= j < cum_i]  (tts_modules.py:178-190); one workgroup per utterance
__global__ void length_regulator_kernel(const long long* __restrict__ dur, const long long* __restrict__ txt,
                                        long long* __restrict__ mel2ph, int Tt, int T) {
  extern __shared__ long long cum[];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    long long c = 0;
    for (int i = 0; i < Tt; ++i) {
      long long d = dur[(long long)b * Tt + i];
      if (txt && txt[(long long)b * Tt + i] == 0) d = 0;
      c += d;
      cum[i] = c;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    // first i with cum[i] > j
    int lo = 0, hi = Tt;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] > j) hi = mid; else lo = mid + 1;
    }
    mel2ph[(long long)b * T + j] = lo < Tt ? lo + 1 : 0;
  }
}

// conv weight [M][Cin][k] -> [k][M][Cin]  (so each tap is a plain [N,K] operand of the GEMM)
__global__ void repack_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int M, int Cin, int k) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)M * Cin * k;
  if (i >= total) return;
  const int c = (int)(i % Cin);
  const int m = (int)((i / Cin) % M);
  const int tap = (int)(i / ((long long)Cin * M));
  out[i] = w[((long long)m * Cin + c) * k + tap];
}

__global__ void mask_rows_by_index_kernel(float* __restrict__ x, const long long* __restrict__ idx, long long rows, int width) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * width) return;
  if (idx[i / width] <= 0) x[i] = 0.f;
}

}  // namespace
}  // namespace bsg

// ================================================================================================
using namespace bsg;

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != BSG_OK) return _rc; \
  } while (0)

struct FftLayerW {
  float *ln1w, *ln1b, *in_proj, *out_proj, *ln2w, *ln2b, *ffn1 /*[k][4H][H]*/, *ffn1b, *ffn2, *ffn2b;
  H2wWeights p_in, p_out, p_ffn1, p_ffn2;   // the four weight matrices once more as pre-split fp16 fragments (gemm_h2w.hip)
};

struct bsg_fs2midi {
  bsg_fs2midi_cfg cfg;
  Guard guard;   // this handle's range-event word and split-fp16 GEMM switch
  std::vector<float*> owned;
  // weights
  float *Etok, *dec_alpha, *dec_lnw, *dec_lnb, *mel_w, *mel_b, *Espk, *dur_lin_w, *dur_lin_b;
  std::vector<FftLayerW> enc, dec;
  std::vector<float*> dur_conv, dur_convb, dur_lnw, dur_lnb;
  float *esm_in_w, *esm_in_b, *esm_out_w, *esm_out_b, *esm_f0w, *esm_f0b, *esm_f2w, *esm_f2b, *esm_ln1w, *esm_ln1b, *esm_ln2w, *esm_ln2b;
  H2wWeights p_esm_q, p_esm_kv, p_esm_out, p_esm_f0, p_esm_f2;   // the ESM's Linear layers as pre-split fragments (K and V projections as one product)
  float *enc_lnw, *enc_lnb, *Emidi, *Wdur, *bdur, *Eslur, *Elang, *Estyle;
  float *dec_table, *rel_table;
  // workspace
  size_t cap_rows = 0, cap_scores = 0;
  float *w_x = nullptr, *w_a = nullptr, *w_b = nullptr, *w_qkv = nullptr, *w_ffn = nullptr, *w_keep = nullptr, *w_scores = nullptr;
  float *w_c = nullptr;
  int* w_pos = nullptr;
  unsigned short *w_ap = nullptr, *w_fp = nullptr;   // activation planes [2][rows][H] / [2][rows][4H] fp16 (operands of gemm_h2w_kernel)
  float* w_fsk = nullptr;                            // key-split partials of flash_attn_planes_kernel (small batches) and their arrival counters
  unsigned* w_fcnt = nullptr;
  size_t cap_fsk = 0, cap_fcnt = 0;
  unsigned* pack_bad = nullptr;                      // device word: packed weights beyond the fp16 range (then the pre-split GEMMs are not used)
  bool h2w_ok = false;
  int last_token_rows = 0;   // rows (utterances x tokens) the last encode ran its encoder on; rows of the last FFT stack (bsg_fs2midi_last_rows)
  int last_stack_rows = 0;
};

static int fs2_alloc(bsg_fs2midi* h, float** p, size_t n) {
  BSG_HIP(hipMalloc((void**)p, n * sizeof(float)));
  h->owned.push_back(*p);
  return BSG_OK;
}
static int fs2_copy(bsg_fs2midi* h, float** dst, const void* src, size_t n, hipStream_t st) {
  TRY(fs2_alloc(h, dst, n));
  BSG_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  return BSG_OK;
}
static int fs2_conv(bsg_fs2midi* h, float** dst, const void* src, int M, int Cin, int k, hipStream_t st) {
  TRY(fs2_alloc(h, dst, (size_t)M * Cin * k));
  const long long total = (long long)M * Cin * k;
  hipLaunchKernelGGL(repack_conv_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)src, *dst, M, Cin, k);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" void bsg_fs2midi_destroy(bsg_fs2midi* h) {
  if (!h) return;
  guard_free(&h->guard);
  for (float* p : h->owned) (void)hipFree(p);
  float* ws[] = {h->w_x, h->w_a, h->w_b, h->w_qkv, h->w_ffn, h->w_keep, h->w_scores, h->w_c};
  for (float* p : ws)
    if (p) (void)hipFree(p);
  if (h->w_pos) (void)hipFree(h->w_pos);
  if (h->w_ap) (void)hipFree(h->w_ap);
  if (h->w_fp) (void)hipFree(h->w_fp);
  if (h->w_fsk) (void)hipFree(h->w_fsk);
  if (h->w_fcnt) (void)hipFree(h->w_fcnt);
  if (h->pack_bad) (void)hipFree(h->pack_bad);
  for (std::vector<FftLayerW>* v : {&h->enc, &h->dec})
    for (FftLayerW& L : *v) { h2w_free(&L.p_in); h2w_free(&L.p_out); h2w_free(&L.p_ffn1); h2w_free(&L.p_ffn2); }
  for (H2wWeights* p : {&h->p_esm_q, &h->p_esm_kv, &h->p_esm_out, &h->p_esm_f0, &h->p_esm_f2}) h2w_free(p);
  delete h;
}

static int load_fft_layers(bsg_fs2midi* h, std::vector<FftLayerW>& out, const void* const* w, int n_layers, int ksz, hipStream_t st) {
  out.resize(n_layers);
  if (!h->pack_bad) {
    BSG_HIP(hipMalloc((void**)&h->pack_bad, sizeof(unsigned)));
    BSG_HIP(hipMemsetAsync(h->pack_bad, 0, sizeof(unsigned), st));
  }
  for (int i = 0; i < n_layers; ++i) {
    const void* const* lw = w + 10 * i;
    FftLayerW& L = out[i];
    TRY(fs2_copy(h, &L.ln1w, lw[0], H, st));
    TRY(fs2_copy(h, &L.ln1b, lw[1], H, st));
    TRY(fs2_copy(h, &L.in_proj, lw[2], (size_t)3 * H * H, st));
    TRY(fs2_copy(h, &L.out_proj, lw[3], (size_t)H * H, st));
    TRY(fs2_copy(h, &L.ln2w, lw[4], H, st));
    TRY(fs2_copy(h, &L.ln2b, lw[5], H, st));
    TRY(fs2_conv(h, &L.ffn1, lw[6], 4 * H, H, ksz, st));
    TRY(fs2_copy(h, &L.ffn1b, lw[7], 4 * H, st));
    TRY(fs2_copy(h, &L.ffn2, lw[8], (size_t)H * 4 * H, st));
    TRY(fs2_copy(h, &L.ffn2b, lw[9], H, st));
    // pre-split fragments: Linear weights [N][K]; the conv from its original [4H][H][k] layout (tap stride 1, k stride `ksz`)
    TRY(h2w_pack(&L.p_in, (const float*)lw[2], 3 * H, H, 1, 0, H, 1, h->pack_bad, st));
    TRY(h2w_pack(&L.p_out, (const float*)lw[3], H, H, 1, 0, H, 1, h->pack_bad, st));
    if (ksz <= 17) TRY(h2w_pack(&L.p_ffn1, (const float*)lw[6], 4 * H, H, ksz, 1, (long long)H * ksz, ksz, h->pack_bad, st));
    TRY(h2w_pack(&L.p_ffn2, (const float*)lw[8], H, 4 * H, 1, 0, 4 * H, 1, h->pack_bad, st));
  }
  return BSG_OK;
}

static int fs2_create_impl(bsg_fs2midi* h, const void* const* w, const float* dec_table, const float* rel_table, hipStream_t st) {
  const bsg_fs2midi_cfg& c = h->cfg;
  int i = 0;
  TRY(fs2_copy(h, &h->Etok, w[i++], (size_t)c.vocab * H, st));
  TRY(fs2_copy(h, &h->dec_alpha, w[i++], 1, st));
  i++;  // decoder.embed_positions._float_tensor: placeholder buffer of the reference, no meaning
  TRY(load_fft_layers(h, h->dec, w + i, c.dec_layers, c.dec_ffn_kernel_size, st));
  i += 10 * c.dec_layers;
  TRY(fs2_copy(h, &h->dec_lnw, w[i++], H, st));
  TRY(fs2_copy(h, &h->dec_lnb, w[i++], H, st));
  TRY(fs2_copy(h, &h->mel_w, w[i++], (size_t)c.out_dims * H, st));
  TRY(fs2_copy(h, &h->mel_b, w[i++], c.out_dims, st));
  TRY(fs2_copy(h, &h->Espk, w[i++], (size_t)c.spk_rows * H, st));
  h->dur_conv.resize(c.dur_layers); h->dur_convb.resize(c.dur_layers); h->dur_lnw.resize(c.dur_layers); h->dur_lnb.resize(c.dur_layers);
  for (int l = 0; l < c.dur_layers; ++l) {
    TRY(fs2_conv(h, &h->dur_conv[l], w[i++], H, H, c.dur_kernel, st));
    TRY(fs2_copy(h, &h->dur_convb[l], w[i++], H, st));
    TRY(fs2_copy(h, &h->dur_lnw[l], w[i++], H, st));
    TRY(fs2_copy(h, &h->dur_lnb[l], w[i++], H, st));
  }
  TRY(fs2_copy(h, &h->dur_lin_w, w[i++], H, st));
  TRY(fs2_copy(h, &h->dur_lin_b, w[i++], 1, st));
  TRY(fs2_copy(h, &h->esm_in_w, w[i++], (size_t)3 * H * H, st));
  TRY(fs2_copy(h, &h->esm_in_b, w[i++], 3 * H, st));
  TRY(fs2_copy(h, &h->esm_out_w, w[i++], (size_t)H * H, st));
  TRY(fs2_copy(h, &h->esm_out_b, w[i++], H, st));
  TRY(fs2_copy(h, &h->esm_f0w, w[i++], (size_t)H * H, st));
  TRY(fs2_copy(h, &h->esm_f0b, w[i++], H, st));
  TRY(fs2_copy(h, &h->esm_f2w, w[i++], (size_t)H * H, st));
  TRY(fs2_copy(h, &h->esm_f2b, w[i++], H, st));
  TRY(fs2_copy(h, &h->esm_ln1w, w[i++], H, st));
  TRY(fs2_copy(h, &h->esm_ln1b, w[i++], H, st));
  TRY(fs2_copy(h, &h->esm_ln2w, w[i++], H, st));
  TRY(fs2_copy(h, &h->esm_ln2b, w[i++], H, st));
  TRY(h2w_pack(&h->p_esm_q, h->esm_in_w, H, H, 1, 0, H, 1, h->pack_bad, st));
  TRY(h2w_pack(&h->p_esm_kv, h->esm_in_w + (size_t)H * H, 2 * H, H, 1, 0, H, 1, h->pack_bad, st));
  TRY(h2w_pack(&h->p_esm_out, h->esm_out_w, H, H, 1, 0, H, 1, h->pack_bad, st));
  TRY(h2w_pack(&h->p_esm_f0, h->esm_f0w, H, H, 1, 0, H, 1, h->pack_bad, st));
  TRY(h2w_pack(&h->p_esm_f2, h->esm_f2w, H, H, 1, 0, H, 1, h->pack_bad, st));
  TRY(load_fft_layers(h, h->enc, w + i, c.enc_layers, c.enc_ffn_kernel_size, st));
  i += 10 * c.enc_layers;
  TRY(fs2_copy(h, &h->enc_lnw, w[i++], H, st));
  TRY(fs2_copy(h, &h->enc_lnb, w[i++], H, st));
  i += 1 + 12;  // encoder.embed_tokens / encoder.esm.*: aliases of encoder_embed_tokens / esm.* (fs2.py:84-87)
  TRY(fs2_copy(h, &h->Emidi, w[i++], (size_t)300 * H, st));
  TRY(fs2_copy(h, &h->Wdur, w[i++], H, st));
  TRY(fs2_copy(h, &h->bdur, w[i++], H, st));
  TRY(fs2_copy(h, &h->Eslur, w[i++], (size_t)2 * H, st));
  TRY(fs2_copy(h, &h->Elang, w[i++], (size_t)2 * H, st));
  TRY(fs2_copy(h, &h->Estyle, w[i++], (size_t)3 * H, st));
  TRY(fs2_copy(h, &h->dec_table, dec_table, (size_t)c.n_pos * H, st));
  TRY(fs2_copy(h, &h->rel_table, rel_table, (size_t)c.n_rel * H, st));
  unsigned bad = 0;
  BSG_HIP(hipMemcpyAsync(&bad, h->pack_bad, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  BSG_HIP(hipStreamSynchronize(st));
  h->h2w_ok = bad == 0 && c.enc_ffn_kernel_size <= 17 && c.dec_ffn_kernel_size <= 17;
  return BSG_OK;
}

extern "C" int bsg_fs2midi_n_weights(const bsg_fs2midi_cfg* c) {
  return 3 + 10 * c->dec_layers + 2 + 2 + 1 + 4 * c->dur_layers + 2 + 12 + 10 * c->enc_layers + 2 + 1 + 12 + 6;
}

extern "C" int bsg_fs2midi_create(bsg_fs2midi** out, const bsg_fs2midi_cfg* cfg, const void* const* dev_weights,
                                  int32_t n_weights, const float* dec_pos_table, const float* rel_pos_table, void* stream) {
  BSG_REQUIRE(out && cfg && dev_weights && dec_pos_table && rel_pos_table, "fs2midi_create: null argument");
  BSG_REQUIRE(cfg->hidden_size == H, "fs2midi_create: hidden_size=%d; kernels are built for 256", cfg->hidden_size);
  BSG_REQUIRE(cfg->num_heads > 0 && H % cfg->num_heads == 0 && (H / cfg->num_heads) % 4 == 0, "fs2midi_create: num_heads=%d", cfg->num_heads);
  BSG_REQUIRE(cfg->esm_heads == 8, "fs2midi_create: esm_heads=%d (the reference fixes 8, fs2.py:83)", cfg->esm_heads);
  BSG_REQUIRE(cfg->enc_layers > 0 && cfg->dec_layers > 0 && cfg->dur_layers > 0 && cfg->vocab > 0 && cfg->out_dims > 0 &&
                  cfg->out_dims % 4 == 0 && cfg->spk_rows > 0 && cfg->n_pos > 1 && cfg->n_rel > 0,
              "fs2midi_create: bad config");
  BSG_REQUIRE(cfg->enc_ffn_kernel_size % 2 == 1 && cfg->dec_ffn_kernel_size % 2 == 1 && cfg->dur_kernel % 2 == 1,
              "fs2midi_create: SAME padding needs odd kernels");
  BSG_REQUIRE(n_weights == bsg_fs2midi_n_weights(cfg), "fs2midi_create: expected %d weight tensors, got %d",
              bsg_fs2midi_n_weights(cfg), n_weights);
  for (int i = 0; i < n_weights; ++i) BSG_REQUIRE(dev_weights[i] != nullptr, "fs2midi_create: weight %d is null", i);
  bsg_fs2midi* h = new bsg_fs2midi();
  h->cfg = *cfg;
  int rc = guard_init(&h->guard, (hipStream_t)stream);
  if (rc == BSG_OK) rc = fs2_create_impl(h, dev_weights, dec_pos_table, rel_pos_table, (hipStream_t)stream);
  if (rc != BSG_OK) {
    bsg_fs2midi_destroy(h);
    return rc;
  }
  *out = h;
  return BSG_OK;
}

// workspace for `rows` token/frame rows and attention scores of batch B, length T
static int ensure_ws(bsg_fs2midi* h, size_t rows, size_t scores, hipStream_t st) {
  if (rows > h->cap_rows) {
    BSG_HIP(hipStreamSynchronize(st));
    float** bufs[] = {&h->w_x, &h->w_a, &h->w_b, &h->w_c, &h->w_qkv, &h->w_ffn, &h->w_keep};
    for (float** p : bufs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (h->w_pos) { (void)hipFree(h->w_pos); h->w_pos = nullptr; }
    if (h->w_ap) { (void)hipFree(h->w_ap); h->w_ap = nullptr; }
    if (h->w_fp) { (void)hipFree(h->w_fp); h->w_fp = nullptr; }
    h->cap_rows = 0;
    BSG_HIP(hipMalloc((void**)&h->w_x, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->w_a, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->w_b, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->w_c, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->w_qkv, rows * 3 * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->w_ffn, rows * 4 * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->w_keep, rows * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->w_pos, rows * sizeof(int)));
    BSG_HIP(hipMalloc((void**)&h->w_ap, 2 * rows * H * sizeof(unsigned short)));
    BSG_HIP(hipMalloc((void**)&h->w_fp, 2 * rows * 4 * H * sizeof(unsigned short)));
    h->cap_rows = rows;
  }
  (void)scores;   // the score tensor of the unfused attention is allocated on demand (ensure_scores)
  return BSG_OK;
}

// [B*heads, T, T] scores: only the unfused attention path (BSG_NO_FLASH_ATTN=1, or a head dim other than 128) needs them
static int ensure_scores(bsg_fs2midi* h, size_t scores, hipStream_t st) {
  if (scores > h->cap_scores) {
    BSG_HIP(hipStreamSynchronize(st));
    if (h->w_scores) (void)hipFree(h->w_scores);
    h->w_scores = nullptr;
    h->cap_scores = 0;
    BSG_HIP(hipMalloc((void**)&h->w_scores, scores * sizeof(float)));
    h->cap_scores = scores;
  }
  return BSG_OK;
}

static int linear(const float* X, const float* W, const float* bias, float* Y, long long rows, int N, int K, int act,
                  const float* R, const float* rowscale, hipStream_t st, float alpha = 1.f, int alpha_ncols = 0) {
  GemmArgs g{};
  g.A = X; g.B = W; g.C = Y; g.M = (int)rows; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.trans_b = 1; g.taps = 1;
  g.bias_n = bias; g.alpha = alpha; g.alpha_ncols = alpha_ncols; g.act = act; g.R = R; g.ldr = N; g.rowscale = rowscale; g.batch = 1;
  return launch_gemm(g, st);
}

static int ln(const float* x, const float* w, const float* b, float* y, const float* rowscale, long long rows, float eps, hipStream_t st) {
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, w, b, y, rowscale, rows, eps, (_Float16*)nullptr,
                     (_Float16*)nullptr, (unsigned*)nullptr);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
// LayerNorm whose result is read by a pre-split GEMM only: hi / lo fp16 planes [2][rows][H]
static int ln_planes(const float* x, const float* w, const float* b, unsigned short* planes, long long rows, float eps, hipStream_t st) {
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, w, b, (float*)nullptr, (const float*)nullptr, rows, eps,
                     reinterpret_cast<_Float16*>(planes), reinterpret_cast<_Float16*>(planes) + rows * H, gemm_range_counter());
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
// y[rows][N] = epi(planes[rows][K] W^T): Linear through gemm_h2w_kernel (W pre-split); `planes_out`: the result as planes [2][rows][N] instead
static int linear_h2w(const unsigned short* planes, const H2wWeights& W, int N, const float* bias, float* Y, unsigned short* planes_out, long long rows,
                      int act, const float* R, const float* rowscale, hipStream_t st, float alpha = 1.f, int alpha_ncols = 0,
                      long long act_plane = 0) {
  H2wArgs g{};
  g.act = planes; g.act_plane = act_plane ? act_plane : rows * W.K; g.lda = W.K; g.wpack = W.pack; g.rows = (int)rows; g.K = W.K; g.Wn = W.Wn; g.taps = 1;
  g.act_is_a = 1; g.C = Y; g.ldc = N; g.out = planes_out; g.out_plane = rows * N; g.ldo = N; g.bias = bias; g.alpha = alpha;
  g.alpha_ncols = alpha_ncols; g.act_fn = act; g.R = R; g.ldr = N; g.rowscale = rowscale; g.batch = 1;
  return launch_gemm_h2w(g, st);
}

// EncSALayer x FFTBlocks tail (common_layers.py:706-730, tts_modules.py:298-305); x [B*T, H] in place
static int fft_stack(bsg_fs2midi* h, const std::vector<FftLayerW>& layers, const float* lnw, const float* lnb, int ksz,
                     float* x, const float* keep, int B, int T, hipStream_t st) {
  const long long rows = (long long)B * T;
  h->last_stack_rows = (int)rows;
  const int heads = h->cfg.num_heads, hd = H / heads;
  const float qscale = (float)sqrt(1.0 / (double)hd);
  static int env_h2w = -1;   // BSG_GEMM_H2W=0: gemm_split_kernel (operands split while staged) instead of the pre-split GEMMs
  if (env_h2w < 0) { const char* e = getenv("BSG_GEMM_H2W"); env_h2w = e ? atoi(e) : 1; }
  static int fsplit_env = -1;
  if (fsplit_env < 0) { const char* e = getenv("BSG_FLASH_SPLIT"); fsplit_env = e ? atoi(e) : 1; }
  const bool h2w = env_h2w && h->h2w_ok && gemm_split_enabled() && fsplit_env && hd == 128 && !getenv("BSG_NO_FLASH_ATTN") &&
                   h2w_supports(T, 4 * H, H, ksz, H) && h2w_supports((int)rows, 3 * H, H, 1, H) && rows * 4 * H * 2 < (1LL << 31);
  if (h2w) {
    // Every product on the 16-bit matrix pipe with PRE-SPLIT operands (gemm_h2w.hip): the weights were split into hi / lo fp16 fragments at
    // create, and each activation is written as hi / lo fp16 planes by the kernel that produces it (LayerNorm, the fused attention, the
    // GELU epilogue of the FFN convolution) — no kernel splits an operand while it stages it.  Same arithmetic as the path below.
    unsigned short* ap = h->w_ap;   // [2][rows][H]
    unsigned short* fp = h->w_fp;   // [2][rows][4H]
    static int fplanes = -1;        // BSG_FLASH_PLANES=0: flash_attn_split_kernel (K / V split while staged) instead of the pre-split attention
    if (fplanes < 0) {
      const char* e = getenv("BSG_FLASH_PLANES");
      fplanes = e ? atoi(e) : 1;
      if (fplanes && hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_planes_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, FLP_LDS) != hipSuccess) {
        (void)hipGetLastError();
        fplanes = 0;
      }
    }
    const int Tp = cdiv(T, 32) * 32;
    static int qkv_fused = -1;   // BSG_QKV_FUSED=0: fp32 QKV tensor + qkv_split_kernel instead of the QKV product's own plane output
    if (qkv_fused < 0) { const char* e = getenv("BSG_QKV_FUSED"); qkv_fused = e ? atoi(e) : 1; }
    for (const FftLayerW& L : layers) {
      TRY(ln_planes(x, L.ln1w, L.ln1b, ap, rows, 1e-5f, st));
      const long long wg4 = (long long)cdiv(T, 128) * B * heads;
      const bool use_planes = fplanes && Tp <= 3 * T && rows >= 32;
      // Q | K planes [2][rows][2H] in the (not yet written) FFN planes buffer, V^T planes [2][B][H][Tp] in the fp32 FFN buffer this path does not use
      _Float16* qk = reinterpret_cast<_Float16*>(fp);
      _Float16* vt = reinterpret_cast<_Float16*>(h->w_ffn);
      const long long vplane = (long long)B * H * Tp;
      unsigned* km = reinterpret_cast<unsigned*>(vt + 2 * vplane);   // key mask words [B][Tp / 32] behind the V^T planes
      if (use_planes && qkv_fused) {
        // the QKV product writes the attention's planes itself (H2wArgs::qkv_T): no fp32 QKV tensor, no split launch; the mask words once per stack
        if (&L == &layers.front()) {
          hipLaunchKernelGGL(key_mask_kernel, dim3(cdiv(B * (Tp / 32), 4)), dim3(256), 0, st, keep, km, B, T, Tp);
          BSG_LAUNCH_CHECK();
        }
        H2wArgs g{};
        g.act = ap; g.act_plane = rows * H; g.lda = H; g.wpack = L.p_in.pack; g.rows = (int)rows; g.K = H; g.Wn = 3 * H; g.taps = 1; g.act_is_a = 1;
        g.out = fp; g.out_plane = rows * 2 * H; g.ldo = 2 * H; g.alpha = qscale; g.alpha_ncols = H; g.act_fn = ACT_NONE; g.batch = 1;
        g.qkv_T = T; g.qkv_Tp = Tp; g.qkv_H = H; g.vt = h->w_ffn ? reinterpret_cast<unsigned short*>(h->w_ffn) : nullptr; g.vt_plane = vplane;
        TRY(launch_gemm_h2w(g, st));
      } else {
        TRY(linear_h2w(ap, L.p_in, 3 * H, nullptr, h->w_qkv, nullptr, rows, ACT_NONE, nullptr, nullptr, st, qscale, H));
      }
      if (use_planes) {
        if (!qkv_fused) {
          hipLaunchKernelGGL(qkv_split_kernel, dim3(Tp / 32, B), dim3(256), 0, st, (const float*)h->w_qkv, qk, rows * 2 * H, vt, vplane, T, Tp, keep, km, gemm_range_counter());
          BSG_LAUNCH_CHECK();
        }
        // 2 waves (64 queries) per workgroup at every size: the pipelined loop keeps two score tiles live and does not fit 4 waves x 2 workgroups.
        // Small batches: the keys of a query tile over KS workgroups (a single utterance is 32 workgroups, each a serial chain of 32 key blocks)
        static int ks_env = -1;   // BSG_FLASH_KS: 0 = auto, 1 = never split, 2 / 4 / 8 = that many splits whenever the sequence allows
        if (ks_env < 0) { const char* e = getenv("BSG_FLASH_KS"); ks_env = e ? atoi(e) : 0; }
        const int units = cdiv(T, 64) * B * heads, nb = Tp / 32;
        int ks = 1;
        if (ks_env == 0) { while (ks < 4 && units * ks * 2 <= 512 && nb / (ks * 2) >= 4) ks *= 2; }
        else { while (ks < ks_env && nb / (ks * 2) >= 1) ks *= 2; }
        if (ks > 1) {
          const size_t need = (size_t)units * ks * (2 * 64 * 66);
          if (need > h->cap_fsk || (size_t)units > h->cap_fcnt) {
            BSG_HIP(hipStreamSynchronize(st));
            if (h->w_fsk) (void)hipFree(h->w_fsk);
            if (h->w_fcnt) (void)hipFree(h->w_fcnt);
            h->w_fsk = nullptr; h->w_fcnt = nullptr; h->cap_fsk = h->cap_fcnt = 0;
            BSG_HIP(hipMalloc((void**)&h->w_fsk, need * sizeof(float)));
            BSG_HIP(hipMalloc((void**)&h->w_fcnt, (size_t)units * sizeof(unsigned)));
            BSG_HIP(hipMemsetAsync(h->w_fcnt, 0, (size_t)units * sizeof(unsigned), st));   // (the kernel leaves every counter at zero)
            h->cap_fsk = need; h->cap_fcnt = (size_t)units;
          }
        }
        hipLaunchKernelGGL(flash_attn_planes_kernel<2>, dim3(cdiv(T, 64), B * heads, ks), dim3(128), FLP_LDS, st, (const _Float16*)qk, rows * 2 * H, (const _Float16*)vt, vplane, (const unsigned*)km, T, Tp, heads, reinterpret_cast<_Float16*>(ap), rows * H, H, gemm_range_counter(), h->w_fsk, h->w_fcnt);
      } else if (wg4 >= 512) hipLaunchKernelGGL(flash_attn_split_kernel<4>, dim3(cdiv(T, 128), B * heads), dim3(256), 0, st, (const float*)h->w_qkv, keep, (float*)nullptr, T, heads, 3 * H, H, gemm_range_counter(), reinterpret_cast<_Float16*>(ap), rows * H);
      else hipLaunchKernelGGL(flash_attn_split_kernel<2>, dim3(cdiv(T, 64), B * heads), dim3(128), 0, st, (const float*)h->w_qkv, keep, (float*)nullptr, T, heads, 3 * H, H, gemm_range_counter(), reinterpret_cast<_Float16*>(ap), rows * H);
      BSG_LAUNCH_CHECK();
      TRY(linear_h2w(ap, L.p_out, H, nullptr, h->w_b, nullptr, rows, ACT_NONE, x, keep, st));   // x1 = (x + attn) * keep
      TRY(ln_planes(h->w_b, L.ln2w, L.ln2b, ap, rows, 1e-5f, st));
      {
        H2wArgs g{};   // Conv1d(H -> 4H, k, SAME) * k^-1/2 -> GELU, written as the planes the second Linear reads
        g.act = ap; g.act_plane = rows * H; g.lda = H; g.sAct = (long long)T * H; g.wpack = L.p_ffn1.pack; g.rows = T; g.K = H; g.Wn = 4 * H;
        g.taps = ksz; g.tap_shift0 = -(ksz / 2); g.act_is_a = 1; g.out = fp; g.out_plane = rows * 4 * H; g.ldo = 4 * H; g.sO = (long long)T * 4 * H;
        g.bias = L.ffn1b; g.alpha = (float)pow((double)ksz, -0.5); g.act_fn = ACT_GELU; g.batch = B;
        TRY(launch_gemm_h2w(g, st));
      }
      TRY(linear_h2w(fp, L.p_ffn2, H, L.ffn2b, x, nullptr, rows, ACT_NONE, h->w_b, keep, st));   // x = (x1 + ffn) * keep
    }
    TRY(ln(x, lnw, lnb, x, keep, rows, 1e-5f, st));
    return BSG_OK;
  }
  for (const FftLayerW& L : layers) {
    // --- self attention ---
    TRY(ln(x, L.ln1w, L.ln1b, h->w_a, nullptr, rows, 1e-5f, st));
    TRY(linear(h->w_a, L.in_proj, nullptr, h->w_qkv, rows, 3 * H, H, ACT_NONE, nullptr, nullptr, st, qscale, H));
    if (hd == 128 && !getenv("BSG_NO_FLASH_ATTN")) {
      // fused attention: no [B*heads, T, T] score tensor (flash_attn_kernel); 2 waves per workgroup when 4 would leave CUs idle
      const long long wg4 = (long long)cdiv(T, 128) * B * heads;
      static int fsplit = -1;   // BSG_FLASH_SPLIT=0: the fp32-MFMA form even while the GEMMs run split-fp16
      if (fsplit < 0) { const char* e = getenv("BSG_FLASH_SPLIT"); fsplit = e ? atoi(e) : 1; }
      if (fsplit && gemm_split_enabled()) {
        if (wg4 >= 512) hipLaunchKernelGGL(flash_attn_split_kernel<4>, dim3(cdiv(T, 128), B * heads), dim3(256), 0, st, (const float*)h->w_qkv, keep, h->w_a, T, heads, 3 * H, H, gemm_range_counter(), (_Float16*)nullptr, 0LL);
        else hipLaunchKernelGGL(flash_attn_split_kernel<2>, dim3(cdiv(T, 64), B * heads), dim3(128), 0, st, (const float*)h->w_qkv, keep, h->w_a, T, heads, 3 * H, H, gemm_range_counter(), (_Float16*)nullptr, 0LL);
      } else if (wg4 >= 512) hipLaunchKernelGGL(flash_attn_kernel<4>, dim3(cdiv(T, 128), B * heads), dim3(256), 0, st, (const float*)h->w_qkv, keep, h->w_a, T, heads, 3 * H, H);
      else hipLaunchKernelGGL(flash_attn_kernel<2>, dim3(cdiv(T, 64), B * heads), dim3(128), 0, st, (const float*)h->w_qkv, keep, h->w_a, T, heads, 3 * H, H);
      BSG_LAUNCH_CHECK();
    } else {
      TRY(ensure_scores(h, (size_t)B * heads * T * T, st));
      {
      GemmArgs g{};   // S[b,h] = Q K^T
      g.A = h->w_qkv; g.B = h->w_qkv + H; g.C = h->w_scores; g.M = T; g.N = T; g.K = hd; g.lda = 3 * H; g.ldb = 3 * H; g.ldc = T;
      g.trans_b = 1; g.taps = 1; g.alpha = 1.f; g.batch = B * heads; g.batch2 = heads;
      g.sA = (long long)T * 3 * H; g.sA2 = hd; g.sB = (long long)T * 3 * H; g.sB2 = hd;
      g.sC = (long long)heads * T * T; g.sC2 = (long long)T * T;
      TRY(launch_gemm(g, st));
      }
      hipLaunchKernelGGL(masked_softmax_kernel, dim3((unsigned)((long long)B * heads * T)), dim3(256), 0, st, h->w_scores, keep, T, T, heads);
      BSG_LAUNCH_CHECK();
      {
      GemmArgs g{};   // O[b,:,h] = P V
      g.A = h->w_scores; g.B = h->w_qkv + 2 * H; g.C = h->w_a; g.M = T; g.N = hd; g.K = T; g.lda = T; g.ldb = 3 * H; g.ldc = H;
      g.trans_b = 0; g.taps = 1; g.alpha = 1.f; g.batch = B * heads; g.batch2 = heads;
      g.sA = (long long)heads * T * T; g.sA2 = (long long)T * T; g.sB = (long long)T * 3 * H; g.sB2 = hd;
      g.sC = (long long)T * H; g.sC2 = hd;
      TRY(launch_gemm(g, st));
      }
    }
    TRY(linear(h->w_a, L.out_proj, nullptr, h->w_b, rows, H, H, ACT_NONE, x, keep, st));   // x1 = (x + attn) * keep
    // --- conv FFN ---
    TRY(ln(h->w_b, L.ln2w, L.ln2b, h->w_a, nullptr, rows, 1e-5f, st));
    {
      GemmArgs g{};   // Conv1d(H -> 4H, k, SAME) * k^-1/2 -> GELU
      g.A = h->w_a; g.B = L.ffn1; g.C = h->w_ffn; g.M = T; g.N = 4 * H; g.K = H; g.lda = H; g.ldb = H; g.ldc = 4 * H;
      g.trans_b = 1; g.taps = ksz; g.tap_shift0 = -(ksz / 2); g.sTapB = (long long)4 * H * H;
      g.bias_n = L.ffn1b; g.alpha = (float)pow((double)ksz, -0.5); g.act = ACT_GELU; g.batch = B;
      g.sA = (long long)T * H; g.sC = (long long)T * 4 * H;
      TRY(launch_gemm(g, st));
    }
    TRY(linear(h->w_ffn, L.ffn2, L.ffn2b, x, rows, H, 4 * H, ACT_NONE, h->w_b, keep, st));   // x = (x1 + ffn) * keep
  }
  TRY(ln(x, lnw, lnb, x, keep, rows, 1e-5f, st));
  return BSG_OK;
}

// Token-level front for the batch rows [row0, row0 + nb) of a batch of B utterances.  Only the ESM couples rows (it attends over the BATCH
// axis, common_layers.py:853), and only through K / V = projections of LN(lang_embed[lang]) of every row (:850-853): the other rows
// contribute their `lang` ints and nothing else.  So K / V are projected for all B rows, and everything else — Q, the attention's queries,
// the ESM's FFN, the embedding sum, the 4-layer FFT encoder, the duration predictor — runs on the nb rows asked for.  row0 = 0, nb = B is
// the whole batch (bsg_fs2midi_encode); a rank of a sharded run asks for its own rows (bsg_fs2midi_encode_rows): the same arithmetic
// on an eighth of the tokens at configs[3].
static int encode_impl(bsg_fs2midi* h, const int64_t* txt, const int64_t* pitch_midi, const float* midi_dur, const int64_t* is_slur,
                       const int64_t* lang, const int64_t* spk_id, int32_t B, int32_t Tt, int32_t row0, int32_t nb, float* enc_out,
                       float* dur_xs, int64_t* dur, hipStream_t st) {
  const long long rows = (long long)B * Tt;        // every utterance: the lang embedding and K / V
  const long long lrows = (long long)nb * Tt;      // the rows asked for
  const long long off = (long long)row0 * Tt;
  TRY(ensure_ws(h, (size_t)rows, (size_t)nb * h->cfg.num_heads * Tt * Tt, st));
  h->last_token_rows = (int)lrows;
  const dim3 rg(cdiv(rows, 4)), lg(cdiv(lrows, 4)), rb(256);
  const float sq = sqrtf((float)H);
  float* x0 = h->w_x;      // sqrt(H) * tok                [rows][H]
  float* lange = h->w_b;   // lang embedding LP            [rows][H]
  static int esm_env = -1;   // BSG_ESM_H2W=0: the ESM's Linear layers on gemm_split_kernel and the thread-per-query attention
  if (esm_env < 0) { const char* e = getenv("BSG_ESM_H2W"); esm_env = e ? atoi(e) : 1; }
  static int esm_gemm_env = -1;
  if (esm_gemm_env < 0) { const char* e = getenv("BSG_GEMM_H2W"); esm_gemm_env = e ? atoi(e) : 1; }
  const bool esm_h2w = esm_env && esm_gemm_env && h->h2w_ok && gemm_split_enabled() && B <= 64 && h2w_supports((int)rows, H, H, 1, H) &&
                       h2w_supports((int)lrows, H, H, 1, H) && rows * 4 * H * 2 < (1LL << 31);
  if (esm_h2w) {
    // ---- ESM (common_layers.py:848-860) on the pre-split GEMM: every operand is written as hi / lo planes by its producer; the K and V
    // projections (both of LN(lang)) are ONE product of 512 weight rows; the attention runs one wave per (position, head)
    unsigned short* ap = h->w_ap;   // [2][rows][H]
    unsigned short* fp = h->w_fp;   // x0 planes, later the FFN's hidden planes ([2][rows][H] of its [2][rows][4H])
    hipLaunchKernelGGL(embed_tokens_kernel, rg, rb, 0, st, (const long long*)txt, (const long long*)lang, h->Etok, h->Elang, x0, lange, rows, sq,
                       reinterpret_cast<_Float16*>(fp), reinterpret_cast<_Float16*>(fp) + rows * H, gemm_range_counter());
    BSG_LAUNCH_CHECK();
    TRY(ln_planes(lange, h->esm_ln1w, h->esm_ln1b, ap, rows, 1e-5f, st));
    float* q = h->w_qkv;                // [lrows][H]
    float* kvp = h->w_qkv + rows * H;   // [rows][2H]: K | V
    TRY(linear_h2w(fp + off * H, h->p_esm_q, H, h->esm_in_b, q, nullptr, lrows, ACT_NONE, nullptr, nullptr, st, 1.f, 0, rows * H));
    TRY(linear_h2w(ap, h->p_esm_kv, 2 * H, h->esm_in_b + H, kvp, nullptr, rows, ACT_NONE, nullptr, nullptr, st));
    hipLaunchKernelGGL(esm_attention_wave_kernel, dim3(cdiv(Tt * 8, 4)), dim3(256), 0, st, (const float*)q, (const float*)kvp, (const float*)(kvp + H), H,
                       2 * H, (float*)nullptr, reinterpret_cast<_Float16*>(ap), lrows * H, B, nb, Tt, 8, (float)sqrt(1.0 / 32.0), gemm_range_counter());
    BSG_LAUNCH_CHECK();
    float* Mo = h->w_c;
    TRY(linear_h2w(ap, h->p_esm_out, H, h->esm_out_b, Mo, nullptr, lrows, ACT_NONE, lange + off * H, nullptr, st));   // Mo = out_proj + LP
    TRY(ln_planes(Mo, h->esm_ln2w, h->esm_ln2b, ap, lrows, 1e-5f, st));
    TRY(linear_h2w(ap, h->p_esm_f0, H, h->esm_f0b, nullptr, fp, lrows, ACT_RELU, nullptr, nullptr, st));
    TRY(linear_h2w(fp, h->p_esm_f2, H, h->esm_f2b, h->w_a, nullptr, lrows, ACT_NONE, Mo, nullptr, st));          // Fo = ffn + Mo
  } else {
  hipLaunchKernelGGL(embed_tokens_kernel, rg, rb, 0, st, (const long long*)txt, (const long long*)lang, h->Etok, h->Elang, x0, lange, rows, sq,
                     (_Float16*)nullptr, (_Float16*)nullptr, (unsigned*)nullptr);
  BSG_LAUNCH_CHECK();
  // ---- ESM (common_layers.py:848-860): attention over the batch axis
  float* lpn = h->w_a;
  TRY(ln(lange, h->esm_ln1w, h->esm_ln1b, lpn, nullptr, rows, 1e-5f, st));
  float* q = h->w_qkv;                     // [lrows][H]
  float* k = h->w_qkv + rows * H;          // [rows][H]
  float* v = h->w_qkv + 2 * rows * H;
  TRY(linear(x0 + off * H, h->esm_in_w, h->esm_in_b, q, lrows, H, H, ACT_NONE, nullptr, nullptr, st));
  TRY(linear(lpn, h->esm_in_w + (size_t)H * H, h->esm_in_b + H, k, rows, H, H, ACT_NONE, nullptr, nullptr, st));
  TRY(linear(lpn, h->esm_in_w + (size_t)2 * H * H, h->esm_in_b + 2 * H, v, rows, H, H, ACT_NONE, nullptr, nullptr, st));
  float* att = h->w_a;   // lpn is dead once k, v exist
  {
    const long long total = lrows * 8;
    hipLaunchKernelGGL(esm_attention_kernel<32>, dim3(cdiv(total, 128)), dim3(128), 0, st, (const float*)q, (const float*)k,
                       (const float*)v, att, B, nb, Tt, 8, (float)sqrt(1.0 / 32.0));
    BSG_LAUNCH_CHECK();
  }
  float* Mo = h->w_c;
  TRY(linear(att, h->esm_out_w, h->esm_out_b, Mo, lrows, H, H, ACT_NONE, lange + off * H, nullptr, st));       // Mo = out_proj + LP
  float* t1 = h->w_a;
  float* t2 = h->w_qkv;   // (q, k, v are dead; w_b still holds the lang embedding of every row)
  TRY(ln(Mo, h->esm_ln2w, h->esm_ln2b, t1, nullptr, lrows, 1e-5f, st));
  TRY(linear(t1, h->esm_f0w, h->esm_f0b, t2, lrows, H, H, ACT_RELU, nullptr, nullptr, st));
  TRY(linear(t2, h->esm_f2w, h->esm_f2b, h->w_a, lrows, H, H, ACT_NONE, Mo, nullptr, st));      // Fo = ffn + Mo
  }
  // ---- sum of embeddings, *sqrt(H) + reversed positional table, mask (the rows asked for; row0 * Tt is a multiple of Tt: the position of a
  // row inside its utterance is unchanged)
  float* x = h->w_c;
  hipLaunchKernelGGL(embed_finish_kernel, lg, rb, 0, st, (const float*)(x0 + off * H), (const float*)h->w_a, (const long long*)txt + off,
                     (const long long*)pitch_midi + off, midi_dur + off, (const long long*)is_slur + off, h->Emidi, h->Wdur, h->bdur, h->Eslur,
                     h->rel_table, x, h->w_keep, lrows, Tt, sq);
  BSG_LAUNCH_CHECK();
  TRY(fft_stack(h, h->enc, h->enc_lnw, h->enc_lnb, h->cfg.enc_ffn_kernel_size, x, h->w_keep, nb, Tt, st));
  BSG_HIP(hipMemcpyAsync(enc_out, x, lrows * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (dur) {
    // duration predictor (tts_modules.py:108-133) on (enc + spk) * keep
    float* a = h->w_a;
    float* b = h->w_b;
    hipLaunchKernelGGL(add_spk_kernel, lg, rb, 0, st, (const float*)x, (const long long*)spk_id + row0, h->Espk, h->w_keep, a, lrows, Tt);
    BSG_LAUNCH_CHECK();
    const int ks = h->cfg.dur_kernel;
    for (int l = 0; l < h->cfg.dur_layers; ++l) {
      GemmArgs g{};
      g.A = a; g.B = h->dur_conv[l]; g.C = b; g.M = Tt; g.N = H; g.K = H; g.lda = H; g.ldb = H; g.ldc = H; g.trans_b = 1;
      g.taps = ks; g.tap_shift0 = -(ks / 2); g.sTapB = (long long)H * H; g.bias_n = h->dur_convb[l]; g.alpha = 1.f;
      g.act = ACT_RELU; g.batch = nb; g.sA = (long long)Tt * H; g.sC = (long long)Tt * H;
      TRY(launch_gemm(g, st));
      TRY(ln(b, h->dur_lnw[l], h->dur_lnb[l], a, h->w_keep, lrows, 1e-12f, st));
    }
    TRY(linear(a, h->dur_lin_w, h->dur_lin_b, dur_xs, lrows, 1, H, ACT_NONE, nullptr, h->w_keep, st));
    hipLaunchKernelGGL(dur_from_log_kernel, dim3(cdiv(lrows, 256)), dim3(256), 0, st, (const float*)dur_xs, (long long*)dur, lrows);
    BSG_LAUNCH_CHECK();
  }
  return BSG_OK;
}

extern "C" int bsg_fs2midi_encode(bsg_fs2midi* h, const int64_t* txt, const int64_t* pitch_midi, const float* midi_dur,
                                  const int64_t* is_slur, const int64_t* lang, const int64_t* spk_id, int32_t B, int32_t Tt,
                                  float* enc_out, float* dur_xs, int64_t* dur, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  BSG_REQUIRE(h && txt && pitch_midi && midi_dur && is_slur && lang && spk_id && enc_out, "fs2midi_encode: null argument");
  BSG_REQUIRE(B > 0 && Tt > 0 && Tt <= h->cfg.n_rel, "fs2midi_encode: B=%d T_txt=%d (rel-pos table has %d rows)", B, Tt, h->cfg.n_rel);
  BSG_REQUIRE((dur_xs == nullptr) == (dur == nullptr), "fs2midi_encode: dur_xs and dur go together");
  return encode_impl(h, txt, pitch_midi, midi_dur, is_slur, lang, spk_id, B, Tt, 0, B, enc_out, dur_xs, dur, (hipStream_t)stream);
}

extern "C" int bsg_fs2midi_encode_rows(bsg_fs2midi* h, const int64_t* txt, const int64_t* pitch_midi, const float* midi_dur,
                                       const int64_t* is_slur, const int64_t* lang, const int64_t* spk_id, int32_t B, int32_t Tt,
                                       int32_t row0, int32_t n_rows, float* enc_out, float* dur_xs, int64_t* dur, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  BSG_REQUIRE(h && txt && pitch_midi && midi_dur && is_slur && lang && spk_id && enc_out, "fs2midi_encode_rows: null argument");
  BSG_REQUIRE(B > 0 && Tt > 0 && Tt <= h->cfg.n_rel, "fs2midi_encode_rows: B=%d T_txt=%d (rel-pos table has %d rows)", B, Tt, h->cfg.n_rel);
  BSG_REQUIRE(row0 >= 0 && n_rows > 0 && row0 + n_rows <= B, "fs2midi_encode_rows: rows [%d, %d) of a batch of %d", row0, row0 + n_rows, B);
  BSG_REQUIRE((dur_xs == nullptr) == (dur == nullptr), "fs2midi_encode_rows: dur_xs and dur go together");
  return encode_impl(h, txt, pitch_midi, midi_dur, is_slur, lang, spk_id, B, Tt, row0, n_rows, enc_out, dur_xs, dur, (hipStream_t)stream);
}

extern "C" int bsg_fs2midi_last_rows(const bsg_fs2midi* h, int32_t* token_rows, int32_t* stack_rows) {
  BSG_REQUIRE(h && token_rows && stack_rows, "fs2midi_last_rows: null argument");
  *token_rows = h->last_token_rows;
  *stack_rows = h->last_stack_rows;
  return BSG_OK;
}

extern "C" int bsg_length_regulator(const int64_t* dur, const int64_t* txt, int64_t* mel2ph, int32_t B, int32_t Tt, int32_t T,
                                    void* stream) {
  BSG_REQUIRE(dur && mel2ph && B > 0 && Tt > 0 && T > 0 && Tt <= 8192, "length_regulator: bad argument (T_txt <= 8192)");
  hipLaunchKernelGGL(length_regulator_kernel, dim3(B), dim3(256), Tt * sizeof(long long), (hipStream_t)stream,
                     (const long long*)dur, (const long long*)txt, (long long*)mel2ph, Tt, T);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" int bsg_fs2midi_decode(bsg_fs2midi* h, const float* enc_out, const int64_t* mel2ph, const int64_t* spk_id,
                                  const int64_t* speechsing, int32_t B, int32_t Tt, int32_t T, float* decoder_inp,
                                  float* mel_out, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  BSG_REQUIRE(h && enc_out && mel2ph && spk_id && speechsing && decoder_inp, "fs2midi_decode: null argument");
  BSG_REQUIRE(B > 0 && Tt > 0 && T > 0 && T < h->cfg.n_pos, "fs2midi_decode: B=%d T_txt=%d T=%d (position table has %d rows)", B, Tt, T, h->cfg.n_pos);
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)B * T;
  TRY(ensure_ws(h, (size_t)rows, mel_out ? (size_t)B * h->cfg.num_heads * T * T : 0, st));
  const dim3 rg(cdiv(rows, 4)), rb(256);
  hipLaunchKernelGGL(gather_frames_kernel, rg, rb, 0, st, enc_out, (const long long*)mel2ph, (const long long*)spk_id,
                     (const long long*)speechsing, h->Espk, h->Estyle, decoder_inp, rows, T, Tt);
  BSG_LAUNCH_CHECK();
  if (!mel_out) return BSG_OK;   // skip_decoder
  float* x = h->w_x;
  BSG_HIP(hipMemcpyAsync(x, decoder_inp, rows * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(decoder_positions_kernel, dim3(B), dim3(64), 0, st, (const float*)x, h->w_pos, h->w_keep, T);
  BSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(decoder_entry_kernel, rg, rb, 0, st, x, (const int*)h->w_pos, h->dec_table, h->dec_alpha, h->w_keep, rows, h->cfg.n_pos);
  BSG_LAUNCH_CHECK();
  TRY(fft_stack(h, h->dec, h->dec_lnw, h->dec_lnb, h->cfg.dec_ffn_kernel_size, x, h->w_keep, B, T, st));
  // mel_out = Linear(H -> M)(x) * (mel2ph > 0)                                        (fastspeech/fs2.py:236-240)
  TRY(linear(x, h->mel_w, h->mel_b, mel_out, rows, h->cfg.out_dims, H, ACT_NONE, nullptr, nullptr, st));
  // ... * tgt_nonpadding, which comes from mel2ph (not from the decoder's own |x| test)
  hipLaunchKernelGGL(mask_rows_by_index_kernel, dim3(cdiv(rows * h->cfg.out_dims, 256)), dim3(256), 0, st, mel_out,
                     (const long long*)mel2ph, rows, h->cfg.out_dims);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}


// ================================================================================================
// the `FFT` candidate denoiser (DIFF_DECODERS['fft'], usr/diff/candidate_decoder.py:39-100):
// input projection -> concat[x, cond, step embedding] -> Linear(3C -> H) -> FastspeechDecoder stack -> Linear(H -> M).
// The concat-Linear is split by columns: the cond part is step-invariant (hoisted to prepare), the step part is one
// vector per utterance, so per call only x * W_x^T runs over all frames.  Reuses fft_stack() of the FS2 decoder.
// ================================================================================================
namespace bsg {
namespace {
// in [B][R][Cc] -> out [B][Cc][R]
__global__ void transpose_brc_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < Cc) ? in[((long long)b * R + r) * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < Cc && r < R) out[((long long)b * Cc + c) * R + r] = tile[tx][j];
  }
}
__global__ void gather_rows_kernel(const float* __restrict__ table, const long long* __restrict__ idx, float* __restrict__ out, int n,
                                   int width, int n_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * width) return;
  long long r = idx[i / width];
  r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
  out[i] = table[r * width + (i % width)];
}
}  // namespace
}  // namespace bsg

struct bsg_fftden {
  bsg_fs2midi* core = nullptr;   // owns weights + the FFT-stack workspaces
  int M = 0, S = 0, n_pos = 0, ksz = 9;
  std::vector<FftLayerW> layers;
  float *alpha, *lnw, *lnb, *in_w, *in_b, *mel_w, *mel_b, *gdi_w, *gdi_b, *dtab, *table;
  size_t cap = 0;
  int B = 0, T = 0;
  float *condpart = nullptr, *xT = nullptr, *xp = nullptr, *te = nullptr, *tvec = nullptr, *mel = nullptr;
};

extern "C" void bsg_fftden_destroy(bsg_fftden* h) {
  if (!h) return;
  float* ws[] = {h->condpart, h->xT, h->xp, h->te, h->tvec, h->mel};
  for (float* p : ws)
    if (p) (void)hipFree(p);
  for (FftLayerW& L : h->layers) { h2w_free(&L.p_in); h2w_free(&L.p_out); h2w_free(&L.p_ffn1); h2w_free(&L.p_ffn2); }
  bsg_fs2midi_destroy(h->core);
  delete h;
}

extern "C" int bsg_fftden_n_weights(int32_t n_layers) { return 2 + 10 * n_layers + 2 + 2 + 4 + 2 + 2; }

extern "C" int bsg_fftden_create(bsg_fftden** out, int32_t in_dims, int32_t n_layers, int32_t num_heads, int32_t ffn_kernel,
                                 int32_t max_steps, int32_t n_pos, const void* const* w, int32_t n_weights, const float* step_table,
                                 const float* pos_table, void* stream) {
  BSG_REQUIRE(out && w && step_table && pos_table, "fftden_create: null argument");
  BSG_REQUIRE(in_dims > 0 && in_dims % 4 == 0 && n_layers > 0 && num_heads > 0 && H % num_heads == 0 && (H / num_heads) % 4 == 0 &&
                  ffn_kernel % 2 == 1 && max_steps > 0 && n_pos > 1, "fftden_create: bad config");
  BSG_REQUIRE(n_weights == bsg_fftden_n_weights(n_layers), "fftden_create: expected %d weight tensors, got %d", bsg_fftden_n_weights(n_layers), n_weights);
  for (int i = 0; i < n_weights; ++i) BSG_REQUIRE(w[i] != nullptr, "fftden_create: weight %d is null", i);
  hipStream_t st = (hipStream_t)stream;
  bsg_fftden* h = new bsg_fftden();
  h->core = new bsg_fs2midi();
  h->core->cfg = bsg_fs2midi_cfg{};
  h->core->cfg.num_heads = num_heads;
  h->M = in_dims; h->S = max_steps; h->n_pos = n_pos; h->ksz = ffn_kernel;
  bsg_fs2midi* c = h->core;
  auto fail = [&](int rc) { bsg_fftden_destroy(h); return rc; };
  int rc, i = 0;
  if ((rc = guard_init(&c->guard, st)) != BSG_OK) return fail(rc);
  GuardScope guard_scope(&c->guard);
  // FFT.state_dict(): pos_embed_alpha, embed_positions._float_tensor, layers.*, layer_norm.{w,b}, input_projection.{w,b},
  // mlp.0.{w,b}, mlp.2.{w,b}, get_mel_out.{w,b}, get_decode_inp.{w,b}
  if ((rc = fs2_copy(c, &h->alpha, w[i++], 1, st)) != BSG_OK) return fail(rc);
  i++;
  if ((rc = load_fft_layers(c, h->layers, w + i, n_layers, ffn_kernel, st)) != BSG_OK) return fail(rc);
  i += 10 * n_layers;
  if ((rc = fs2_copy(c, &h->lnw, w[i++], H, st)) != BSG_OK) return fail(rc);
  if ((rc = fs2_copy(c, &h->lnb, w[i++], H, st)) != BSG_OK) return fail(rc);
  if ((rc = fs2_copy(c, &h->in_w, w[i++], (size_t)H * in_dims, st)) != BSG_OK) return fail(rc);
  if ((rc = fs2_copy(c, &h->in_b, w[i++], H, st)) != BSG_OK) return fail(rc);
  const float *m0w = (const float*)w[i], *m0b = (const float*)w[i + 1], *m2w = (const float*)w[i + 2], *m2b = (const float*)w[i + 3];
  i += 4;
  if ((rc = fs2_copy(c, &h->mel_w, w[i++], (size_t)in_dims * H, st)) != BSG_OK) return fail(rc);
  if ((rc = fs2_copy(c, &h->mel_b, w[i++], in_dims, st)) != BSG_OK) return fail(rc);
  if ((rc = fs2_copy(c, &h->gdi_w, w[i++], (size_t)H * 3 * H, st)) != BSG_OK) return fail(rc);
  if ((rc = fs2_copy(c, &h->gdi_b, w[i++], H, st)) != BSG_OK) return fail(rc);
  if ((rc = fs2_copy(c, &h->table, pos_table, (size_t)n_pos * H, st)) != BSG_OK) return fail(rc);
  // step-embedding MLP tabulated for every timestep (candidate_decoder.py:60-62)
  float* hid = nullptr;
  if ((rc = fs2_alloc(c, &hid, (size_t)max_steps * 4 * H)) != BSG_OK) return fail(rc);
  if ((rc = fs2_alloc(c, &h->dtab, (size_t)max_steps * H)) != BSG_OK) return fail(rc);
  if ((rc = linear(step_table, m0w, m0b, hid, max_steps, 4 * H, H, ACT_MISH, nullptr, nullptr, st)) != BSG_OK) return fail(rc);
  if ((rc = linear(hid, m2w, m2b, h->dtab, max_steps, H, 4 * H, ACT_NONE, nullptr, nullptr, st)) != BSG_OK) return fail(rc);
  unsigned pack_bad = 1;
  if (hipMemcpyAsync(&pack_bad, c->pack_bad, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) { set_error("fftden_create: sync failed"); return fail(BSG_EHIP); }
  c->h2w_ok = pack_bad == 0 && ffn_kernel <= 17;
  *out = h;
  return BSG_OK;
}

extern "C" int bsg_fftden_prepare(bsg_fftden* h, const float* cond, int32_t B, int32_t T, void* stream) {
  GuardScope guard_scope(h && h->core ? &h->core->guard : nullptr);
  BSG_REQUIRE(h && cond && B > 0 && T > 0 && T < h->n_pos, "fftden_prepare: bad argument (T=%d)", T);
  hipStream_t st = (hipStream_t)stream;
  const size_t rows = (size_t)B * T;
  if (rows > h->cap) {
    BSG_HIP(hipStreamSynchronize(st));
    float** bufs[] = {&h->condpart, &h->xT, &h->xp, &h->te, &h->tvec, &h->mel};
    for (float** p : bufs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    h->cap = 0;
    BSG_HIP(hipMalloc((void**)&h->condpart, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->xT, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->xp, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->te, rows * H * sizeof(float)));      // [B][H] used; rows >= B for any later (B,T) that fits
    BSG_HIP(hipMalloc((void**)&h->tvec, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->mel, rows * h->M * sizeof(float)));
    h->cap = rows;
  }
  TRY(ensure_ws(h->core, rows, (size_t)B * h->core->cfg.num_heads * T * T, st));
  h->B = B; h->T = T;
  // cond [B][H][T] -> [B*T][H]; condpart = cond_t * W[:, C:2C]^T                           (candidate_decoder.py:63-70)
  hipLaunchKernelGGL(transpose_brc_kernel, dim3(cdiv(T, 32), cdiv(H, 32), B), dim3(256), 0, st, cond, h->xT, H, T);
  BSG_LAUNCH_CHECK();
  GemmArgs g{};
  g.A = h->xT; g.B = h->gdi_w + H; g.C = h->condpart; g.M = (int)rows; g.N = H; g.K = H; g.lda = H; g.ldb = 3 * H; g.ldc = H;
  g.trans_b = 1; g.taps = 1; g.alpha = 1.f; g.batch = 1;
  return launch_gemm(g, st);
}

extern "C" int bsg_fftden_forward(bsg_fftden* h, const float* x, const int64_t* t, float* eps, int32_t B, int32_t T, void* stream) {
  GuardScope guard_scope(h && h->core ? &h->core->guard : nullptr);
  BSG_REQUIRE(h && x && t && eps, "fftden_forward: null argument");
  BSG_REQUIRE(h->cap > 0 && h->B == B && h->T == T, "fftden_forward: (B=%d,T=%d) does not match bsg_fftden_prepare (B=%d,T=%d)", B, T, h->B, h->T);
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)B * T;
  bsg_fs2midi* c = h->core;
  // x [B][M][T] -> [B*T][M]; xp = input_projection                                        (:57-58)
  hipLaunchKernelGGL(transpose_brc_kernel, dim3(cdiv(T, 32), cdiv(h->M, 32), B), dim3(256), 0, st, x, h->xT, h->M, T);
  BSG_LAUNCH_CHECK();
  TRY(linear(h->xT, h->in_w, h->in_b, h->xp, rows, H, h->M, ACT_NONE, nullptr, nullptr, st));
  // step part: tvec[b] = mlp(emb(t_b)) * W[:, 2C:3C]^T + bias                               (:59-66)
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(B * H, 256)), dim3(256), 0, st, (const float*)h->dtab, (const long long*)t, h->te, B, H, h->S);
  BSG_LAUNCH_CHECK();
  {
    GemmArgs g{};
    g.A = h->te; g.B = h->gdi_w + 2 * H; g.C = h->tvec; g.M = B; g.N = H; g.K = H; g.lda = H; g.ldb = 3 * H; g.ldc = H; g.trans_b = 1;
    g.taps = 1; g.bias_n = h->gdi_b; g.alpha = 1.f; g.batch = 1;
    TRY(launch_gemm(g, st));
  }
  float* xs = c->w_x;
  {
    GemmArgs g{};   // decoder_inp = xp W_x^T + condpart + tvec[b]
    g.A = h->xp; g.B = h->gdi_w; g.C = xs; g.M = T; g.N = H; g.K = H; g.lda = H; g.ldb = 3 * H; g.ldc = H; g.trans_b = 1; g.taps = 1;
    g.bias_n = h->tvec; g.sBiasN = H; g.alpha = 1.f; g.R = h->condpart; g.ldr = H; g.sR = (long long)T * H; g.batch = B;
    g.sA = (long long)T * H; g.sC = (long long)T * H;
    TRY(launch_gemm(g, st));
  }
  const dim3 rg(cdiv(rows, 4)), rb(256);
  hipLaunchKernelGGL(decoder_positions_kernel, dim3(B), dim3(64), 0, st, (const float*)xs, c->w_pos, c->w_keep, T);
  BSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(decoder_entry_kernel, rg, rb, 0, st, xs, (const int*)c->w_pos, h->table, h->alpha, c->w_keep, rows, h->n_pos);
  BSG_LAUNCH_CHECK();
  TRY(fft_stack(c, h->layers, h->lnw, h->lnb, h->ksz, xs, c->w_keep, B, T, st));
  TRY(linear(xs, h->mel_w, h->mel_b, h->mel, rows, h->M, H, ACT_NONE, nullptr, nullptr, st));       // get_mel_out (:98)
  hipLaunchKernelGGL(transpose_brc_kernel, dim3(cdiv(h->M, 32), cdiv(T, 32), B), dim3(256), 0, st, (const float*)h->mel, eps, T, h->M);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

namespace bsg {
Guard* guard_of_fs2midi(void* h) { return &static_cast<bsg_fs2midi*>(h)->guard; }
Guard* guard_of_fftden(void* h) { return &static_cast<bsg_fftden*>(h)->core->guard; }
}
