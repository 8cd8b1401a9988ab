// Sampler arithmetic and the arguments of the fused step tail, shared by the fp32 tail (diffnet.hip) and the bf16-operand tail
// (diffnet_bf16.hip).
#pragma once
#include "diffnet_res.h"

namespace bsg {

struct Philox {
  unsigned c[4];
};
__device__ __forceinline__ Philox philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = (unsigned long long)c0 * 0xD2511F53ull;
    const unsigned long long p1 = (unsigned long long)c2 * 0xCD9E8D57ull;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox{{c0, c1, c2, c3}};
}
// element i of stream `stream` = lane i%4 of counter (i/4, stream, 0, 0); Box-Muller on (0,1),(2,3)
// — the layout bisinger_amd/synth.py:philox_normal reproduces on the host.
__device__ __forceinline__ f32x4 philox_normal4(unsigned long long seed, unsigned stream, unsigned long long quad) {
  const Philox r = philox4x32_10((unsigned)quad, stream, 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32));
  float u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = fminf(((float)r.c[i] + 1.0f) * 2.3283064365386963e-10f, 1.0f);
  const float r0 = sqrtf(-2.0f * logf(u[0])), r1 = sqrtf(-2.0f * logf(u[2]));
  const float a0 = 6.283185307179586f * u[1], a1 = 6.283185307179586f * u[3];
  return f32x4{r0 * cosf(a0), r0 * sinf(a0), r1 * cosf(a1), r1 * sinf(a1)};
}

struct StepCoef {
  float recip, recipm1, pc1, pc2, sigma;
};

// p_sample_plms coefficients of one step (shallow_diffusion_tts.py:168-201)
struct PlmsCoef {
  float a_t, a_prev;
  float w0, w1, w2, w3, inv;  // eps' = (w0*e0 + w1*e1 + w2*e2 + w3*e3) / inv
};

// p_sample_plms update of one element, shared by plms_step_kernel and the fused tail (same rounding sequence in both)
__device__ __forceinline__ float plms_update(float x, float e0, float e1, float e2, float e3, int n_hist, const PlmsCoef& k, float* ep_out) {
  float ep = e0;
  if (n_hist > 0) {   // multistep blends, evaluated left to right like the reference expressions
    if (n_hist >= 3) ep = __fsub_rn(__fadd_rn(__fsub_rn(__fmul_rn(k.w0, e0), __fmul_rn(-k.w1, e1)), __fmul_rn(k.w2, e2)), __fmul_rn(-k.w3, e3));
    else if (n_hist == 2) ep = __fadd_rn(__fsub_rn(__fmul_rn(k.w0, e0), __fmul_rn(-k.w1, e1)), __fmul_rn(k.w2, e2));
    else if (k.w0 == 1.0f) ep = __fadd_rn(e0, e1);                       // (eps + eps_prev) / 2
    else ep = __fsub_rn(__fmul_rn(k.w0, e0), e1);                        // (3*eps - h[-1]) / 2
    ep = ep / k.inv;
  }
  const float a_t = k.a_t, a_prev = k.a_prev;
  const float a_t_sq = sqrtf(a_t), a_prev_sq = sqrtf(a_prev);
  const float cx = 1.0f / __fmul_rn(a_t_sq, __fadd_rn(a_t_sq, a_prev_sq));
  const float ce = 1.0f / __fmul_rn(a_t_sq, __fadd_rn(sqrtf(__fmul_rn(__fsub_rn(1.0f, a_prev), a_t)),
                                                       sqrtf(__fmul_rn(__fsub_rn(1.0f, a_t), a_prev))));
  const float xd = __fmul_rn(__fsub_rn(a_prev, a_t), __fsub_rn(__fmul_rn(cx, x), __fmul_rn(ce, ep)));
  if (ep_out) *ep_out = ep;
  return __fadd_rn(x, xd);
}

struct TailArgs {
  const float* skip;    // [B][C][T]
  const unsigned short* skip_h;  // bf16 mode: [B][C/4][T][4] instead of `skip`
  float* x;             // [B][M][T] in/out
  const float* noise;   // [B][M][T] or null (Philox)
  float* xa_next;       // [B][C][T]
  const float* ws_pack; // [8][32][64][4]
  const float* wo_pack; // [3][32][64][4]  (rows >= M are zero)
  const float* wi_pack; // [8][MP/8][64][4]
  const float* b_skip;  // [C]
  const float* b_fin;   // [96] (zero padded)
  const float* b_in;    // [C]
  // bf16-operand tail (step_tail_bf16_kernel): A fragments of v_mfma_f32_32x32x16_bf16, k-step major (pack_a_frag_bf16)
  const unsigned short* ws_h;  // skip projection   [16 k-steps][8 row tiles][64][8]
  const unsigned short* wo_h;  // output projection [16][3][64][8]   (rows >= M zero)
  const unsigned short* wi_h;  // input projection  [6][8][64][8]    (K = in_dims padded to 96 with zero columns)
  // split-fp16 tail (fused into residual_stack_h2_kernel, diffnet_h2.hip): hi / lo fp16 fragments [ks][plane][row tiles][64][8]
  const unsigned short* ws_s;  // skip projection   [16][2][8][64][8]
  const unsigned short* wo_s;  // output projection [16][2][3][64][8]   (rows >= M zero)
  const unsigned short* wi_s;  // input projection  [6][2][8][64][8]    (K padded to 96)
  const float* tail_scale;     // [3][2]: power-of-two scale and its reciprocal of the three projections
  unsigned* status;            // step_tail_h2_kernel: the launch status words (word 1 += range events), or null
  StepCoef k;
  unsigned long long seed, quad_row0;   // Philox: key, and the flat element index of this shard's row 0
  unsigned stream;
  int B, T, M, tiles_per_row, do_head;
  // PLMS form (plms_hist > 0): eps is stored to e_new and x <- p_sample_plms(x, eps, history)   (shallow_diffusion_tts.py:168-201)
  int plms_hist;          // 0: DDPM ancestral update; 1..3: number of history entries blended
  PlmsCoef pk;
  float* e_new;           // [B][M][T]
  const float* h1;        // newest history entry, then older
  const float* h2;
  const float* h3;
};

__device__ __forceinline__ float philox_normal1(unsigned long long seed, unsigned stream, unsigned long long idx) {
  const f32x4 z = philox_normal4(seed, stream, idx >> 2);
  const int s = (int)(idx & 3);
  return s == 0 ? z[0] : s == 1 ? z[1] : s == 2 ? z[2] : z[3];
}


// split-fp16 stack launch (diffnet_h2.hip), optionally with the step tail in the same launch
int launch_residual_stack_h2(const StackArgs& p, const TailArgs* tail, hipStream_t st, int nct);
// the same launch on 16-row matrix tiles (diffnet_h2q.hip): p.apack1q / p.apack2q hold its weight fragments
int launch_residual_stack_h2q(const StackArgs& p, const TailArgs* tail, hipStream_t st, int nct);
// the step tail behind a pair / quad launch, on the 16-bit matrix pipe (diffnet_h2.hip step_tail_h2_kernel)
int launch_step_tail_h2(const TailArgs& a, hipStream_t st);
int h2_tail_pack(const float* ws, const float* wo96, const float* wi96, unsigned short* out_ws, unsigned short* out_wo, unsigned short* out_wi,
                 unsigned* maxbits, float* tab, hipStream_t st);

// bf16-operand form of the fused step tail (diffnet_bf16.hip): 64-frame tiles, skip sum read as bf16 channel quads (a.skip_h)
int launch_step_tail_bf16(const TailArgs& a, hipStream_t st);

}  // namespace bsg
