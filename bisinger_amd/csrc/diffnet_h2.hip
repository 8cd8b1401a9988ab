// fp32 DiffNet residual stack on the 16-bit matrix pipe: every fp32 operand split into two fp16 terms (to one fp32 ulp; usually exactly).
//
// Same contract, tensors and results (to fp32 rounding) as the fp32-matrix-pipe stack launch (diffnet_f43.hip; reference semantics
// /root/reference/train_bisinger/usr/diff/net.py:66-78,107-130; with TAIL also net.py:126-129 and the sampler update,
// usr/diff/shallow_diffusion_tts.py:149-201).  gfx950 multiplies fp32 operands at 256 FLOP/clk/CU (v_mfma_f32_32x32x2_f32) and fp16 operands at 4096 (v_mfma_f32_32x32x16_f16, fp32 accumulate).  An
// fp32 value a is hi + lo with hi = fp16(a), lo = fp16(a - hi): 11 + 11 significand bits plus the sign of lo cover 23 of fp32's 24 (the
// residual a - hi can have 12 significant bits; rounding it to 11 loses at most the last), so |a - hi - lo| <= 2^-23 |a| (one fp32 ulp, in
// most cases 0) as long as lo is a normal fp16.  A product of two such values is
//     a b = ah bh + ah bl + al bh   (+ al bl, dropped: |al| <= 2^-11 |a|, so <= 2^-22 |a b| in the worst case, ~2^-24 in the mean)
// and each of the three terms is a product of fp16 numbers — EXACT in the fp32 the matrix pipe accumulates in.  Three fp16 MFMAs of
// K = 16 replace eight fp32 MFMAs of K = 2: 3/16 of the matrix cycles of the direct fp32 form (the F(4,3) Winograd form needs 10/16).
// The error of a product is <= 2^-21 relative in the worst case and a few 2^-24 typically — the size of an fp32 rounding — and a dot product of K = 256..768 terms is
// dominated by the roundings of its fp32 accumulation either way (measured: tests/test_gpu_h2.py compares both against float64).
//   Range.  fp16 normals span 2^-14 .. 65504.  Weights are multiplied by a power of two per layer and GEMM (chosen at create so that
// max |w| lands in [2^13, 2^14)) and z, which lies in (-1, 1), by 2^10; the accumulators start from (initial value) x (scale) and are
// multiplied by 1 / scale afterwards — all exact.  Activations x + d are split unscaled: below 0.5 their lo term is a subnormal
// fp16 and carries an ABSOLUTE error of <= 2^-25 (3e-8, half an fp32 ulp of 0.5); |x + d| must stay below 65504 — a RANGE GUARD
// watches every value that is split in the kernel (image, skip sum, hidden tile, updated x): one beyond 60000 (or not finite) raises
// the launch's status word, and the host repeats the call on the fp32 matrix pipe (the path of a hand-off give-up, DiffNet.guarded).
//
// Structure: the on-chip stack launch of the bf16 configuration (diffnet_bf16.hip residual_stack_bf16_kernel) with two fp16 planes
// per LDS image and per weight slab: one workgroup of 8 waves x 256 registers per CU owns a 64-frame tile for all L layers; x and the
// running skip sum live in registers (fp32, accumulator layout); the conv input image x + d_l is rewritten in LDS (hi and lo plane,
// channels-last) by the waves that own the channels; neighbours exchange the two 8-frame edges of both planes through L2 (16 KB per
// tile and layer) under the centre tap of GEMM1; the conditioner term (fp32, 2 KB per frame and layer: the only HBM stream) is
// requested into the free accumulators a phase ahead.  LDS: 2 x 42,240 (image) + 2 x 33,792 (z) + 3 KB of tables = 155,136 B.
// Template NCT = 1 is the same program on 32-frame tiles (one column tile per wave; for batches whose 64-frame tiles would leave CUs
// idle); TAIL appends the sampler step's tail on the tile (skip / output / input projections, DDPM or PLMS update).
// NOTE for whoever edits the kernel: the NCT = 2 form needs all 256 VGPRs and its register allocation has no slack — check the spill
// count after every change (tests/test_build_resources.py; ~10-17 spilled registers outside the matrix loops are the good state).
#include "diffnet_h2_shared.h"
#include <type_traits>

namespace bsg {

namespace {


// scale table, per layer: [0] s1 (GEMM1 weights x s1), [1] 1 / s1, [2] s2 x 2^10 (GEMM2 weights x s2, z x 2^10), [3] its reciprocal
__global__ void h2_absmax_kernel(const float* __restrict__ src, long long n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(fabsf(src[i])));   // non-negative floats order like their bit patterns
  atomicMax(out, m);
}
__global__ void h2_scale_kernel(const unsigned* __restrict__ maxbits, float* __restrict__ tab, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * L) return;
  const float mx = __uint_as_float(maxbits[i]);
  // max |w| x s in [2^13, 2^14): hi never overflows, and lo = (w - hi) x s is a normal fp16 for every |w| >= 2^-15 max |w|
  const float s = (mx > 0.f && mx < 3.0e38f) ? ldexpf(1.0f, 13 - ilogbf(mx)) : 1.0f;
  const float tot = (i & 1) ? s * ZSCALE : s;
  tab[2 * i] = tot;
  tab[2 * i + 1] = 1.0f / tot;   // a power of two: exact
}

__global__ void h2_tail_scale_kernel(const unsigned* __restrict__ maxbits, float* __restrict__ tab) {
  const int i = threadIdx.x;
  if (i >= 3) return;
  const float mx = __uint_as_float(maxbits[i]);
  const float s = (mx > 0.f && mx < 3.0e38f) ? ldexpf(1.0f, 13 - ilogbf(mx)) : 1.0f;
  tab[2 * i] = s;
  tab[2 * i + 1] = 1.0f / s;
}

// out[(((ks*2 + plane)*(M/32) + rt)*64 + lane)*8 + j] = plane ? lo : hi of  s x W(m = 32 rt + (lane & 31), k = 16 ks + 8 (lane >> 5) + j)
// with W(m,k) at src[m*sm + (k % Kc)*sc + (k / Kc)*st]   (dilated conv: k = tap*C + ci, src [2C][C][3]); s = tab[0] / ZSCALE or tab[0]
__global__ void pack_a_frag_h2_kernel(const float* __restrict__ src, _Float16* __restrict__ out, int M, int K, int Kc, long long sm,
                                      long long sc, long long st, const float* __restrict__ tab, int is_gemm2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * K) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  const long long rest = i >> 9;
  const int RT = M / 32;
  const int rt = (int)(rest % RT), ks = (int)(rest / RT);
  const int m = 32 * rt + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
  const float s = is_gemm2 ? tab[0] / ZSCALE : tab[0];
  const float v = src[(long long)m * sm + (long long)(k % Kc) * sc + (long long)(k / Kc) * st] * s;
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  const long long base = ((long long)(ks * 2) * RT + rt) * 512 + lane * 8 + j;
  out[base] = hi;
  out[base + (long long)RT * 512] = lo;
}


// i-th executed k-step -> k-step index: GEMM1 (ROT = 16, 48 k-steps, tap-major) starts with the CENTRE tap, whose B operand is the
// tile's own 64 frames, and visits the two outer taps (which read the neighbours' halo frames) afterwards
template <int ROT>
__device__ __forceinline__ int kmap(int i) {
  if (ROT == 0) return i;
  return i < ROT ? i + ROT : (i < 2 * ROT ? i - ROT : i);
}

// k-step pipeline over two row tiles x two column tiles, 12 MFMAs per k-step (lo hi, hi hi, hi lo for each of the 4 accumulators; an
// accumulator is revisited every 4th MFMA).  A[s] = {row tile 0 hi, row tile 0 lo, row tile 1 hi, row tile 1 lo} of ring slot s,
// refilled right after use (NSH k-steps = 48 MFMAs ahead); the B fragments of the next k-step (LDS: column tile 0 hi, lo, column tile
// 1 hi, lo) are read before the MFMAs of the current one.  `mid()` runs after the first ROT k-steps have been issued (ROT = 0:
// never): the hand-off with the neighbours sits there, under the centre tap's MFMAs; the ring keeps prefetching across it.
// FAIRB: the two waves of a SIMD take turns at issue priority (see f43_gemm1, diffnet_f43.hip).
template <int ROT, bool FAIRB, int NCT, typename LDB, typename MID>
__device__ __forceinline__ void mfma_pipe_h2(f32x16 (&c0)[NCT], f32x16 (&c1)[NCT], f16x8 (&A)[NSH][4], rsrc_t rs, int vfrag,
                                             int sa0, int sa1, int n_ks, LDB ldb, MID mid, int half) {
  // c0[ct] / c1[ct]: row tile 0 / 1 x column tile ct.  B[..][2 ct] = hi, [2 ct + 1] = lo of column tile ct
  f16x8 B[2][2 * NCT];
  ldb(kmap<ROT>(0), B[0]);
  const int last = n_ks - 1;
#pragma unroll 1
  for (int ks = 0; ks < n_ks; ks += NSH) {
    if (FAIRB) {
      const unsigned tnow = (unsigned)__builtin_amdgcn_s_memtime();
      if (((tnow >> 12) & 1u) == (unsigned)half) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
    if (ROT > 0 && ks == ROT) {
      mid();
      ldb(kmap<ROT>(ks), B[0]);   // the B operand of the next k-step was read before the halo rows arrived: read it again
    }
#pragma unroll
    for (int s = 0; s < NSH; ++s) {
      // One k-step: the next step's B fragments and the ring's reloads are issued INSIDE the MFMA sequence (sched_group_barrier), not around it:
      // lo weights x hi operand first — their registers are reloaded right behind that group — and the hi weights' reload one k-step later,
      // inside the next step's second group (residual_part_h2_kernel's pipe, where it matters more: one wave per SIMD).  +0.65 % at B = 16
      const int in = ks + s + 1 <= last ? ks + s + 1 : last;
      ldb(kmap<ROT>(in), B[(s + 1) & 1]);
      const f16x8(&Bc)[2 * NCT] = B[s & 1];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {   // lo hi
        BSG_MFMA_H(c0[ct], A[s][1], Bc[2 * ct]);
        BSG_MFMA_H(c1[ct], A[s][3], Bc[2 * ct]);
      }
      {
        const int ir = ks + s + NSH <= last ? ks + s + NSH : last;
        const int kr = kmap<ROT>(ir);
        A[s][1] = lda8(rs, vfrag, sa0 + kr * KSB2 + PLB);
        A[s][3] = lda8(rs, vfrag, sa1 + kr * KSB2 + PLB);
        const int sp = (s + NSH - 1) % NSH;
        const int ip = ks + s - 1 + NSH <= last ? ks + s - 1 + NSH : last;
        const int kp = kmap<ROT>(ip);
        A[sp][0] = lda8(rs, vfrag, sa0 + kp * KSB2);
        A[sp][2] = lda8(rs, vfrag, sa1 + kp * KSB2);
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {   // hi hi
        BSG_MFMA_H(c0[ct], A[s][0], Bc[2 * ct]);
        BSG_MFMA_H(c1[ct], A[s][2], Bc[2 * ct]);
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {   // hi lo
        BSG_MFMA_H(c0[ct], A[s][0], Bc[2 * ct + 1]);
        BSG_MFMA_H(c1[ct], A[s][2], Bc[2 * ct + 1]);
      }
#pragma unroll
      for (int i = 0; i < 2 * NCT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 2 * NCT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 4 / (2 * NCT), 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCT, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}


// TAIL: the tail of the sampler step runs on the tile while it is still on chip (the fp32 launch pair it replaces: this kernel without
// TAIL + step_tail_kernel, diffnet.hip; net.py:126-129, shallow_diffusion_tts.py:149-201): skip projection + ReLU, output projection,
// sampler update of x (DDPM ancestral or PLMS), and the next evaluation's input projection — the same split-fp16 products, biases
// and sampler arithmetic (diffnet_tail.h) — so that a step is ONE launch and neither the skip sum nor the hidden tile touch HBM.
template <bool FAIRB, bool TAIL, int NCT>
__global__ __launch_bounds__(512, 2) void residual_stack_h2_kernel(StackArgs p, TailArgs a) {
  constexpr int NT = 32 * NCT, XP = h2_xp(NCT), ZP = h2_zp(NCT);   // frames per workgroup; bytes per plane of the image / of z
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;                  // [2 planes][80 frames][528 B]: hi / lo of x + d_l, frames t0-8 .. t0+71
  char* zs = lds_raw + 2 * XP;         // [2 planes][NT frames][528 B]: hi / lo of 2^10 x gated activation
  float* dtab = reinterpret_cast<float*>(lds_raw + 2 * XP + 2 * ZP);   // [256]: d_{l+1} per channel, fetched a layer ahead
  float* btab = dtab + C;                                              // [512]: output-projection bias of the current layer

  const int n_tiles = p.n_tiles, per_xcd = (n_tiles + 7) >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile_id >= n_tiles) return;
  p.fbase = stack_epoch_take(p);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int tpr = p.tiles_per_row, L = p.L, T = p.T;
  const int b = tile_id / tpr, j = tile_id - b * tpr;
  const int t0 = j * NT;
  const int tb = p.t_dev ? (int)p.t_dev[b] : p.t_uniform;
  const bool has_left = j > 0, has_right = j + 1 < tpr;

  const unsigned plane = (unsigned)C * T * 4;
  const rsrc_t rs_x = mk_rsrc(p.x_in + (long long)b * C * T, plane);
  const int rowT = T * 4, vfrag = lane * 16;
  int vcol[NCT], vst[NCT];
  bool col_ok[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int col = t0 + 32 * ct + l31;
    col_ok[ct] = col < T;
    vcol[ct] = (lh * 4 * T + (col_ok[ct] ? col : T - 1)) * 4;
    vst[ct] = (lh * 4 * T + col) * 4;
  }
  const int sa_g = wave * 1024, sa_f = (8 + wave) * 1024;   // gate / filter row tile inside a plane of a k-step slab
  const int sb_r = wave * 1024, sb_s = (8 + wave) * 1024;   // residual / skip row tile

  float xr[NCT][16];      // x, accumulator layout: registers 4g..4g+3 = channels 32w + 8g + 4 lh + (0..3) of frame 32 ct + l31
  float sk[NCT][16];      // running skip sum (fp32), same layout (skip rows C + 32w + ..)
  f32x16 yg[NCT], yf[NCT];     // GEMM1 accumulators (gate / filter rows x column tile); they start from the conditioner term x s1
  // range guard: a value whose hi term would leave the fp16 range (or is not finite) is reported through the hand-off status word, and
  // the host repeats the call on the fp32 matrix pipe (DiffNet.guarded) — the split never returns a clipped result silently.  `worst`
  // collects the largest |value| bit pattern a phase splits (NaN and inf order above every finite float); the flag is wave-uniform
  int range_flag = 0;
  auto range_check = [&](unsigned worst) {
    if (__builtin_amdgcn_ballot_w64(worst >= 0x476A6000u) != 0ull) range_flag = 1;   // 60000.0f
  };
  auto absbits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };

  // the conditioner term of a layer (fp32 [2C][T] rows of this utterance): 64 dword loads per lane, 128 B coalesced per half-wave,
  // requested straight into the accumulators a phase before they are used
  auto cond_request = [&](int l) {
    const rsrc_t rs_ct = mk_rsrc(p.condterm + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int so = (32 * wave + acc_row0(r)) * rowT;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        yg[ct][r] = ldf(rs_ct, vcol[ct], so);
        yf[ct][r] = ldf(rs_ct, vcol[ct], so + C * rowT);
      }
    }
  };
  // image core (frames t0 .. t0+63, this wave's 32 channels) = hi / lo of x + d_l, zero beyond T (the conv pads x + d)
  auto write_core = [&]() {   // d of the layer being prepared is in dtab (written a phase earlier, behind a barrier)
    float dv[16];
    unsigned worst = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = dtab[32 * wave + acc_row(r, lh)];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = xr[ct][4 * g] + dv[4 * g], v1 = xr[ct][4 * g + 1] + dv[4 * g + 1];
        const float v2 = xr[ct][4 * g + 2] + dv[4 * g + 2], v3 = xr[ct][4 * g + 3] + dv[4 * g + 3];
        worst = max(max(worst, max(absbits(v0), absbits(v1))), max(absbits(v2), absbits(v3)));
        const HiLo s0 = split2(v0, v1);
        const HiLo s1_ = split2(v2, v3);
        u32x2 wh = u32x2{s0.hi, s1_.hi}, wl = u32x2{s0.lo, s1_.lo};
        if (!col_ok[ct]) { wh = u32x2{0u, 0u}; wl = u32x2{0u, 0u}; }
        char* dst = xs + (HALO + 32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2;
        *reinterpret_cast<u32x2*>(dst) = wh;
        *reinterpret_cast<u32x2*>(dst + XP) = wl;
      }
    range_check(worst);
  };

  // ---- layer 0: x from HBM (the whole input exists, halo included) ------------------------------------------------------
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xr[ct][r] = ldf(rs_x, vcol[ct], (32 * wave + acc_row0(r)) * rowT);
      sk[ct][r] = 0.f;
    }
  {
    const rsrc_t rs_dp = mk_rsrc(p.dproj + ((long long)tb * L + 0) * C, C * 4);
    const int hf = tid & 15, hc = tid >> 4;   // 16 halo frames x 32 chunks of 8 channels
    const int th = hf < 8 ? t0 - HALO + hf : t0 + NT - 8 + hf;
    const int hrow = hf < 8 ? hf : NT + hf;
    const bool hok = th >= 0 && th < T;
    float hv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) hv[k] = ldf(rs_x, hok ? ((8 * hc + k) * T + th) * 4 : 0, 0) + ldf(rs_dp, (8 * hc + k) * 4, 0);
    range_check(max(max(max(absbits(hv[0]), absbits(hv[1])), max(absbits(hv[2]), absbits(hv[3]))),
                    max(max(absbits(hv[4]), absbits(hv[5])), max(absbits(hv[6]), absbits(hv[7])))));
    const HiLo h0 = split2(hv[0], hv[1]), h1 = split2(hv[2], hv[3]), h2 = split2(hv[4], hv[5]), h3 = split2(hv[6], hv[7]);
    u32x4 wh = u32x4{h0.hi, h1.hi, h2.hi, h3.hi}, wl = u32x4{h0.lo, h1.lo, h2.lo, h3.lo};
    if (!hok) { wh = u32x4{0u, 0u, 0u, 0u}; wl = u32x4{0u, 0u, 0u, 0u}; }
    *reinterpret_cast<u32x4*>(xs + hrow * ROWB + hc * 16) = wh;
    *reinterpret_cast<u32x4*>(xs + XP + hrow * ROWB + hc * 16) = wl;
  }
  if (tid < C) dtab[tid] = p.dproj[((long long)tb * L + 0) * C + tid];
  btab[tid] = p.bias_out[tid];
  cond_request(0);
  __syncthreads();
  write_core();
  // weight ring, shared by both GEMMs.  GEMM1's first k-steps (it starts with the centre tap: kmap) are requested a phase ahead —
  // right behind the previous layer's GEMM2 — so that the L2 latency of the weight stream is never on the layer's critical path
  f16x8 A[NSH][4];
  auto prefetch_a1 = [&](int l) {
    const rsrc_t rs = mk_rsrc(p.apack1s + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const int kr = kmap<16>(k);
      A[k][0] = lda8(rs, vfrag, sa_g + kr * KSB2);
      A[k][1] = lda8(rs, vfrag, sa_g + kr * KSB2 + PLB);
      A[k][2] = lda8(rs, vfrag, sa_f + kr * KSB2);
      A[k][3] = lda8(rs, vfrag, sa_f + kr * KSB2 + PLB);
    }
  };
  prefetch_a1(0);

#define STK_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (p.stamps && lane == 0 && (wave == 0 || p.stamp_mode >= 4)) {                                              \
      unsigned long long sv_ = __builtin_amdgcn_s_memrealtime();                                                  \
      if (p.stamp_mode >= 2) sv_ = (sv_ & 0xffffffffull) | ((unsigned long long)__builtin_amdgcn_s_memtime() << 32); /* + shader cycles */ \
      /* modes >= 4: every wave stamps, [tile][layer][wave][8] (tools/wave_stamps.py) */                           \
      p.stamps[p.stamp_mode >= 4 ? (((long long)tile_id * L + l) * 8 + wave) * 8 + (i) : ((long long)tile_id * L + l) * 8 + (i)] = sv_; \
    }                                                                                                             \
  } while (0)
  if (p.stamps && tid == 0 && p.stamp_mode < 4) p.stamps[((long long)tile_id * L + L - 1) * 8 + 6] = __builtin_amdgcn_s_memtime();   // ... and start of the first (tools/stack_stamps.py)
  if (p.clk && tile_id == 0 && tid == 0) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int dil = 1 << (l % p.cycle);
    const rsrc_t rs_a1 = mk_rsrc(p.apack1s + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
    const rsrc_t rs_a2 = mk_rsrc(p.apack2s + (long long)l * (2 * 2 * C * C), 2 * 2 * C * C * 2);
    const float s1 = p.h2_scale[4 * l], inv1 = p.h2_scale[4 * l + 1], s2 = p.h2_scale[4 * l + 2], inv2 = p.h2_scale[4 * l + 3];
    const float dnext = (tid < C && l + 1 < L) ? p.dproj[((long long)tb * L + l + 1) * C + tid] : 0.f;   // lands during GEMM1
    const float bnext = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + tid] : 0.f;
    // GEMM1 accumulates (conditioner term + W x) x s1: the requested term is scaled on arrival (its first use)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) { yg[ct][r] *= s1; yf[ct][r] *= s1; }
    if (l == 0) __syncthreads();   // layer 0: the staged image (core + halo rows); later layers: barrier (C) below covers the core rows
    STK_STAMP(0);
    // ---- GEMM1: 48 k-steps.  The centre tap (16 k-steps) reads the tile's own frames only, so it runs while the neighbours'
    // edges of this layer are still in flight; the wait for them, and the copy of the halo rows, sit behind it (mid) -------------
    {
      const char* xb = xs + (HALO + l31) * ROWB + lh * 16;
      auto ldb = [&](int ks, f16x8 (&Bf)[2 * NCT]) {
        const int tap = ks >> 4, kc = ks & 15;
        const char* q = xb + ((tap - 1) * dil) * ROWB + kc * 32;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          Bf[2 * ct] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB);
          Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB + XP);
        }
      };
      auto mid = [&]() {
        if (l == 0) return;   // layer 0 staged its halo rows from HBM
        if constexpr (NCT == 1) {   // (measured: +1.1 % for the 32-frame form at B=8, -0.8 % for the 64-frame form at B=16: profiles/r03_ab_poll.log)
        if (wave == 0) {
          // lane 0 polls the left neighbour's flag, lane 1 the right one's — both loads in flight together (one L2 round trip, not two)
          const unsigned want = p.fbase + (unsigned)l;
          const bool mine = lane == 0 ? has_left : (lane == 1 ? has_right : false);
          const unsigned* fl = p.flags + (lane == 0 ? tile_id - 1 : tile_id + 1);
          bool pend = mine;
          if (p.inject == 1) {
            if (pend) atomicAdd(p.status, 1u);
          } else {
            unsigned spins = 0;
            while (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
              if (pend) pend = (int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0;
              if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
              __builtin_amdgcn_s_sleep(2);
              // ~ seconds: never reached unless a workgroup is not resident.  Once ANY wait of this handle has given up (status != 0: the host
              // repeats the call without hand-offs anyway) the others stop waiting within a thousand polls instead of seconds each
              if (++spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (pend) atomicAdd(p.status, 1u);
                break;
              }
            }
          }
        }
        } else if (tid == 0) {
          const unsigned want = p.fbase + (unsigned)l;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            if (side == 0 ? !has_left : !has_right) continue;
            const unsigned* fl = p.flags + (side == 0 ? tile_id - 1 : tile_id + 1);
            if (p.inject == 1) { atomicAdd(p.status, 1u); continue; }
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                atomicAdd(p.status, 1u);
                break;
              }
            }
          }
        }
        __syncthreads();   // (D) the polling wave has seen both flags
        if (p.stamp_mode != 3 && p.stamp_mode < 5) STK_STAMP(1);
        {
          // halo rows of this layer, both planes: rows 0..7 = the left neighbour's last 8 frames, rows 72..79 = the right neighbour's
          // first 8.  Write-through (sc1) stores, drained before the flag, one workgroup per CU, and EVERY load of the handed-off
          // bytes an sc1 buffer load to registers: the hand-off form that needs no agent-scope acquire (MI355X_MICROARCH.md)
          const int side = tid >> 8, f = (tid >> 5) & 7, c16 = tid & 31;
          const bool have = side == 0 ? has_left : has_right;
          u32x4 vh = u32x4{0u, 0u, 0u, 0u}, vl = u32x4{0u, 0u, 0u, 0u};
          if (have) {
            const unsigned short* src = reinterpret_cast<const unsigned short*>(p.hx) +
                                        ((long long)(l & 1) * n_tiles + (side == 0 ? tile_id - 1 : tile_id + 1)) * (4 * 8 * C);
            const rsrc_t rs_h = mk_rsrc(src, 4 * 8 * C * 2);
            const int o = (((side == 0 ? 8 : 0) + f) * C + c16 * 8) * 2;   // the neighbour's side 1 (its last frames) for our left halo
            vh = __builtin_amdgcn_raw_buffer_load_b128(rs_h, o, 0, 16);                 // sc1
            vl = __builtin_amdgcn_raw_buffer_load_b128(rs_h, o + 2 * 8 * C * 2, 0, 16);   // lo plane
          }
          char* dst = xs + ((side ? HALO + NT : 0) + f) * ROWB + c16 * 16;
          *reinterpret_cast<u32x4*>(dst) = vh;
          *reinterpret_cast<u32x4*>(dst + XP) = vl;
        }
        __syncthreads();   // (A) halo rows in place
        if (p.stamp_mode != 3 && p.stamp_mode < 5) STK_STAMP(2);
      };
      mfma_pipe_h2<16, FAIRB, NCT>(yg, yf, A, rs_a1, vfrag, sa_g, sa_f, 48, ldb, mid, wave >> 2);
      if (FAIRB) __builtin_amdgcn_s_setprio(0);
    }
    STK_STAMP(3);
    // ---- gate -> zs (hi / lo of 2^10 z); GEMM2's first weights fly meanwhile ------------------------------------------------
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      A[k][0] = lda8(rs_a2, vfrag, sb_r + k * KSB2);
      A[k][1] = lda8(rs_a2, vfrag, sb_r + k * KSB2 + PLB);
      A[k][2] = lda8(rs_a2, vfrag, sb_s + k * KSB2);
      A[k][3] = lda8(rs_a2, vfrag, sb_s + k * KSB2 + PLB);
    }
    if (tid < C) dtab[tid] = dnext;   // read by write_core() behind barrier (B)
    const float rs2 = inv2 * 0.70710678118654752440f;
    const float gcg = -1.44269504088896340736f * inv1, gcf = -2.88539008177792681472f * inv1, glim = 15.0f * s1;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x2 z01 = gate2_scaled(f32x2{yg[ct][4 * g], yg[ct][4 * g + 1]}, f32x2{yf[ct][4 * g], yf[ct][4 * g + 1]}, gcg, gcf, glim, ZSCALE);   // 2^10 z from the raw (scaled) accumulators
        const f32x2 z23 = gate2_scaled(f32x2{yg[ct][4 * g + 2], yg[ct][4 * g + 3]}, f32x2{yf[ct][4 * g + 2], yf[ct][4 * g + 3]}, gcg, gcf, glim, ZSCALE);
        const HiLo s0 = split2(z01[0], z01[1]), s1_ = split2(z23[0], z23[1]);
        const u32x2 wh = u32x2{s0.hi, s1_.hi}, wl = u32x2{s0.lo, s1_.lo};
        char* dst = zs + (32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2;
        *reinterpret_cast<u32x2*>(dst) = wh;
        *reinterpret_cast<u32x2*>(dst + ZP) = wl;
      }
    }
    // residual rows start from (x + b_out) x s2', skip rows from b_out x s2' (the accumulators of GEMM1 are free now)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float br = btab[32 * wave + acc_row(r, lh)], bs = btab[C + 32 * wave + acc_row(r, lh)];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        yg[ct][r] = (xr[ct][r] + br) * s2;
        yf[ct][r] = bs * s2;
      }
    }
    __syncthreads();   // (B) zs complete; every wave is done reading xs and this layer's biases
    btab[tid] = bnext;
    STK_STAMP(4);
    // ---- GEMM2: 16 k-steps; yg = residual rows, yf = skip rows -----------------------------------------------------------
    {
      const char* zb = zs + l31 * ROWB + lh * 16;
      auto ldb = [&](int ks, f16x8 (&Bf)[2 * NCT]) {
        const char* q = zb + ks * 32;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          Bf[2 * ct] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB);
          Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB + ZP);
        }
      };
      mfma_pipe_h2<0, FAIRB, NCT>(yg, yf, A, rs_a2, vfrag, sb_r, sb_s, 16, ldb, [] {}, wave >> 2);
      if (FAIRB) __builtin_amdgcn_s_setprio(0);
    }
    if (l + 1 < L) prefetch_a1(l + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        xr[ct][r] = yg[ct][r] * rs2;   // (x + residual) / sqrt(2), net.py:78: un-scaling and 1 / sqrt(2) in one factor (a product: the IEEE
        sk[ct][r] += yf[ct][r] * inv2; // division is ~10 instructions per element)
      }
    }
    STK_STAMP(5);
    if (l + 1 == L) break;

    // ---- next layer: its conditioner term (128 KB per tile, the only HBM stream) is requested into the free accumulators NOW, so
    // that it lands under the image / publish phase; then the image, the edges for the neighbours, the flag ------------------------
    cond_request(l + 1);
    if (p.stamp_mode == 3 || p.stamp_mode >= 5) STK_STAMP(1);   // diagnostics: the image phase's inner boundaries instead of GEMM1's
    write_core();
    if (p.stamp_mode == 3 || p.stamp_mode >= 5) STK_STAMP(2);
    __syncthreads();   // (C1) the core rows are complete (every wave wrote its 32 channels of every frame)
    STK_STAMP(6);
    {
      // publish the first and the last 8 frames of both planes: [plane][side][8 frames][256 ch] fp16 = 16 KB, write-through
      unsigned short* hx_t = reinterpret_cast<unsigned short*>(p.hx) + ((long long)((l + 1) & 1) * n_tiles + tile_id) * (4 * 8 * C);
      const int side = tid >> 8, f = (tid >> 5) & 7, c16 = tid & 31;
      const char* srcp = xs + (HALO + (side ? NT - 8 : 0) + f) * ROWB + c16 * 16;
      const u32x4 vh = *reinterpret_cast<const u32x4*>(srcp);
      const u32x4 vl = *reinterpret_cast<const u32x4*>(srcp + XP);
      if (!(p.inject == 1 && (tile_id & 1))) {
        const rsrc_t rs_hx = mk_rsrc(hx_t, 4 * 8 * C * 2);
        const int o = ((side * 8 + f) * C + c16 * 8) * 2;
        __builtin_amdgcn_raw_buffer_store_b128(vh, rs_hx, o, 0, 16);                   // sc1
        __builtin_amdgcn_raw_buffer_store_b128(vl, rs_hx, o + 2 * 8 * C * 2, 0, 16);   // lo plane
      }
    }
    // every storing wave drains its write-through stores before the flag goes up.  vmcnt counts in order: the 64 conditioner loads
    // of this wave are older than its edge stores, so this also waits for them (they have had the image phase to land)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (C)
    if (tid == 0) __hip_atomic_store(p.flags + tile_id, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    STK_STAMP(7);
  }
#undef STK_STAMP
  if (tid == 0) stack_epoch_done(p, p.fbase, n_tiles);
  if (p.stamps && tid == 0 && p.stamp_mode < 4) p.stamps[((long long)tile_id * L + L - 1) * 8 + 7] = __builtin_amdgcn_s_memtime();   // shader clock: end of the last layer ...
  if (p.clk && tile_id == 0 && tid == 0) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); }
  if constexpr (!TAIL) {
    if (range_flag && lane == 0) atomicAdd(p.status + 1, 1u);   // word 1: range events (word 0: hand-off give-ups)
    // ---- the skip sum / sqrt(L) (net.py:126), fp32 [C][T] rows: what the step tail (diffnet.hip step_tail_kernel) reads -------------
    const rsrc_t rs_sk = mk_rsrc(p.skip + (long long)b * C * T, plane);
    const float rdiv = 1.0f / sqrtf((float)L);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
      if (col_ok[ct]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stf(sk[ct][r] * rdiv, rs_sk, vst[ct], (32 * wave + acc_row0(r)) * rowT);
      }
  } else {
    // ================= fused step tail =================================================================================================
    const int M = a.M;
    const float* tsc = a.tail_scale;   // [3][2]: scale, 1 / scale of the skip / output / input projection
    // ---- s = skip sum / sqrt(L) -> hi / lo image rows (the conv image is dead: every wave is behind barrier (B) of the last layer) ----
    {
      const float rdiv = 1.0f / sqrtf((float)L);
      unsigned worst = 0;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) worst = max(worst, absbits(sk[ct][r]));   // |s| <= |skip sum|
      range_check(worst);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const HiLo s0 = split2(sk[ct][4 * g] * rdiv, sk[ct][4 * g + 1] * rdiv), s1_ = split2(sk[ct][4 * g + 2] * rdiv, sk[ct][4 * g + 3] * rdiv);
          char* dst = xs + (HALO + 32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2;
          *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
          *reinterpret_cast<u32x2*>(dst + XP) = u32x2{s0.lo, s1_.lo};
        }
    }
    const char* xcore = xs + (HALO + l31) * ROWB + lh * 16;
    auto ldb_x2 = [&](int ks, f16x8 (&Bf)[2 * NCT]) {   // every column tile of the image rows
      const char* q = xcore + ks * 32;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        Bf[2 * ct] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB);
        Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB + XP);
      }
    };
    // ---- h = relu(W_skip s + b) -> zs (hi / lo) -----------------------------------------------------------------------------------
    {
      const rsrc_t rs_ws = mk_rsrc(a.ws_s, 2 * C * C * 2);
      const rsrc_t rs_bs = mk_rsrc(a.b_skip, C * 4);
      const float sc = tsc[0], inv = tsc[1];
      f32x16 hc[NCT];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) hc[ct][r] = ldf(rs_bs, lh * 16, (32 * wave + acc_row0(r)) * 4) * sc;
      __syncthreads();   // (T1) s complete; every wave is done with GEMM2 of the last layer (zs is free)
      tail_gemm_h2<NCT, 16>(hc, rs_ws, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024, ldb_x2);
      {
        unsigned worst = 0;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) worst = max(worst, absbits(hc[ct][r] * inv));
        range_check(worst);
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const HiLo s0 = split2(fmaxf(hc[ct][4 * g] * inv, 0.f), fmaxf(hc[ct][4 * g + 1] * inv, 0.f));
          const HiLo s1_ = split2(fmaxf(hc[ct][4 * g + 2] * inv, 0.f), fmaxf(hc[ct][4 * g + 3] * inv, 0.f));
          char* dst = zs + (32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2;
          *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
          *reinterpret_cast<u32x2*>(dst + ZP) = u32x2{s0.lo, s1_.lo};
        }
    }
    __syncthreads();   // (T2) h complete; every wave is done reading s
    // ---- the step's noise, by ALL waves: the Philox quads that cover the tile's frames of each mel row (element idx = quad idx >> 2, lane
    // idx & 3: the values philox_normal1 returns).  Evaluated per element by the 3 NCT updating waves it was 16 Philox rounds + Box-Muller
    // per lane.  Where: the dead s image, bytes 192.. of the hi plane's core rows (channels 96.. — the updated x below uses 0..95): 80 floats
    // per frame, so for in_dims <= 80 only (else per element as before) ------------------------------------------------------------------
    const bool lds_noise = !a.noise && a.k.sigma != 0.f && !a.plms_hist && M <= 80;
    if (lds_noise) {
      constexpr int QPR = NT / 4 + 1;
#pragma unroll 1
      for (int item = tid; item < M * QPR; item += 512) {
        const int m = item / QPR, jq = item - m * QPR;
        const unsigned long long base = a.quad_row0 + ((unsigned long long)b * M + m) * T + t0;
        const unsigned long long qd = (base >> 2) + jq;
        const f32x4 z = philox_normal4(a.seed, a.stream, qd);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const long long cx = (long long)(4 * qd + c) - (long long)base;
          if (cx >= 0 && cx < NT) *reinterpret_cast<float*>(xs + (HALO + (int)cx) * ROWB + 192 + 4 * m) = z[c];
        }
      }
      __syncthreads();   // (T2b)
    }
    // ---- eps = W_out h + b and the sampler update, fp32, on the 3 row tiles that cover the M mel bins x 2 column tiles: waves 0..5 ----
    if (wave < 3 * NCT) {
      const int rt = wave % 3, ct2 = wave / 3;
      const int col = t0 + 32 * ct2 + l31;
      const bool cok = col < T;
      const int vc = (lh * 4 * T + (cok ? col : T - 1)) * 4, vs = (lh * 4 * T + col) * 4;
      const rsrc_t rs_wo = mk_rsrc(a.wo_s, 2 * 96 * C * 2);
      const rsrc_t rs_bf = mk_rsrc(a.b_fin, 96 * 4);
      const rsrc_t rs_xx = mk_rsrc(a.x + (long long)b * M * T, (unsigned)M * T * 4);
      const rsrc_t rs_n = mk_rsrc(a.noise ? a.noise + (long long)b * M * T : a.x, a.noise ? (unsigned)M * T * 4 : 0u);
      const float sc = tsc[2], inv = tsc[3];
      f32x16 e[1];
      float xv[16], nv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // the lane's row is m0 + 4 lh; rows >= M fall outside the descriptor's range and read as 0, and are never stored
        const int m0 = 32 * rt + acc_row0(r);
        e[0][r] = ldf(rs_bf, lh * 16, m0 * 4) * sc;
        xv[r] = ldf(rs_xx, vc, m0 * rowT);
        nv[r] = a.noise ? ldf(rs_n, vc, m0 * rowT) : 0.f;
      }
      float h1v[16], h2v[16], h3v[16];
      if (a.plms_hist) {
        const unsigned hb = (unsigned)M * T * 4;
        const rsrc_t rs_h1 = mk_rsrc(a.h1 + (long long)b * M * T, hb);
        const rsrc_t rs_h2 = mk_rsrc(a.plms_hist > 1 ? a.h2 + (long long)b * M * T : a.x, a.plms_hist > 1 ? hb : 0u);
        const rsrc_t rs_h3 = mk_rsrc(a.plms_hist > 2 ? a.h3 + (long long)b * M * T : a.x, a.plms_hist > 2 ? hb : 0u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int so = (32 * rt + acc_row0(r)) * rowT;
          h1v[r] = ldf(rs_h1, vc, so);
          h2v[r] = ldf(rs_h2, vc, so);   // zero-size descriptors read as 0
          h3v[r] = ldf(rs_h3, vc, so);
        }
      }
      const char* zb = zs + (32 * ct2 + l31) * ROWB + lh * 16;
      auto ldb_h = [&](int ks, f16x8 (&Bf)[2]) {
        Bf[0] = *reinterpret_cast<const f16x8*>(zb + ks * 32);
        Bf[1] = *reinterpret_cast<const f16x8*>(zb + ks * 32 + ZP);
      };
      tail_gemm_h2<1, 16>(e, rs_wo, vfrag, rt * 1024, 2 * 3 * 1024, 3 * 1024, ldb_h);
      const rsrc_t rs_en = mk_rsrc(a.plms_hist ? a.e_new + (long long)b * M * T : a.x, a.plms_hist ? (unsigned)M * T * 4 : 0u);
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * rt + acc_row(r, lh);
        const float ev = e[0][r] * inv;
        o[r] = 0.f;
        if (m < M) {
          if (a.plms_hist) {
            o[r] = plms_update(xv[r], ev, h1v[r], h2v[r], h3v[r], a.plms_hist, a.pk, nullptr);
            if (cok) stf(ev, rs_en, vs, (32 * rt + acc_row0(r)) * rowT);
          } else {
            float nz = nv[r];
            if (lds_noise) nz = *reinterpret_cast<const float*>(xs + (HALO + 32 * ct2 + l31) * ROWB + 192 + 4 * m);
            else if (!a.noise && a.k.sigma != 0.f)
              nz = philox_normal1(a.seed, a.stream, a.quad_row0 + ((unsigned long long)b * M + m) * T + (cok ? col : T - 1));
            float x0 = __fsub_rn(__fmul_rn(a.k.recip, xv[r]), __fmul_rn(a.k.recipm1, ev));
            x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            const float mean = __fadd_rn(__fmul_rn(a.k.pc1, x0), __fmul_rn(a.k.pc2, xv[r]));
            o[r] = __fadd_rn(mean, __fmul_rn(a.k.sigma, nz));
          }
          if (cok) stf(o[r], rs_xx, vs, (32 * rt + acc_row0(r)) * rowT);
        }
      }
      {
        unsigned worst = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) worst = max(worst, absbits(o[r]));
        range_check(worst);
      }
      // the updated x as the input projection's B operand: channels-last rows of the image region (channels 0..95; rows >= M zero)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const HiLo s0 = split2(o[4 * g], o[4 * g + 1]), s1_ = split2(o[4 * g + 2], o[4 * g + 3]);
        char* dst = xs + (HALO + 32 * ct2 + l31) * ROWB + (32 * rt + 8 * g + 4 * lh) * 2;
        *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
        *reinterpret_cast<u32x2*>(dst + XP) = u32x2{s0.lo, s1_.lo};
      }
    }
    if (range_flag && lane == 0) atomicAdd(p.status + 1, 1u);   // word 1: range events (word 0: hand-off give-ups)
    if (!a.do_head) return;
    // ---- next evaluation's input projection: xa = relu(W_in x + b), K = 96 (in_dims zero-padded) ------------------------------------
    {
      const rsrc_t rs_wi = mk_rsrc(a.wi_s, 2 * C * 96 * 2);
      const rsrc_t rs_bi = mk_rsrc(a.b_in, C * 4);
      const float sc = tsc[4], inv = tsc[5];
      f32x16 hc[NCT];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) hc[ct][r] = ldf(rs_bi, lh * 16, (32 * wave + acc_row0(r)) * 4) * sc;
      __syncthreads();   // (T3) the updated x tile is complete
      tail_gemm_h2<NCT, 6>(hc, rs_wi, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024, ldb_x2);
      const rsrc_t rs_xa = mk_rsrc(a.xa_next + (long long)b * C * T, plane);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        if (col_ok[ct]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stf(fmaxf(hc[ct][r] * inv, 0.f), rs_xa, vst[ct], (32 * wave + acc_row0(r)) * rowT);
        }
    }
  }
}
// ------------------------------------------------------------------------------------------------
// The step tail of the part forms as its own launch on the 16-bit matrix pipe: what the TAIL branch of residual_stack_h2_kernel does on
// chip, from the skip sum in HBM (fp32 [C][T], written by the last layer of the part launch).  One workgroup of 8 waves per 32-frame
// tile; the three projections as split-fp16 GEMMs (the fragments and scales of h2_tail_pack), the sampler update in fp32 with the
// reference's rounding sequence (diffnet_tail.h).  It replaces step_tail_kernel (fp32 matrix pipe: 39.8 us per step at B = 1, where the
// 32 workgroups of a single utterance run three serial GEMMs of 128 + 128 + 40 16-pass MFMAs) behind those launches only; operands beyond
// the fp16 range count a range event in a.status[1] like the launches in front of it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void step_tail_h2_kernel(TailArgs a) {
  constexpr int XP = h2_xp(1), ZP = h2_zp(1);
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;                  // [2 planes][48 rows][528 B]: hi / lo of s, later of the updated x (channels 0..95); rows HALO..HALO+31 used
  char* zs = lds_raw + 2 * XP;         // [2 planes][32 rows][528 B]: hi / lo of h
  float* nzs = reinterpret_cast<float*>(lds_raw + 2 * XP + 2 * ZP);   // [96][32]: the step's Philox normals of the tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int T = a.T, M = a.M;
  const int b = blockIdx.x / a.tiles_per_row;
  const int t0 = (blockIdx.x - b * a.tiles_per_row) * 32;
  const int col = t0 + l31;
  const bool col_ok = col < T;
  const int rowT = T * 4, vfrag = lane * 16;
  const int vcol = (lh * 4 * T + (col_ok ? col : T - 1)) * 4, vst = (lh * 4 * T + col) * 4;
  const unsigned plane = (unsigned)C * T * 4;
  const float* tsc = a.tail_scale;   // [3][2]: scale, 1 / scale of the skip / output / input projection
  int range_flag = 0;
  auto range_check = [&](unsigned worst) {
    if (__builtin_amdgcn_ballot_w64(worst >= 0x476A6000u) != 0ull) range_flag = 1;   // 60000.0f
  };
  auto absbits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };
  // the first weight fragments of the skip projection (all waves) and of the output projection (waves 0..2), requested before anything else:
  // they arrive while the skip sum is on its way and the noise is generated (each projection used to fill its ring behind its barrier)
  const rsrc_t rs_ws = mk_rsrc(a.ws_s, 2 * C * C * 2);
  const rsrc_t rs_wo = mk_rsrc(a.wo_s, 2 * 96 * C * 2);
  const rsrc_t rs_wi = mk_rsrc(a.wi_s, 2 * C * 96 * 2);
  TailRing<16> ring_s, ring_o;
  TailRing<6> ring_i;
  tail_ring_fill<16>(ring_s, rs_ws, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024);
  if (wave < 3) tail_ring_fill<16>(ring_o, rs_wo, vfrag, wave * 1024, 2 * 3 * 1024, 3 * 1024);
  // ---- s (skip sum / sqrt(L), fp32 rows) -> hi / lo image rows: 32 chunks of 8 channels x 32 frames, lanes = consecutive frames.  The
  // loads are requested first; the noise below is generated while they are on their way from HBM ------------------------------------
  float sv[2][8];
  {
    const rsrc_t rs_s = mk_rsrc(a.skip + (long long)b * C * T, plane);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = it * 512 + tid;
      const int hc = item >> 5, t = t0 + (item & 31);
#pragma unroll
      for (int k = 0; k < 8; ++k) sv[it][k] = ldf(rs_s, t < T ? ((8 * hc + k) * T + t) * 4 : 0, 0);
    }
  }
  // ---- the step's noise, by ALL waves: the Philox quads that cover the tile's 32 frames of each mel row (element idx = quad idx >> 2, lane
  // idx & 3: the values philox_normal1 returns; evaluated per element by the three updating waves it was 16 evaluations per lane) --------
  const bool philox = !a.noise && a.k.sigma != 0.f && !a.plms_hist;
  if (philox) {
#pragma unroll 1
    for (int item = tid; item < M * 9; item += 512) {
      const int m = item / 9, jq = item - m * 9;
      const unsigned long long base = a.quad_row0 + ((unsigned long long)b * M + m) * T + t0;
      const unsigned long long qd = (base >> 2) + jq;
      const f32x4 z = philox_normal4(a.seed, a.stream, qd);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long long cx = (long long)(4 * qd + c) - (long long)base;
        if (cx >= 0 && cx < 32) nzs[m * 32 + (int)cx] = z[c];
      }
    }
  }
  {
    unsigned worst = 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = it * 512 + tid;
      const int hc = item >> 5, f = item & 31;
      const bool ok = t0 + f < T;
      const float(&v)[8] = sv[it];
      if (ok) worst = max(worst, max(max(max(absbits(v[0]), absbits(v[1])), max(absbits(v[2]), absbits(v[3]))),
                                     max(max(absbits(v[4]), absbits(v[5])), max(absbits(v[6]), absbits(v[7])))));
      const HiLo h0 = split2(v[0], v[1]), h1 = split2(v[2], v[3]), h2 = split2(v[4], v[5]), h3 = split2(v[6], v[7]);
      u32x4 wh = u32x4{h0.hi, h1.hi, h2.hi, h3.hi}, wl = u32x4{h0.lo, h1.lo, h2.lo, h3.lo};
      if (!ok) { wh = u32x4{0u, 0u, 0u, 0u}; wl = u32x4{0u, 0u, 0u, 0u}; }
      *reinterpret_cast<u32x4*>(xs + (HALO + f) * ROWB + hc * 16) = wh;
      *reinterpret_cast<u32x4*>(xs + XP + (HALO + f) * ROWB + hc * 16) = wl;
    }
    range_check(worst);
  }
  const char* xcore = xs + (HALO + l31) * ROWB + lh * 16;
  auto ldb_x = [&](int ks, f16x8 (&Bf)[2]) {
    Bf[0] = *reinterpret_cast<const f16x8*>(xcore + ks * 32);
    Bf[1] = *reinterpret_cast<const f16x8*>(xcore + ks * 32 + XP);
  };
  // ---- h = relu(W_skip s + b) -> zs (hi / lo)                                                                          (net.py:126-128) ----
  {
    const rsrc_t rs_bs = mk_rsrc(a.b_skip, C * 4);
    const float sc = tsc[0], inv = tsc[1];
    f32x16 hc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) hc[0][r] = ldf(rs_bs, lh * 16, (32 * wave + acc_row0(r)) * 4) * sc;
    __syncthreads();   // (T1) s complete
    tail_gemm_h2_run<1, 16>(hc, ring_s, rs_ws, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024, ldb_x);
    if (a.do_head) tail_ring_fill<6>(ring_i, rs_wi, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024);   // (the skip projection's ring is free)
    unsigned worst = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) worst = max(worst, absbits(hc[0][r] * inv));
    range_check(worst);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const HiLo s0 = split2(fmaxf(hc[0][4 * g] * inv, 0.f), fmaxf(hc[0][4 * g + 1] * inv, 0.f));
      const HiLo s1_ = split2(fmaxf(hc[0][4 * g + 2] * inv, 0.f), fmaxf(hc[0][4 * g + 3] * inv, 0.f));
      char* dst = zs + l31 * ROWB + (32 * wave + 8 * g + 4 * lh) * 2;
      *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
      *reinterpret_cast<u32x2*>(dst + ZP) = u32x2{s0.lo, s1_.lo};
    }
  }
  __syncthreads();   // (T2) h complete; every wave is done reading s
  // ---- eps = W_out h + b and the sampler update, fp32, on the 3 row tiles that cover the M mel bins: waves 0..2     (net.py:129) --------------
  if (wave < 3) {
    const int rt = wave;
    const rsrc_t rs_bf = mk_rsrc(a.b_fin, 96 * 4);
    const rsrc_t rs_xx = mk_rsrc(a.x + (long long)b * M * T, (unsigned)M * T * 4);
    const rsrc_t rs_n = mk_rsrc(a.noise ? a.noise + (long long)b * M * T : a.x, a.noise ? (unsigned)M * T * 4 : 0u);
    const float sc = tsc[2], inv = tsc[3];
    f32x16 e[1];
    float xv[16], nv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // the lane's row is m0 + 4 lh; rows >= M fall outside the descriptor's range and read as 0, and are never stored
      const int m0 = 32 * rt + acc_row0(r);
      e[0][r] = ldf(rs_bf, lh * 16, m0 * 4) * sc;
      xv[r] = ldf(rs_xx, vcol, m0 * rowT);
      nv[r] = a.noise ? ldf(rs_n, vcol, m0 * rowT) : 0.f;
    }
    float h1v[16], h2v[16], h3v[16];
    if (a.plms_hist) {
      const unsigned hb = (unsigned)M * T * 4;
      const rsrc_t rs_h1 = mk_rsrc(a.h1 + (long long)b * M * T, hb);
      const rsrc_t rs_h2 = mk_rsrc(a.plms_hist > 1 ? a.h2 + (long long)b * M * T : a.x, a.plms_hist > 1 ? hb : 0u);
      const rsrc_t rs_h3 = mk_rsrc(a.plms_hist > 2 ? a.h3 + (long long)b * M * T : a.x, a.plms_hist > 2 ? hb : 0u);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int so = (32 * rt + acc_row0(r)) * rowT;
        h1v[r] = ldf(rs_h1, vcol, so);
        h2v[r] = ldf(rs_h2, vcol, so);   // zero-size descriptors read as 0
        h3v[r] = ldf(rs_h3, vcol, so);
      }
    }
    const char* zb = zs + l31 * ROWB + lh * 16;
    auto ldb_h = [&](int ks, f16x8 (&Bf)[2]) {
      Bf[0] = *reinterpret_cast<const f16x8*>(zb + ks * 32);
      Bf[1] = *reinterpret_cast<const f16x8*>(zb + ks * 32 + ZP);
    };
    tail_gemm_h2_run<1, 16>(e, ring_o, rs_wo, vfrag, rt * 1024, 2 * 3 * 1024, 3 * 1024, ldb_h);
    const rsrc_t rs_en = mk_rsrc(a.plms_hist ? a.e_new + (long long)b * M * T : a.x, a.plms_hist ? (unsigned)M * T * 4 : 0u);
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * rt + acc_row(r, lh);
      const float ev = e[0][r] * inv;
      o[r] = 0.f;
      if (m < M) {
        if (a.plms_hist) {
          o[r] = plms_update(xv[r], ev, h1v[r], h2v[r], h3v[r], a.plms_hist, a.pk, nullptr);
          if (col_ok) stf(ev, rs_en, vst, (32 * rt + acc_row0(r)) * rowT);
        } else {
          const float nz = philox ? nzs[m * 32 + l31] : nv[r];
          float x0 = __fsub_rn(__fmul_rn(a.k.recip, xv[r]), __fmul_rn(a.k.recipm1, ev));
          x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
          const float mean = __fadd_rn(__fmul_rn(a.k.pc1, x0), __fmul_rn(a.k.pc2, xv[r]));
          o[r] = __fadd_rn(mean, __fmul_rn(a.k.sigma, nz));
        }
        if (col_ok) stf(o[r], rs_xx, vst, (32 * rt + acc_row0(r)) * rowT);
      }
    }
    {
      unsigned worst = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) worst = max(worst, absbits(o[r]));
      range_check(worst);
    }
    // the updated x as the input projection's B operand: channels-last rows of the image region (channels 0..95; rows >= M zero)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const HiLo s0 = split2(o[4 * g], o[4 * g + 1]), s1_ = split2(o[4 * g + 2], o[4 * g + 3]);
      char* dst = xs + (HALO + l31) * ROWB + (32 * rt + 8 * g + 4 * lh) * 2;
      *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
      *reinterpret_cast<u32x2*>(dst + XP) = u32x2{s0.lo, s1_.lo};
    }
  }
  if (range_flag && lane == 0 && a.status) atomicAdd(a.status + 1, 1u);   // word 1: range events
  if (!a.do_head) return;
  // ---- next evaluation's input projection: xa = relu(W_in x + b), K = 96 (in_dims zero-padded)                     (net.py:116-118) ----------
  {
    const rsrc_t rs_bi = mk_rsrc(a.b_in, C * 4);
    const float sc = tsc[4], inv = tsc[5];
    f32x16 hc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) hc[0][r] = ldf(rs_bi, lh * 16, (32 * wave + acc_row0(r)) * 4) * sc;
    __syncthreads();   // (T3) the updated x tile is complete
    tail_gemm_h2_run<1, 6>(hc, ring_i, rs_wi, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024, ldb_x);
    const rsrc_t rs_xa = mk_rsrc(a.xa_next + (long long)b * C * T, plane);
    if (col_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stf(fmaxf(hc[0][r] * inv, 0.f), rs_xa, vst, (32 * wave + acc_row0(r)) * rowT);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PART forms for small batches (residual_part_h2_kernel<P, W, NC>).  With one workgroup per tile a single utterance keeps 32 of the 256
// CUs busy, each of them bound by streaming the layer's 2.1 MB of weight fragments out of L2 (tools/l2_fill.hip: a CU pulls at most
// ~64 B/clk).  Here a tile is computed by P workgroups of W waves on P CUs of one XCD, each owning C / P channels — its gate / filter rows
// of GEMM1, its residual / skip rows of GEMM2, its channels of x and of the skip sum in registers — so every CU streams 1 / P of the
// weights.  A wave owns 16 channels as 16-row matrix tiles over NC column tiles of 16 frames (v_mfma_f32_16x16x32_f16: lane l holds
// A[row l & 15][k = 8 (l >> 4) + j], B[k][column l & 15]; C/D column l & 15, rows 4 (l >> 4) + r; the weights are packed a second time in that
// fragment order, pack_a_frag_q_kernel).  What a workgroup lacks it gets from its partners through L2, twice per layer, with the hand-off
// protocol of the launch above (write-through stores, drain, barrier, flag = launch epoch + layer, bounded poll by one lane per flag, sc1
// loads):
//   * after the gate: the partners' parts of z — GEMM2 contracts over all 256 channels;
//   * after GEMM2: the partners' parts of the next image and, from the two neighbouring tiles' workgroups, the 8-frame edges of all parts —
//     GEMM1 contracts over all 256 channels of NT + 16 frames.  GEMM1 starts with the 32-deep k-steps of the centre tap over its OWN
//     channels, the only part of the image a workgroup has without waiting.
// The step tail is not fused (its skip projection contracts over all parts): the skip sum goes to HBM and step_tail_h2_kernel follows.
// Instantiated:
//   <4, 4, 2>  QUAD of a 32-frame tile: 4 x B * ceil(T / 32) <= CUs (one or two utterances at T = 1000)
//   <4, 4, 4>  QUAD of a 64-frame tile: 4 x B * ceil(T / 64) <= CUs (B <= 4 at T = 1000)
//   <2, 8, 4>  PAIR of a 64-frame tile, 8 waves each (two per SIMD, 256 registers: one set of B fragments): 2 x B * ceil(T / 64) <= CUs (B <= 8 at T = 1000; 59.4 ms per pass at B = 8 against 69.1 for one
//              workgroup per 32-frame tile — with every CU busy the conditioner burst and the exchange phases grow while the GEMMs shrink)
// Measured and not kept (profiles/r03_part_forms/): pairs of 32-frame tiles on 32-row matrix tiles (the first form of this idea: B=1 44.4 ms
// per 100-step pass against 24.1 for the quad, B=4 47.3 against 35.0 for the quad of 64-frame tiles), pairs of 64-frame tiles with 4 waves
// of 32-row tiles (B=8: 77.4 ms), octets of 2 waves (B=1: 27.8 against 25.7 ms at the time: the exchange among eight costs more than the halved
// weight stream saves).
// ------------------------------------------------------------------------------------------------
constexpr int QPLB = 32 * 1024;            // bytes per plane of a k-step slab: 32 row tiles of 16 x 1 KB
constexpr int QKSB = 2 * QPLB;             // bytes per k-step (32 deep): hi slab, lo slab

// out[(((ks*2 + plane)*(M/16) + rt)*64 + lane)*8 + j] = plane ? lo : hi of  s x W(m = 16 rt + (lane & 15), k = 32 ks + 8 (lane >> 4) + j)
__global__ void pack_a_frag_q_kernel(const float* __restrict__ src, _Float16* __restrict__ out, int M, int K, int Kc, long long sm, long long sc,
                                     long long st, const float* __restrict__ tab, int is_gemm2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * K) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  const long long rest = i >> 9;
  const int RT = M / 16;
  const int rt = (int)(rest % RT), ks = (int)(rest / RT);
  const int m = 16 * rt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
  const float s = is_gemm2 ? tab[0] / ZSCALE : tab[0];
  const float v = src[(long long)m * sm + (long long)(k % Kc) * sc + (long long)(k / Kc) * st] * s;
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  const long long base = ((long long)(ks * 2) * RT + rt) * 512 + lane * 8 + j;
  out[base] = hi;
  out[base + (long long)RT * 512] = lo;
}

// LDS rows of the part forms: 256 fp16 + 32 B pad = 544 B.  A B fragment of v_mfma_f32_16x16x32_f16 is read as 16 bytes per lane at row
// (lane & 15), byte 16 (lane >> 4): with the 528-byte rows of the 32-row forms the 16-lane groups of ds_read_b128 meet 2-way bank conflicts
// (rows 4 dwords apart); with rows 8 dwords apart mod 64 every group covers the 64 banks once.
constexpr int ROWQ = 2 * C + 32;
constexpr int part_xp(int nc) { return (16 * nc + 2 * HALO) * ROWQ; }   // bytes per plane of the image
constexpr int part_zp(int nc) { return 16 * nc * ROWQ; }                // bytes per plane of z
constexpr size_t part_lds(int nc) { return (size_t)2 * part_xp(nc) + 2 * part_zp(nc) + 3 * C * sizeof(float); }
static_assert(part_lds(4) <= 160 * 1024 && part_lds(2) > 80 * 1024, "the 64-frame forms fill the LDS; the 32-frame ones take more than half of it, so that a CU holds ONE workgroup");
#define BSG_MFMA_Q(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_, B_, ACC, 0, 0, 0)
using f32x4q = __attribute__((ext_vector_type(4))) float;

// Order of GEMM1's k-steps (32 deep; 8 per tap) for a part whose own channels start at k-step `base` of a tap: the centre tap first, cyclically
// from the own channels (the only part of the image a workgroup has without waiting), then tap 0, then tap 2.  The i-th executed k-step is
// kb0 + ((rot + (i & 7)) & 7) with (kb0, rot) constant over the 8 steps of a tap — a pass of the weight ring never straddles two taps, so
// the scalar index arithmetic of a k-step is two adds and a mask (with a general map it outweighed the k-step's MFMA issue time).
struct KPass {
  int kb0;    // first k-step of the tap in the packed weights / the image's tap order (tap 0: 0, centre: 8, tap 2: 16)
  int rot;    // cyclic start inside the tap
  int boff;   // LDS byte offset of the tap's rows relative to the centre tap: (tap - 1) x dilation x ROWQ
};
template <bool GEMM1>
__device__ __forceinline__ KPass k_pass(int i0, int base, int dilrow) {
  if (!GEMM1) return KPass{i0 & ~7, base, 0};   // GEMM2: its 8 k-steps cyclically from the own channels of z too (round 4)
  const int t = i0 >> 3;   // 0: centre tap, 1: tap 0, 2: tap 2
  return KPass{t == 0 ? 8 : (t == 1 ? 0 : 16), t == 0 ? base : 0, t == 0 ? 0 : (t == 1 ? -dilrow : dilrow)};
}
__device__ __forceinline__ int k_of(const KPass& d, int i) { return d.kb0 + ((d.rot + i) & 7); }   // i: any index congruent to the step's mod 8

// k-step pipeline: two row tiles of 16 (c0: gate / residual, c1: filter / skip) x NC column tiles of 16 frames, 6 NC MFMAs per k-step;
// weight ring of NSQ k-steps (NSQ divides 8).  B fragments: 16-byte reads at bptr + tap offset + 64 x (k-step inside the tap) + 16 ct rows,
// the lo plane `bplane` bytes behind.  The hand-off `mid` sits behind the first ROT k-steps.  The ring runs THROUGH the GEMMs: the reloads of
// the last pass fetch the first NSQ k-steps of the GEMM that follows (`rs_next`; NEXT1: it is a GEMM1, in its order) — requested after this
// GEMM instead, the first slot's way from L2 stood in front of every GEMM (1.2 us per layer).
template <bool GEMM1, bool NEXT1, int ROT, int NC, int NSQ, bool BS1, typename MID>
__device__ __forceinline__ void mfma_pipe_part(f32x4q (&c0)[NC], f32x4q (&c1)[NC], f16x8 (&A)[NSQ][4], rsrc_t rs, rsrc_t rs_next, int vfrag, int sa0,
                                               int sa1, int n_ks, int base, int dilrow, const char* bptr, int bplane, MID mid) {
  static_assert(8 % NSQ == 0, "a ring pass stays inside a tap");
  if constexpr (BS1) {
    // Two waves per SIMD (256 registers each): ONE set of B fragments instead of two.  The hi plane's registers are free behind the second
    // MFMA group of a k-step and are refilled during the third (lo weights first: hi x hi is the second group); the lo plane's are free behind
    // the third and are refilled during the next step's first.  The LDS latency of either is a group of 2 NC MFMAs — and the other wave's.
    f16x8 Bh[NC], Bl[NC];
    auto ldh = [&](const KPass& d, int i) {
      const char* qp = bptr + d.boff + ((d.rot + i) & 7) * 64;
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) Bh[ct] = *reinterpret_cast<const f16x8*>(qp + 16 * ct * ROWQ);
    };
    auto ldl = [&](const KPass& d, int i) {
      const char* qp = bptr + d.boff + ((d.rot + i) & 7) * 64 + bplane;
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) Bl[ct] = *reinterpret_cast<const f16x8*>(qp + 16 * ct * ROWQ);
    };
    ldh(k_pass<GEMM1>(0, base, dilrow), 0);
#pragma unroll 1
    for (int ks = 0; ks < n_ks; ks += NSQ) {
      const KPass dc = k_pass<GEMM1>(ks, base, dilrow);
      const bool fin = ks + NSQ >= n_ks;
      const KPass dn = fin ? k_pass<NEXT1>(0, base, 0) : k_pass<GEMM1>(ks + NSQ, base, dilrow);
      const rsrc_t rsn = fin ? rs_next : rs;
      const int o = ks & 7;
#pragma unroll
      for (int s = 0; s < NSQ; ++s) {
        if (ROT > 0 && s == ROT % NSQ && ks == ROT - ROT % NSQ) {
          mid();
          ldh(dc, o + s);
        }
        ldl(dc, o + s);
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          BSG_MFMA_Q(c0[ct], A[s][1], Bh[ct]);
          BSG_MFMA_Q(c1[ct], A[s][3], Bh[ct]);
        }
        {
          const int kr = k_of(dn, o + s + NSQ) * QKSB;
          A[s][1] = lda8(rsn, vfrag, sa0 + kr + QPLB);
          A[s][3] = lda8(rsn, vfrag, sa1 + kr + QPLB);
          const int sp = (s + NSQ - 1) % NSQ;
          const int kp = k_of(s > 0 ? dn : dc, o + s - 1 + NSQ) * QKSB;
          A[sp][0] = lda8(s > 0 ? rsn : rs, vfrag, sa0 + kp);
          A[sp][2] = lda8(s > 0 ? rsn : rs, vfrag, sa1 + kp);
        }
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          BSG_MFMA_Q(c0[ct], A[s][0], Bh[ct]);
          BSG_MFMA_Q(c1[ct], A[s][2], Bh[ct]);
        }
        ldh(s + 1 < NSQ ? dc : dn, o + s + 1);
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          BSG_MFMA_Q(c0[ct], A[s][0], Bl[ct]);
          BSG_MFMA_Q(c1[ct], A[s][2], Bl[ct]);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read (lo plane of this step)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, NC / 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read (hi plane of the next step)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int kpe = k_of(k_pass<NEXT1>(0, base, 0), NSQ - 1) * QKSB;
    A[NSQ - 1][0] = lda8(rs_next, vfrag, sa0 + kpe);
    A[NSQ - 1][2] = lda8(rs_next, vfrag, sa1 + kpe);
    return;
  }
  f16x8 B[2][2 * NC];
  auto ldb = [&](const KPass& d, int i, f16x8 (&Bf)[2 * NC]) {
    const char* qp = bptr + d.boff + ((d.rot + i) & 7) * 64;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      Bf[2 * ct] = *reinterpret_cast<const f16x8*>(qp + 16 * ct * ROWQ);
      Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(qp + 16 * ct * ROWQ + bplane);
    }
  };
  ldb(k_pass<GEMM1>(0, base, dilrow), 0, B[0]);
#pragma unroll 1
  for (int ks = 0; ks < n_ks; ks += NSQ) {
    const KPass dc = k_pass<GEMM1>(ks, base, dilrow);
    const bool fin = ks + NSQ >= n_ks;   // the last pass: its reloads belong to the next GEMM
    const KPass dn = fin ? k_pass<NEXT1>(0, base, 0) : k_pass<GEMM1>(ks + NSQ, base, dilrow);
    const rsrc_t rsn = fin ? rs_next : rs;
    const int o = ks & 7;
#pragma unroll
    for (int s = 0; s < NSQ; ++s) {
      if (ROT > 0 && s == ROT % NSQ && ks == ROT - ROT % NSQ) {
        mid();
        ldb(dc, o + s, B[s & 1]);
      }
      // One k-step.  An MFMA of this shape occupies the pipe for 16 cycles and holds the vector issue for 8 of them: everything else of the
      // k-step (the next step's B fragments from LDS, the ring's reloads) is issued INSIDE those gaps — one LDS read behind each MFMA of the
      // first group (lo weights x hi operand: their registers are free for the reload right behind it), one reload behind every NC / 2 MFMAs of
      // the second; the hi weights' reload follows one k-step later, inside the next step's second group.  (With the loads outside the MFMA
      // sequence a k-step took 300 cycles for 192 cycles of matrix work.)
      ldb(s + 1 < NSQ ? dc : dn, o + s + 1, B[(s + 1) & 1]);
      const f16x8(&Bc)[2 * NC] = B[s & 1];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        BSG_MFMA_Q(c0[ct], A[s][1], Bc[2 * ct]);
        BSG_MFMA_Q(c1[ct], A[s][3], Bc[2 * ct]);
      }
      {
        const int kr = k_of(dn, o + s + NSQ) * QKSB;
        A[s][1] = lda8(rsn, vfrag, sa0 + kr + QPLB);
        A[s][3] = lda8(rsn, vfrag, sa1 + kr + QPLB);
        const int sp = (s + NSQ - 1) % NSQ;       // the previous k-step's slot: its hi weights (a constant in the unrolled loop)
        const int kp = k_of(s > 0 ? dn : dc, o + s - 1 + NSQ) * QKSB;
        A[sp][0] = lda8(s > 0 ? rsn : rs, vfrag, sa0 + kp);
        A[sp][2] = lda8(s > 0 ? rsn : rs, vfrag, sa1 + kp);
      }
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        BSG_MFMA_Q(c0[ct], A[s][0], Bc[2 * ct]);
        BSG_MFMA_Q(c1[ct], A[s][2], Bc[2 * ct]);
      }
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        BSG_MFMA_Q(c0[ct], A[s][0], Bc[2 * ct + 1]);
        BSG_MFMA_Q(c1[ct], A[s][2], Bc[2 * ct + 1]);
      }
#pragma unroll
      for (int i = 0; i < 2 * NC; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NC / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NC, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {   // the last k-step's hi weights: slot NSQ - 1 of the next GEMM
    const int kp = k_of(k_pass<NEXT1>(0, base, 0), NSQ - 1) * QKSB;
    A[NSQ - 1][0] = lda8(rs_next, vfrag, sa0 + kp);
    A[NSQ - 1][2] = lda8(rs_next, vfrag, sa1 + kp);
  }
}

template <int P, int W, int NC>
__global__ __launch_bounds__(64 * W, 1) void residual_part_h2_kernel(StackArgs p) {
  static_assert(P * W * 16 == C && (NC == 2 || NC == 4), "a wave owns 16 channels; 32- or 64-frame tiles");
  constexpr int NT = 16 * NC, XP = part_xp(NC), ZP = part_zp(NC);
  constexpr int NTH = 64 * W;                  // threads
  constexpr int TPT = NTH >= C ? 1 : C / NTH;  // table entries per thread
  constexpr int QCH = C / P;                   // channels per workgroup
  constexpr int CH8 = QCH / 8;                 // 16-byte chunks of 8 channels per frame of a slot
  constexpr int NSQ = W == 8 ? 4 : 8;          // weight ring in k-steps of 32 (one wave per SIMD: 512 registers)
  constexpr int OWN = 8 / P;                   // k-steps of a tap over the own channels
  constexpr int NPIECE = 2 * NT * CH8;         // 16-byte pieces of one exchange slot: 2 planes x NT frames x CH8 chunks
  static_assert(NPIECE % NTH == 0 && 1024 % NTH == 0 && (32 * (NT + 2 * HALO)) % NTH == 0, "copy loops");
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;                  // [2 planes][NT + 16 frames][528 B]: hi / lo of x + d_l, ALL channels
  char* zs = lds_raw + 2 * XP;         // [2 planes][NT frames][528 B]: hi / lo of 2^10 x gated activation, ALL channels
  float* dtab = reinterpret_cast<float*>(lds_raw + 2 * XP + 2 * ZP);   // [256]
  float* btab = dtab + C;                                              // [512]

  // workgroup -> (tile, part): the P parts of a tile sit on the same XCD (workgroup i runs on XCD i mod 8)
  const int n_tiles = p.n_tiles, per_xcd = (n_tiles + 7) >> 3;
  const int slot = (int)blockIdx.x >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + slot / P;
  const int q = slot % P;
  if (slot / P >= per_xcd || tile_id >= n_tiles) return;
  p.fbase = stack_epoch_take(p);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kb = lane >> 4;
  const int tpr = p.tiles_per_row, L = p.L, T = p.T;
  const int b = tile_id / tpr, j = tile_id - b * tpr;
  const int t0 = j * NT;
  const int tb = p.t_dev ? (int)p.t_dev[b] : p.t_uniform;
  const bool has_left = j > 0, has_right = j + 1 < tpr;
  const int cb = QCH * q + 16 * wave;   // first channel of this wave

  const unsigned plane = (unsigned)C * T * 4;
  const rsrc_t rs_x = mk_rsrc(p.x_in + (long long)b * C * T, plane);
  const int rowT = T * 4, vfrag = lane * 16;
  int vcol[NC], vst[NC], vq[NC];
  bool col_ok[NC];
#pragma unroll
  for (int ct = 0; ct < NC; ++ct) {
    const int col = t0 + 16 * ct + l15;
    col_ok[ct] = col < T;
    vcol[ct] = (kb * 4 * T + (col_ok[ct] ? col : T - 1)) * 4;   // accumulator rows 4 kb + r
    vst[ct] = (kb * 4 * T + col) * 4;
    vq[ct] = (kb * T + (col_ok[ct] ? col : T - 1)) * 16;        // the same rows as one channel quad
  }
  const int rt_g = W * q + wave;                                  // gate / residual row tile (of 16); filter / skip: + 16
  const int sa_g = rt_g * 1024, sa_f = (16 + rt_g) * 1024;

  float xr[NC][4], sk[NC][4];
  f32x4q yg[NC], yf[NC];
  int range_flag = 0;
  auto range_check = [&](unsigned worst) {
    if (__builtin_amdgcn_ballot_w64(worst >= 0x476A6000u) != 0ull) range_flag = 1;   // 60000.0f
  };
  auto absbits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };
  unsigned* fx = p.pflags;                   // image flags [n_tiles][P]
  unsigned* fz = p.pflags + P * n_tiles;     // z flags     [n_tiles][P]
  unsigned* xcc_tab = p.pflags + 2 * P * n_tiles;   // [n_tiles][P]: launch epoch + the XCC id the part runs on
  const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
  // A give-up of THIS launch is recorded twice: counted in status word 0 (cumulative until the host takes it) and, as the launch's own flag
  // base, in status word 2.  Only the second makes the other waits of the launch return at once: word 0 may still hold a give-up of an
  // EARLIER launch that the host has not taken yet (guard_mode 'deferred', captured replays, ABI users who only poll bsg_diffnet_status) —
  // a later launch must then wait for its partners as usual, not skip every hand-off and produce garbage too
  auto give_up = [&]() {
    atomicAdd(p.status, 1u);
    __hip_atomic_store(p.status + 2, p.fbase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto launch_gave_up = [&]() { return __hip_atomic_load(p.status + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.fbase; };
  auto wait_flags = [&](const unsigned* fl, unsigned want) {   // a whole wave: every lane with a flag polls its own; bounded
    bool pend = fl != nullptr;
    if (p.inject == 1) {
      if (pend) give_up();
      return;
    }
    if (p.inject == 3) return;   // timing experiment (tools/part_nowait.py; wrong results): no wait at all, nothing counted
    // a launch that has already counted a give-up (the host repeats the call anyway) waits for nothing any more — in particular not for
    // flags that partners on ANOTHER XCD store plainly and that never become visible here (checked first, below)
    if (launch_gave_up()) return;
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
      if (pend) pend = (int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0;
      if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
      __builtin_amdgcn_s_sleep(2);
      ++spins;
      const bool quit = spins > (1u << 22) || ((spins & 1023u) == 0u && launch_gave_up());
      if (quit) {
        if (pend) give_up();
        break;
      }
    }
  };

  // cache-policy bits of these loads: nt + sc0 for the pair form — the term is a stream read once per step that should not displace the weight
  // fragments in L2 (same-box A/B, profiles/r05_cq_aux_ab.log: B = 8 55.6 -> 53.1 ms per pass; the quads, whose 4 parts share one XCD's L2 with a
  // quarter of the weight stream each, lose 1-2 % with it)
#ifndef BSG_PQ_AUX
#define BSG_PQ_AUX 3
#endif
  constexpr int CQ_AUX = P == 2 ? BSG_PQ_AUX : 0;
  auto cond_request = [&](int l) {
    if (p.condterm_q) {
      // channel-quad order [2C/4][T][4] (round 5, as residual_stack_q_kernel): the 4 registers of an accumulator tile are ONE 16-byte load,
      // 256 B contiguous per 16 lanes — 2 NC requests per lane instead of 8 NC
      const rsrc_t rs_cq = mk_rsrc(p.condterm_q + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
      const int so = (cb >> 2) * T * 16;
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        yg[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cq, vq[ct], so, CQ_AUX));
        yf[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cq, vq[ct], so + (C / 4) * T * 16, CQ_AUX));
      }
      return;
    }
    const rsrc_t rs_ct = mk_rsrc(p.condterm + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int so = (cb + r) * rowT;
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        yg[ct][r] = ldf(rs_ct, vcol[ct], so);
        yf[ct][r] = ldf(rs_ct, vcol[ct], so + C * rowT);
      }
    }
  };
  auto write_core = [&]() {
    float dv[4];
    unsigned worst = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) dv[r] = dtab[cb + 4 * kb + r];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const float v0 = xr[ct][0] + dv[0], v1 = xr[ct][1] + dv[1], v2 = xr[ct][2] + dv[2], v3 = xr[ct][3] + dv[3];
      worst = max(max(worst, max(absbits(v0), absbits(v1))), max(absbits(v2), absbits(v3)));
      const HiLo s0 = split2(v0, v1);
      const HiLo s1_ = split2(v2, v3);
      u32x2 wh = u32x2{s0.hi, s1_.hi}, wl = u32x2{s0.lo, s1_.lo};
      if (!col_ok[ct]) { wh = u32x2{0u, 0u}; wl = u32x2{0u, 0u}; }
      char* dst = xs + (HALO + 16 * ct + l15) * ROWQ + (cb + 4 * kb) * 2;
      *reinterpret_cast<u32x2*>(dst) = wh;
      *reinterpret_cast<u32x2*>(dst + XP) = wl;
    }
    range_check(worst);
  };
  // a quarter (QCH channels) of NT LDS rows starting at row r0, both planes, to / from an exchange slot [plane][NT][QCH]
  auto part_out = [&](const char* img, int plane_bytes, int r0, int part, unsigned short* slot_p, auto aux_c) {
    constexpr int AUX = decltype(aux_c)::value;
    const rsrc_t rs = mk_rsrc(slot_p, 2 * NT * QCH * 2);
#pragma unroll
    for (int k = 0; k < NPIECE / NTH; ++k) {
      const int piece = k * NTH + tid;
      const int pl = piece / (NT * CH8), f = (piece / CH8) % NT, c8 = piece % CH8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + pl * plane_bytes + (r0 + f) * ROWQ + (QCH * part + 8 * c8) * 2);
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((pl * NT + f) * QCH + 8 * c8) * 2, 0, AUX);   // 16: sc1
    }
  };
  auto parts_in = [&](char* img, int plane_bytes, int r0, auto slot_of) {   // the P - 1 partners' parts
    u32x4 v[(P - 1) * NPIECE / NTH];
#pragma unroll
    for (int k = 0; k < (P - 1) * NPIECE / NTH; ++k) {
      const int piece = k * NTH + tid;
      const int o = piece / NPIECE, part = o < q ? o : o + 1;   // the o-th partner
      const int pc = piece % NPIECE, pl = pc / (NT * CH8), f = (pc / CH8) % NT, c8 = pc % CH8;
      const rsrc_t rs = mk_rsrc(slot_of(part), 2 * NT * QCH * 2);
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((pl * NT + f) * QCH + 8 * c8) * 2, 0, 16);   // sc1
    }
#pragma unroll
    for (int k = 0; k < (P - 1) * NPIECE / NTH; ++k) {
      const int piece = k * NTH + tid;
      const int o = piece / NPIECE, part = o < q ? o : o + 1;
      const int pc = piece % NPIECE, pl = pc / (NT * CH8), f = (pc / CH8) % NT, c8 = pc % CH8;
      *reinterpret_cast<u32x4*>(img + pl * plane_bytes + (r0 + f) * ROWQ + (QCH * part + 8 * c8) * 2) = v[k];
    }
  };
  const size_t slot_halfs = (size_t)2 * NT * QCH;   // fp16 elements of one exchange slot
  // the 8-frame edges of the image for the neighbouring tiles: [parity][tile][2 sides: first / last 8 core frames][2 planes][8][C] fp16 in
  // the edge-exchange array of the one-workgroup launches (p.hx: the same 16 KB per tile and parity)
  constexpr int EDGE_HALFS = 2 * 2 * 8 * C;
  auto edge_slot = [&](int par, int tile) { return reinterpret_cast<unsigned short*>(p.hx) + ((size_t)par * n_tiles + tile) * EDGE_HALFS; };
  auto edges_out = [&](int par) {   // the own channels of the first and the last 8 core frames, both planes: write-through
    const rsrc_t rs = mk_rsrc(edge_slot(par, tile_id), EDGE_HALFS * 2);
    static_assert(32 * CH8 <= NTH, "one 16-byte piece per thread");
    if (tid < 32 * CH8) {
      const int side = tid / (16 * CH8), pl = (tid / (8 * CH8)) & 1, f = (tid / CH8) & 7, c8 = tid % CH8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(xs + pl * XP + (HALO + (side ? NT - 8 : 0) + f) * ROWQ + (QCH * q + 8 * c8) * 2);
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (((side * 2 + pl) * 8 + f) * C + QCH * q + 8 * c8) * 2, 0, 16);   // sc1
    }
  };
  auto zx_slot = [&](int tile, int part) { return p.zx + ((size_t)tile * P + part) * slot_halfs; };
  // ONE image slot per part (round 5; until then one per layer parity): a part writes its image of layer l + 2 behind its GEMM2 of layer l + 1, which
  // needed the partners' z of that layer, which they published behind THEIR GEMM1 of layer l + 1 — i.e. after they had copied this part's image of
  // layer l + 1 into LDS.  Half the exchange footprint in the XCD's L2, which the weight fragments share (BSG_IX_SINGLE=0 at build time: two).
#ifndef BSG_IX_SINGLE
#define BSG_IX_SINGLE 1
#endif
  auto ix_slot = [&](int par, int tile, int part) { return p.ix + (((size_t)(BSG_IX_SINGLE ? 0 : par) * n_tiles + tile) * P + part) * slot_halfs; };

  // ---- layer 0: x from HBM — this wave's channels into registers, the WHOLE image (all channels, halo frames included) into LDS ----------
#pragma unroll
  for (int ct = 0; ct < NC; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xr[ct][r] = ldf(rs_x, vcol[ct], (cb + r) * rowT);
      sk[ct][r] = 0.f;
    }
  {
    const rsrc_t rs_dp = mk_rsrc(p.dproj + ((long long)tb * L + 0) * C, C * 4);
    unsigned worst = 0;
    constexpr int ROWS = NT + 2 * HALO;
#pragma unroll 1
    for (int it = 0; it < 32 * ROWS / NTH; ++it) {   // 32 chunks of 8 channels x ROWS frames, lanes = consecutive frames
      const int item = it * NTH + tid;
      const int hc = item / ROWS, row = item - hc * ROWS;
      const int th = t0 - HALO + row;
      const bool hok = th >= 0 && th < T;
      float hv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) hv[k] = ldf(rs_x, hok ? ((8 * hc + k) * T + th) * 4 : 0, 0) + ldf(rs_dp, (8 * hc + k) * 4, 0);
      if (hok) worst = max(worst, max(max(max(absbits(hv[0]), absbits(hv[1])), max(absbits(hv[2]), absbits(hv[3]))),
                                      max(max(absbits(hv[4]), absbits(hv[5])), max(absbits(hv[6]), absbits(hv[7])))));
      const HiLo h0 = split2(hv[0], hv[1]), h1 = split2(hv[2], hv[3]), h2 = split2(hv[4], hv[5]), h3 = split2(hv[6], hv[7]);
      u32x4 wh = u32x4{h0.hi, h1.hi, h2.hi, h3.hi}, wl = u32x4{h0.lo, h1.lo, h2.lo, h3.lo};
      if (!hok) { wh = u32x4{0u, 0u, 0u, 0u}; wl = u32x4{0u, 0u, 0u, 0u}; }
      *reinterpret_cast<u32x4*>(xs + row * ROWQ + hc * 16) = wh;
      *reinterpret_cast<u32x4*>(xs + XP + row * ROWQ + hc * 16) = wl;
    }
    range_check(worst);
  }
  for (int i = tid; i < C; i += NTH) dtab[i] = p.dproj[((long long)tb * L + 0) * C + i];
  for (int i = tid; i < 2 * C; i += NTH) btab[i] = p.bias_out[i];
  cond_request(0);
  f16x8 A[NSQ][4];
  auto prefetch_a1 = [&](int l) {
    const rsrc_t rs = mk_rsrc(p.apack1q + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
#pragma unroll
    for (int k = 0; k < NSQ; ++k) {
      const int kr = k_of(k_pass<true>(0, OWN * q, 0), k);
      A[k][0] = lda8(rs, vfrag, sa_g + kr * QKSB);
      A[k][1] = lda8(rs, vfrag, sa_g + kr * QKSB + QPLB);
      A[k][2] = lda8(rs, vfrag, sa_f + kr * QKSB);
      A[k][3] = lda8(rs, vfrag, sa_f + kr * QKSB + QPLB);
    }
  };
  prefetch_a1(0);
  // (p.inject == 2, fault injection: odd parts publish ANOTHER id, as if the dispatcher had placed them on another XCD)
  if (tid == 0) __hip_atomic_store(xcc_tab + P * tile_id + q, p.fbase + (p.inject == 2 && (q & 1) ? my_xcc ^ 1u : my_xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p.clk && tile_id == 0 && q == 0 && tid == 0) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  __syncthreads();   // the staged image and the tables

#define QUAD_STAMP(i)                                                                                                      \
  do {                                                                                                                    \
    if (p.stamps && tid == 0 && q == 0) p.stamps[((long long)tile_id * L + l) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int dil = 1 << (l % p.cycle);
    const rsrc_t rs_a1 = mk_rsrc(p.apack1q + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
    const rsrc_t rs_a2 = mk_rsrc(p.apack2q + (long long)l * (2 * 2 * C * C), 2 * 2 * C * C * 2);
    const rsrc_t rs_a1n = mk_rsrc(p.apack1q + (long long)(l + 1 < L ? l + 1 : 0) * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);   // the next layer's GEMM1
    const float s1 = p.h2_scale[4 * l], inv1 = p.h2_scale[4 * l + 1], s2 = p.h2_scale[4 * l + 2], inv2 = p.h2_scale[4 * l + 3];
    float dnext[TPT], bnext0[TPT], bnext1[TPT];   // the next layer's tables: entries tid + k NTH (W = 8: threads 256.. repeat the first half's)
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int tc = (tid + k * NTH) & (C - 1);
      dnext[k] = l + 1 < L ? p.dproj[((long long)tb * L + l + 1) * C + tc] : 0.f;
      bnext0[k] = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + tc] : 0.f;
      bnext1[k] = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + 256 + tc] : 0.f;
    }
    // the conditioner term landed in the accumulators (requested behind the previous layer's image flag).  (Requested inside GEMM1 and added
    // behind it — tried for the forms with one wave per SIMD — it blocks the weight ring: vmcnt counts in order, so every ring load issued
    // behind the HBM request completes behind it.  Quad of 64-frame tiles: 16.1 -> 13.8 us per layer without it)
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) { yg[ct] *= s1; yf[ct] *= s1; }
    QUAD_STAMP(0);
    // ---- GEMM1: 24 k-steps of 32; those of the centre tap over the own channels first, behind them the partners' parts + the halo ----
    {
      const char* xb = xs + (HALO + l15) * ROWQ + kb * 16;
      auto mid = [&]() {
        if (l == 0) return;   // layer 0 staged the whole image from HBM
        if (wave == 0) {
          // 3 P - 1 flags (the partners, the P parts of each neighbouring tile), polled by as many lanes at once
          const unsigned want = p.fbase + (unsigned)l;
          const unsigned* fl = nullptr;
          if (lane < P - 1) fl = fx + P * tile_id + (lane < q ? lane : lane + 1);
          else if (lane < 2 * P - 1) fl = has_left ? fx + P * (tile_id - 1) + (lane - (P - 1)) : nullptr;
          else if (lane < 3 * P - 1) fl = has_right ? fx + P * (tile_id + 1) + (lane - (2 * P - 1)) : nullptr;
          wait_flags(fl, want);
        }
        __syncthreads();   // (D) the polling lanes have seen the flags
        if (!p.stamp_mode) QUAD_STAMP(1);
        parts_in(xs, XP, HALO, [&](int part) { return ix_slot(l & 1, tile_id, part); });   // the partners' channels of the core frames
        {
          // halo rows, both planes, all channels: rows 0..7 = the left tile's last 8 frames (its edge slot 1), rows NT+8..NT+15 = the right
          // tile's first 8 (its edge slot 0); a neighbour may sit on another XCD: write-through stores there, L1-bypassing loads here
          u32x4 v[1024 / NTH];
#pragma unroll
          for (int k = 0; k < 1024 / NTH; ++k) {
            const int piece = k * NTH + tid;   // 2 sides x 2 planes x 8 frames x 32 chunks of 8 channels = 1024
            const int side = piece >> 9, pl = (piece >> 8) & 1, f = (piece >> 5) & 7, c8 = piece & 31;
            const bool have = side == 0 ? has_left : has_right;
            v[k] = u32x4{0u, 0u, 0u, 0u};
            if (have) {
              const rsrc_t rs = mk_rsrc(edge_slot(l & 1, side == 0 ? tile_id - 1 : tile_id + 1), EDGE_HALFS * 2);
              v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((((1 - side) * 2 + pl) * 8 + f) * C + 8 * c8) * 2, 0, 16);   // sc1
            }
          }
#pragma unroll
          for (int k = 0; k < 1024 / NTH; ++k) {
            const int piece = k * NTH + tid;
            const int side = piece >> 9, pl = (piece >> 8) & 1, f = (piece >> 5) & 7, c8 = piece & 31;
            *reinterpret_cast<u32x4*>(xs + pl * XP + ((side ? HALO + NT : 0) + f) * ROWQ + 8 * c8 * 2) = v[k];
          }
        }
        __syncthreads();   // (A) the whole image is in place
        if (!p.stamp_mode) QUAD_STAMP(2);
      };
      mfma_pipe_part<true, false, OWN, NC, NSQ, W == 8>(yg, yf, A, rs_a1, rs_a2, vfrag, sa_g, sa_f, 24, OWN * q, dil * ROWQ, xb, XP, mid);
    }
    QUAD_STAMP(3);
    // ---- gate -> own quarter of zs (hi / lo of 2^10 z) ----------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < TPT; ++k)
      if (tid + k * NTH < C) dtab[tid + k * NTH] = dnext[k];
    const float rs2 = inv2 * 0.70710678118654752440f;
    const float gcg = -1.44269504088896340736f * inv1, gcf = -2.88539008177792681472f * inv1, glim = 15.0f * s1;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const f32x2 z01 = gate2_scaled(f32x2{yg[ct][0], yg[ct][1]}, f32x2{yf[ct][0], yf[ct][1]}, gcg, gcf, glim, ZSCALE);
      const f32x2 z23 = gate2_scaled(f32x2{yg[ct][2], yg[ct][3]}, f32x2{yf[ct][2], yf[ct][3]}, gcg, gcf, glim, ZSCALE);
      const HiLo s0 = split2(z01[0], z01[1]), s1_ = split2(z23[0], z23[1]);
      char* dst = zs + (16 * ct + l15) * ROWQ + (cb + 4 * kb) * 2;
      *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
      *reinterpret_cast<u32x2*>(dst + ZP) = u32x2{s0.lo, s1_.lo};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float br = btab[cb + 4 * kb + r], bs = btab[C + cb + 4 * kb + r];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        yg[ct][r] = (xr[ct][r] + br) * s2;
        yf[ct][r] = bs * s2;
      }
    }
    if (p.stamp_mode) QUAD_STAMP(1);   // gate evaluated, own z in LDS (this wave)
    __syncthreads();   // (Z1) the own quarter of z is complete in LDS; every wave is done reading xs and this layer's biases
#pragma unroll
    for (int k = 0; k < TPT; ++k)
      if (tid + k * NTH < C) {
        btab[tid + k * NTH] = bnext0[k];
        btab[tid + k * NTH + 256] = bnext1[k];
      }
    // z is read by the partners only, and they sit on THIS XCD (checked below): plain stores, which keep the lines in the XCD's L2 — the
    // partners' L1-bypassing loads are served there at the same-XCD rate; write-through stores would drop them from L2
    // (MI355X_MICROARCH.md, stores of each flavour)
    if (!(p.inject == 1 && (tile_id & 1))) part_out(zs, ZP, 0, q, zx_slot(tile_id, q), std::integral_constant<int, 0>{});
    if (p.stamp_mode) QUAD_STAMP(2);   // barrier Z1 passed, the z part copied out of LDS (stores issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have reached L2
    __syncthreads();
    if (tid == 0) {   // the z flag is polled by the partners only (this XCD): a plain store too — it stays in L2, where their polls are served
      const rsrc_t rs_f = mk_rsrc(fz + P * tile_id, P * 4);
      __builtin_amdgcn_raw_buffer_store_b32(p.fbase + (unsigned)(l + 1), rs_f, q * 4, 0, 0);
    }
    QUAD_STAMP(4);
    // ---- GEMM2: 8 k-steps of 32; yg = residual rows, yf = skip rows of the own channels.  Like GEMM1 it starts with the k-steps over the
    // OWN channels of z — in LDS since (Z1) — and takes the partners' parts behind them: the way of the own z flag to the partners, their
    // polls and the way of their parts back run under those MFMAs (round 4; before, GEMM2 started behind the whole exchange) -------------
    {
      const char* zb = zs + l15 * ROWQ + kb * 16;
      auto mid2 = [&]() {
        if (wave == 0) {
          if (l == 0) {
            // once per launch, BEFORE the first wait on a plainly stored flag: do the partners really run on this XCD?  Each part stored
            // launch epoch + XCC id with an agent-scope store at its start (visible from every XCD), so a bounded poll of those words
            // terminates wherever the partner runs; another id than ours — another dispatch order than workgroup i -> XCD i mod 8 —
            // counts a give-up at once (round 4 looked only behind the first flag wait: the partner's plain flag store never became
            // visible here and the mismatch was noticed after a full bounded spin, seconds).  The host then takes the part forms off
            // this handle and repeats the call on the one-workgroup-per-tile launch (DiffNet.guarded)
            const unsigned* xw = lane < P - 1 ? xcc_tab + P * tile_id + (lane < q ? lane : lane + 1) : nullptr;
            bool pend = xw != nullptr && p.inject != 1 && p.inject != 3;
            unsigned theirs = 0, spins = 0;
            while (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
              if (pend) {
                theirs = __hip_atomic_load(xw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend = theirs - p.fbase >= 16u;   // not yet this launch's word (epoch x 64 + id)
              }
              if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
              __builtin_amdgcn_s_sleep(2);
              if (++spins > (1u << 22) || ((spins & 1023u) == 0u && launch_gave_up())) break;
            }
            if (xw != nullptr && p.inject != 1 && p.inject != 3 && (pend || theirs != p.fbase + my_xcc)) give_up();
          }
          wait_flags(lane < P - 1 ? fz + P * tile_id + (lane < q ? lane : lane + 1) : nullptr, p.fbase + (unsigned)(l + 1));
        }
        __syncthreads();   // the partners' parts of z are published
        parts_in(zs, ZP, 0, [&](int part) { return zx_slot(tile_id, part); });
        __syncthreads();   // (B) zs complete
        QUAD_STAMP(5);
      };
      mfma_pipe_part<false, true, OWN, NC, NSQ, W == 8>(yg, yf, A, rs_a2, rs_a1n, vfrag, sa_g, sa_f, 8, OWN * q, 0, zb, ZP, mid2);
    }
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xr[ct][r] = yg[ct][r] * rs2;
        sk[ct][r] += yf[ct][r] * inv2;
      }
    QUAD_STAMP(6);
    if (l + 1 == L) break;
    // ---- next layer: the own quarter of the image into LDS and to the exchange slot, the flag, then the conditioner term and the weights ----
    write_core();
    __syncthreads();   // (C1) the own quarter of the core rows is complete
    // the own part of the image: plain stores for the partners (this XCD, like z), the edges once more write-through for the neighbours
    if (!(p.inject == 1 && (tile_id & 1))) {
      part_out(xs, XP, HALO, q, ix_slot((l + 1) & 1, tile_id, q), std::integral_constant<int, 0>{});
      edges_out((l + 1) & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (C)
    if (tid == 0) __hip_atomic_store(fx + P * tile_id + q, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    QUAD_STAMP(7);
    cond_request(l + 1);   // (behind the flag: in front of the drain its way from HBM would delay the flag — vmcnt counts in order)
  }
#undef QUAD_STAMP
  if (tid == 0) stack_epoch_done(p, p.fbase, P * n_tiles);
  if (p.clk && tile_id == 0 && q == 0 && tid == 0) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); }
  if (range_flag && lane == 0) atomicAdd(p.status + 1, 1u);
  // ---- the skip sum / sqrt(L) of the own channels (net.py:126), fp32 [C][T] rows: what step_tail_kernel reads ---------------------------
  const rsrc_t rs_sk = mk_rsrc(p.skip + (long long)b * C * T, plane);
  const float rdiv = 1.0f / sqrtf((float)L);
#pragma unroll
  for (int ct = 0; ct < NC; ++ct)
    if (col_ok[ct]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) stf(sk[ct][r] * rdiv, rs_sk, vst[ct], (cb + r) * rowT);
    }
}
#undef BSG_MFMA_Q


}  // namespace

template <int NCT>
static int h2_occupancy() {
  int o = 0;
  const int lds = (int)h2_lds(NCT);
  if (hipFuncSetAttribute((const void*)residual_stack_h2_kernel<true, false, NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
      hipFuncSetAttribute((const void*)residual_stack_h2_kernel<true, true, NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)residual_stack_h2_kernel<true, true, NCT>, 512, h2_lds(NCT)) != hipSuccess)
    return 0;
  return o;
}
// resident workgroups per CU (0 on error) of the form with `nct` column tiles of 32 frames per workgroup (1 or 2)
int stack_h2_occupancy(int nct) { return nct == 1 ? h2_occupancy<1>() : h2_occupancy<2>(); }

// (FAIRB = the time-sliced issue priority between the two waves of a SIMD is always on: +1.2 % in round 2's A/B; the switch is gone)
template <int NCT>
static int h2_launch(const StackArgs& p, const TailArgs* tail, hipStream_t st) {
  const dim3 grid(8 * cdiv(p.n_tiles, 8)), block(512);
  const TailArgs a = tail ? *tail : TailArgs{};
  const size_t lds = h2_lds(NCT);
  if (tail) hipLaunchKernelGGL((residual_stack_h2_kernel<true, true, NCT>), grid, block, lds, st, p, a);
  else hipLaunchKernelGGL((residual_stack_h2_kernel<true, false, NCT>), grid, block, lds, st, p, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// tail == nullptr: the residual stack only (skip sum to p.skip); else the sampler step's tail runs in the same launch (TailArgs of THIS
// launch's rows: x, noise, xa_next, history pointers and quad_row0 already offset to its first row).  nct = column tiles of 32 frames per
// workgroup: p.tiles_per_row / p.n_tiles count tiles of 32 * nct frames
int launch_residual_stack_h2(const StackArgs& p, const TailArgs* tail, hipStream_t st, int nct) {
  return nct == 1 ? h2_launch<1>(p, tail, st) : h2_launch<2>(p, tail, st);
}

// the step tail behind a part launch: one workgroup per 32-frame tile (a.tiles_per_row = ceil(T / 32)); a.status: the launch's status words
int launch_step_tail_h2(const TailArgs& a, hipStream_t st) {
  static bool attr = false;
  const size_t lds = (size_t)2 * h2_xp(1) + 2 * h2_zp(1) + 96 * 32 * sizeof(float);
  if (!attr) {
    BSG_HIP(hipFuncSetAttribute((const void*)step_tail_h2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  BSG_REQUIRE(a.ws_s && a.wo_s && a.wi_s && a.tail_scale && a.M <= 96, "split-fp16 step tail: fragments missing or in_dims > 96");
  hipLaunchKernelGGL(step_tail_h2_kernel, dim3(a.B * a.tiles_per_row), dim3(512), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

template <int P, int W, int NC>
static int part_occ() {
  int o = 0;
  const size_t lds = part_lds(NC);
  if (hipFuncSetAttribute((const void*)residual_part_h2_kernel<P, W, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)residual_part_h2_kernel<P, W, NC>, 64 * W, lds) != hipSuccess)
    return 0;
  return o;
}
// resident workgroups per CU of the part form (parts per tile, tile width in units of 32 frames): (4, 1) quad of 32 frames, (4, 2) quad of 64,
// (2, 2) pair of 64
int part_h2_occupancy(int parts, int nct) {
  if (parts == 4 && nct == 1) return part_occ<4, 4, 2>();
  if (parts == 4 && nct == 2) return part_occ<4, 4, 4>();
  if (parts == 2 && nct == 2) return part_occ<2, 8, 4>();
  return 0;
}
// part forms: `parts` workgroups per tile of 32 nct frames; grid = 8 parts ceil(n_tiles / 8) workgroups, all resident (one per CU)
int launch_residual_part_h2(const StackArgs& p, hipStream_t st, int parts, int nct) {
  BSG_REQUIRE(p.zx && p.ix && p.hx && p.pflags && p.apack1q && p.apack2q, "part launch: exchange buffers / 16-row weight fragments missing");
  const dim3 grid(8 * parts * cdiv(p.n_tiles, 8));
  if (parts == 4 && nct == 1) hipLaunchKernelGGL((residual_part_h2_kernel<4, 4, 2>), grid, dim3(256), part_lds(2), st, p);
  else if (parts == 4 && nct == 2) hipLaunchKernelGGL((residual_part_h2_kernel<4, 4, 4>), grid, dim3(256), part_lds(4), st, p);
  else if (parts == 2 && nct == 2) hipLaunchKernelGGL((residual_part_h2_kernel<2, 8, 4>), grid, dim3(512), part_lds(4), st, p);
  else BSG_REQUIRE(false, "part launch: no form for %d parts of %d-frame tiles", parts, 32 * nct);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
// fp32 [M][K] weights -> hi / lo fp16 A fragments of v_mfma_f32_16x16x32_f16 (16-row tiles), scaled by the layer's table entry
int pack_a_frag_q(const float* src, unsigned short* out, int M, int K, int Kc, long long sm, long long sc, long long stp, const float* tab,
                  int is_gemm2, hipStream_t st) {
  const long long total = (long long)M * K;
  hipLaunchKernelGGL(pack_a_frag_q_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, src, reinterpret_cast<_Float16*>(out), M, K, Kc, sm, sc,
                     stp, is_gemm2 ? tab + 2 : tab, is_gemm2);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// the three projections of the step tail as split-fp16 fragments + their scale table [3][2]; maxbits: [3] scratch
int h2_tail_pack(const float* ws, const float* wo96, const float* wi96, unsigned short* out_ws, unsigned short* out_wo, unsigned short* out_wi,
                 unsigned* maxbits, float* tab, hipStream_t st) {
  BSG_HIP(hipMemsetAsync(maxbits, 0, 3 * sizeof(unsigned), st));
  hipLaunchKernelGGL(h2_absmax_kernel, dim3(64), dim3(256), 0, st, ws, (long long)C * C, maxbits);
  hipLaunchKernelGGL(h2_absmax_kernel, dim3(64), dim3(256), 0, st, wo96, (long long)96 * C, maxbits + 1);
  hipLaunchKernelGGL(h2_absmax_kernel, dim3(64), dim3(256), 0, st, wi96, (long long)C * 96, maxbits + 2);
  hipLaunchKernelGGL(h2_tail_scale_kernel, dim3(1), dim3(64), 0, st, (const unsigned*)maxbits, tab);
  hipLaunchKernelGGL(pack_a_frag_h2_kernel, dim3(cdiv((long long)C * C, 256)), dim3(256), 0, st, ws, reinterpret_cast<_Float16*>(out_ws), C, C, C,
                     (long long)C, 1LL, 0LL, (const float*)tab, 0);
  hipLaunchKernelGGL(pack_a_frag_h2_kernel, dim3(cdiv((long long)96 * C, 256)), dim3(256), 0, st, wo96, reinterpret_cast<_Float16*>(out_wo), 96, C, C,
                     (long long)C, 1LL, 0LL, (const float*)(tab + 2), 0);
  hipLaunchKernelGGL(pack_a_frag_h2_kernel, dim3(cdiv((long long)C * 96, 256)), dim3(256), 0, st, wi96, reinterpret_cast<_Float16*>(out_wi), C, 96, 96,
                     96LL, 1LL, 0LL, (const float*)(tab + 4), 0);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// scale table of all layers: maxbits [2L] scratch (zeroed here), tab [4L]; w1[l] = dilated conv [2C][C][3], w2[l] = output projection [2C][C]
int h2_scales(const float* const* w1, const float* const* w2, int L, unsigned* maxbits, float* tab, hipStream_t st) {
  BSG_HIP(hipMemsetAsync(maxbits, 0, (size_t)2 * L * sizeof(unsigned), st));
  for (int l = 0; l < L; ++l) {
    hipLaunchKernelGGL(h2_absmax_kernel, dim3(64), dim3(256), 0, st, w1[l], (long long)2 * C * 3 * C, maxbits + 2 * l);
    hipLaunchKernelGGL(h2_absmax_kernel, dim3(64), dim3(256), 0, st, w2[l], (long long)2 * C * C, maxbits + 2 * l + 1);
  }
  hipLaunchKernelGGL(h2_scale_kernel, dim3(cdiv(2 * L, 64)), dim3(64), 0, st, (const unsigned*)maxbits, tab, L);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// fp32 [M][K] weights -> hi / lo fp16 A fragments of v_mfma_f32_32x32x16_f16, scaled by the layer's table entry (`tab` = the 4 floats
// of the layer; is_gemm2 selects s2)
int pack_a_frag_h2(const float* src, unsigned short* out, int M, int K, int Kc, long long sm, long long sc, long long stp, const float* tab,
                   int is_gemm2, hipStream_t st) {
  const long long total = (long long)M * K;
  hipLaunchKernelGGL(pack_a_frag_h2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, src, reinterpret_cast<_Float16*>(out), M, K, Kc, sm, sc,
                     stp, is_gemm2 ? tab + 2 : tab, is_gemm2);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

}  // namespace bsg
