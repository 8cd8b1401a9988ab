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
#include "diffnet_res.h"
#include "diffnet_tail.h"

namespace bsg {

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// NCT = column tiles of 32 frames per workgroup: 2 (64-frame tiles, the form described above) or 1 (32-frame tiles: twice the tiles for
// batches that would leave CUs without one — half the matrix work per tile against the same weight stream)
constexpr int ROWB = 2 * C + 16;          // LDS image row: 256 fp16 + 16 B pad = 528 B (132 dwords = 4 mod 64)
constexpr int h2_xp(int nct) { return (32 * nct + 2 * HALO) * ROWB; }   // bytes per plane of the image: 42,240 (NCT = 2)
constexpr int h2_zp(int nct) { return 32 * nct * ROWB; }                // bytes per plane of z: 33,792 (NCT = 2)
constexpr int NSH = 4;                    // weight ring (k-steps).  A ring of 8 for the 32-frame form (it has the registers) measured 5-7 % SLOWER at
                                          // B = 1, 4, 8 (profiles/r03_ab_ring.log): the L2 latency of the weight stream is already covered
constexpr int PLB = 16 * 1024;            // bytes per plane of a k-step slab (16 row tiles x 1 KB)
constexpr int KSB2 = 2 * PLB;             // bytes per k-step: hi slab, lo slab
constexpr float ZSCALE = 1024.0f;         // z in (-1, 1) is split as z x 2^10
constexpr size_t h2_lds(int nct) { return (size_t)2 * h2_xp(nct) + 2 * h2_zp(nct) + 3 * C * sizeof(float); }

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

__device__ __forceinline__ f16x8 lda8(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// hi / lo split of two values into two packed dwords
struct HiLo { unsigned hi, lo; };
__device__ __forceinline__ HiLo split2(float a, float b) {
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  return HiLo{__builtin_bit_cast(unsigned, f16x2{ha, hb}),
              __builtin_bit_cast(unsigned, f16x2{(_Float16)(a - (float)ha), (_Float16)(b - (float)hb)})};
}

#define BSG_MFMA_H(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, ACC, 0, 0, 0)

// i-th executed k-step -> k-step index: GEMM1 (ROT = 16, 48 k-steps, tap-major) starts with the CENTRE tap, whose B operand is the
// tile's own 64 frames, and visits the two outer taps (which read the neighbours' halo frames) afterwards
template <int ROT>
__device__ __forceinline__ int kmap(int i) {
  if (ROT == 0) return i;
  return i < ROT ? i + ROT : (i < 2 * ROT ? i - ROT : i);
}

// k-step pipeline over two row tiles x two column tiles, 12 MFMAs per k-step (hi hi, hi lo, lo hi for each of the 4 accumulators; an
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
      const int in = ks + s + 1 <= last ? ks + s + 1 : last;
      ldb(kmap<ROT>(in), B[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const f16x8(&Bc)[2 * NCT] = B[s & 1];
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
      for (int ct = 0; ct < NCT; ++ct) {   // lo hi
        BSG_MFMA_H(c0[ct], A[s][1], Bc[2 * ct]);
        BSG_MFMA_H(c1[ct], A[s][3], Bc[2 * ct]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int ir = ks + s + NSH <= last ? ks + s + NSH : last;
      const int kr = kmap<ROT>(ir);
      A[s][0] = lda8(rs, vfrag, sa0 + kr * KSB2);
      A[s][1] = lda8(rs, vfrag, sa0 + kr * KSB2 + PLB);
      A[s][2] = lda8(rs, vfrag, sa1 + kr * KSB2);
      A[s][3] = lda8(rs, vfrag, sa1 + kr * KSB2 + PLB);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Small split-fp16 GEMM of the fused step tail: one row tile of 32 x NC column tiles of 32, N_KS k-steps of 16, fully unrolled; A fragments
// (hi at sa + ks*ksb, lo at + plb) in a ring of up to 8 k-steps, B fragments from `ldb(ks, Bf)`: Bf[2 nc] = hi, Bf[2 nc + 1] = lo of column
// tile nc.
template <int NC, int N_KS, typename LDB>
__device__ __forceinline__ void tail_gemm_h2(f32x16 (&c)[NC], rsrc_t rs, int vfrag, int sa, int ksb, int plb, LDB ldb) {
  constexpr int R = N_KS < 8 ? N_KS : 8;
  f16x8 Ar[R][2];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    Ar[k][0] = lda8(rs, vfrag, sa + k * ksb);
    Ar[k][1] = lda8(rs, vfrag, sa + k * ksb + plb);
  }
  f16x8 Bf[2][2 * NC];
  ldb(0, Bf[0]);
#pragma unroll
  for (int ks = 0; ks < N_KS; ++ks) {
    if (ks + 1 < N_KS) ldb(ks + 1, Bf[(ks + 1) & 1]);
    const f16x8(&Bc)[2 * NC] = Bf[ks & 1];
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) BSG_MFMA_H(c[nc], Ar[ks % R][0], Bc[2 * nc]);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) BSG_MFMA_H(c[nc], Ar[ks % R][0], Bc[2 * nc + 1]);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) BSG_MFMA_H(c[nc], Ar[ks % R][1], Bc[2 * nc]);
    if (ks + R < N_KS) {
      Ar[ks % R][0] = lda8(rs, vfrag, sa + (ks + R) * ksb);
      Ar[ks % R][1] = lda8(rs, vfrag, sa + (ks + R) * ksb + plb);
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
    if (p.stamps && tid == 0) p.stamps[((long long)tile_id * L + l) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
  if (p.stamps && tid == 0) p.stamps[((long long)tile_id * L + L - 1) * 8 + 6] = __builtin_amdgcn_s_memtime();   // ... and start of the first (tools/stack_stamps.py)
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
          if (p.inject) {
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
            if (p.inject) { atomicAdd(p.status, 1u); continue; }
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
        STK_STAMP(1);
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
        STK_STAMP(2);
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
    write_core();
    __syncthreads();   // (C1) the core rows are complete (every wave wrote its 32 channels of every frame)
    STK_STAMP(6);
    {
      // publish the first and the last 8 frames of both planes: [plane][side][8 frames][256 ch] fp16 = 16 KB, write-through
      unsigned short* hx_t = reinterpret_cast<unsigned short*>(p.hx) + ((long long)((l + 1) & 1) * n_tiles + tile_id) * (4 * 8 * C);
      const int side = tid >> 8, f = (tid >> 5) & 7, c16 = tid & 31;
      const char* srcp = xs + (HALO + (side ? NT - 8 : 0) + f) * ROWB + c16 * 16;
      const u32x4 vh = *reinterpret_cast<const u32x4*>(srcp);
      const u32x4 vl = *reinterpret_cast<const u32x4*>(srcp + XP);
      if (!(p.inject && (tile_id & 1))) {
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
  if (p.stamps && tid == 0) p.stamps[((long long)tile_id * L + L - 1) * 8 + 7] = __builtin_amdgcn_s_memtime();   // shader clock: end of the last layer ...
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
            if (!a.noise && a.k.sigma != 0.f)
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
// The step tail of the pair / quad forms as its own launch on the 16-bit matrix pipe: what the TAIL branch of residual_stack_h2_kernel does on
// chip, from the skip sum in HBM (fp32 [C][T], written by the last layer of the pair / quad launch).  One workgroup of 8 waves per 32-frame
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
  // ---- s (skip sum / sqrt(L), fp32 rows) -> hi / lo image rows: 32 chunks of 8 channels x 32 frames, lanes = consecutive frames ------------
  {
    const rsrc_t rs_s = mk_rsrc(a.skip + (long long)b * C * T, plane);
    unsigned worst = 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = it * 512 + tid;
      const int hc = item >> 5, f = item & 31;
      const int t = t0 + f;
      const bool ok = t < T;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ldf(rs_s, ok ? ((8 * hc + k) * T + t) * 4 : 0, 0);
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
  const char* xcore = xs + (HALO + l31) * ROWB + lh * 16;
  auto ldb_x = [&](int ks, f16x8 (&Bf)[2]) {
    Bf[0] = *reinterpret_cast<const f16x8*>(xcore + ks * 32);
    Bf[1] = *reinterpret_cast<const f16x8*>(xcore + ks * 32 + XP);
  };
  // ---- h = relu(W_skip s + b) -> zs (hi / lo)                                                                          (net.py:126-128) ----
  {
    const rsrc_t rs_ws = mk_rsrc(a.ws_s, 2 * C * C * 2);
    const rsrc_t rs_bs = mk_rsrc(a.b_skip, C * 4);
    const float sc = tsc[0], inv = tsc[1];
    f32x16 hc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) hc[0][r] = ldf(rs_bs, lh * 16, (32 * wave + acc_row0(r)) * 4) * sc;
    __syncthreads();   // (T1) s complete
    tail_gemm_h2<1, 16>(hc, rs_ws, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024, ldb_x);
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
    const rsrc_t rs_wi = mk_rsrc(a.wi_s, 2 * C * 96 * 2);
    const rsrc_t rs_bi = mk_rsrc(a.b_in, C * 4);
    const float sc = tsc[4], inv = tsc[5];
    f32x16 hc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) hc[0][r] = ldf(rs_bi, lh * 16, (32 * wave + acc_row0(r)) * 4) * sc;
    __syncthreads();   // (T3) the updated x tile is complete
    tail_gemm_h2<1, 6>(hc, rs_wi, vfrag, wave * 1024, 2 * 8 * 1024, 8 * 1024, ldb_x);
    const rsrc_t rs_xa = mk_rsrc(a.xa_next + (long long)b * C * T, plane);
    if (col_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stf(fmaxf(hc[0][r] * inv, 0.f), rs_xa, vst, (32 * wave + acc_row0(r)) * rowT);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PAIR form for small batches (B * ceil(T / 32) <= CUs / 2; B <= 4 at T = 1000).  With one workgroup per 32-frame tile a single utterance
// keeps 32 of the 256 CUs busy, and each of them is bound by streaming the layer's 2.1 MB of weight fragments out of L2 (the two GEMMs
// take 18.8 us of a 25.5-us layer where their MFMAs need 10; tools/l2_fill.hip: a CU pulls at most ~64 B/clk).  Here a tile is computed by
// TWO workgroups of 4 waves on two CUs of one XCD, each owning HALF of the channels — its gate / filter rows of GEMM1, its residual / skip
// rows of GEMM2, its 128 channels of x and of the skip sum in registers — so every CU streams half of the weights.  What a workgroup
// lacks it gets from its partner through L2, twice per layer, with the hand-off protocol of the launch above (write-through stores,
// drain, barrier, flag = launch epoch + layer, bounded poll, sc1 loads):
//   * after the gate: the partner's half of z (16 KB, both planes) — GEMM2 contracts over all 256 channels;
//   * after GEMM2: the partner's half of the next image (16 KB) and, from the two neighbouring tiles' workgroups, the 8-frame edges of
//     both halves — GEMM1 contracts over all 256 channels of 48 frames.  GEMM1 starts with the centre tap of its OWN channels (8 k-steps),
//     the only part of the image a workgroup has without waiting.
// The step tail is not fused (its skip projection contracts over both halves): the skip sum goes to HBM and step_tail_kernel follows.
// ------------------------------------------------------------------------------------------------
constexpr int PCH = C / 2;   // channels per workgroup of a pair

// i-th executed k-step of GEMM1 -> k-step index (tap-major, 16 channel groups per tap) for part q
__device__ __forceinline__ int kmap_pair(int i, int q) {
  if (i < 8) return 16 + 8 * q + i;                  // centre tap, own channels
  if (i < 16) return 16 + 8 * (1 - q) + (i - 8);     // centre tap, the partner's channels
  if (i < 32) return i - 16;                         // tap 0
  return i;                                          // tap 2
}

// weight ring of the pair form in k-steps: one wave per SIMD has the registers, and with half as many waves per CU the bytes in flight —
// not the L2 — bound the weight stream (phase stamps at B=1 with a ring of 4: 0.22 us per k-step where the MFMAs need 0.08)
constexpr int NSP = 8;   // (measured at B=1, us per layer: ring 4 18.6, ring 8 17.5, ring 16 18.9; three accumulators per row tile instead of one, or an
                         // L2 prefetch of the next layer's weights, change nothing: profiles/r03_pair_form/)
// the k-step pipeline of mfma_pipe_h2 with the k-step order given by `km` (NCT column tiles of 32 frames)
template <int ROT, int NCT, typename KM, typename LDB, typename MID>
__device__ __forceinline__ void mfma_pipe_pair(f32x16 (&c0)[NCT], f32x16 (&c1)[NCT], f16x8 (&A)[NSP][4], rsrc_t rs, int vfrag, int sa0, int sa1,
                                               int n_ks, KM km, LDB ldb, MID mid) {
  f16x8 B[2][2 * NCT];
  ldb(km(0), B[0]);
  const int last = n_ks - 1;
#pragma unroll 1
  for (int ks = 0; ks < n_ks; ks += NSP) {
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      if (ROT > 0 && s == ROT % NSP && ks == ROT - ROT % NSP) {   // the hand-off sits behind the first ROT k-steps
        mid();
        ldb(km(ks + s), B[s & 1]);
      }
      const int in = ks + s + 1 <= last ? ks + s + 1 : last;
      // (never across the hand-off: the k-step behind it is read again above)
      ldb(km(in), B[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const f16x8(&Bc)[2 * NCT] = B[s & 1];
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
      for (int ct = 0; ct < NCT; ++ct) {   // lo hi
        BSG_MFMA_H(c0[ct], A[s][1], Bc[2 * ct]);
        BSG_MFMA_H(c1[ct], A[s][3], Bc[2 * ct]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int ir = ks + s + NSP <= last ? ks + s + NSP : last;
      const int kr = km(ir);
      A[s][0] = lda8(rs, vfrag, sa0 + kr * KSB2);
      A[s][1] = lda8(rs, vfrag, sa0 + kr * KSB2 + PLB);
      A[s][2] = lda8(rs, vfrag, sa1 + kr * KSB2);
      A[s][3] = lda8(rs, vfrag, sa1 + kr * KSB2 + PLB);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// NCT = column tiles of 32 frames per tile: 1 (32-frame tiles: while those fill at most half of the CUs, B <= 4 at T = 1000) or 2 (64-frame
// tiles: B = 5 .. 8 at T = 1000, where one workgroup per 32-frame tile had every CU stream the whole layer for half the matrix work)
template <int NCT>
__global__ __launch_bounds__(256, 1) void residual_pair_h2_kernel(StackArgs p) {
  constexpr int NT = 32 * NCT, XP = h2_xp(NCT), ZP = h2_zp(NCT);
  constexpr int NPIECE = 2 * NT * 16;   // 16-byte pieces of one exchange slot: 2 planes x NT frames x 16 chunks of 8 channels
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;                  // [2 planes][NT + 16 frames][528 B]: hi / lo of x + d_l, ALL channels, frames t0-8 .. t0+NT+7
  char* zs = lds_raw + 2 * XP;         // [2 planes][NT frames][528 B]: hi / lo of 2^10 x gated activation, ALL channels
  float* dtab = reinterpret_cast<float*>(lds_raw + 2 * XP + 2 * ZP);   // [256]
  float* btab = dtab + C;                                              // [512]

  // workgroup -> (tile, part): the two parts of a tile sit on the same XCD (workgroup i runs on XCD i mod 8)
  const int n_tiles = p.n_tiles, per_xcd = (n_tiles + 7) >> 3;
  const int slot = (int)blockIdx.x >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + (slot >> 1);
  const int q = slot & 1;
  if ((slot >> 1) >= per_xcd || tile_id >= n_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int tpr = p.tiles_per_row, L = p.L, T = p.T;
  const int b = tile_id / tpr, j = tile_id - b * tpr;
  const int t0 = j * NT;
  const int tb = p.t_dev ? (int)p.t_dev[b] : p.t_uniform;
  const bool has_left = j > 0, has_right = j + 1 < tpr;
  const int cb = PCH * q + 32 * wave;   // first channel of this wave

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
  const int sa_g = (4 * q + wave) * 1024, sa_f = (8 + 4 * q + wave) * 1024;   // gate / filter (= residual / skip) row tile inside a plane of a k-step slab

  float xr[NCT][16], sk[NCT][16];
  f32x16 yg[NCT], yf[NCT];
  int range_flag = 0;
  auto range_check = [&](unsigned worst) {
    if (__builtin_amdgcn_ballot_w64(worst >= 0x476A6000u) != 0ull) range_flag = 1;   // 60000.0f
  };
  auto absbits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };
  unsigned* fx = p.pflags;                   // image flags [n_tiles][2]
  unsigned* fz = p.pflags + 2 * n_tiles;     // z flags     [n_tiles][2]
  // a whole wave: every lane with a flag polls its own (all in flight together); bounded like the polls of residual_stack_h2_kernel
  auto wait_flags = [&](const unsigned* fl, unsigned want) {
    bool pend = fl != nullptr;
    if (p.inject) {
      if (pend) atomicAdd(p.status, 1u);
      return;
    }
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
      if (pend) pend = (int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0;
      if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
      __builtin_amdgcn_s_sleep(2);
      ++spins;
      const bool quit = spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
      if (quit) {
        if (pend) atomicAdd(p.status, 1u);
        break;
      }
    }
  };
  auto wait_flag = [&](const unsigned* fl, unsigned want) {   // one lane
    if (p.inject) { atomicAdd(p.status, 1u); return; }
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        atomicAdd(p.status, 1u);
        break;
      }
    }
  };

  auto cond_request = [&](int l) {
    const rsrc_t rs_ct = mk_rsrc(p.condterm + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int so = (cb + acc_row0(r)) * rowT;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        yg[ct][r] = ldf(rs_ct, vcol[ct], so);
        yf[ct][r] = ldf(rs_ct, vcol[ct], so + C * rowT);
      }
    }
  };
  auto write_core = [&]() {
    float dv[16];
    unsigned worst = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = dtab[cb + acc_row(r, lh)];
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
        char* dst = xs + (HALO + 32 * ct + l31) * ROWB + (cb + 8 * g + 4 * lh) * 2;
        *reinterpret_cast<u32x2*>(dst) = wh;
        *reinterpret_cast<u32x2*>(dst + XP) = wl;
      }
    range_check(worst);
  };
  // a half (PCH channels) of NT LDS rows starting at row r0, both planes, to / from an exchange slot [plane][NT][PCH]: 16-byte pieces,
  // write-through stores / sc1 loads (the hand-off form that needs no acquire)
  auto half_out = [&](const char* img, int plane_bytes, int r0, int part, unsigned short* slot_p) {
    const rsrc_t rs = mk_rsrc(slot_p, 2 * NT * PCH * 2);
#pragma unroll
    for (int k = 0; k < NPIECE / 256; ++k) {
      const int piece = k * 256 + tid;
      const int pl = piece / (NT * 16), f = (piece >> 4) % NT, c16 = piece & 15;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + pl * plane_bytes + (r0 + f) * ROWB + (PCH * part + 8 * c16) * 2);
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((pl * NT + f) * PCH + 8 * c16) * 2, 0, 16);   // sc1
    }
  };
  auto half_in = [&](char* img, int plane_bytes, int r0, int part, const unsigned short* slot_p) {
    const rsrc_t rs = mk_rsrc(slot_p, 2 * NT * PCH * 2);
    u32x4 v[NPIECE / 256];
#pragma unroll
    for (int k = 0; k < NPIECE / 256; ++k) {
      const int piece = k * 256 + tid;
      const int pl = piece / (NT * 16), f = (piece >> 4) % NT, c16 = piece & 15;
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((pl * NT + f) * PCH + 8 * c16) * 2, 0, 16);   // sc1
    }
#pragma unroll
    for (int k = 0; k < NPIECE / 256; ++k) {
      const int piece = k * 256 + tid;
      const int pl = piece / (NT * 16), f = (piece >> 4) % NT, c16 = piece & 15;
      *reinterpret_cast<u32x4*>(img + pl * plane_bytes + (r0 + f) * ROWB + (PCH * part + 8 * c16) * 2) = v[k];
    }
  };
  const size_t slot_halfs = (size_t)2 * NT * PCH;   // fp16 elements of one exchange slot
  auto zx_slot = [&](int tile, int part) { return p.zx + ((size_t)tile * 2 + part) * slot_halfs; };
  auto ix_slot = [&](int par, int tile, int part) { return p.ix + (((size_t)par * n_tiles + tile) * 2 + part) * slot_halfs; };

  // ---- layer 0: x from HBM — this wave's channels into registers, the WHOLE image (all channels, halo frames included) into LDS ----------
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xr[ct][r] = ldf(rs_x, vcol[ct], (cb + acc_row0(r)) * rowT);
      sk[ct][r] = 0.f;
    }
  {
    const rsrc_t rs_dp = mk_rsrc(p.dproj + ((long long)tb * L + 0) * C, C * 4);
    unsigned worst = 0;
    constexpr int ROWS = NT + 2 * HALO;
#pragma unroll 1
    for (int it = 0; it < 32 * ROWS / 256; ++it) {   // 32 chunks of 8 channels x ROWS frames, lanes = consecutive frames
      const int item = it * 256 + tid;
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
      *reinterpret_cast<u32x4*>(xs + row * ROWB + hc * 16) = wh;
      *reinterpret_cast<u32x4*>(xs + XP + row * ROWB + hc * 16) = wl;
    }
    range_check(worst);
  }
  dtab[tid] = p.dproj[((long long)tb * L + 0) * C + tid];
  btab[tid] = p.bias_out[tid];
  btab[tid + 256] = p.bias_out[tid + 256];
  cond_request(0);
  f16x8 A[NSP][4];
  auto prefetch_a1 = [&](int l) {
    const rsrc_t rs = mk_rsrc(p.apack1s + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
#pragma unroll
    for (int k = 0; k < NSP; ++k) {
      const int kr = kmap_pair(k, q);
      A[k][0] = lda8(rs, vfrag, sa_g + kr * KSB2);
      A[k][1] = lda8(rs, vfrag, sa_g + kr * KSB2 + PLB);
      A[k][2] = lda8(rs, vfrag, sa_f + kr * KSB2);
      A[k][3] = lda8(rs, vfrag, sa_f + kr * KSB2 + PLB);
    }
  };
  prefetch_a1(0);
  __syncthreads();   // the staged image and the tables

#define PAIR_STAMP(i)                                                                                                      \
  do {                                                                                                                    \
    if (p.stamps && tid == 0 && q == 0) p.stamps[((long long)tile_id * L + l) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int dil = 1 << (l % p.cycle);
    const rsrc_t rs_a1 = mk_rsrc(p.apack1s + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
    const rsrc_t rs_a2 = mk_rsrc(p.apack2s + (long long)l * (2 * 2 * C * C), 2 * 2 * C * C * 2);
    const float s1 = p.h2_scale[4 * l], inv1 = p.h2_scale[4 * l + 1], s2 = p.h2_scale[4 * l + 2], inv2 = p.h2_scale[4 * l + 3];
    const float dnext = l + 1 < L ? p.dproj[((long long)tb * L + l + 1) * C + tid] : 0.f;
    const float bnext0 = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + tid] : 0.f;
    const float bnext1 = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + 256 + tid] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) { yg[ct][r] *= s1; yf[ct][r] *= s1; }
    PAIR_STAMP(0);
    // ---- GEMM1: the centre tap of the own channels first (8 k-steps); behind it the partner's half of the image and the neighbours' edges
    {
      const char* xb = xs + (HALO + l31) * ROWB + lh * 16;
      auto ldb = [&](int ks, f16x8 (&Bf)[2 * NCT]) {
        const int tap = ks >> 4, kc = ks & 15;
        const char* qp = xb + ((tap - 1) * dil) * ROWB + kc * 32;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          Bf[2 * ct] = *reinterpret_cast<const f16x8*>(qp + 32 * ct * ROWB);
          Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(qp + 32 * ct * ROWB + XP);
        }
      };
      auto mid = [&]() {
        if (l == 0) return;   // layer 0 staged the whole image from HBM
        if (wave == 0) {
          // five flags (the partner's and the two neighbouring tiles' halves), polled by five lanes AT ONCE: one L2 round trip, not five
          const unsigned want = p.fbase + (unsigned)l;
          const unsigned* fl = lane == 0 ? fx + 2 * tile_id + (1 - q)
                               : (lane <= 2 ? (has_left ? fx + 2 * (tile_id - 1) + (lane - 1) : nullptr)
                                            : (lane <= 4 ? (has_right ? fx + 2 * (tile_id + 1) + (lane - 3) : nullptr) : nullptr));
          wait_flags(fl, want);
        }
        __syncthreads();   // (D) the polling lanes have seen the flags
        PAIR_STAMP(1);
        half_in(xs, XP, HALO, 1 - q, ix_slot(l & 1, tile_id, 1 - q));   // the partner's channels of the core frames
        {
          // halo rows, both planes, both halves: rows 0..7 = the left tile's last 8 frames, rows NT+8..NT+15 = the right tile's first 8
          u32x4 v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int piece = k * 256 + tid;   // 2 sides x 2 parts x 2 planes x 8 frames x 16 chunks
            const int side = piece >> 9, part = (piece >> 8) & 1, pl = (piece >> 7) & 1, f = (piece >> 4) & 7, c16 = piece & 15;
            const bool have = side == 0 ? has_left : has_right;
            v[k] = u32x4{0u, 0u, 0u, 0u};
            if (have) {
              const rsrc_t rs = mk_rsrc(ix_slot(l & 1, side == 0 ? tile_id - 1 : tile_id + 1, part), 2 * NT * PCH * 2);
              v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((pl * NT + (side == 0 ? NT - 8 : 0) + f) * PCH + 8 * c16) * 2, 0, 16);   // sc1
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int piece = k * 256 + tid;
            const int side = piece >> 9, part = (piece >> 8) & 1, pl = (piece >> 7) & 1, f = (piece >> 4) & 7, c16 = piece & 15;
            *reinterpret_cast<u32x4*>(xs + pl * XP + ((side ? HALO + NT : 0) + f) * ROWB + (PCH * part + 8 * c16) * 2) = v[k];
          }
        }
        __syncthreads();   // (A) the whole image is in place
        PAIR_STAMP(2);
      };
      mfma_pipe_pair<8, NCT>(yg, yf, A, rs_a1, vfrag, sa_g, sa_f, 48, [&](int i) { return kmap_pair(i, q); }, ldb, mid);
    }
    PAIR_STAMP(3);
    // ---- gate -> own half of zs (hi / lo of 2^10 z) ---------------------------------------------------------------------------------
    dtab[tid] = dnext;   // read by write_core() behind the barriers below
    const float rs2 = inv2 * 0.70710678118654752440f;
    const float gcg = -1.44269504088896340736f * inv1, gcf = -2.88539008177792681472f * inv1, glim = 15.0f * s1;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x2 z01 = gate2_scaled(f32x2{yg[ct][4 * g], yg[ct][4 * g + 1]}, f32x2{yf[ct][4 * g], yf[ct][4 * g + 1]}, gcg, gcf, glim, ZSCALE);
        const f32x2 z23 = gate2_scaled(f32x2{yg[ct][4 * g + 2], yg[ct][4 * g + 3]}, f32x2{yf[ct][4 * g + 2], yf[ct][4 * g + 3]}, gcg, gcf, glim, ZSCALE);
        const HiLo s0 = split2(z01[0], z01[1]), s1_ = split2(z23[0], z23[1]);
        char* dst = zs + (32 * ct + l31) * ROWB + (cb + 8 * g + 4 * lh) * 2;
        *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
        *reinterpret_cast<u32x2*>(dst + ZP) = u32x2{s0.lo, s1_.lo};
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float br = btab[cb + acc_row(r, lh)], bs = btab[C + cb + acc_row(r, lh)];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        yg[ct][r] = (xr[ct][r] + br) * s2;
        yf[ct][r] = bs * s2;
      }
    }
    __syncthreads();   // (Z1) the own half of z is complete in LDS; every wave is done reading xs and this layer's biases
    btab[tid] = bnext0;
    btab[tid + 256] = bnext1;
    if (!(p.inject && (tile_id & 1))) half_out(zs, ZP, 0, q, zx_slot(tile_id, q));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have landed
    __syncthreads();
    // GEMM2's first weights are requested only now: in front of the drain they would delay the flag by their own latency (vmcnt counts in
    // order), here they land while the partner's half of z is on its way
#pragma unroll
    for (int k = 0; k < NSP; ++k) {
      A[k][0] = lda8(rs_a2, vfrag, sa_g + k * KSB2);
      A[k][1] = lda8(rs_a2, vfrag, sa_g + k * KSB2 + PLB);
      A[k][2] = lda8(rs_a2, vfrag, sa_f + k * KSB2);
      A[k][3] = lda8(rs_a2, vfrag, sa_f + k * KSB2 + PLB);
    }
    if (tid == 0) {
      __hip_atomic_store(fz + 2 * tile_id + q, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_flag(fz + 2 * tile_id + (1 - q), p.fbase + (unsigned)(l + 1));
    }
    __syncthreads();   // the partner's half of z is published
    PAIR_STAMP(4);
    half_in(zs, ZP, 0, 1 - q, zx_slot(tile_id, 1 - q));
    __syncthreads();   // (B) zs complete
    PAIR_STAMP(5);
    // ---- GEMM2: 16 k-steps; yg = residual rows, yf = skip rows of the own channels ----------------------------------------------------
    {
      const char* zb = zs + l31 * ROWB + lh * 16;
      auto ldb = [&](int ks, f16x8 (&Bf)[2 * NCT]) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          Bf[2 * ct] = *reinterpret_cast<const f16x8*>(zb + 32 * ct * ROWB + ks * 32);
          Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(zb + 32 * ct * ROWB + ks * 32 + ZP);
        }
      };
      mfma_pipe_pair<0, NCT>(yg, yf, A, rs_a2, vfrag, sa_g, sa_f, 16, [](int i) { return i; }, ldb, [] {});
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        xr[ct][r] = yg[ct][r] * rs2;
        sk[ct][r] += yf[ct][r] * inv2;
      }
    PAIR_STAMP(6);
    if (l + 1 == L) break;
    // ---- next layer: the own half of the image into LDS and to the exchange slot, the flag — and only then the conditioner term into the
    // free accumulators: its HBM latency overlaps the wait for the partner's half instead of delaying the own flag (vmcnt counts in order)
    write_core();
    __syncthreads();   // (C1) the own half of the core rows is complete
    if (!(p.inject && (tile_id & 1))) half_out(xs, XP, HALO, q, ix_slot((l + 1) & 1, tile_id, q));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (C)
    if (tid == 0) __hip_atomic_store(fx + 2 * tile_id + q, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PAIR_STAMP(7);
    cond_request(l + 1);
    prefetch_a1(l + 1);   // (behind the drain, like the conditioner term)
  }
#undef PAIR_STAMP
  if (range_flag && lane == 0) atomicAdd(p.status + 1, 1u);
  // ---- the skip sum / sqrt(L) of the own channels (net.py:126), fp32 [C][T] rows: what step_tail_kernel reads ---------------------------
  const rsrc_t rs_sk = mk_rsrc(p.skip + (long long)b * C * T, plane);
  const float rdiv = 1.0f / sqrtf((float)L);
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    if (col_ok[ct]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stf(sk[ct][r] * rdiv, rs_sk, vst[ct], (cb + acc_row0(r)) * rowT);
    }
}

// ------------------------------------------------------------------------------------------------
// QUAD form for one or two utterances (4 x B * ceil(T / 32) <= CUs; B <= 2 at T = 1000): the pair form's idea once more.  A 32-frame tile is
// computed by FOUR workgroups of 4 waves on four CUs of one XCD, each owning a QUARTER of the channels (64): a wave owns 16 channels — 16
// gate + 16 filter rows of GEMM1, 16 residual + 16 skip rows of GEMM2 — as 16-row matrix tiles (v_mfma_f32_16x16x32_f16: lane l holds
// A[row l & 15][k = 8 (l >> 4) + j], B[k][column l & 15]; C/D column l & 15, rows 4 (l >> 4) + r; weights packed a second time in that
// fragment order, pack_a_frag_q_kernel).  A CU streams a quarter of the layer's weights (0.52 MB); z and the image are all-gathered among
// the four through L2 (8 KB per part) with the pair form's protocol; GEMM1 starts with the two 32-deep k-steps of the centre tap that
// cover the own 64 channels.
// ------------------------------------------------------------------------------------------------
constexpr int QCH = C / 4;     // channels per workgroup of a quad
constexpr int NSQ = 8;         // weight ring in k-steps of 32
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

// i-th executed k-step (32 deep; 8 per tap) of GEMM1 for part q
__device__ __forceinline__ int kmap_quad(int i, int q) {
  if (i < 2) return 8 + 2 * q + i;                     // centre tap, own 64 channels
  if (i < 8) { const int c = i - 2; return 8 + (c < 2 * q ? c : c + 2); }   // centre tap, the partners' channels
  if (i < 16) return i - 8;                            // tap 0
  return i;                                            // tap 2
}

#define BSG_MFMA_Q(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_, B_, ACC, 0, 0, 0)
using f32x4q = __attribute__((ext_vector_type(4))) float;

// k-step pipeline: two row tiles of 16 (c0: gate / residual, c1: filter / skip) x two column tiles of 16 frames, 12 MFMAs per k-step
template <int ROT, typename KM, typename LDB, typename MID>
__device__ __forceinline__ void mfma_pipe_quad(f32x4q (&c0)[2], f32x4q (&c1)[2], f16x8 (&A)[NSQ][4], rsrc_t rs, int vfrag, int sa0, int sa1, int n_ks,
                                               KM km, LDB ldb, MID mid) {
  f16x8 B[2][4];
  ldb(km(0), B[0]);
  const int last = n_ks - 1;
#pragma unroll 1
  for (int ks = 0; ks < n_ks; ks += NSQ) {
#pragma unroll
    for (int s = 0; s < NSQ; ++s) {
      if (ROT > 0 && s == ROT % NSQ && ks == ROT - ROT % NSQ) {   // the hand-off sits behind the first ROT k-steps
        mid();
        ldb(km(ks + s), B[s & 1]);
      }
      const int in = ks + s + 1 <= last ? ks + s + 1 : last;
      ldb(km(in), B[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const f16x8(&Bc)[4] = B[s & 1];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        BSG_MFMA_Q(c0[ct], A[s][0], Bc[2 * ct]);
        BSG_MFMA_Q(c1[ct], A[s][2], Bc[2 * ct]);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        BSG_MFMA_Q(c0[ct], A[s][0], Bc[2 * ct + 1]);
        BSG_MFMA_Q(c1[ct], A[s][2], Bc[2 * ct + 1]);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        BSG_MFMA_Q(c0[ct], A[s][1], Bc[2 * ct]);
        BSG_MFMA_Q(c1[ct], A[s][3], Bc[2 * ct]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int ir = ks + s + NSQ <= last ? ks + s + NSQ : last;
      const int kr = km(ir);
      A[s][0] = lda8(rs, vfrag, sa0 + kr * QKSB);
      A[s][1] = lda8(rs, vfrag, sa0 + kr * QKSB + QPLB);
      A[s][2] = lda8(rs, vfrag, sa1 + kr * QKSB);
      A[s][3] = lda8(rs, vfrag, sa1 + kr * QKSB + QPLB);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__global__ __launch_bounds__(256, 1) void residual_quad_h2_kernel(StackArgs p) {
  constexpr int NT = 32, XP = h2_xp(1), ZP = h2_zp(1);
  constexpr int NPIECE = 2 * NT * (QCH / 8);   // 16-byte pieces of one exchange slot: 2 planes x 32 frames x 8 chunks of 8 channels = 512
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;                  // [2 planes][48 frames][528 B]: hi / lo of x + d_l, ALL channels
  char* zs = lds_raw + 2 * XP;         // [2 planes][32 frames][528 B]: hi / lo of 2^10 x gated activation, ALL channels
  float* dtab = reinterpret_cast<float*>(lds_raw + 2 * XP + 2 * ZP);   // [256]
  float* btab = dtab + C;                                              // [512]

  // workgroup -> (tile, part): the four parts of a tile sit on the same XCD (workgroup i runs on XCD i mod 8)
  const int n_tiles = p.n_tiles, per_xcd = (n_tiles + 7) >> 3;
  const int slot = (int)blockIdx.x >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + (slot >> 2);
  const int q = slot & 3;
  if ((slot >> 2) >= per_xcd || tile_id >= n_tiles) return;
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
  int vcol[2], vst[2];
  bool col_ok[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = t0 + 16 * ct + l15;
    col_ok[ct] = col < T;
    vcol[ct] = (kb * 4 * T + (col_ok[ct] ? col : T - 1)) * 4;   // accumulator rows 4 kb + r
    vst[ct] = (kb * 4 * T + col) * 4;
  }
  const int rt_g = 4 * q + wave;                                  // gate / residual row tile (of 16); filter / skip: + 16
  const int sa_g = rt_g * 1024, sa_f = (16 + rt_g) * 1024;

  float xr[2][4], sk[2][4];
  f32x4q yg[2], yf[2];
  int range_flag = 0;
  auto range_check = [&](unsigned worst) {
    if (__builtin_amdgcn_ballot_w64(worst >= 0x476A6000u) != 0ull) range_flag = 1;   // 60000.0f
  };
  auto absbits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };
  unsigned* fx = p.pflags;                   // image flags [n_tiles][4]
  unsigned* fz = p.pflags + 4 * n_tiles;     // z flags     [n_tiles][4]
  auto wait_flags = [&](const unsigned* fl, unsigned want) {   // a whole wave: every lane with a flag polls its own; bounded
    bool pend = fl != nullptr;
    if (p.inject) {
      if (pend) atomicAdd(p.status, 1u);
      return;
    }
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
      if (pend) pend = (int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0;
      if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
      __builtin_amdgcn_s_sleep(2);
      ++spins;
      const bool quit = spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
      if (quit) {
        if (pend) atomicAdd(p.status, 1u);
        break;
      }
    }
  };

  auto cond_request = [&](int l) {
    const rsrc_t rs_ct = mk_rsrc(p.condterm + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int so = (cb + r) * rowT;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
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
    for (int ct = 0; ct < 2; ++ct) {
      const float v0 = xr[ct][0] + dv[0], v1 = xr[ct][1] + dv[1], v2 = xr[ct][2] + dv[2], v3 = xr[ct][3] + dv[3];
      worst = max(max(worst, max(absbits(v0), absbits(v1))), max(absbits(v2), absbits(v3)));
      const HiLo s0 = split2(v0, v1);
      const HiLo s1_ = split2(v2, v3);
      u32x2 wh = u32x2{s0.hi, s1_.hi}, wl = u32x2{s0.lo, s1_.lo};
      if (!col_ok[ct]) { wh = u32x2{0u, 0u}; wl = u32x2{0u, 0u}; }
      char* dst = xs + (HALO + 16 * ct + l15) * ROWB + (cb + 4 * kb) * 2;
      *reinterpret_cast<u32x2*>(dst) = wh;
      *reinterpret_cast<u32x2*>(dst + XP) = wl;
    }
    range_check(worst);
  };
  // a quarter (QCH channels) of NT LDS rows starting at row r0, both planes, to / from an exchange slot [plane][NT][QCH]
  auto part_out = [&](const char* img, int plane_bytes, int r0, int part, unsigned short* slot_p) {
    const rsrc_t rs = mk_rsrc(slot_p, 2 * NT * QCH * 2);
#pragma unroll
    for (int k = 0; k < NPIECE / 256; ++k) {
      const int piece = k * 256 + tid;
      const int pl = piece >> 8, f = (piece >> 3) & 31, c8 = piece & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + pl * plane_bytes + (r0 + f) * ROWB + (QCH * part + 8 * c8) * 2);
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((pl * NT + f) * QCH + 8 * c8) * 2, 0, 16);   // sc1
    }
  };
  auto parts_in = [&](char* img, int plane_bytes, int r0, auto slot_of) {   // the three partners' quarters
    u32x4 v[3 * NPIECE / 256];
#pragma unroll
    for (int k = 0; k < 3 * NPIECE / 256; ++k) {
      const int piece = k * 256 + tid;
      const int o = piece >> 9, part = o < q ? o : o + 1;   // the o-th partner
      const int pc = piece & 511, pl = pc >> 8, f = (pc >> 3) & 31, c8 = pc & 7;
      const rsrc_t rs = mk_rsrc(slot_of(part), 2 * NT * QCH * 2);
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((pl * NT + f) * QCH + 8 * c8) * 2, 0, 16);   // sc1
    }
#pragma unroll
    for (int k = 0; k < 3 * NPIECE / 256; ++k) {
      const int piece = k * 256 + tid;
      const int o = piece >> 9, part = o < q ? o : o + 1;
      const int pc = piece & 511, pl = pc >> 8, f = (pc >> 3) & 31, c8 = pc & 7;
      *reinterpret_cast<u32x4*>(img + pl * plane_bytes + (r0 + f) * ROWB + (QCH * part + 8 * c8) * 2) = v[k];
    }
  };
  const size_t slot_halfs = (size_t)2 * NT * QCH;   // fp16 elements of one exchange slot (8 KB)
  auto zx_slot = [&](int tile, int part) { return p.zx + ((size_t)tile * 4 + part) * slot_halfs; };
  auto ix_slot = [&](int par, int tile, int part) { return p.ix + (((size_t)par * n_tiles + tile) * 4 + part) * slot_halfs; };

  // ---- layer 0: x from HBM — this wave's channels into registers, the WHOLE image (all channels, halo frames included) into LDS ----------
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
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
    for (int it = 0; it < 32 * ROWS / 256; ++it) {   // 32 chunks of 8 channels x 48 frames, lanes = consecutive frames
      const int item = it * 256 + tid;
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
      *reinterpret_cast<u32x4*>(xs + row * ROWB + hc * 16) = wh;
      *reinterpret_cast<u32x4*>(xs + XP + row * ROWB + hc * 16) = wl;
    }
    range_check(worst);
  }
  dtab[tid] = p.dproj[((long long)tb * L + 0) * C + tid];
  btab[tid] = p.bias_out[tid];
  btab[tid + 256] = p.bias_out[tid + 256];
  cond_request(0);
  f16x8 A[NSQ][4];
  auto prefetch_a1 = [&](int l) {
    const rsrc_t rs = mk_rsrc(p.apack1q + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
#pragma unroll
    for (int k = 0; k < NSQ; ++k) {
      const int kr = kmap_quad(k, q);
      A[k][0] = lda8(rs, vfrag, sa_g + kr * QKSB);
      A[k][1] = lda8(rs, vfrag, sa_g + kr * QKSB + QPLB);
      A[k][2] = lda8(rs, vfrag, sa_f + kr * QKSB);
      A[k][3] = lda8(rs, vfrag, sa_f + kr * QKSB + QPLB);
    }
  };
  prefetch_a1(0);
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
    const float s1 = p.h2_scale[4 * l], inv1 = p.h2_scale[4 * l + 1], s2 = p.h2_scale[4 * l + 2], inv2 = p.h2_scale[4 * l + 3];
    const float dnext = l + 1 < L ? p.dproj[((long long)tb * L + l + 1) * C + tid] : 0.f;
    const float bnext0 = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + tid] : 0.f;
    const float bnext1 = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + 256 + tid] : 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { yg[ct] *= s1; yf[ct] *= s1; }
    QUAD_STAMP(0);
    // ---- GEMM1: 24 k-steps of 32; the two of the centre tap over the own channels first, behind them the partners' quarters + the halo ----
    {
      const char* xb = xs + (HALO + l15) * ROWB + kb * 16;
      auto ldb = [&](int ks, f16x8 (&Bf)[4]) {
        const int tap = ks >> 3, kc = ks & 7;
        const char* qp = xb + ((tap - 1) * dil) * ROWB + kc * 64;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          Bf[2 * ct] = *reinterpret_cast<const f16x8*>(qp + 16 * ct * ROWB);
          Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(qp + 16 * ct * ROWB + XP);
        }
      };
      auto mid = [&]() {
        if (l == 0) return;   // layer 0 staged the whole image from HBM
        if (wave == 0) {
          // eleven flags (three partners, the four parts of each neighbouring tile), polled by eleven lanes at once
          const unsigned want = p.fbase + (unsigned)l;
          const unsigned* fl = nullptr;
          if (lane < 3) fl = fx + 4 * tile_id + (lane < q ? lane : lane + 1);
          else if (lane < 7) fl = has_left ? fx + 4 * (tile_id - 1) + (lane - 3) : nullptr;
          else if (lane < 11) fl = has_right ? fx + 4 * (tile_id + 1) + (lane - 7) : nullptr;
          wait_flags(fl, want);
        }
        __syncthreads();   // (D) the polling lanes have seen the flags
        QUAD_STAMP(1);
        parts_in(xs, XP, HALO, [&](int part) { return ix_slot(l & 1, tile_id, part); });   // the partners' channels of the core frames
        {
          // halo rows, both planes, all four parts: rows 0..7 = the left tile's last 8 frames, rows 40..47 = the right tile's first 8
          u32x4 v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int piece = k * 256 + tid;   // 2 sides x 4 parts x 2 planes x 8 frames x 8 chunks
            const int side = piece >> 9, part = (piece >> 7) & 3, pl = (piece >> 6) & 1, f = (piece >> 3) & 7, c8 = piece & 7;
            const bool have = side == 0 ? has_left : has_right;
            v[k] = u32x4{0u, 0u, 0u, 0u};
            if (have) {
              const rsrc_t rs = mk_rsrc(ix_slot(l & 1, side == 0 ? tile_id - 1 : tile_id + 1, part), 2 * NT * QCH * 2);
              v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((pl * NT + (side == 0 ? NT - 8 : 0) + f) * QCH + 8 * c8) * 2, 0, 16);   // sc1
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int piece = k * 256 + tid;
            const int side = piece >> 9, part = (piece >> 7) & 3, pl = (piece >> 6) & 1, f = (piece >> 3) & 7, c8 = piece & 7;
            *reinterpret_cast<u32x4*>(xs + pl * XP + ((side ? HALO + NT : 0) + f) * ROWB + (QCH * part + 8 * c8) * 2) = v[k];
          }
        }
        __syncthreads();   // (A) the whole image is in place
        QUAD_STAMP(2);
      };
      mfma_pipe_quad<2>(yg, yf, A, rs_a1, vfrag, sa_g, sa_f, 24, [&](int i) { return kmap_quad(i, q); }, ldb, mid);
    }
    QUAD_STAMP(3);
    // ---- gate -> own quarter of zs (hi / lo of 2^10 z) ----------------------------------------------------------------------------------
    dtab[tid] = dnext;
    const float rs2 = inv2 * 0.70710678118654752440f;
    const float gcg = -1.44269504088896340736f * inv1, gcf = -2.88539008177792681472f * inv1, glim = 15.0f * s1;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const f32x2 z01 = gate2_scaled(f32x2{yg[ct][0], yg[ct][1]}, f32x2{yf[ct][0], yf[ct][1]}, gcg, gcf, glim, ZSCALE);
      const f32x2 z23 = gate2_scaled(f32x2{yg[ct][2], yg[ct][3]}, f32x2{yf[ct][2], yf[ct][3]}, gcg, gcf, glim, ZSCALE);
      const HiLo s0 = split2(z01[0], z01[1]), s1_ = split2(z23[0], z23[1]);
      char* dst = zs + (16 * ct + l15) * ROWB + (cb + 4 * kb) * 2;
      *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
      *reinterpret_cast<u32x2*>(dst + ZP) = u32x2{s0.lo, s1_.lo};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float br = btab[cb + 4 * kb + r], bs = btab[C + cb + 4 * kb + r];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        yg[ct][r] = (xr[ct][r] + br) * s2;
        yf[ct][r] = bs * s2;
      }
    }
    __syncthreads();   // (Z1) the own quarter of z is complete in LDS; every wave is done reading xs and this layer's biases
    btab[tid] = bnext0;
    btab[tid + 256] = bnext1;
    if (!(p.inject && (tile_id & 1))) part_out(zs, ZP, 0, q, zx_slot(tile_id, q));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have landed
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NSQ; ++k) {   // GEMM2's weights (all 8 k-steps), requested behind the drain
      A[k][0] = lda8(rs_a2, vfrag, sa_g + k * QKSB);
      A[k][1] = lda8(rs_a2, vfrag, sa_g + k * QKSB + QPLB);
      A[k][2] = lda8(rs_a2, vfrag, sa_f + k * QKSB);
      A[k][3] = lda8(rs_a2, vfrag, sa_f + k * QKSB + QPLB);
    }
    if (tid == 0) __hip_atomic_store(fz + 4 * tile_id + q, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) wait_flags(lane < 3 ? fz + 4 * tile_id + (lane < q ? lane : lane + 1) : nullptr, p.fbase + (unsigned)(l + 1));
    __syncthreads();   // the partners' quarters of z are published
    QUAD_STAMP(4);
    parts_in(zs, ZP, 0, [&](int part) { return zx_slot(tile_id, part); });
    __syncthreads();   // (B) zs complete
    QUAD_STAMP(5);
    // ---- GEMM2: 8 k-steps of 32; yg = residual rows, yf = skip rows of the own channels ---------------------------------------------------
    {
      const char* zb = zs + l15 * ROWB + kb * 16;
      auto ldb = [&](int ks, f16x8 (&Bf)[4]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          Bf[2 * ct] = *reinterpret_cast<const f16x8*>(zb + 16 * ct * ROWB + ks * 64);
          Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(zb + 16 * ct * ROWB + ks * 64 + ZP);
        }
      };
      mfma_pipe_quad<0>(yg, yf, A, rs_a2, vfrag, sa_g, sa_f, 8, [](int i) { return i; }, ldb, [] {});
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
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
    if (!(p.inject && (tile_id & 1))) part_out(xs, XP, HALO, q, ix_slot((l + 1) & 1, tile_id, q));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (C)
    if (tid == 0) __hip_atomic_store(fx + 4 * tile_id + q, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    QUAD_STAMP(7);
    cond_request(l + 1);
    prefetch_a1(l + 1);
  }
#undef QUAD_STAMP
  if (range_flag && lane == 0) atomicAdd(p.status + 1, 1u);
  // ---- the skip sum / sqrt(L) of the own channels (net.py:126), fp32 [C][T] rows: what step_tail_kernel reads ---------------------------
  const rsrc_t rs_sk = mk_rsrc(p.skip + (long long)b * C * T, plane);
  const float rdiv = 1.0f / sqrtf((float)L);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
    if (col_ok[ct]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) stf(sk[ct][r] * rdiv, rs_sk, vst[ct], (cb + r) * rowT);
    }
}
#undef BSG_MFMA_Q

#undef BSG_MFMA_H

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

// LDS of the pair form: the image and z of the whole tile (both channel halves), and more than half of the CU's LDS so that a CU holds one workgroup
constexpr size_t pair_lds(int nct) { return h2_lds(nct) > 84 * 1024 ? h2_lds(nct) : 84 * 1024; }
template <int NCT>
static int pair_occ() {
  int o = 0;
  if (hipFuncSetAttribute((const void*)residual_pair_h2_kernel<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds(NCT)) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)residual_pair_h2_kernel<NCT>, 256, pair_lds(NCT)) != hipSuccess)
    return 0;
  return o;
}
int pair_h2_occupancy(int nct) { return nct == 1 ? pair_occ<1>() : 0; }   // (pairs of 64-frame tiles: measured slower, not instantiated)
int launch_residual_pair_h2(const StackArgs& p, hipStream_t st, int nct) {
  BSG_REQUIRE(p.zx && p.ix && p.pflags, "pair launch: exchange buffers missing");
  const dim3 grid(16 * cdiv(p.n_tiles, 8)), block(256);
  BSG_REQUIRE(nct == 1, "pair launch: 32-frame tiles only");
  hipLaunchKernelGGL(residual_pair_h2_kernel<1>, grid, block, pair_lds(1), st, p);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// the step tail behind a pair / quad launch: one workgroup per 32-frame tile (a.tiles_per_row = ceil(T / 32)); a.status: the launch's status words
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
int quad_h2_occupancy() {
  int o = 0;
  if (hipFuncSetAttribute((const void*)residual_quad_h2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds(1)) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)residual_quad_h2_kernel, 256, pair_lds(1)) != hipSuccess)
    return 0;
  return o;
}
// quad form: FOUR workgroups of 4 waves per 32-frame tile; grid = 32 * ceil(n_tiles / 8) workgroups, all resident (one per CU)
int launch_residual_quad_h2(const StackArgs& p, hipStream_t st) {
  BSG_REQUIRE(p.zx && p.ix && p.pflags && p.apack1q && p.apack2q, "quad launch: exchange buffers / 16-row weight fragments missing");
  hipLaunchKernelGGL(residual_quad_h2_kernel, dim3(32 * cdiv(p.n_tiles, 8)), dim3(256), pair_lds(1), st, p);
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
