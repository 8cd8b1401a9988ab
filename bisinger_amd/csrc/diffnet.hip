// DiffNet (WaveNet noise predictor) on gfx950 — the >97 %-of-FLOPs hot loop of BiSinger's mel generation.
//
// Reference semantics: /root/reference/train_bisinger/usr/diff/net.py
//   ResidualBlock.forward :66-78, DiffNet.forward :107-130, SinusoidalPosEmb :32-44.
//
// One launch of residual_layer_kernel computes a whole ResidualBlock for a tile of N_TILE frames of one
// utterance, with every intermediate ([2C,T] pre-activation, gated z, [2C,T] output projection) kept on
// chip:
//
//   stage  xs = x[:, t0-8 : t0+N_TILE+8] + diffusion_projection(step)   (zero outside [0,T): the conv's
//          zero padding pads x + d, net.py:69-71)                                       -> LDS  [C][N_TILE+16]
//   GEMM1  y = W_dil[2C x 3C] * im2col(xs)  (3 dilated taps = 3 shifted reads of the same LDS image),
//          accumulators initialised with the step-invariant conditioner term (hoisted to prepare()),
//          v_mfma_f32_32x32x2_f32, A fragments streamed from L2 in pre-packed fragment order (one
//          global_load_dwordx4 per lane = 4 k-steps), B fragments = conflict-free ds_read_b32 rows.
//   gate   z = sigmoid(y[:C]) * tanh(y[C:])  in registers (a wave owns gate rows c and filter rows C+c)
//          -> LDS [C][N_TILE] (aliases xs)
//   GEMM2  o = W_out[2C x C] * z ; residual half: x_out = (x + o)/sqrt(2) ; skip half: skip += o
//
// 8 waves per workgroup, each a 32-row slice of the gate/filter (GEMM1) and residual/skip (GEMM2) rows.  Algorithmic work:
// 2*(2C*3C + 2C*C) = 1,048,576 FLOP per frame per layer (C = 256); algorithmic HBM bytes per frame per
// layer = 6*C*4 = 6 KB (x in, x out, conditioner term 2C, skip read+write)  -> AI = 171 FLOP/B: bound by
// the fp32 MFMA roof (157.3 TFLOP/s), not HBM.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "bsg_common.h"
#include "diffnet_res.h"
#include "diffnet_tail.h"

namespace bsg {

namespace {


// ------------------------------------------------------------------------------------------------
// weight packing: out[((mt*(K/8) + q)*64 + lane)*4 + j] = W(m = 32*mt + (lane&31), k = 8*q + 2*j + (lane>>5))
// with W(m,k) at src[m*sm + (k % Kc)*sc + (k / Kc)*st]  (dilated conv: k = tap*C + ci, src [2C][C][3]).
// ------------------------------------------------------------------------------------------------
__global__ void pack_a_frag_kernel(const float* __restrict__ src, float* __restrict__ out, int M, int K, int Kc,
                                   long long sm, long long sc, long long st) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)M * K;
  if (i >= total) return;
  const int j = (int)(i & 3);
  const int lane = (int)((i >> 2) & 63);
  const long long rest = i >> 8;
  const int Kq = K / 8;
  const int q = (int)(rest % Kq);
  const int mt = (int)(rest / Kq);
  const int m = 32 * mt + (lane & 31);
  const int k = 8 * q + 2 * j + (lane >> 5);
  out[i] = src[(long long)m * sm + (long long)(k % Kc) * sc + (long long)(k / Kc) * st];
}

// Winograd F(2,3) weights of the k=3 dilated conv, in 16x16x4 A-fragment order:
//   U0 = w0, U1 = (w0+w1+w2)/2, U2 = (w0-w1+w2)/2, U3 = w2;   out[comp][mt16][q][lane][jj] = U_comp(m, k),
//   m = 16*mt16 + (lane&15), k = 16*q + 4*jj + (lane>>4)       (src [2C][C][3])
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * 2 * C * C) return;
  const int jj = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
  const int q = rest & 15, mt = (rest >> 4) & 31, comp = rest >> 9;
  const int m = 16 * mt + (lane & 15), k = 16 * q + 4 * jj + (lane >> 4);
  const float* p = w + ((long long)m * C + k) * 3;
  const float w0 = p[0], w1 = p[1], w2 = p[2];
  float u;
  if (comp == 0) u = w0;
  else if (comp == 1) u = (w0 + w1 + w2) * 0.5f;
  else if (comp == 2) u = (w0 - w1 + w2) * 0.5f;
  else u = w2;
  out[i] = u;
}

__global__ void vec_add_kernel(const float* a, const float* b, float* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

// ------------------------------------------------------------------------------------------------
// fused residual block
// ------------------------------------------------------------------------------------------------

#define BSG_STAMP(i)                                                                                   \
  do {                                                                                                 \
    if (STAMP) {                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      const unsigned long long _t = __builtin_amdgcn_s_memtime();                                      \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                                              \
      if (lane == 0) a.stamps[((long long)tile_id * 8 + wave) * 10 + (i)] = _t;                      \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
  } while (0)

#define BSG_MFMA8(ACC0, ACC1, A0_, A1_, B_)                                                  \
  ACC0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0_[0], B_[0], ACC0, 0, 0, 0);                  \
  ACC1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1_[0], B_[0], ACC1, 0, 0, 0);                  \
  ACC0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0_[1], B_[1], ACC0, 0, 0, 0);                  \
  ACC1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1_[1], B_[1], ACC1, 0, 0, 0);                  \
  ACC0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0_[2], B_[2], ACC0, 0, 0, 0);                  \
  ACC1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1_[2], B_[2], ACC1, 0, 0, 0);                  \
  ACC0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0_[3], B_[3], ACC0, 0, 0, 0);                  \
  ACC1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1_[3], B_[3], ACC1, 0, 0, 0);

// Software-pipelined MFMA stream over two row tiles that share the B fragment.  Group q = 4 k-steps = 8 MFMAs.
// A fragments live in a ring of NS register sets: set s holds group q+s and is refilled with group q+s+NS right
// after its MFMAs have issued, i.e. NS-1 groups (>= 512 cycles each) before it is needed again; the B fragment of
// the next group is read from LDS while the current group multiplies.  sched_barrier pins this order: left alone,
// hipcc sinks the refills next to their consumers and every trip waits a full L2 latency.
template <int NS, typename LDB>
__device__ __forceinline__ void mfma_pipe(f32x16& acc0, f32x16& acc1, f32x4 (&A0)[NS], f32x4 (&A1)[NS], rsrc_t rs, int vfrag,
                                          int sa0, int sa1, int q_begin, int q_end, int q_last, LDB ldb) {
  f32x4 Bf[2];
  Bf[0] = ldb(q_begin);
#pragma unroll 1
  for (int q = q_begin; q < q_end; q += NS) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int qn = q + s + 1 <= q_last ? q + s + 1 : q_last;
      Bf[(s + 1) & 1] = ldb(qn);
      __builtin_amdgcn_sched_barrier(0);
      BSG_MFMA8(acc0, acc1, A0[s], A1[s], Bf[s & 1])
      __builtin_amdgcn_sched_barrier(0);
      const int qr = q + s + NS <= q_last ? q + s + NS : q_last;
      A0[s] = ldf4(rs, vfrag, sa0 + qr * 1024);
      A1[s] = ldf4(rs, vfrag, sa1 + qr * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GEMM1 of the residual block as Winograd F(2,3) over the dilated taps, for one wave: 4 row tiles of 16 (sw[] = their byte
// offsets in the packed Winograd weights) x 16 output pairs, K = 4 components x 256 channels.
//   y0 += M0 + M1 + M2   (output frame t of the pair),   y1 += M1 - M2 - M3   (frame t + d),   M_comp = U_comp * V_comp
//   V0 = x[t-d] - x[t+d],  V1 = x[t] + x[t+d],  V2 = x[t+d] - x[t],  V3 = x[t] - x[t+2d]     (x = the staged, zero-padded xs)
// The B fragment of group (comp, q) — channels 16 q + 4 jj + lq — is formed in registers: V = x[xo + oa] + sgn * x[xo + ob], one FMA
// with sgn = +-1 (the exact sum / difference).  Loop structure: components outside, the 16 channel groups inside, so that inside
// the loop the LDS addresses are two running offsets plus immediates and a group costs NO address arithmetic (the first form
// of this loop recomputed component, offsets and sign per group: ~15 scalar / vector instructions between two MFMA blocks, which
// a lone workgroup on a CU — 2 waves per SIMD — could not hide: GEMM1 phase 58 -> 43 us per layer in the stack launch).  The
// last group of a component prefetches the first of the next one (peeled tail).  One block = 16 MFMAs (v_mfma_f32_16x16x4_f32);
// the next group's LDS reads are issued before them (their latency passes under the MFMAs), its transform and the weight loads
// of the group NR ahead behind them; sched_barrier pins that order.  On entry AW[0..NR) hold the weights of groups 0..NR-1.
// ------------------------------------------------------------------------------------------------
template <int NTI, int NR>   // NTI row tiles of 16 per wave (sw[] = their byte offsets), NR = depth of the weight ring (groups in flight)
__device__ __forceinline__ void wino_gemm1(f32x4 (&y0)[NTI], f32x4 (&y1)[NTI], f32x4 (&AW)[NR][NTI], const float* xs, const int xo, const int dil,
                                           const rsrc_t rs_aw, const int vfrag, const int (&sw)[NTI]) {
  constexpr int LDX = 32 + 2 * HALO;
  static_assert(NR == 2 || NR == 4, "ring depth");
  f32x4 M[NTI];
#pragma unroll
  for (int i = 0; i < NTI; ++i) M[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto offs = [&](int comp, int& oa, int& ob, float& sgn) {
    oa = xo + ((comp == 2) - (comp == 0)) * dil;
    ob = xo + ((comp < 2) + 2 * (comp == 3)) * dil;
    sgn = comp == 1 ? 1.0f : -1.0f;
  };
  // byte offset of group g's weights inside a row tile of the packed matrix [comp][32 tiles][16 q][64 lanes][4] (clamped: a repeat at the end)
  auto wso = [](int g) {
    g = g < 63 ? g : 63;
    return ((g >> 4) * 512 + (g & 15)) * 1024;
  };
  auto block = [&](f32x4 (&A)[NTI], const f32x4& Bcur, f32x4& Bnext, int ia, int ib, float sgn, int so) {
    f32x4 ra, rb;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { ra[jj] = xs[ia + 4 * jj * LDX]; rb[jj] = xs[ib + 4 * jj * LDX]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < NTI; ++i) M[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i][jj], Bcur[jj], M[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) Bnext[jj] = __builtin_fmaf(sgn, rb[jj], ra[jj]);
#pragma unroll
    for (int i = 0; i < NTI; ++i) A[i] = ldf4(rs_aw, vfrag, sw[i] + so);
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x4 Bw[2];
  int na, nb;
  float nsgn;
  offs(0, na, nb, nsgn);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) Bw[0][jj] = __builtin_fmaf(nsgn, xs[nb + 4 * jj * LDX], xs[na + 4 * jj * LDX]);
#pragma unroll 1
  for (int comp = 0; comp < 4; ++comp) {
    int ia = na + 16 * LDX, ib = nb + 16 * LDX;   // group q + 1 of this component
    const float sgn = nsgn;
    offs(comp < 3 ? comp + 1 : 3, na, nb, nsgn);
    int g = comp * 16 + NR;                       // group whose weights the first block requests
#pragma unroll 1
    for (int q = 0; q < 16 - NR; q += NR) {
#pragma unroll
      for (int s = 0; s < NR; ++s) block(AW[s], Bw[s & 1], Bw[(s + 1) & 1], ia + 16 * s * LDX, ib + 16 * s * LDX, sgn, wso(g + s));
      ia += 16 * NR * LDX; ib += 16 * NR * LDX; g += NR;
    }
#pragma unroll
    for (int s = 0; s < NR - 1; ++s) block(AW[s], Bw[s & 1], Bw[(s + 1) & 1], ia + 16 * s * LDX, ib + 16 * s * LDX, sgn, wso(g + s));
    block(AW[NR - 1], Bw[(NR - 1) & 1], Bw[NR & 1], na, nb, nsgn, wso(g + NR - 1));   // (comp, 15) reads (comp + 1, 0)
    // the component is complete (FMAs with 0 / +-1: exact, branch-free)
    const float c0 = comp < 3 ? 1.0f : 0.0f, c1 = comp == 1 ? 1.0f : comp >= 2 ? -1.0f : 0.0f;
#pragma unroll
    for (int i = 0; i < NTI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y0[i][r] = __builtin_fmaf(c0, M[i][r], y0[i][r]);
        y1[i][r] = __builtin_fmaf(c1, M[i][r], y1[i][r]);
        M[i][r] = 0.f;
      }
  }
}

// One workgroup = 8 waves = one 32-frame tile of one utterance; wave w owns gate rows [32w,32w+32) and filter
// rows [256+32w, ...) in GEMM1 and residual rows [32w, ...) + skip rows [256+32w, ...) in GEMM2.
// Both GEMM loops are software-pipelined by hand: A fragments (global, L2) are requested a full 8-MFMA group
// before use and B fragments (LDS) one group before use; sched_barrier pins that order (left alone, hipcc
// sinks the prefetch loads next to their consumers, which exposes the L2 latency on every trip).
template <bool STAMP, int NS, bool WINO>   // NS: depth of the A-fragment ring of GEMM2 (and of GEMM1 when !WINO); WINO: GEMM1 as
                                         // Winograd F(2,3) over the dilated taps (2/3 of the MFMA work)
__device__ __forceinline__ void residual_tile(const ResArgs& a, const int tile_id) {
  constexpr int NT = 32;               // frames per workgroup
  constexpr int LDX = NT + 2 * HALO;   // xs row stride (48 floats = 12 x 16 B)
  constexpr int LDZ = NT;              // zs row stride

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* zs = lds;  // aliases xs after the barrier that ends GEMM1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = tile_id / a.tiles_per_row;
  const int t0 = (tile_id - b * a.tiles_per_row) * NT;
  const int T = a.T;
  const int tb = a.t_dev ? (int)a.t_dev[b] : a.t_uniform;
  const int col = t0 + l31;            // this lane's frame in every accumulator tile
  const bool col_ok = col < T;
  const int colc = col_ok ? col : T - 1;   // clamped: loads stay unconditional, results are masked

  const unsigned plane = (unsigned)C * T * 4;
  const rsrc_t rs_x = mk_rsrc(a.x_in + (long long)b * C * T, plane);
  const rsrc_t rs_xo = mk_rsrc(a.x_out + (long long)b * C * T, plane);
  const rsrc_t rs_sk = mk_rsrc(a.skip + (long long)b * C * T, plane);
  const rsrc_t rs_ct = mk_rsrc(a.condterm + (long long)b * 2 * C * T, 2 * plane);
  const rsrc_t rs_a1 = mk_rsrc(a.apack1, 2 * C * 3 * C * 4);
  const rsrc_t rs_a2 = mk_rsrc(a.apack2, 2 * C * C * 4);
  const rsrc_t rs_bo = mk_rsrc(a.bias_out, 2 * C * 4);
  const rsrc_t rs_dp = mk_rsrc(a.dproj + ((long long)tb * a.L + a.layer) * C, C * 4);
  const int vcol = (lh * 4 * T + colc) * 4;   // per-lane byte offset inside a [rows][T] plane (row half + frame)
  const int rowT = T * 4;                     // bytes per row
  const int vfrag = lane * 16;                // per-lane byte offset inside a packed 1-KB A fragment group

  BSG_STAMP(0);
  if (STAMP && lane == 0) a.stamps[((long long)tile_id * 8 + wave) * 10 + 8] = __builtin_amdgcn_s_memrealtime();
  // ---- (1) the first A fragments fly while the x tile is staged ------------------------------------
  f32x16 acc0, acc1;
  f32x4 Ag[NS], Af[NS];
  const int sa_g = wave * 96 * 1024, sa_f = (8 + wave) * 96 * 1024;   // gate / filter tile of the packed dilated conv
  // Winograd: 4 row tiles of 16 (gate 32w..+15, gate +16, filter 256+32w.., filter +16), ring of 2 groups
  const rsrc_t rs_aw = mk_rsrc(a.apackw, 4 * 2 * C * C * 4);
  f32x4 AW[2][4];
  int sw[4];
  if constexpr (WINO) {
    sw[0] = (2 * wave) * 16 * 1024; sw[1] = (2 * wave + 1) * 16 * 1024;
    sw[2] = (16 + 2 * wave) * 16 * 1024; sw[3] = (16 + 2 * wave + 1) * 16 * 1024;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) AW[k][i] = ldf4(rs_aw, vfrag, sw[i] + k * 1024);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      Ag[k] = ldf4(rs_a1, vfrag, sa_g + k * 1024);
      Af[k] = ldf4(rs_a1, vfrag, sa_f + k * 1024);
    }
  }

  // ---- (2) stage xs = x + d (zero padded) ------------------------------------------------------
  if ((T & 3) == 0) {
    // 256 rows x 12 float4; a float4 is entirely inside or entirely outside [0,T)
#pragma unroll 3
    for (int k = 0; k < 6; ++k) {
      const int idx = tid + 512 * k;
      const int c = idx / 12, j4 = idx - c * 12;
      const int t = t0 - HALO + 4 * j4;
      const bool ok = t >= 0 && t < T;
      f32x4 v = ldf4(rs_x, ok ? (c * T + t) * 4 : 0, 0);
      const float d = ldf(rs_dp, c * 4, 0);
      v += d;
      if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(xs + c * LDX + 4 * j4) = v;
    }
  } else {
#pragma unroll 4
    for (int idx = tid; idx < C * LDX; idx += 512) {
      const int c = idx / LDX, j = idx - c * LDX;
      const int t = t0 - HALO + j;
      const bool ok = t >= 0 && t < T;
      const float v = ldf(rs_x, ok ? (c * T + t) * 4 : 0, 0) + ldf(rs_dp, c * 4, 0);
      xs[idx] = ok ? v : 0.f;
    }
  }
  // Winograd form (registers to spare below its 128-VGPR budget): the residual input x of this wave's rows — the initial
  // accumulator of GEMM2 — is requested together with the staging loads, while the tile's lines are in flight / L2-hot, instead
  // of after the gate (≈50 us later, when they have left L2 and cost 16 MB of fabric reads per launch a second time)
  if constexpr (WINO) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = ldf(rs_x, vcol, (32 * wave + acc_row0(r)) * rowT);
  }
  __syncthreads();
  BSG_STAMP(1);

  float z[16];
  int zoff[16];   // LDS element offsets of z[] (compile-time pattern + per-lane part; folded by the compiler)
  if constexpr (!WINO) {
    // ---- (3) GEMM1: y = W_dil * im2col(xs): 96 groups of 8 MFMAs ----------------------------------
    {
      const float* xrow = xs + lh * LDX + HALO + l31;
      const int dil = a.dil;
      auto ldb = [&](int q) {
        const int tap = q >> 5, cg = q & 31;
        const float* p = xrow + 8 * cg * LDX + (tap - 1) * dil;
        return f32x4{p[0], p[2 * LDX], p[4 * LDX], p[6 * LDX]};
      };
      BSG_STAMP(2);
      mfma_pipe<NS>(acc0, acc1, Ag, Af, rs_a1, vfrag, sa_g, sa_f, 0, 96, 95, ldb);
    }
    BSG_STAMP(3);
    // ---- (4) + hoisted conditioner term, gate: z = sigmoid(gate) * tanh(filter)   (net.py:71-74) --------
    // The 64 KB/workgroup conditioner tile is the largest HBM stream of the layer.  It is requested here, not at
    // kernel start: all workgroups of a launch start together, and 32 MB requested at t=0 by an idle chip costs
    // ~15k cycles with the MFMA pipe empty; by now the co-resident workgroups have drifted apart, so this
    // latency hides under their MFMA work.
    float cg[16], cf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      cg[r] = ldf(rs_ct, vcol, (32 * wave + acc_row0(r)) * rowT);
      cf[r] = ldf(rs_ct, vcol, (C + 32 * wave + acc_row0(r)) * rowT);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      z[r] = fast_sigmoid(acc0[r] + cg[r]) * fast_tanh(acc1[r] + cf[r]);
      zoff[r] = (32 * wave + acc_row(r, lh)) * LDZ + l31;
    }
  } else {
    // ---- (3w) GEMM1 as Winograd F(2,3) along the dilated taps -------------------------------------
    // For the output pair (t, t+d):  y[t] = M0+M1+M2,  y[t+d] = M1-M2-M3,  M_j = sum_c U_j[.,c] * V_j[c], with
    //   V0 = x[t-d]-x[t+d], V1 = x[t]+x[t+d], V2 = x[t+d]-x[t], V3 = x[t]-x[t+2d]   (x = the staged, zero-padded tile)
    // 4 GEMMs of K = 256 over 16 pairs instead of one of K = 768 over 32 frames: 2/3 of the MFMA work.  16x16x4 MFMAs
    // (N = 16 pairs); lane = (pair p = lane&15, k-row lq = lane>>4); pair p of dilation d sits at tp = (p/d)*2d + p%d.
    const int p16 = lane & 15, lq = lane >> 4, dil = a.dil;
    const int ld = dil == 1 ? 0 : dil == 2 ? 1 : dil == 4 ? 2 : 3;
    const int tp = ((p16 >> ld) << (ld + 1)) + (p16 & (dil - 1));
    f32x4 y0[4], y1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y0[i] = y1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    BSG_STAMP(2);
    wino_gemm1<4, 2>(y0, y1, AW, xs, lq * LDX + HALO + tp, dil, rs_aw, vfrag, sw);
    BSG_STAMP(3);
    // ---- (4w) + hoisted conditioner term, gate ----------------------------------------------------
    const int f0c = t0 + tp < T ? t0 + tp : T - 1, f1c = t0 + tp + dil < T ? t0 + tp + dil : T - 1;
    const int vc0 = (lq * 4 * T + f0c) * 4, vc1 = (lq * 4 * T + f1c) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int so_g = (32 * wave + 16 * i + r) * rowT, so_f = so_g + C * rowT;
        const float cg0 = ldf(rs_ct, vc0, so_g), cf0 = ldf(rs_ct, vc0, so_f);
        const float cg1 = ldf(rs_ct, vc1, so_g), cf1 = ldf(rs_ct, vc1, so_f);
        z[i * 8 + r] = fast_sigmoid(y0[i][r] + cg0) * fast_tanh(y0[2 + i][r] + cf0);
        z[i * 8 + 4 + r] = fast_sigmoid(y1[i][r] + cg1) * fast_tanh(y1[2 + i][r] + cf1);
        zoff[i * 8 + r] = (32 * wave + 16 * i + 4 * lq + r) * LDZ + tp;
        zoff[i * 8 + 4 + r] = (32 * wave + 16 * i + 4 * lq + r) * LDZ + tp + dil;
      }
  }
  BSG_STAMP(4);
  // loads GEMM2 needs first fly across the two barriers: its first A fragments and the residual input x,
  // which becomes the initial accumulator of the residual rows (x + b_out + W_out z, then / sqrt(2))
  const int sb_r = wave * 32 * 1024, sb_s = (8 + wave) * 32 * 1024;   // residual / skip tile of the packed output projection
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    Ag[k] = ldf4(rs_a2, vfrag, sb_r + k * 1024);
    Af[k] = ldf4(rs_a2, vfrag, sb_s + k * 1024);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if constexpr (!WINO) acc0[r] = ldf(rs_x, vcol, (32 * wave + acc_row0(r)) * rowT);
    acc1[r] = ldf(rs_bo, lh * 16, (C + 32 * wave + acc_row0(r)) * 4);
  }
  __syncthreads();  // every wave is done reading xs
#pragma unroll
  for (int r = 0; r < 16; ++r) zs[zoff[r]] = z[r];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] += ldf(rs_bo, lh * 16, (32 * wave + acc_row0(r)) * 4);
  __syncthreads();
  BSG_STAMP(5);

  // ---- (5) GEMM2: o = W_out * z; residual rows and skip rows share the B fragment ---------------
  float prevs[16];
  {
    const float* zrow = zs + lh * LDZ + l31;
    auto ldb = [&](int q) {
      const float* p = zrow + 8 * q * LDZ;
      return f32x4{p[0], p[2 * LDZ], p[4 * LDZ], p[6 * LDZ]};
    };
    mfma_pipe<NS>(acc0, acc1, Ag, Af, rs_a2, vfrag, sb_r, sb_s, 0, 16, 31, ldb);
    // the running skip sum is added after the chain (keeps the small products out of a large accumulator)
#pragma unroll
    for (int r = 0; r < 16; ++r) prevs[r] = ldf(rs_sk, vcol, (32 * wave + acc_row0(r)) * rowT);
    mfma_pipe<NS>(acc0, acc1, Ag, Af, rs_a2, vfrag, sb_r, sb_s, 16, 32, 31, ldb);
  }
  BSG_STAMP(6);
  if (col_ok) {
    const int vst = (lh * 4 * T + col) * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int so = (32 * wave + acc_row0(r)) * rowT;
      stf(acc0[r] / 1.41421356237309504880f, rs_xo, vst, so);                      // (x + residual) / sqrt(2), net.py:78
      stf(((a.first ? 0.f : prevs[r]) + acc1[r]) / a.skip_div, rs_sk, vst, so);    // running skip sum (/ sqrt(L) last, :126)
    }
  }
  BSG_STAMP(7);
  if (STAMP && lane == 0) a.stamps[((long long)tile_id * 8 + wave) * 10 + 9] = __builtin_amdgcn_s_memrealtime();
}
#undef BSG_MFMA8

template <bool STAMP, bool WINO>
__global__ __launch_bounds__(512, WINO ? 4 : 6) void residual_layer_kernel(ResArgs a) {
  // XCD-aware tile order: workgroup i runs on XCD i % 8; a contiguous run of tiles per XCD lets neighbouring tiles share
  // their halo lines in one L2 (the halo of a tile is its neighbours' core)
  const int n_tiles = a.B * a.tiles_per_row, per_xcd = (n_tiles + 7) >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile_id >= n_tiles) return;
  residual_tile<STAMP, 2, WINO>(a, tile_id);
}

// ------------------------------------------------------------------------------------------------
// Small launches (B * ceil(T/32) <= the number of CUs): one tile per workgroup leaves most of the chip idle and the layer
// is bound by ONE CU's matrix rate (393 k MFMA cycles per tile over 4 SIMDs = 42 us).  Here a tile is computed by a PAIR
// of workgroups (2*tile, 2*tile + 1), each owning half of the channels:
//   part p: gate rows [128p, 128p+128) + filter rows [256+128p, ...)  ->  z channels [128p, 128p+128)   (Winograd GEMM1)
//   exchange: each part stores its z half to a global scratch tile write-through (sc1), every wave drains (vmcnt(0)),
//             barrier, one relaxed agent-scope flag store (value = launch epoch); one lane polls the partner's flag (bounded),
//             ONE agent-scope acquire, barrier, plain loads of the partner half   (same protocol as the stack launch)
//   part p: residual rows [128p, ...) + skip rows [256+128p, ...) of GEMM2 over all 256 z channels
// Wave w owns ONE 16-row tile (gate/filter, then residual/skip): 16x16x4 MFMAs throughout.  The host only takes this
// path when every workgroup of the launch is resident (2 * tiles <= 2 per CU), so the partner is always running.
// ------------------------------------------------------------------------------------------------
struct SplitArgs {
  ResArgs base;
  const float* apack2w;   // output projection packed for 16x16x4 MFMAs [32 row tiles][16 k-groups][64][4]
  float* zbuf;            // [tiles][C][32] scratch for the z exchange
  unsigned* flags;        // [tiles][parts][parts]: (producer, consumer)
  unsigned* status;       // += 1 for a poll that gave up
  unsigned epoch;         // launch counter (never 0): the value a flag takes in this launch
  int inject;             // fault injection (bsg_diffnet_debug_inject_giveup): consumers give up at once, without waiting
};

// dconv[((s*L + l)*4 + v)*2C + row] = sum_ci w[row][ci][tap] * dproj[s][l][ci] for tap = v < 3, all three taps for v = 3 (fp64 accumulation):
// the share of the step term d (constant over the frames) in dilated_conv_l(x + d), which the 16-row stack launch adds to the conditioner
// term instead of adding d to every frame of its conv image (net.py:67,72-74).  One workgroup per 4 weight rows: the layer's step terms
// [S][C] (padded rows: the threads of a row group read one column of it) and the 4 rows' weights in LDS, a thread per (row, step).
constexpr int DCONV_ROWS = 4, DCONV_SMAX = 128;
__global__ __launch_bounds__(DCONV_ROWS * DCONV_SMAX) void dconv_kernel(const float* __restrict__ w, const float* __restrict__ dproj, float* __restrict__ out,
                                                                           int S, int L, int l) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float* ds = reinterpret_cast<float*>(lds_raw);            // [DCONV_SMAX][C + 1]: the steps blockIdx.y * DCONV_SMAX ..
  float* ws = ds + (size_t)DCONV_SMAX * (C + 1);            // [DCONV_ROWS][3 C]
  const int tid = threadIdx.x, row0 = blockIdx.x * DCONV_ROWS, s0 = blockIdx.y * DCONV_SMAX;
  const int ns = S - s0 < DCONV_SMAX ? S - s0 : DCONV_SMAX;
  for (int i = tid; i < ns * C; i += blockDim.x) ds[(i / C) * (C + 1) + i % C] = dproj[((long long)(s0 + i / C) * L + l) * C + i % C];
  for (int i = tid; i < DCONV_ROWS * 3 * C; i += blockDim.x) ws[i] = w[(long long)row0 * 3 * C + i];
  __syncthreads();
  const int r = tid / DCONV_SMAX, sl = tid % DCONV_SMAX, st = s0 + sl;
  if (sl >= ns) return;
  double acc[3] = {0.0, 0.0, 0.0};
  const float* wr = ws + r * 3 * C;
  const float* dr = ds + sl * (C + 1);
  for (int ci = 0; ci < C; ++ci) {
    const double d = (double)dr[ci];
    acc[0] += (double)wr[ci * 3] * d;
    acc[1] += (double)wr[ci * 3 + 1] * d;
    acc[2] += (double)wr[ci * 3 + 2] * d;
  }
  float* o = out + ((long long)st * L + l) * 4 * (2 * C) + row0 + r;
  o[0] = (float)acc[0];
  o[2 * C] = (float)acc[1];
  o[2 * 2 * C] = (float)acc[2];
  o[3 * 2 * C] = (float)(acc[0] + acc[1] + acc[2]);
}

// out[((mt*(K/16) + q)*64 + lane)*4 + jj] = W(m = 16*mt + (lane&15), k = 16*q + 4*jj + (lane>>4)),  W row-major [M][K]
__global__ void pack_a16_kernel(const float* __restrict__ w, float* __restrict__ out, int M, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * K) return;
  const int jj = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
  const int KQ = K / 16;
  const int q = rest % KQ, mt = rest / KQ;
  out[i] = w[(long long)(16 * mt + (lane & 15)) * K + 16 * q + 4 * jj + (lane >> 4)];
}

// WIDE = false: the pair form described above (2 workgroups x 8 waves per tile).  WIDE = true: ONE workgroup of 16 waves per
// tile — the same wave program, z exchanged through LDS only — for launches of 129..256 tiles (B = 5..8 at T=1000), where the
// regular launch has one 8-wave workgroup per CU: 4 waves per SIMD instead of 2 for the same matrix work.
// NPART = 4 (launches of <= 64 tiles, B <= 2): four workgroups of 4 waves per tile, each a quarter of the channels; every part
// publishes its z quarter to the three others through per-(producer, consumer) flags.
template <bool WIDE, int NPART>
__global__ __launch_bounds__(WIDE ? 1024 : 1024 / NPART, 4) void residual_split_kernel(SplitArgs s) {
  const ResArgs& a = s.base;
  constexpr int NT = 32, LDX = NT + 2 * HALO, LDZ = 48, NR = 4;   // NR: depth of the A-fragment rings (k-groups)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* zs = lds;   // [C][LDZ], aliases xs after the barrier that ends GEMM1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, lq = lane >> 4;
  constexpr int WAVES = WIDE ? 16 : 16 / NPART, NTHR = 64 * WAVES;
  const int tile_id = WIDE ? (int)blockIdx.x : (int)blockIdx.x / NPART;
  const int part = WIDE ? 0 : (int)blockIdx.x % NPART;
  const int b = tile_id / a.tiles_per_row;
  const int t0 = (tile_id - b * a.tiles_per_row) * NT;
  const int T = a.T;
  const int tb = a.t_dev ? (int)a.t_dev[b] : a.t_uniform;

  const unsigned plane = (unsigned)C * T * 4;
  const rsrc_t rs_x = mk_rsrc(a.x_in + (long long)b * C * T, plane);
  const rsrc_t rs_xo = mk_rsrc(a.x_out + (long long)b * C * T, plane);
  const rsrc_t rs_sk = mk_rsrc(a.skip + (long long)b * C * T, plane);
  const rsrc_t rs_ct = mk_rsrc(a.condterm + (long long)b * 2 * C * T, 2 * plane);
  const rsrc_t rs_aw = mk_rsrc(a.apackw, 4 * 2 * C * C * 4);
  const rsrc_t rs_a2 = mk_rsrc(s.apack2w, 2 * C * C * 4);
  const rsrc_t rs_bo = mk_rsrc(a.bias_out, 2 * C * 4);
  const rsrc_t rs_dp = mk_rsrc(a.dproj + ((long long)tb * a.L + a.layer) * C, C * 4);
  const rsrc_t rs_zb = mk_rsrc(s.zbuf + (long long)tile_id * C * NT, C * NT * 4);
  const int rowT = T * 4, vfrag = lane * 16;
  const int gt = WIDE ? wave : WAVES * part + wave;   // this wave's 16-row tile: gate 16gt.., filter C+16gt..; later residual 16gt.., skip C+16gt..

  // ---- (1) first A fragments fly while the x tile is staged ---------------------------------------
  const int sw[2] = {gt * 16 * 1024, (16 + gt) * 16 * 1024};
  f32x4 AW[NR][2];
#pragma unroll
  for (int k = 0; k < NR; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i) AW[k][i] = ldf4(rs_aw, vfrag, sw[i] + k * 1024);

  // ---- (2) stage xs = x + d (zero padded), as residual_tile ---------------------------------------
  if ((T & 3) == 0) {
#pragma unroll 3
    for (int k = 0; k < 3072 / NTHR; ++k) {
      const int idx = tid + NTHR * k;
      const int c = idx / 12, j4 = idx - c * 12;
      const int t = t0 - HALO + 4 * j4;
      const bool ok = t >= 0 && t < T;
      f32x4 v = ldf4(rs_x, ok ? (c * T + t) * 4 : 0, 0);
      const float d = ldf(rs_dp, c * 4, 0);
      v += d;
      if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(xs + c * LDX + 4 * j4) = v;
    }
  } else {
#pragma unroll 4
    for (int idx = tid; idx < C * LDX; idx += NTHR) {
      const int c = idx / LDX, j = idx - c * LDX;
      const int t = t0 - HALO + j;
      const bool ok = t >= 0 && t < T;
      const float v = ldf(rs_x, ok ? (c * T + t) * 4 : 0, 0) + ldf(rs_dp, c * 4, 0);
      xs[idx] = ok ? v : 0.f;
    }
  }
  __syncthreads();

  // ---- (3) GEMM1, Winograd F(2,3) over the dilated taps (see residual_tile), one gate + one filter tile of 16 rows ----
  const int dil = a.dil;
  const int ld = dil == 1 ? 0 : dil == 2 ? 1 : dil == 4 ? 2 : 3;
  const int tp = ((p16 >> ld) << (ld + 1)) + (p16 & (dil - 1));
  f32x4 y0[2], y1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) y0[i] = y1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  wino_gemm1<2, NR>(y0, y1, AW, xs, lq * LDX + HALO + tp, dil, rs_aw, vfrag, sw);
  // ---- (4) + conditioner term, gate --------------------------------------------------------------------
  float z0[4], z1[4];
  {
    const int f0c = t0 + tp < T ? t0 + tp : T - 1, f1c = t0 + tp + dil < T ? t0 + tp + dil : T - 1;
    const int vc0 = (lq * 4 * T + f0c) * 4, vc1 = (lq * 4 * T + f1c) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int so_g = (16 * gt + r) * rowT, so_f = so_g + C * rowT;
      const float cg0 = ldf(rs_ct, vc0, so_g), cf0 = ldf(rs_ct, vc0, so_f);
      const float cg1 = ldf(rs_ct, vc1, so_g), cf1 = ldf(rs_ct, vc1, so_f);
      z0[r] = fast_sigmoid(y0[0][r] + cg0) * fast_tanh(y0[1][r] + cf0);
      z1[r] = fast_sigmoid(y1[0][r] + cg1) * fast_tanh(y1[1][r] + cf1);
    }
  }
  // loads GEMM2 needs first: its A fragments, the residual input (initial accumulator) and the biases
  const int sr = gt * 16 * 1024, ss = (16 + gt) * 16 * 1024;
  f32x4 AR[NR], AS[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    AR[k] = ldf4(rs_a2, vfrag, sr + k * 1024);
    AS[k] = ldf4(rs_a2, vfrag, ss + k * 1024);
  }
  int vcol[2];
  bool col_ok[2];
  f32x4 accR[2], accS[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = t0 + 16 * ct + p16;
    col_ok[ct] = col < T;
    vcol[ct] = (lq * 4 * T + (col_ok[ct] ? col : T - 1)) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      accR[ct][r] = ldf(rs_x, vcol[ct], (16 * gt + r) * rowT) + ldf(rs_bo, lq * 16, (16 * gt + r) * 4);
      accS[ct][r] = ldf(rs_bo, lq * 16, (C + 16 * gt + r) * 4);
    }
  }
  __syncthreads();   // every wave is done reading xs
  // ---- (5) z: own half -> LDS and (write-through) -> the exchange tile; partner half <- exchange tile ------
  if constexpr (WIDE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * gt + 4 * lq + r;
      zs[row * LDZ + tp] = z0[r];
      zs[row * LDZ + tp + dil] = z1[r];
    }
    __syncthreads();
  } else {
  {
    const int vz = ((4 * lq) * NT + tp) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * gt + 4 * lq + r;
      zs[row * LDZ + tp] = z0[r];
      zs[row * LDZ + tp + dil] = z1[r];
      if (s.inject && part == 0) continue;   // injected fault: part 0 never publishes, its partners consume whatever the tile held
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, z0[r]), rs_zb, vz, (16 * gt + r) * NT * 4, 16);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, z1[r]), rs_zb, vz + dil * 4, (16 * gt + r) * NT * 4, 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // flags[tile][producer][consumer]: a producer publishes to each of the other parts; a consumer clears what it has consumed,
    // so that a replay of this very launch (a captured graph holding a single layer launch replays the same epoch) cannot match
    // a stale value; the producer sets it again only in a later launch
    unsigned* f = s.flags + (long long)tile_id * (NPART * NPART);
#pragma unroll
    for (int o = 0; o < NPART; ++o)
      if (o != part) __hip_atomic_store(f + part * NPART + o, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 0; o < NPART; ++o) {
      if (o == part) continue;
      unsigned spins = s.inject ? (1u << 22) : 0u;   // injected fault: the first failed poll gives up (or none is made at all)
      if (s.inject) atomicAdd(s.status, 1u);
      while (!s.inject && __hip_atomic_load(f + o * NPART + part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != s.epoch) {
        __builtin_amdgcn_s_sleep(2);
        // ~ seconds: never reached unless a partner is not resident; once any wait of this handle has given up (status != 0: the host
        // repeats the call without hand-offs anyway) the others stop within a thousand polls
        if (++spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load(s.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
          atomicAdd(s.status, 1u);
          break;
        }
      }
      __hip_atomic_store(f + o * NPART + part, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  {
    // the other parts' channels: (NPART-1) * C/NPART rows of 8 float4
    constexpr int CP = C / NPART, ITEMS = (NPART - 1) * CP * 8;
#pragma unroll
    for (int k = 0; k < ITEMS / NTHR; ++k) {
      const int idx = tid + NTHR * k;
      int c = idx >> 3;
      const int j4 = idx & 7;
      c += c >= part * CP ? CP : 0;   // skip the own block
      *reinterpret_cast<f32x4*>(zs + c * LDZ + 4 * j4) = ldf4(rs_zb, (c * NT + 4 * j4) * 4, 0);
    }
  }
  __syncthreads();
  }
  // ---- (6) GEMM2: residual tile + skip tile of 16 rows x 2 column tiles of 16 frames -------------------------
  {
    const float* zb = zs + lq * LDZ + p16;
    auto ldbz = [&](int q, int ct) {
      const float* p = zb + 16 * q * LDZ + 16 * ct;
      return f32x4{p[0], p[4 * LDZ], p[8 * LDZ], p[12 * LDZ]};
    };
    f32x4 Bz[2][2];
    Bz[0][0] = ldbz(0, 0);
    Bz[0][1] = ldbz(0, 1);
#pragma unroll 1
    for (int q = 0; q < 16; q += NR) {
#pragma unroll
      for (int s2 = 0; s2 < NR; ++s2) {
        const int qn = q + s2 + 1 < 16 ? q + s2 + 1 : 15;
        Bz[(s2 + 1) & 1][0] = ldbz(qn, 0);
        Bz[(s2 + 1) & 1][1] = ldbz(qn, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          accR[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AR[s2][jj], Bz[s2 & 1][0][jj], accR[0], 0, 0, 0);
          accS[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AS[s2][jj], Bz[s2 & 1][0][jj], accS[0], 0, 0, 0);
          accR[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AR[s2][jj], Bz[s2 & 1][1][jj], accR[1], 0, 0, 0);
          accS[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AS[s2][jj], Bz[s2 & 1][1][jj], accS[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int qr = q + s2 + NR < 16 ? q + s2 + NR : 15;
        AR[s2] = ldf4(rs_a2, vfrag, sr + qr * 1024);
        AS[s2] = ldf4(rs_a2, vfrag, ss + qr * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // ---- (7) epilogue --------------------------------------------------------------------------------
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float prevs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prevs[r] = a.first ? 0.f : ldf(rs_sk, vcol[ct], (16 * gt + r) * rowT);
    if (col_ok[ct]) {
      const int vst = (lq * 4 * T + t0 + 16 * ct + p16) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int so = (16 * gt + r) * rowT;
        stf(accR[ct][r] / 1.41421356237309504880f, rs_xo, vst, so);   // (x + residual) / sqrt(2), net.py:78
        stf((prevs[r] + accS[ct][r]) / a.skip_div, rs_sk, vst, so);   // running skip sum (/ sqrt(L) last, :126)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stack launches: all L residual layers of a group of rows in ONE launch with the residual stream on chip (diffnet_h2.hip: split-fp16
// form, the fp32 default; diffnet_f43.hip: Winograd F(4,3) on the fp32 matrix pipe; diffnet_bf16.hip: bf16-operand configuration; the
// host side — stack_rows / launch_stack — is below).  (The first form, a F(2,3) kernel with two 32-frame workgroups per CU, lived here
// in the first half of round 2; it was 3 % slower than two chains of per-layer launches, was superseded by the forms above and removed.)
// The hand-off protocol all of them share: per layer the only inter-workgroup traffic is the 8-frame edge of the new conv image that
// each of the two neighbour tiles needs as its halo (dilation <= 8): published write-through (sc1) into an exchange array that is
// double-buffered by layer parity; flag[tile] = launch epoch * 64 + number of layers published.  Producer: sc1 stores -> every storing
// wave s_waitcnt vmcnt(0) -> barrier -> one relaxed agent-scope flag store.  Consumer: one lane polls the two neighbour flags (relaxed,
// s_sleep, bounded: a give-up is counted in `status` and handled by the host in the same call) -> barrier -> sc1 buffer loads of the
// handed-off bytes to registers (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md hand-off "valid forms": no agent-scope
// acquire, which would invalidate the CU's L1 — the weight stream's — every layer).  Double buffering is enough: a tile can only
// publish layer l+2 after its neighbours published l+1, which they do after reading its l.  Every workgroup of a launch must be
// resident (neighbours wait for each other): the host launches whole rows, at most one workgroup per CU, never inside a stream capture.
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// sampler: one ancestral step, elementwise over [B][M][T]   (shallow_diffusion_tts.py:134-166)
// ------------------------------------------------------------------------------------------------
__global__ void ddpm_step_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                                 StepCoef k, long long n4, unsigned long long seed, unsigned stream,
                                 unsigned long long quad0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 ev = reinterpret_cast<const f32x4*>(eps)[i];
  f32x4 nv;
  if (noise) nv = reinterpret_cast<const f32x4*>(noise)[i];
  else if (k.sigma != 0.f) nv = philox_normal4(seed, stream, quad0 + (unsigned long long)i);
  else nv = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // every product/sum rounded separately, like the reference's elementwise ATen ops (no FMA contraction)
    float x0 = __fsub_rn(__fmul_rn(k.recip, xv[e]), __fmul_rn(k.recipm1, ev[e]));   // :134-138
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                                              // :154
    const float mean = __fadd_rn(__fmul_rn(k.pc1, x0), __fmul_rn(k.pc2, xv[e]));    // :141-144
    o[e] = __fadd_rn(mean, __fmul_rn(k.sigma, nv[e]));                               // :166
  }
  reinterpret_cast<f32x4*>(x)[i] = o;
}

// ------------------------------------------------------------------------------------------------
// step tail + next head, fused (DDPM loop only): for one 32-frame tile
//   h    = relu(W_skip * s + b)            s = skip sum / sqrt(L), written by the last residual layer   (net.py:126-128)
//   eps  = W_out * h + b                                                                               (net.py:129)
//   x   <- p_sample(x, eps, noise)         same arithmetic as ddpm_step_kernel                  (shallow_diffusion_tts.py:149-166)
//   xa   = relu(W_in * x + b)              the NEXT step's input projection                            (net.py:116-118)
// Replaces three GEMM launches + the sampler launch per step (~150 us -> ~20 us at B=16, T=1000): the three
// projections are too small (0.04-0.13 MFLOP/frame) to fill the chip as separate 128x128-tile GEMMs.
// ------------------------------------------------------------------------------------------------
#define BSG_MFMA4(ACC, A_, B_)                                                \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[0], B_[0], ACC, 0, 0, 0);      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[1], B_[1], ACC, 0, 0, 0);      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[2], B_[2], ACC, 0, 0, 0);      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[3], B_[3], ACC, 0, 0, 0);

// acc += A[tile rows] * Bs, K = 8*NQ, A fragments from `rs` at byte offset `sbase`, B rows from LDS `bs` (stride 32)
template <int NQ>
__device__ __forceinline__ void tile_gemm(f32x16& acc, rsrc_t rs, int vfrag, int sbase, const float* brow) {
  auto ldb = [&](int q) {
    const float* p = brow + 8 * q * 32;
    return f32x4{p[0], p[2 * 32], p[4 * 32], p[6 * 32]};
  };
  f32x4 A0 = ldf4(rs, vfrag, sbase), A1 = ldf4(rs, vfrag, sbase + (NQ > 1 ? 1024 : 0));
  f32x4 B0 = ldb(0), B1;
#pragma unroll 1
  for (int q = 0; q < NQ; q += 2) {
    B1 = ldb(q + 1 < NQ ? q + 1 : NQ - 1);
    __builtin_amdgcn_sched_barrier(0);
    BSG_MFMA4(acc, A0, B0)
    __builtin_amdgcn_sched_barrier(0);
    const int q2 = q + 2 < NQ ? q + 2 : NQ - 1;
    A0 = ldf4(rs, vfrag, sbase + q2 * 1024);
    B0 = ldb(q2);
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < NQ) { BSG_MFMA4(acc, A1, B1) }
    __builtin_amdgcn_sched_barrier(0);
    const int q3 = q + 3 < NQ ? q + 3 : NQ - 1;
    A1 = ldf4(rs, vfrag, sbase + q3 * 1024);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MP, bool PLMS>   // MP = in_dims padded to a multiple of 8 (K of the head GEMM), in_dims <= 96; PLMS: multistep update form
__global__ __launch_bounds__(512, 4) void step_tail_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ss = lds;               // [C][32]  skip tile, then h tile
  float* xin = lds + C * 32;     // [96][32] updated x tile (rows >= M zero)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / a.tiles_per_row;
  const int t0 = (blockIdx.x - b * a.tiles_per_row) * 32;
  const int T = a.T, M = a.M;
  const int col = t0 + l31;
  const bool col_ok = col < T;
  const int colc = col_ok ? col : T - 1;
  const rsrc_t rs_s = mk_rsrc(a.skip + (long long)b * C * T, (unsigned)C * T * 4);
  const rsrc_t rs_xa = mk_rsrc(a.xa_next + (long long)b * C * T, (unsigned)C * T * 4);
  const rsrc_t rs_x = mk_rsrc(a.x + (long long)b * M * T, (unsigned)M * T * 4);
  const rsrc_t rs_ws = mk_rsrc(a.ws_pack, C * C * 4);
  const rsrc_t rs_wo = mk_rsrc(a.wo_pack, 96 * C * 4);
  const rsrc_t rs_wi = mk_rsrc(a.wi_pack, C * MP * 4);
  const rsrc_t rs_bs = mk_rsrc(a.b_skip, C * 4);
  const rsrc_t rs_bf = mk_rsrc(a.b_fin, 96 * 4);
  const rsrc_t rs_bi = mk_rsrc(a.b_in, C * 4);
  const int vfrag = lane * 16, rowT = T * 4;
  const int vcol = (lh * 4 * T + colc) * 4;

  // ---- stage the skip tile ---------------------------------------------------------------------
  if (a.skip_h) {
    // channel-quad bf16: 64 quads x 32 frames, one 8-byte load per item
    const rsrc_t rs_sh = mk_rsrc(a.skip_h + (long long)b * C * T, (unsigned)C * T * 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + 512 * k;
      const int q = idx >> 5, f = idx & 31;
      const int t = t0 + f;
      using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
      u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_sh, t < T ? (q * T + t) * 8 : 0, 0, 0));
      if (t >= T) v = u32x2{0u, 0u};
      ss[(4 * q) * 32 + f] = bf16_lo(v[0]);
      ss[(4 * q + 1) * 32 + f] = bf16_hi(v[0]);
      ss[(4 * q + 2) * 32 + f] = bf16_lo(v[1]);
      ss[(4 * q + 3) * 32 + f] = bf16_hi(v[1]);
    }
  } else if ((T & 3) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + 512 * k;
      const int c = idx >> 3, j4 = idx & 7;
      const int t = t0 + 4 * j4;
      f32x4 v = ldf4(rs_s, t < T ? (c * T + t) * 4 : 0, 0);
      if (t >= T) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(ss + c * 32 + 4 * j4) = v;
    }
  } else {
    for (int idx = tid; idx < C * 32; idx += 512) {
      const int c = idx >> 5, t = t0 + (idx & 31);
      const float v = ldf(rs_s, t < T ? (c * T + t) * 4 : 0, 0);
      ss[idx] = t < T ? v : 0.f;
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = ldf(rs_bs, lh * 16, (32 * wave + acc_row0(r)) * 4);
  __syncthreads();
  // ---- h = relu(W_skip s + b) ------------------------------------------------------------------
  tile_gemm<32>(acc, rs_ws, vfrag, wave * 32 * 1024, ss + lh * 32 + l31);
  __syncthreads();   // every wave is done reading s
#pragma unroll
  for (int r = 0; r < 16; ++r) ss[(32 * wave + acc_row(r, lh)) * 32 + l31] = fmaxf(acc[r], 0.f);
  __syncthreads();
  // ---- eps = W_out h + b; sampler update on the 3 row tiles that cover the M mel bins -----------
  if (wave < 3) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ldf(rs_bf, lh * 16, (32 * wave + acc_row0(r)) * 4);
    float xv[16], nv[16];
    const rsrc_t rs_n = mk_rsrc(a.noise ? a.noise + (long long)b * M * T : a.x, a.noise ? (unsigned)M * T * 4 : 0u);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // the lane's row is m0 + 4*lh; rows >= M fall outside the descriptor's range and read as 0 (raw-buffer
      // range check), and are never stored
      const int m0 = 32 * wave + acc_row0(r);
      xv[r] = ldf(rs_x, vcol, m0 * rowT);
      nv[r] = a.noise ? ldf(rs_n, vcol, m0 * rowT) : 0.f;
    }
    float h1v[PLMS ? 16 : 1], h2v[PLMS ? 16 : 1], h3v[PLMS ? 16 : 1];
    if constexpr (PLMS) {
      const unsigned hb = (unsigned)M * T * 4;
      const rsrc_t rs_h1 = mk_rsrc(a.h1 + (long long)b * M * T, hb);
      const rsrc_t rs_h2 = mk_rsrc(a.plms_hist > 1 ? a.h2 + (long long)b * M * T : a.x, a.plms_hist > 1 ? hb : 0u);
      const rsrc_t rs_h3 = mk_rsrc(a.plms_hist > 2 ? a.h3 + (long long)b * M * T : a.x, a.plms_hist > 2 ? hb : 0u);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int so = (32 * wave + acc_row0(r)) * rowT;
        h1v[r] = ldf(rs_h1, vcol, so);
        h2v[r] = ldf(rs_h2, vcol, so);   // zero-size descriptors read as 0
        h3v[r] = ldf(rs_h3, vcol, so);
      }
    }
    tile_gemm<32>(acc, rs_wo, vfrag, wave * 32 * 1024, ss + lh * 32 + l31);
    const int vst = (lh * 4 * T + col) * 4;
    const rsrc_t rs_en = mk_rsrc(PLMS ? a.e_new + (long long)b * M * T : a.x, PLMS ? (unsigned)M * T * 4 : 0u);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * wave + acc_row(r, lh);
      float o = 0.f;
      if (m < M) {
        if constexpr (PLMS) {
          o = plms_update(xv[r], acc[r], h1v[r], h2v[r], h3v[r], a.plms_hist, a.pk, nullptr);
          if (col_ok) stf(acc[r], rs_en, vst, (32 * wave + acc_row0(r)) * rowT);
        } else {
          float nz = nv[r];
          if (!a.noise && a.k.sigma != 0.f)
            nz = philox_normal1(a.seed, a.stream, a.quad_row0 + ((unsigned long long)b * M + m) * T + colc);
          float x0 = __fsub_rn(__fmul_rn(a.k.recip, xv[r]), __fmul_rn(a.k.recipm1, acc[r]));
          x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
          const float mean = __fadd_rn(__fmul_rn(a.k.pc1, x0), __fmul_rn(a.k.pc2, xv[r]));
          o = __fadd_rn(mean, __fmul_rn(a.k.sigma, nz));
        }
        if (col_ok) stf(o, rs_x, vst, (32 * wave + acc_row0(r)) * rowT);
      }
      xin[m * 32 + l31] = o;
    }
  }
  if (!a.do_head) return;
  __syncthreads();
  // ---- next step's input projection: xa = relu(W_in x + b) --------------------------------------
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = ldf(rs_bi, lh * 16, (32 * wave + acc_row0(r)) * 4);
  tile_gemm<MP / 8>(acc, rs_wi, vfrag, wave * (MP / 8) * 1024, xin + lh * 32 + l31);
  if (col_ok) {
    const int vst = (lh * 4 * T + col) * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) stf(fmaxf(acc[r], 0.f), rs_xa, vst, (32 * wave + acc_row0(r)) * rowT);
  }
}
#undef BSG_MFMA4

__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows_src, int rows_dst, int cols_src,
                                int cols_dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_dst * cols_dst) return;
  const int r = i / cols_dst, c = i - r * cols_dst;
  dst[i] = (r < rows_src && c < cols_src) ? src[(long long)r * cols_src + c] : 0.f;
}

__global__ void philox_fill_kernel(float* __restrict__ x, long long n4, unsigned long long seed, unsigned stream,
                                   unsigned long long quad0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  reinterpret_cast<f32x4*>(x)[i] = philox_normal4(seed, stream, quad0 + (unsigned long long)i);
}

// mel_out[b,t,m] = denorm(x[b,m,t]) * (mel2ph[b,t] > 0)     (shallow_diffusion_tts.py:268-272, :278-279)
// [M][64]-frame tile through LDS: reads coalesced along t, writes coalesced along m.
__global__ __launch_bounds__(256) void mel_finish_kernel(const float* __restrict__ x, const float* __restrict__ smin,
                                                         const float* __restrict__ smax, const long long* __restrict__ mel2ph,
                                                         float* __restrict__ out, int M, int T) {
  extern __shared__ float tile[];   // [M][65]
  const int b = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int i = tid; i < M * 64; i += 256) {
    const int m = i >> 6, j = i & 63;
    const int t = t0 + j;
    tile[m * 65 + j] = t < T ? x[((long long)b * M + m) * T + t] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < 64 * M; i += 256) {
    const int j = i / M, m = i - j * M;
    const int t = t0 + j;
    if (t >= T) continue;
    const float lo = smin[m], hi = smax[m];
    float v = __fadd_rn(__fmul_rn(__fadd_rn(tile[m * 65 + j], 1.0f) / 2.0f, __fsub_rn(hi, lo)), lo);
    if (mel2ph && mel2ph[(long long)b * T + t] <= 0) v = v * 0.f;
    out[((long long)b * T + t) * M + m] = v;
  }
}

// shallow-diffusion start x = q_sample(norm_spec(fs2_mel)^T, K_step-1)          (:249-252, :203-208, :275-276)
__global__ __launch_bounds__(256) void mel_start_kernel(const float* __restrict__ mel, const float* __restrict__ smin,
                                                        const float* __restrict__ smax, const float* __restrict__ noise,
                                                        float ca, float cb, float* __restrict__ x, int M, int T) {
  extern __shared__ float tile[];   // [64][M+1]
  const int b = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int i = tid; i < 64 * M; i += 256) {
    const int j = i / M, m = i - j * M;
    const int t = t0 + j;
    float v = 0.f;
    if (t < T) {
      const float lo = smin[m], hi = smax[m];
      v = __fsub_rn(__fmul_rn(__fsub_rn(mel[((long long)b * T + t) * M + m], lo) / __fsub_rn(hi, lo), 2.0f), 1.0f);
    }
    tile[j * (M + 1) + m] = v;
  }
  __syncthreads();
  for (int i = tid; i < M * 64; i += 256) {
    const int m = i >> 6, j = i & 63;
    const int t = t0 + j;
    if (t >= T) continue;
    const long long o = ((long long)b * M + m) * T + t;
    x[o] = __fadd_rn(__fmul_rn(ca, tile[j * (M + 1) + m]), __fmul_rn(cb, noise[o]));
  }
}

// PLMS transfer x_pred = x + x_delta  (shallow_diffusion_tts.py:174-182); eps' = blend of eps history (:191-198)
__global__ void plms_step_kernel(const float* __restrict__ x, float* __restrict__ xo, const float* __restrict__ e0,
                                 const float* __restrict__ e1, const float* __restrict__ e2,
                                 const float* __restrict__ e3, PlmsCoef k, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int n_hist = e3 ? 3 : e2 ? 2 : e1 ? 1 : 0;
  xo[i] = plms_update(x[i], e0[i], e1 ? e1[i] : 0.f, e2 ? e2[i] : 0.f, e3 ? e3[i] : 0.f, n_hist, k, nullptr);
}

}  // namespace

}  // namespace bsg

// ================================================================================================
// host side
// ================================================================================================
using namespace bsg;

struct bsg_diffnet {
  bsg_diffnet_cfg cfg;
  int M, L;
  Guard guard;   // this handle's range-event word and split-fp16 GEMM switch (input / conditioner projections)
  // packed / derived weights (library-owned)
  float* w_in = nullptr;    // [C][M]
  float* b_in = nullptr;    // [C]
  float* apack1 = nullptr;  // [L][2C*3C]
  float* apack2 = nullptr;  // [L][2C*C]
  float* apackw = nullptr;  // [L][4*2C*C]  Winograd form of the dilated conv
  float* apackw43 = nullptr;  // [L][6*2C*C]  Winograd F(4,3) form (diffnet_f43.hip)
  unsigned short* apack1h = nullptr;  // [L][2C*3C] bf16 fragments (bf16-operand form, diffnet_bf16.hip)
  unsigned short* apack2h = nullptr;  // [L][2C*C]
  unsigned short* apack1s = nullptr;  // [L][2 planes][2C*3C] hi / lo fp16 fragments (split-fp16 form of the fp32 stack launch, diffnet_h2.hip)
  unsigned short* apack2s = nullptr;  // [L][2 planes][2C*C]
  float* h2_scale = nullptr;          // [L][4] power-of-two scales of the split-fp16 form
  unsigned short* tail_s = nullptr;   // split-fp16 step tail (fused into the h2 stack launch): skip [2*C*C], output [2*96*C], input projection [2*C*96]
  float* tail_scale = nullptr;        // [3][2] scales of the three projections (+ 3 words of scratch)
  unsigned short* tail_h = nullptr;   // bf16 step tail (step_tail_bf16_kernel): skip projection [C*C], output projection [96*C], input projection [C*96]
  int compute = BSG_COMPUTE_F32;      // bsg_diffnet_set_compute
  unsigned short* condterm_h = nullptr;  // [L][B][2C/4][T][4] bf16 (bf16 mode only)
  unsigned short* skip_h = nullptr;      // [B][C/4][T][4] bf16
  size_t cap_bt_h = 0;
  int prepared_compute = BSG_COMPUTE_F32;   // mode the bound condition was prepared for
  unsigned short* wcond_pack = nullptr;  // [L][2 x 2C x H] the conditioner projections as pre-split fp16 fragments (gemm_h2w.hip)
  unsigned* pack_bad = nullptr;          // device word: packed conditioner weights beyond the fp16 range (then wcond_pack is not used)
  bool cond_h2w_ok = false;
  unsigned short* cond_planes = nullptr; // [2][B][T][H] fp16: hi / lo of 16 x cond, transposed (the activation operand of the pre-split GEMM)
  size_t cap_cond_planes = 0;            // B x T the planes are sized for
  float* w_cond = nullptr;  // [L][2C][H]
  float* b_cond = nullptr;  // [L][2C]  (b_cond + b_dil)
  float* b_out = nullptr;   // [L][2C]
  float* w_skip = nullptr;  // [C][C]
  float* b_skip = nullptr;
  float* w_fin = nullptr;   // [M][C]
  float* b_fin = nullptr;
  float* dproj = nullptr;   // [S][L][C]
  float* dconv = nullptr;   // [S][L][4][2C]: dilated_conv_l's taps applied to dproj[s][l] (tap 0, 1, 2, their sum): diffnet_h2q.hip
  // fused step tail/head (DDPM loop): packed skip / output / input projections
  float* ws_pack = nullptr;  // [C*C]
  float* wo_pack = nullptr;  // [96*C]
  float* wi_pack = nullptr;  // [C*MP]
  float* b_fin96 = nullptr;  // [96]
  int MP = 0;
  // workspaces for the bound (B,T)
  int B = 0, T = 0;
  size_t cap_bt = 0;
  float* condterm = nullptr;  // [L][B][2C][T]
  float* condterm_q = nullptr;   // the same in channel-quad order [L][B][2C/4][T][4]: what the 16-row stack launch loads (16 bytes per lane)
  size_t cap_cond_q = 0;         // frames it holds
  bool cond_q_valid = false;     // written by the last prepare
  // the ROW layout (condterm) is what the fallback launches read (32-row stack launch, F(4,3), per-layer and channel-split kernels); the
  // default launches read the quads.  Round 6: prepare writes ONE layout — the quads when the shape's default launch reads them — and the
  // rows are derived from the quads by ensure_cond_rows() the first time a launch that reads them runs (655 MB less per pass at B = 16)
  size_t cap_cond_rows = 0;      // frames condterm holds (allocated on demand)
  bool cond_rows_valid = false;  // condterm holds the term of the bound condition
  float* xa = nullptr;        // [B][C][T]
  float* xb = nullptr;
  float* skip = nullptr;
  float* hid = nullptr;
  float* eps = nullptr;       // [B][M][T]
  float* eps_hist[4] = {nullptr, nullptr, nullptr, nullptr};  // PLMS history + x_pred scratch
  float* xpred = nullptr;
  // live timing of the dominant kernel: one event pair around the L residual-layer launches of a forward
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  size_t prof_launches = 0;   // layer-equivalents covered by the recorded pairs (L per evaluation)
  // on-chip stack launches: edge exchange [2][tiles][2][C][8], flags [tiles] + status word
  float* hx = nullptr;
  unsigned* flags = nullptr;
  size_t flags_cap = 0;                // tiles the exchange array and the flags are sized for
  unsigned* epoch_dev = nullptr;       // [2] launch epoch of the stack / part launches in device memory + workgroups done (diffnet_res.h StackArgs::epoch)
  unsigned long long* clk = nullptr;   // [4] s_memtime / s_memrealtime at the start and end of tile 0 of the last profiled stack launch
  int occ_stack_h = -1;                // the same for residual_stack_bf16_kernel
  int occ_stack43 = -1;                // the same for residual_stack_f43_kernel
  bool stack_is_f43 = false;           // the last stack_rows() chose the F(4,3) stack launch
  bool stack_is_h2 = false;            // ... the split-fp16 stack launch (diffnet_h2.hip)
  bool h2_off = false;                 // bsg_diffnet_set_h2(h, 0): this handle multiplies on the fp32 matrix pipe only
  bool q_off = false;                  // bsg_diffnet_set_h2q(h, 0): the 32-row stack launch (|x + d| < 60000) instead of the 16-row one (|x| < 3750)
  int occ_stack_h2q[3] = {-1, -1, -1}; // the same for residual_stack_q_kernel (16-row matrix tiles, diffnet_h2q.hip)
  bool stack_q = false;                // the stack launch of the current shape runs on 16-row matrix tiles
  int occ_stack_h2[3] = {-1, -1, -1};  // resident workgroups per CU of residual_stack_h2_kernel<.., NCT> by NCT (-1: not queried)
  int stack_nct = 2;                   // column tiles of 32 frames per workgroup the last stack_rows() chose for the split-fp16 launch
  int stack_parts = 0;                 // ... a part form (diffnet_h2.hip residual_part_h2_kernel): workgroups per tile (4 / 2), else 0
  int occ_part[3] = {-1, -1, -1};      // resident workgroups per CU: [1] / [2] quad of 32- / 64-frame tiles, [0] pair of 64-frame tiles (-1: not queried)
  unsigned short *apack1q = nullptr, *apack2q = nullptr;   // the split-fp16 weights once more as 16-row fragments (part forms)
  unsigned short* part_zx = nullptr;   // part forms: exchange slots of the z parts [tiles][P][2 planes][tile frames][C/P] fp16
  unsigned short* part_ix = nullptr;   //             ... of the image parts [2 parities][tiles][P][2 planes][tile frames][C/P]
  unsigned* part_flags = nullptr;      //             [2][tiles][P] image / z flags
  size_t part_cap = 0;                 // 32-frame tile equivalents the exchange buffers are sized for
  int num_cus = 0;
  const char* last_path = "none";      // form of the last residual-layer launch (bsg_diffnet_last_path)
  // channel-split launch for small batches (residual_split_kernel)
  float* apack2w = nullptr;            // [L][2C*C] output projection packed for 16x16x4 MFMAs
  float* zbuf = nullptr;               // [tiles][C][32] z exchange scratch
  unsigned* split_flags = nullptr;     // [tiles][16] (producer, consumer) flags + status word
  size_t split_cap = 0;                // tiles the scratch is sized for
  unsigned split_epoch = 0;
  bool split_off = false;              // bsg_diffnet_set_split(h, 0): regular launches only (the self-heal path after a give-up)
  int inject_giveup = 0;               // bsg_diffnet_debug_inject_giveup: split launches left that give up without waiting
  int inject_nowait = 0;               // the same entry with a NEGATIVE count: part launches left that skip their waits silently (timing experiment)
  int inject_xcc = 0;                  // bsg_diffnet_debug_inject_xcc: part launches left in which odd parts report another XCD
  bool parts_off = false;              // bsg_diffnet_set_parts(h, 0): no part forms (several workgroups per tile on CUs of ONE XCD); the one-workgroup-per-tile launches stay
  // residency of the split kernels on this handle's device (workgroups per CU; -1 = not queried yet): pair / 4-way form with the
  // padded LDS size (one workgroup per CU by construction) and with the plain size (two chains share a CU), 16-wave form
  int occ2 = -1, occ4 = -1, occw = -1, occ2s = -1, occ4s = -1;
  // two half-batches on two streams (bsg_ddpm_sample): rows [row_off, row_off + B_sub) of the bound batch
  int row_off = 0;                     // row offset the launch helpers add to the handle's buffers
  bool no_split = false;               // half-batch launches of a large batch use the regular one-workgroup-per-tile kernel
  bool split_small_lds = false;        // half-batch launches of a small batch: split kernels without the LDS padding, so that
                                       // workgroups of the two chains can share a CU
  hipStream_t st2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static int dev_alloc(float** p, size_t n) {
  BSG_HIP(hipMalloc((void**)p, n * sizeof(float)));
  return BSG_OK;
}
static void dev_free(float*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

extern "C" void bsg_diffnet_destroy(bsg_diffnet* h) {
  if (!h) return;
  guard_free(&h->guard);
  float** all[] = {&h->w_in, &h->b_in, &h->apack1, &h->apack2, &h->apackw, &h->apackw43, &h->w_cond, &h->b_cond, &h->b_out, &h->w_skip,
                   &h->b_skip, &h->w_fin, &h->b_fin, &h->dproj, &h->dconv, &h->ws_pack, &h->wo_pack, &h->wi_pack, &h->b_fin96, &h->apack2w, &h->zbuf, &h->condterm, &h->xa, &h->xb, &h->skip, &h->hid,
                   &h->eps, &h->eps_hist[0], &h->eps_hist[1], &h->eps_hist[2], &h->eps_hist[3], &h->xpred};
  for (float** p : all) dev_free(*p);
  for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
  if (h->flags) (void)hipFree(h->flags);
  if (h->epoch_dev) (void)hipFree(h->epoch_dev);
  if (h->hx) (void)hipFree(h->hx);
  if (h->split_flags) (void)hipFree(h->split_flags);
  if (h->st2) (void)hipStreamDestroy(h->st2);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->apack1h) (void)hipFree(h->apack1h);
  if (h->condterm_h) (void)hipFree(h->condterm_h);
  if (h->condterm_q) (void)hipFree(h->condterm_q);
  if (h->skip_h) (void)hipFree(h->skip_h);
  if (h->apack2h) (void)hipFree(h->apack2h);
  if (h->tail_h) (void)hipFree(h->tail_h);
  if (h->apack1s) (void)hipFree(h->apack1s);
  if (h->apack2s) (void)hipFree(h->apack2s);
  if (h->h2_scale) (void)hipFree(h->h2_scale);
  if (h->tail_s) (void)hipFree(h->tail_s);
  if (h->tail_scale) (void)hipFree(h->tail_scale);
  if (h->clk) (void)hipFree(h->clk);
  if (h->part_zx) (void)hipFree(h->part_zx);
  if (h->part_ix) (void)hipFree(h->part_ix);
  if (h->part_flags) (void)hipFree(h->part_flags);
  if (h->apack1q) (void)hipFree(h->apack1q);
  if (h->apack2q) (void)hipFree(h->apack2q);
  if (h->wcond_pack) (void)hipFree(h->wcond_pack);
  if (h->pack_bad) (void)hipFree(h->pack_bad);
  if (h->cond_planes) (void)hipFree(h->cond_planes);
  delete h;
}

static int copy_dev(float* dst, const void* src, size_t n, hipStream_t st) {
  BSG_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  return BSG_OK;
}

#define TRY(expr)              \
  do {                         \
    int _rc = (expr);          \
    if (_rc != BSG_OK) return _rc; \
  } while (0)

static int gemm_nt(const float* A, const float* W, float* Cc, const float* bias_n, int M, int N, int K, int lda, int ldc,
                   int act, hipStream_t st) {
  GemmArgs g{};
  g.A = A; g.B = W; g.C = Cc; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.ldc = ldc; g.trans_b = 1;
  g.taps = 1; g.bias_n = bias_n; g.alpha = 1.f; g.act = act; g.batch = 1;
  return launch_gemm(g, st);
}

// conv1x1 over [B][K][T] -> [B][M][T]
static int conv1x1(const float* W, const float* bias, const float* X, float* Y, int M, int K, int B, int T, int act,
                   hipStream_t st) {
  GemmArgs g{};
  g.A = W; g.B = X; g.C = Y; g.M = M; g.N = T; g.K = K; g.lda = K; g.ldb = T; g.ldc = T; g.trans_b = 0;
  g.sA = 0; g.sB = (long long)K * T; g.sC = (long long)M * T; g.taps = 1; g.bias_m = bias; g.alpha = 1.f; g.act = act;
  g.batch = B;
  return launch_gemm(g, st);
}

static int create_impl(bsg_diffnet* h, const void* const* w, const float* step_table, hipStream_t st) {
  const int M = h->M, L = h->L, S = h->cfg.max_steps;
  TRY(dev_alloc(&h->w_in, (size_t)C * M));
  TRY(dev_alloc(&h->b_in, C));
  TRY(dev_alloc(&h->apack1, (size_t)L * 2 * C * 3 * C));
  TRY(dev_alloc(&h->apack2, (size_t)L * 2 * C * C));
  TRY(dev_alloc(&h->apackw, (size_t)L * 4 * 2 * C * C));
  TRY(dev_alloc(&h->apackw43, (size_t)L * 6 * 2 * C * C));
  TRY(dev_alloc(&h->apack2w, (size_t)L * 2 * C * C));
  BSG_HIP(hipMalloc((void**)&h->apack1h, (size_t)L * 2 * C * 3 * C * sizeof(unsigned short)));
  BSG_HIP(hipMalloc((void**)&h->apack2h, (size_t)L * 2 * C * C * sizeof(unsigned short)));
  BSG_HIP(hipMalloc((void**)&h->apack1s, (size_t)L * 2 * 2 * C * 3 * C * sizeof(unsigned short)));
  BSG_HIP(hipMalloc((void**)&h->apack2s, (size_t)L * 2 * 2 * C * C * sizeof(unsigned short)));
  BSG_HIP(hipMalloc((void**)&h->apack1q, (size_t)L * 2 * 2 * C * 3 * C * sizeof(unsigned short)));
  BSG_HIP(hipMalloc((void**)&h->apack2q, (size_t)L * 2 * 2 * C * C * sizeof(unsigned short)));
  BSG_HIP(hipMalloc((void**)&h->h2_scale, (size_t)(4 * L + 2 * L) * sizeof(float)));   // table + [2L] scratch of the max reduction
  TRY(dev_alloc(&h->w_cond, (size_t)L * 2 * C * C));
  BSG_HIP(hipMalloc((void**)&h->wcond_pack, (size_t)L * 2 * 2 * C * C * sizeof(unsigned short)));
  BSG_HIP(hipMalloc((void**)&h->pack_bad, sizeof(unsigned)));
  BSG_HIP(hipMemsetAsync(h->pack_bad, 0, sizeof(unsigned), st));
  TRY(dev_alloc(&h->b_cond, (size_t)L * 2 * C));
  TRY(dev_alloc(&h->b_out, (size_t)L * 2 * C));
  TRY(dev_alloc(&h->w_skip, (size_t)C * C));
  TRY(dev_alloc(&h->b_skip, C));
  TRY(dev_alloc(&h->w_fin, (size_t)M * C));
  TRY(dev_alloc(&h->b_fin, M));
  TRY(dev_alloc(&h->dproj, (size_t)S * L * C));
  TRY(dev_alloc(&h->dconv, (size_t)S * L * 4 * 2 * C));
  TRY(copy_dev(h->w_in, w[0], (size_t)C * M, st));
  TRY(copy_dev(h->b_in, w[1], C, st));
  // step-embedding MLP over the whole table: D = W2 * mish(W1 * e + b1) + b2        (net.py:92-96,120)
  float *hid = nullptr, *dtab = nullptr;
  TRY(dev_alloc(&hid, (size_t)S * 4 * C));
  TRY(dev_alloc(&dtab, (size_t)S * C));
  int rc = gemm_nt(step_table, (const float*)w[2], hid, (const float*)w[3], S, 4 * C, C, C, 4 * C, ACT_MISH, st);
  if (rc == BSG_OK) rc = gemm_nt(hid, (const float*)w[4], dtab, (const float*)w[5], S, C, 4 * C, 4 * C, C, ACT_NONE, st);
  if (rc == BSG_OK) {   // split-fp16 form: per-layer power-of-two scales from max |w|, then the hi / lo fragments below
    std::vector<const float*> w1(L), w2(L);
    for (int l = 0; l < L; ++l) { w1[l] = (const float*)w[6 + 8 * l]; w2[l] = (const float*)w[6 + 8 * l + 6]; }
    rc = h2_scales(w1.data(), w2.data(), L, reinterpret_cast<unsigned*>(h->h2_scale + 4 * L), h->h2_scale, st);
  }
  for (int l = 0; l < L && rc == BSG_OK; ++l) {
    const void* const* lw = w + 6 + 8 * l;
    // dilated conv [2C][C][3] -> fragment order with k = tap*C + ci
    {
      const long long total = (long long)2 * C * 3 * C;
      hipLaunchKernelGGL(pack_a_frag_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)lw[0],
                         h->apack1 + (size_t)l * total, 2 * C, 3 * C, C, (long long)3 * C, 3LL, 1LL);
    }
    hipLaunchKernelGGL(pack_wino_kernel, dim3(cdiv(4LL * 2 * C * C, 256)), dim3(256), 0, st, (const float*)lw[0],
                       h->apackw + (size_t)l * 4 * 2 * C * C);
    {
      const long long total = (long long)2 * C * C;
      hipLaunchKernelGGL(pack_a_frag_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)lw[6],
                         h->apack2 + (size_t)l * total, 2 * C, C, C, (long long)C, 1LL, 0LL);
    }
    rc = pack_wino43((const float*)lw[0], h->apackw43 + (size_t)l * 6 * 2 * C * C, st);
    if (rc == BSG_OK) rc = pack_a_frag_bf16((const float*)lw[0], h->apack1h + (size_t)l * 2 * C * 3 * C, 2 * C, 3 * C, C, (long long)3 * C, 3LL, 1LL, st);
    if (rc == BSG_OK) rc = pack_a_frag_bf16((const float*)lw[6], h->apack2h + (size_t)l * 2 * C * C, 2 * C, C, C, (long long)C, 1LL, 0LL, st);
    if (rc == BSG_OK) rc = pack_a_frag_h2((const float*)lw[0], h->apack1s + (size_t)l * 2 * 2 * C * 3 * C, 2 * C, 3 * C, C, (long long)3 * C, 3LL, 1LL, h->h2_scale + 4 * l, 0, st);
    if (rc == BSG_OK) rc = pack_a_frag_h2((const float*)lw[6], h->apack2s + (size_t)l * 2 * 2 * C * C, 2 * C, C, C, (long long)C, 1LL, 0LL, h->h2_scale + 4 * l, 1, st);
    if (rc == BSG_OK) rc = pack_a_frag_q((const float*)lw[0], h->apack1q + (size_t)l * 2 * 2 * C * 3 * C, 2 * C, 3 * C, C, (long long)3 * C, 3LL, 1LL, h->h2_scale + 4 * l, 0, st);
    if (rc == BSG_OK) rc = pack_a_frag_q((const float*)lw[6], h->apack2q + (size_t)l * 2 * 2 * C * C, 2 * C, C, C, (long long)C, 1LL, 0LL, h->h2_scale + 4 * l, 1, st);
    if (rc != BSG_OK) break;
    hipLaunchKernelGGL(pack_a16_kernel, dim3(cdiv(2 * C * C, 256)), dim3(256), 0, st, (const float*)lw[6], h->apack2w + (size_t)l * 2 * C * C,
                       2 * C, C);
    hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, st, (const float*)lw[5], (const float*)lw[1],
                       h->b_cond + (size_t)l * 2 * C, 2 * C);
    if (hipGetLastError() != hipSuccess) { set_error("diffnet_create: pack kernels failed"); rc = BSG_EHIP; break; }
    rc = copy_dev(h->w_cond + (size_t)l * 2 * C * C, lw[4], (size_t)2 * C * C, st);
    if (rc == BSG_OK) rc = h2w_pack_into(h->wcond_pack + (size_t)l * 2 * 2 * C * C, (const float*)lw[4], 2 * C, C, 1, 0, C, 1, h->pack_bad, st);
    if (rc == BSG_OK) rc = copy_dev(h->b_out + (size_t)l * 2 * C, lw[7], 2 * C, st);
    // diffusion_projection of the tabulated step embedding -> dproj[s][l][:]            (net.py:67)
    if (rc == BSG_OK) rc = gemm_nt(dtab, (const float*)lw[2], h->dproj + (size_t)l * C, (const float*)lw[3], S, C, C, C, L * C, ACT_NONE, st);
    if (rc == BSG_OK) {
      const size_t dl = ((size_t)DCONV_SMAX * (C + 1) + DCONV_ROWS * 3 * C) * sizeof(float);
      (void)hipFuncSetAttribute((const void*)dconv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dl);
      hipLaunchKernelGGL(dconv_kernel, dim3(2 * C / DCONV_ROWS, cdiv(S, DCONV_SMAX)), dim3(DCONV_ROWS * DCONV_SMAX), dl, st, (const float*)lw[0], (const float*)h->dproj,
                         h->dconv, S, L, l);
      if (hipGetLastError() != hipSuccess) { set_error("diffnet_create: dconv kernel failed"); rc = BSG_EHIP; }
    }
  }
  const void* const* tw = w + 6 + 8 * L;
  if (rc == BSG_OK) rc = copy_dev(h->w_skip, tw[0], (size_t)C * C, st);
  if (rc == BSG_OK) rc = copy_dev(h->b_skip, tw[1], C, st);
  if (rc == BSG_OK) rc = copy_dev(h->w_fin, tw[2], (size_t)M * C, st);
  if (rc == BSG_OK) rc = copy_dev(h->b_fin, tw[3], M, st);
  float *wo_pad = nullptr, *wi_pad = nullptr, *wi_pad96 = nullptr;
  if (rc == BSG_OK && M <= 96) {
    const int MP = (M + 7) / 8 * 8;
    h->MP = MP;
    rc = dev_alloc(&h->ws_pack, (size_t)C * C);
    if (rc == BSG_OK) rc = dev_alloc(&h->wo_pack, (size_t)96 * C);
    if (rc == BSG_OK) rc = dev_alloc(&h->wi_pack, (size_t)C * MP);
    if (rc == BSG_OK) rc = dev_alloc(&h->b_fin96, 96);
    if (rc == BSG_OK) rc = dev_alloc(&wo_pad, (size_t)96 * C);
    if (rc == BSG_OK) rc = dev_alloc(&wi_pad, (size_t)C * MP);
    if (rc == BSG_OK) {
      hipLaunchKernelGGL(pack_a_frag_kernel, dim3(cdiv((long long)C * C, 256)), dim3(256), 0, st, (const float*)tw[0], h->ws_pack, C, C, C, (long long)C, 1LL, 0LL);
      hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv(96 * C, 256)), dim3(256), 0, st, (const float*)tw[2], wo_pad, M, 96, C, C);
      hipLaunchKernelGGL(pack_a_frag_kernel, dim3(cdiv((long long)96 * C, 256)), dim3(256), 0, st, (const float*)wo_pad, h->wo_pack, 96, C, C, (long long)C, 1LL, 0LL);
      hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv(C * MP, 256)), dim3(256), 0, st, (const float*)w[0], wi_pad, C, C, M, MP);
      hipLaunchKernelGGL(pack_a_frag_kernel, dim3(cdiv((long long)C * MP, 256)), dim3(256), 0, st, (const float*)wi_pad, h->wi_pack, C, MP, MP, (long long)MP, 1LL, 0LL);
      hipLaunchKernelGGL(pad_rows_kernel, dim3(1), dim3(256), 0, st, (const float*)tw[3], h->b_fin96, M, 96, 1, 1);
      if (hipGetLastError() != hipSuccess) { set_error("diffnet_create: tail pack kernels failed"); rc = BSG_EHIP; }
    }
    // the same three projections as bf16 fragments (bf16-operand configuration); the input projection's K is padded to 96
    if (rc == BSG_OK) rc = dev_alloc(&wi_pad96, (size_t)C * 96);
    if (rc == BSG_OK && hipMalloc((void**)&h->tail_h, (size_t)(C * C + 96 * C + C * 96) * sizeof(unsigned short)) != hipSuccess) {
      set_error("diffnet_create: out of device memory");
      rc = BSG_EHIP;
    }
    if (rc == BSG_OK) {
      hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv(C * 96, 256)), dim3(256), 0, st, (const float*)w[0], wi_pad96, C, C, M, 96);
      rc = pack_a_frag_bf16((const float*)tw[0], h->tail_h, C, C, C, (long long)C, 1LL, 0LL, st);
      if (rc == BSG_OK) rc = pack_a_frag_bf16(wo_pad, h->tail_h + C * C, 96, C, C, (long long)C, 1LL, 0LL, st);
      if (rc == BSG_OK) rc = pack_a_frag_bf16(wi_pad96, h->tail_h + C * C + 96 * C, C, 96, 96, 96LL, 1LL, 0LL, st);
    }
    // and as split-fp16 fragments (hi / lo planes) for the tail fused into the split-fp16 stack launch
    if (rc == BSG_OK && (hipMalloc((void**)&h->tail_s, (size_t)2 * (C * C + 96 * C + C * 96) * sizeof(unsigned short)) != hipSuccess ||
                         hipMalloc((void**)&h->tail_scale, 9 * sizeof(float)) != hipSuccess)) {
      set_error("diffnet_create: out of device memory");
      rc = BSG_EHIP;
    }
    if (rc == BSG_OK)
      rc = h2_tail_pack((const float*)tw[0], wo_pad, wi_pad96, h->tail_s, h->tail_s + 2 * C * C, h->tail_s + 2 * C * C + 2 * 96 * C,
                        reinterpret_cast<unsigned*>(h->tail_scale + 6), h->tail_scale, st);
  }
  unsigned pack_bad = 1;
  if (rc == BSG_OK && hipMemcpyAsync(&pack_bad, h->pack_bad, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) pack_bad = 1;
  hipError_t e = hipStreamSynchronize(st);
  h->cond_h2w_ok = rc == BSG_OK && e == hipSuccess && pack_bad == 0;
  dev_free(hid);
  dev_free(dtab);
  dev_free(wo_pad);
  dev_free(wi_pad);
  dev_free(wi_pad96);
  if (rc != BSG_OK) return rc;
  BSG_HIP(e);
  return BSG_OK;
}

extern "C" int bsg_diffnet_create(bsg_diffnet** out, const bsg_diffnet_cfg* cfg, const void* const* dev_weights,
                                  int32_t n_weights, const float* step_table, void* stream) {
  BSG_REQUIRE(out && cfg && dev_weights && step_table, "diffnet_create: null argument");
  BSG_REQUIRE(cfg->residual_channels == C && cfg->encoder_hidden == C,
              "diffnet_create: residual_channels=%d hidden=%d; kernels are built for 256/256", cfg->residual_channels,
              cfg->encoder_hidden);
  BSG_REQUIRE(cfg->in_dims > 0 && cfg->in_dims % 4 == 0 && cfg->in_dims <= 128, "diffnet_create: in_dims=%d unsupported", cfg->in_dims);
  BSG_REQUIRE(cfg->residual_layers > 0 && cfg->residual_layers <= 64, "diffnet_create: residual_layers=%d", cfg->residual_layers);
  BSG_REQUIRE(cfg->dilation_cycle_length >= 1 && cfg->dilation_cycle_length <= 4,
              "diffnet_create: dilation_cycle_length=%d (max dilation 8 supported)", cfg->dilation_cycle_length);
  BSG_REQUIRE(cfg->max_steps > 0, "diffnet_create: max_steps=%d", cfg->max_steps);
  BSG_REQUIRE(n_weights == 10 + 8 * cfg->residual_layers, "diffnet_create: expected %d weight tensors, got %d",
              10 + 8 * cfg->residual_layers, n_weights);
  for (int i = 0; i < n_weights; ++i) BSG_REQUIRE(dev_weights[i] != nullptr, "diffnet_create: weight %d is null", i);
  bsg_diffnet* h = new bsg_diffnet();
  h->cfg = *cfg;
  h->M = cfg->in_dims;
  h->L = cfg->residual_layers;
  int rc = guard_init(&h->guard, (hipStream_t)stream);
  if (rc == BSG_OK) rc = create_impl(h, dev_weights, step_table, (hipStream_t)stream);
  if (rc != BSG_OK) {
    bsg_diffnet_destroy(h);
    return rc;
  }
  *out = h;
  return BSG_OK;
}

// condterm (row layout [L][B][2C][T]) on demand: only the fallback launches read it (bsg_diffnet.cond_rows_valid)
static int alloc_cond_rows(bsg_diffnet* h, size_t bt, hipStream_t st) {
  if (bt <= h->cap_cond_rows) return BSG_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st) (void)hipStreamIsCapturing(st, &cap);
  BSG_REQUIRE(cap == hipStreamCaptureStatusNone, "diffnet: the row layout of the conditioner term is allocated by the first eager call of a fallback launch; run it once before capturing");
  BSG_HIP(hipStreamSynchronize(st));
  dev_free(h->condterm);
  h->cap_cond_rows = 0;
  TRY(dev_alloc(&h->condterm, (size_t)h->L * 2 * C * bt));
  h->cap_cond_rows = bt;
  return BSG_OK;
}

// [z][R/4][T][4] -> [z][R][T]: lanes along T, one 16-byte load and four dword stores (256 B contiguous per row and wave)
__global__ __launch_bounds__(256) void quad_to_rows_kernel(const float* __restrict__ q, float* __restrict__ rows, int R, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const long long z = blockIdx.z, qr = blockIdx.y;
  const f32x4 v = *reinterpret_cast<const f32x4*>(q + ((z * (R / 4) + qr) * T + t) * 4);
  float* o = rows + (z * R + 4 * qr) * T + t;
  o[0] = v[0]; o[(long long)T] = v[1]; o[2LL * T] = v[2]; o[3LL * T] = v[3];
}

// the rows of the bound condition's term, derived from its quads the first time a launch that reads rows runs behind a prepare that wrote
// only the quads (a fallback after a range event or a give-up, BSG_H2_Q=0 shapes, the per-layer hooks): 2 x 655 MB at B = 16, once
static int ensure_cond_rows(bsg_diffnet* h, hipStream_t st) {
  if (h->cond_rows_valid) return BSG_OK;
  if (!h->cond_q_valid) {
    set_error("diffnet: the conditioner term of the bound condition exists in the bf16 layout only (bound under BSG_COMPUTE_BF16); call bsg_diffnet_prepare again");
    return BSG_ESTATE;
  }
  const size_t bt = (size_t)h->B * h->T;
  TRY(alloc_cond_rows(h, bt, st));
  const long long zs = (long long)h->L * h->B;
  BSG_REQUIRE(zs <= 65535, "diffnet: L * B = %lld launch slices", zs);
  hipLaunchKernelGGL(quad_to_rows_kernel, dim3(cdiv(h->T, 256), 2 * C / 4, (unsigned)zs), dim3(256), 0, st, (const float*)h->condterm_q, h->condterm, 2 * C, h->T);
  BSG_LAUNCH_CHECK();
  h->cond_rows_valid = true;
  return BSG_OK;
}

extern "C" int bsg_diffnet_prepare(bsg_diffnet* h, const float* cond, int32_t B, int32_t T, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  BSG_REQUIRE(h && cond, "diffnet_prepare: null argument");
  BSG_REQUIRE(B > 0 && T > 0, "diffnet_prepare: B=%d T=%d", B, T);
  BSG_REQUIRE((long long)B * T < (1LL << 31) / (2 * C), "diffnet_prepare: B*T=%lld too large", (long long)B * T);
  hipStream_t st = (hipStream_t)stream;
  const size_t bt = (size_t)B * T;
  if (bt > h->cap_bt) {
    BSG_HIP(hipStreamSynchronize(st));
    float** bufs[] = {&h->xa, &h->xb, &h->skip, &h->hid, &h->eps, &h->eps_hist[0], &h->eps_hist[1],
                      &h->eps_hist[2], &h->eps_hist[3], &h->xpred};
    for (float** p : bufs) dev_free(*p);
    h->cap_bt = 0;
    TRY(dev_alloc(&h->xa, C * bt));
    TRY(dev_alloc(&h->xb, C * bt));
    TRY(dev_alloc(&h->skip, C * bt));
    TRY(dev_alloc(&h->hid, C * bt));
    TRY(dev_alloc(&h->eps, (size_t)h->M * bt));
    // PLMS history ring + x_pred scratch (5 x [B][M][T]: 2 % of the conditioner term) — here, so that no sampler call allocates
    for (int i = 0; i < 4; ++i) TRY(dev_alloc(&h->eps_hist[i], (size_t)h->M * bt));
    TRY(dev_alloc(&h->xpred, (size_t)h->M * bt));
    h->cap_bt = bt;
  }
  h->B = B;
  h->T = T;
  {
    if (!h->num_cus) {
      int dev = 0;
      BSG_HIP(hipGetDevice(&dev));
      BSG_HIP(hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    // the stack launch handles at most 2 workgroups per CU at a time (larger batches run as groups of rows, one after the other)
    const size_t all_tiles = (size_t)B * cdiv(T, 32);
    const size_t need = all_tiles < 2 * (size_t)h->num_cus ? all_tiles : 2 * (size_t)h->num_cus;
    if (need > h->flags_cap) {
      BSG_HIP(hipStreamSynchronize(st));
      if (h->flags) (void)hipFree(h->flags);
      if (h->hx) (void)hipFree(h->hx);
      h->flags = nullptr;
      h->hx = nullptr;
      h->flags_cap = 0;
      BSG_HIP(hipMalloc((void**)&h->flags, (need + 4) * sizeof(unsigned)));
      BSG_HIP(hipMemsetAsync(h->flags, 0, (need + 4) * sizeof(unsigned), st));
      if (!h->epoch_dev) {   // epochs start at 1: a zeroed flag word is older than every launch (flag values = epoch * 64 + layers published)
        BSG_HIP(hipMalloc((void**)&h->epoch_dev, 2 * sizeof(unsigned)));
        const unsigned init[2] = {1u, 0u};
        BSG_HIP(hipMemcpy(h->epoch_dev, init, sizeof(init), hipMemcpyHostToDevice));
      }
      BSG_HIP(hipMalloc((void**)&h->hx, 2 * need * 2 * C * 8 * sizeof(float)));
      h->flags_cap = need;
      if (!h->clk) {
        BSG_HIP(hipMalloc((void**)&h->clk, 4 * sizeof(unsigned long long)));
        BSG_HIP(hipMemsetAsync(h->clk, 0, 4 * sizeof(unsigned long long), st));
      }
    }
    const size_t tiles = (size_t)B * cdiv(T, 32);
    if (tiles > h->split_cap) {
      BSG_HIP(hipStreamSynchronize(st));
      dev_free(h->zbuf);
      if (h->split_flags) (void)hipFree(h->split_flags);
      h->split_flags = nullptr;
      h->split_cap = 0;
      TRY(dev_alloc(&h->zbuf, tiles * C * 32));
      BSG_HIP(hipMalloc((void**)&h->split_flags, (16 * tiles + 4) * sizeof(unsigned)));
      BSG_HIP(hipMemsetAsync(h->split_flags, 0, (16 * tiles + 4) * sizeof(unsigned), st));
      h->split_cap = tiles;
    }
  }
  if (h->compute == BSG_COMPUTE_BF16 && bt > h->cap_bt_h) {
    BSG_HIP(hipStreamSynchronize(st));
    if (h->condterm_h) (void)hipFree(h->condterm_h);
    if (h->skip_h) (void)hipFree(h->skip_h);
    h->condterm_h = h->skip_h = nullptr;
    h->cap_bt_h = 0;
    BSG_HIP(hipMalloc((void**)&h->condterm_h, (size_t)h->L * 2 * C * bt * sizeof(unsigned short)));
    BSG_HIP(hipMalloc((void**)&h->skip_h, (size_t)C * bt * sizeof(unsigned short)));
    h->cap_bt_h = bt;
  }
  h->cond_q_valid = false;
  h->cond_rows_valid = false;
  static int env_h2w = -1;   // BSG_GEMM_H2W=0: gemm_split_kernel (operands split while staged) instead of the pre-split GEMM
  if (env_h2w < 0) { const char* e = getenv("BSG_GEMM_H2W"); env_h2w = e ? atoi(e) : 1; }
  bool bf16_direct = false;
  const bool h2w = env_h2w && h->cond_h2w_ok && gemm_split_enabled() && h2w_supports(T, 2 * C, C, 1, C) && (long long)h->L * B <= 65535;
  if (h2w) {
    // All L projections of all B rows as ONE launch on the 16-bit matrix pipe with pre-split operands (gemm_h2w.hip): cond is transposed and
    // split once into hi / lo fp16 planes [B][T][H]; the weights were split at create; batch index z = l B + b writes condterm[l][b] =
    // W_l cond_b + (b_cond + b_dil) as [2C][T] rows — lanes run along T
    if (bt > h->cap_cond_planes) {
      BSG_HIP(hipStreamSynchronize(st));
      if (h->cond_planes) (void)hipFree(h->cond_planes);
      h->cond_planes = nullptr;
      h->cap_cond_planes = 0;
      BSG_HIP(hipMalloc((void**)&h->cond_planes, 2 * bt * C * sizeof(unsigned short)));
      h->cap_cond_planes = bt;
    }
    TRY(h2w_split_transposed(cond, h->cond_planes, h->cond_planes + bt * C, B, C, T, st));
    // the 16-row stack launch (diffnet_h2q.hip) loads the term as channel quads: the same epilogue stores it once more in that order
    // (BSG_H2_Q=0 / BSG_COND_QUAD=0: not)
    static int env_cq = -1;
    if (env_cq < 0) { const char* e = getenv("BSG_COND_QUAD"); const char* q = getenv("BSG_H2_Q"); env_cq = (e ? atoi(e) : 1) && (q ? atoi(q) : 1); }
    bool want_q = env_cq && h->compute == BSG_COMPUTE_F32 && h->apack1q && !h->h2_off && !h->split_off;
    if (want_q && bt > h->cap_cond_q) {
      BSG_HIP(hipStreamSynchronize(st));
      if (h->condterm_q) (void)hipFree(h->condterm_q);
      h->condterm_q = nullptr;
      h->cap_cond_q = 0;
      if (hipMalloc((void**)&h->condterm_q, (size_t)h->L * 2 * C * bt * sizeof(float)) != hipSuccess) {
        // out of memory for the quads: the launches read the rows instead (dword loads: a few per cent slower), nothing fails
        (void)hipGetLastError();
        h->condterm_q = nullptr;
        want_q = false;
      } else {
        h->cap_cond_q = bt;
      }
    }
    h->cond_q_valid = want_q;
    H2wArgs g{};
    g.Cq = want_q ? h->condterm_q : nullptr;
    // bf16-operand configuration: the epilogue rounds the term to bf16 quads itself and writes NO fp32 copy (until round 5 the 2.6 GB of fp32 at
    // B = 64 went to HBM and 20 conversion launches read them back; BSG_COND_BF16_DIRECT=0: that form)
    static int env_hd = -1;
    if (env_hd < 0) { const char* e = getenv("BSG_COND_BF16_DIRECT"); env_hd = e ? atoi(e) : 1; }
    bf16_direct = env_hd && h->compute == BSG_COMPUTE_BF16 && h->condterm_h;
    g.Ch = bf16_direct ? h->condterm_h : nullptr;
    // ONE layout per pass: the rows only when neither the quads nor the bf16 quads are written (BSG_COND_ROWS=1: the rows as well, round 5's way)
    static int env_rows = -1;
    if (env_rows < 0) { const char* e = getenv("BSG_COND_ROWS"); env_rows = e ? atoi(e) : 0; }
    const bool rows_now = env_rows || !(want_q || bf16_direct);
    if (rows_now) TRY(alloc_cond_rows(h, bt, st));
    h->cond_rows_valid = rows_now;
    g.act = h->cond_planes; g.act_plane = (long long)bt * C; g.lda = C; g.sAct = (long long)T * C; g.wpack = h->wcond_pack;
    g.sW = (long long)2 * 2 * C * C; g.zdiv = B; g.rows = T; g.K = C; g.Wn = 2 * C; g.taps = 1; g.act_is_a = 0; g.C = rows_now ? h->condterm : nullptr; g.ldc = T;
    g.sC = (long long)2 * C * T; g.bias = h->b_cond; g.sBias = 2 * C; g.alpha = 1.f; g.act_fn = ACT_NONE; g.batch = h->L * B;
    TRY(launch_gemm_h2w(g, st));
  } else if (B == 1) {
    TRY(alloc_cond_rows(h, bt, st));
    h->cond_rows_valid = true;
    // a single utterance: the L projections are ONE GEMM of L x 2C rows ([L][2C][C] weights and [L][1][2C][T] outputs are contiguous) — twenty
    // launches of 64 workgroups each left most of the chip idle (1.0 -> 0.15 ms of a 25-ms pass)
    TRY(conv1x1(h->w_cond, h->b_cond, cond, h->condterm, h->L * 2 * C, C, 1, T, ACT_NONE, st));
  }
  if (B != 1 && !h2w) {
    TRY(alloc_cond_rows(h, bt, st));
    h->cond_rows_valid = true;
  }
  for (int l = 0; l < h->L; ++l) {
    if (B != 1 && !h2w)
      TRY(conv1x1(h->w_cond + (size_t)l * 2 * C * C, h->b_cond + (size_t)l * 2 * C, cond,
                  h->condterm + (size_t)l * 2 * C * bt, 2 * C, C, B, T, ACT_NONE, st));
    if (h->compute == BSG_COMPUTE_BF16 && !bf16_direct)
      TRY(f32_to_quad_bf16(h->condterm + (size_t)l * 2 * C * bt, h->condterm_h + (size_t)l * 2 * C * bt, B, 2 * C, T, st));
  }
  h->prepared_compute = h->compute;
  return BSG_OK;
}

// GEMM1 of the residual block: BSG_WINO unset = 2: Winograd F(2,3) kernels, and the F(4,3) stack launch (diffnet_f43.hip) for launches
// that fill the chip with 64-frame tiles (stack_rows; BSG_STACK43=2: for any shape); 1: F(2,3) only; 0: the direct K=768 form
static int wino_env() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("BSG_WINO"); v = e ? atoi(e) : 2; }
  return v;
}
static bool use_wino() { return wino_env() != 0; }

// A pair of workgroups per tile pays (a z exchange through L2) only when single workgroups would leave CUs idle: measured on
// MI355X at T=1000, B = 1 / 2 / 4 (32 / 64 / 128 tiles): 63 -> 41, 64 -> 42, 65 -> 47 us per layer; B = 6 (192 tiles): 68 -> 70.
// So: at most one workgroup per CU after the split, which also keeps every workgroup of the launch resident (the partner
// of a polling workgroup is always running).  Between 129 and 256 tiles the 16-wave form (no inter-workgroup traffic) takes over.
// BSG_SPLIT=0 disables both.
// The pair form runs with at most one workgroup per CU by construction of its threshold, but the dispatcher may still place two
// of them on one CU while others stay empty; asking for more than half of the CU's 160 KB of LDS (only 48 KB are used) makes
// that impossible.
static constexpr size_t kSplitLds = 84 * 1024;

static int split_env() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("BSG_SPLIT"); env = e ? atoi(e) : 1; }
  return env;
}

static void query_split_occupancy(bsg_diffnet* h) {
  const int lds = C * 48 * (int)sizeof(float);
  auto occ_of = [](const void* fn, int threads, size_t bytes) {
    int o = 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplitLds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, threads, bytes) != hipSuccess)
      return 0;
    return o;
  };
  h->occ2 = occ_of((const void*)residual_split_kernel<false, 2>, 512, kSplitLds);
  h->occ4 = occ_of((const void*)residual_split_kernel<false, 4>, 256, kSplitLds);
  h->occ2s = occ_of((const void*)residual_split_kernel<false, 2>, 512, (size_t)lds);
  h->occ4s = occ_of((const void*)residual_split_kernel<false, 4>, 256, (size_t)lds);
  h->occw = occ_of((const void*)residual_split_kernel<true, 1>, 1024, (size_t)lds);
}

static int use_split(bsg_diffnet* h, int B, int T) {   // 0: regular launch; 1 / 4: 2 / 4 workgroups per tile; 2: one 16-wave workgroup per tile
  const int env = split_env();
  if (!env || h->split_off || !h->num_cus || !h->zbuf) return 0;
  const long long tiles = (long long)B * cdiv(T, 32);
  if ((size_t)tiles > h->split_cap) return 0;
  if (h->occ2 < 0) query_split_occupancy(h);
  // residency is what makes a hand-off terminate: every workgroup of the launch (and, for two chains sharing CUs, of both
  // launches: dual_fork checks that sum) must fit the device at the LDS size actually launched
  const int o4 = h->split_small_lds ? h->occ4s : h->occ4, o2 = h->split_small_lds ? h->occ2s : h->occ2;
  if (4 * tiles <= h->num_cus && env != 2 && o4 >= 1) return 4;   // BSG_SPLIT=2: no 4-way split (A/B measurements)
  if (2 * tiles <= h->num_cus) return o2 >= 1 ? 1 : 0;
  if (tiles <= h->num_cus && env != 3) return h->occw >= 1 ? 2 : 0;   // BSG_SPLIT=3: pair form only (A/B measurements)
  return 0;
}

static int launch_layer(bsg_diffnet* h, int layer, const float* x_in, const long long* t_dev, int t_uniform, float* x_out,
                        float* skip, int B, int T, hipStream_t st, unsigned long long* stamps = nullptr) {
  ResArgs a{};
  a.x_in = x_in; a.x_out = x_out; a.skip = skip;
  a.condterm = h->condterm + ((size_t)layer * h->B + h->row_off) * 2 * C * (size_t)T;   // [L][B bound][2C][T], this launch's rows
  a.dproj = h->dproj; a.t_dev = t_dev; a.t_uniform = t_uniform;
  a.apack1 = h->apack1 + (size_t)layer * 2 * C * 3 * C;
  a.apack2 = h->apack2 + (size_t)layer * 2 * C * C;
  a.apackw = h->apackw + (size_t)layer * 4 * 2 * C * C;
  a.apackw43 = h->apackw43 + (size_t)layer * 6 * 2 * C * C;
  a.apack1h = h->apack1h + (size_t)layer * 2 * C * 3 * C;
  a.apack2h = h->apack2h + (size_t)layer * 2 * C * C;
  a.bias_out = h->b_out + (size_t)layer * 2 * C;
  a.B = B; a.T = T; a.L = h->L; a.layer = layer;
  a.dil = 1 << (layer % h->cfg.dilation_cycle_length);
  a.first = layer == 0;
  a.skip_div = layer == h->L - 1 ? sqrtf((float)h->L) : 1.0f;
  a.stamps = stamps;
  if (h->compute == BSG_COMPUTE_F32 && !stamps && use_wino() && !h->no_split && use_split(h, B, T)) {
    // small launch: a pair of workgroups per tile, each half of the channels (residual_split_kernel)
    SplitArgs s{};
    a.tiles_per_row = cdiv(T, 32);
    s.base = a;
    s.apack2w = h->apack2w + (size_t)layer * 2 * C * C;
    // exchange tiles and flags of this launch's rows (two half-batch chains may be in flight at once)
    const size_t tile0 = (size_t)h->row_off * a.tiles_per_row;
    s.zbuf = h->zbuf + tile0 * C * 32;
    s.flags = h->split_flags + tile0 * 16;
    s.status = h->split_flags + 16 * h->split_cap;
    if (++h->split_epoch == 0) h->split_epoch = 1;
    s.epoch = h->split_epoch;
    const int mode = use_split(h, B, T);
    if (h->inject_giveup > 0 && mode != 2) { s.inject = 1; --h->inject_giveup; }
    const size_t slds = h->split_small_lds ? (size_t)C * 48 * sizeof(float) : kSplitLds;
    if (mode == 2) hipLaunchKernelGGL((residual_split_kernel<true, 1>), dim3(B * a.tiles_per_row), dim3(1024), (size_t)C * 48 * sizeof(float), st, s);
    else if (mode == 4) hipLaunchKernelGGL((residual_split_kernel<false, 4>), dim3(4 * B * a.tiles_per_row), dim3(256), slds, st, s);
    else hipLaunchKernelGGL((residual_split_kernel<false, 2>), dim3(2 * B * a.tiles_per_row), dim3(512), slds, st, s);
    BSG_LAUNCH_CHECK();
    h->last_path = mode == 2 ? "wide" : mode == 4 ? "split4" : "split2";
    return BSG_OK;
  }
  if (h->compute == BSG_COMPUTE_BF16) {
    // the running skip sum lives in h->skip_h (bf16); a caller-supplied fp32 buffer (the unit-test hook) is converted
    // in and out around the launch
    a.condterm_h = h->condterm_h + ((size_t)layer * h->B + h->row_off) * 2 * C * (size_t)T;
    a.skip_h = h->skip_h + (size_t)h->row_off * C * T;
    const bool ext = skip != h->skip + (size_t)h->row_off * C * T;
    if (ext && !a.first) TRY(f32_to_quad_bf16(skip, h->skip_h, B, C, T, st));
    TRY(launch_residual_layer_bf16(a, st));
    h->last_path = "bf16";
    if (ext) TRY(quad_bf16_to_f32(h->skip_h, skip, B, C, T, st));
    return BSG_OK;
  }
  // 32-frame tiles: 48 KB of LDS -> 3 workgroups per CU.  (Wider tiles of 64 / 128 frames were built and
  // measured in round 1: 0-50 % slower at every batch size, because fewer workgroups per CU hide less of the
  // L2 latency of the weight stream.)
  a.tiles_per_row = cdiv(T, 32);
  const dim3 grid(8 * cdiv(B * a.tiles_per_row, 8)), block(512);   // see the tile order in residual_layer_kernel
  const size_t lds = (size_t)C * (32 + 2 * HALO) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    BSG_HIP(hipFuncSetAttribute((const void*)residual_layer_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BSG_HIP(hipFuncSetAttribute((const void*)residual_layer_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BSG_HIP(hipFuncSetAttribute((const void*)residual_layer_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BSG_HIP(hipFuncSetAttribute((const void*)residual_layer_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (use_wino()) {
    if (stamps) hipLaunchKernelGGL((residual_layer_kernel<true, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((residual_layer_kernel<false, true>), grid, block, lds, st, a);
  } else {
    if (stamps) hipLaunchKernelGGL((residual_layer_kernel<true, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((residual_layer_kernel<false, false>), grid, block, lds, st, a);
  }
  BSG_LAUNCH_CHECK();
  h->last_path = "layer";
  return BSG_OK;
}

// rows per launch group (0: the stack launch is not used for this shape).  A tile row is ceil(T/32) workgroups that wait for
// each other, and every workgroup of a launch must be resident: at most occ x CUs workgroups, whole rows only.  It pays when a
// launch has more workgroups than CUs (two per CU overlap each other's waits); smaller launches keep the channel-split kernels.
static int stack_rows(bsg_diffnet* h, int B, int T, hipStream_t st) {
  h->stack_is_f43 = false;
  h->stack_is_h2 = false;
  {
    // split-fp16 form (diffnet_h2.hip): fp32 operands as hi + lo fp16 terms on the 16-bit matrix pipe; 64-frame tiles, one workgroup per
    // CU, whole rows per launch group, every batch size.  BSG_H2=0 / bsg_diffnet_set_h2(h, 0): off (the kernels of the fp32 matrix pipe)
    static int envh2 = -1;
    if (envh2 < 0) { const char* e = getenv("BSG_H2"); envh2 = e ? atoi(e) : 1; }
    // (round 4: the launch epoch of the flags lives in device memory, so these launches can be captured and replayed; only the part forms'
    // exchange buffers must exist already — their first eager call allocates them)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st) (void)hipStreamIsCapturing(st, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (envh2 && !h->h2_off && h->compute == BSG_COMPUTE_F32 && !h->split_off && h->num_cus && h->hx && h->apack1s && h->epoch_dev) {
      // tile width: 32 frames (one column tile per workgroup) while those tiles fit ONE launch group, else 64 frames (two column tiles:
      // half the weight stream per frame).  ms per 100-step pass at T = 1000 on one box (tools/bench_small.py), 32-frame / 64-frame /
      // fp32-pipe kernels (BSG_H2=0):  B=1 55 / 70 / 60,  B=2 53 / 70 / 64,  B=4 52 / 70 / 84,  B=6 56 / 74 / 119,  B=8 69 / 76 / 132,
      // B=10 115 / 80 / 187,  B=12 113 / 87 / 198,  B=16 134 / 103 / 200 — so the launch is taken for every batch size (BSG_H2_NCT=1 / 2
      // forces a width)
      static int env_nct = -1;
      if (env_nct < 0) { const char* e = getenv("BSG_H2_NCT"); env_nct = e ? atoi(e) : 0; }
      int nct = (long long)B * cdiv(T, 32) > h->num_cus ? 2 : 1;
      if (env_nct == 1 || env_nct == 2) nct = env_nct;
      h->stack_parts = 0;
      {
        // part forms (residual_part_h2_kernel): several workgroups on as many CUs of an XCD share a tile, each a part of the channels and of
        // the weight stream; the whole batch in one launch.  Quads of 32-frame tiles while B * ceil(T / 32) <= CUs / 4 (one or two utterances
        // at T = 1000), quads of 64-frame tiles while B * ceil(T / 64) <= CUs / 4 (B <= 4), pairs of 64-frame tiles (8 waves each) while
        // B * ceil(T / 64) <= CUs / 2 (B <= 8).  ms per 100-step pass at T = 1000, one workgroup per tile / part form: B=1 54.5 / 24.1,
        // B=2 52.6 / 25.4, B=3 52.0 / 32.0, B=4 51.5 / 35.0, B=5 54.8 / 46.0, B=6 57.1 / 49.6, B=8 69.1 / 59.4.  BSG_H2_PART=0: none
        // (BSG_H2_QUAD=0 / BSG_H2_QUAD64=0 / BSG_H2_PAIR64=0: not that form)
        static int env_part = -1, env_quad = -1, env_quad64 = -1;
        if (env_part < 0) { const char* e = getenv("BSG_H2_PART"); env_part = e ? atoi(e) : 1; }
        if (env_quad < 0) { const char* e = getenv("BSG_H2_QUAD"); env_quad = e ? atoi(e) : 1; }
        if (env_quad64 < 0) { const char* e = getenv("BSG_H2_QUAD64"); env_quad64 = e ? atoi(e) : 1; }
        auto take_quad = [&](int pn) {
          if (h->occ_part[pn] < 0) h->occ_part[pn] = part_h2_occupancy(4, pn) >= 1 ? 1 : 0;
          if (h->occ_part[pn] < 1) return false;
          h->stack_is_h2 = true;
          h->stack_nct = pn;
          h->stack_parts = 4;
          return true;
        };
        const long long t32 = (long long)B * cdiv(T, 32), t64 = (long long)B * cdiv(T, 64);
        const bool part_bufs = !capturing || (size_t)(2 * t64) <= h->part_cap;   // (sized in 32-frame tile equivalents; 2 t64 >= t32)
        if (env_part && !h->parts_off && env_nct == 0 && h->apack1q && part_bufs) {
          if (env_quad && 4 * 8 * cdiv(t32, 8) <= h->num_cus && take_quad(1)) return B;
          if (env_quad64 && 4 * 8 * cdiv(t64, 8) <= h->num_cus && take_quad(2)) return B;
          static int env_pair64 = -1;
          if (env_pair64 < 0) { const char* e = getenv("BSG_H2_PAIR64"); env_pair64 = e ? atoi(e) : 1; }
          if (env_pair64 && 2 * 8 * cdiv(t64, 8) <= h->num_cus) {
            if (h->occ_part[0] < 0) h->occ_part[0] = part_h2_occupancy(2, 2) >= 1 ? 1 : 0;
            if (h->occ_part[0] >= 1) {
              h->stack_is_h2 = true;
              h->stack_nct = 2;
              h->stack_parts = 2;
              return B;
            }
          }
        }
      }
      // 16-row matrix tiles (residual_stack_q_kernel, diffnet_h2q.hip; round 5): the same launch with every product a v_mfma_f32_16x16x32_f16 —
      // the same matrix cycles, but the chip holds a higher clock under that shape.  BSG_H2_Q=0: the 32-row form (residual_stack_h2_kernel)
      static int env_q = -1;
      if (env_q < 0) { const char* e = getenv("BSG_H2_Q"); env_q = e ? atoi(e) : 1; }
      h->stack_q = false;
      if (env_q && !h->q_off && h->apack1q && h->apack2q) {
        if (h->occ_stack_h2q[nct] < 0) h->occ_stack_h2q[nct] = stack_h2q_occupancy(nct) >= 1 ? 1 : 0;
        h->stack_q = h->occ_stack_h2q[nct] >= 1;
      }
      if (!h->stack_q && h->occ_stack_h2[nct] < 0) h->occ_stack_h2[nct] = stack_h2_occupancy(nct) >= 1 ? 1 : 0;
      const int tpr = cdiv(T, 32 * nct);
      if ((h->stack_q || h->occ_stack_h2[nct] >= 1) && tpr <= h->num_cus) {
        int rows = h->num_cus / tpr;
        if (rows > B) rows = B;
        h->stack_is_h2 = true;
        h->stack_nct = nct;
        return rows;
      }
    }
  }
  if (wino_env() == 2) {
    // F(4,3) form (diffnet_f43.hip): 64-frame tiles, one workgroup per CU, whole rows per launch group; BSG_STACK43=0 keeps per-layer
    // launches.  A launch group takes the same time whatever part of the chip it fills, so the form is taken when the groups are
    // >= 90 % full (B = 15, 16, 29..32, .. at T = 1000): it is ~6 % faster than two chains of per-layer F(2,3) launches, not more.
    static int env43 = -1;
    if (env43 < 0) { const char* e = getenv("BSG_STACK43"); env43 = e ? atoi(e) : 1; }
    if (env43 && h->compute == BSG_COMPUTE_F32 && !h->split_off && h->num_cus && h->hx && h->epoch_dev) {
      if (h->occ_stack43 < 0) h->occ_stack43 = stack_f43_occupancy() >= 1 ? 1 : 0;
      const int tpr43 = cdiv(T, 64);
      if (h->occ_stack43 >= 1 && tpr43 <= h->num_cus) {
        int rows43 = h->num_cus / tpr43;
        if (rows43 > B) rows43 = B;
        const int groups = cdiv(B, rows43);
        if (env43 == 2 || (long long)B * tpr43 * 10 >= (long long)groups * h->num_cus * 9) {   // BSG_STACK43=2: any shape (tests)
          h->stack_is_f43 = true;
          return rows43;
        }
      }
    }
  }
  return 0;
}

// flag values of one stack launch: launch epoch x 64 + layers published (L < 64)
// The launch epoch of the handle's flags lives in device memory (round 4: a captured launch can be replayed): every workgroup reads it at
// entry, the last one through its layers advances it and, before the 32-bit flag values could come round to a slot that was last written
// long ago (a large tile index after > 2^25 launches of smaller batches), zeroes the flags and starts the epochs again (diffnet_res.h).
static int next_stack_epoch(bsg_diffnet* h, StackArgs& p) {
  BSG_REQUIRE(h->epoch_dev, "stack launch: no launch epoch (bsg_diffnet_prepare allocates it)");
  p.epoch = h->epoch_dev;
  p.fbase = 0;
  p.flag_words = (int)h->flags_cap;
  p.pflag_words = 0;   // part forms: set with p.pflags
  // the flags a wrapping launch zeroes: BOTH arrays of the handle, whichever form wraps (a part launch that has to grow the part arrays
  // replaces them below, zeroed, and updates these two)
  p.wrap_pflags = h->part_flags;
  p.wrap_pflag_words = h->part_flags ? (int)(2 * h->part_cap * 4) : 0;
  static int env_old = -1;   // BSG_DEBUG_WRAP_R04=1: round 4's behaviour (only a PART launch zeroes the part flags at a wrap) — the negative control of tests/test_gpu_handoff.py
  if (env_old < 0) { const char* e = getenv("BSG_DEBUG_WRAP_R04"); env_old = e ? atoi(e) : 0; }
  if (env_old) { p.wrap_pflags = nullptr; p.wrap_pflag_words = 0; }
  return BSG_OK;
}

static int launch_stack(bsg_diffnet* h, const long long* t_dev, int t_uniform, int B, int T, int rows_per_launch, hipStream_t st,
                        unsigned long long* stamps = nullptr, const TailArgs* tail = nullptr) {
  const bool f43 = h->stack_is_f43, h2 = h->stack_is_h2;   // the decision of the stack_rows() call that returned rows_per_launch
  const int nct = h2 ? h->stack_nct : 2;
  const int tpr = cdiv(T, 32 * nct);
  const size_t bt = (size_t)h->B * T;   // bound batch: per-layer stride of the conditioner term
  for (int r0 = 0; r0 < B; r0 += rows_per_launch) {
    const int nb = B - r0 < rows_per_launch ? B - r0 : rows_per_launch;
    const size_t row = (size_t)h->row_off + r0;
    StackArgs p{};
    p.x_in = h->xa + row * C * T;
    p.skip = h->skip + row * C * T;
    p.condterm = h->condterm + row * 2 * C * T;
    p.dproj = h->dproj; p.dconv = h->dconv; p.t_dev = t_dev ? t_dev + r0 : nullptr; p.t_uniform = t_uniform;
    p.apackw = h->apackw; p.apack2 = h->apack2; p.bias_out = h->b_out;
    p.T = T; p.L = h->L; p.tiles_per_row = tpr;
    p.ct_stride = (long long)2 * C * (long long)bt;
    p.n_tiles = nb * tpr; p.cycle = h->cfg.dilation_cycle_length;
    BSG_REQUIRE((size_t)p.n_tiles <= h->flags_cap && h->L < 64, "stack launch: %d tiles exceed the exchange array (%zu)", p.n_tiles, h->flags_cap);
    p.hx = h->hx; p.flags = h->flags; p.status = h->flags + h->flags_cap;
    TRY(next_stack_epoch(h, p));
    if (h->inject_giveup > 0) { p.inject = 1; --h->inject_giveup; }
    else if (h->inject_xcc > 0 && h2 && h->stack_parts) { p.inject = 2; --h->inject_xcc; }
    else if (h->inject_nowait > 0 && h2 && h->stack_parts) { p.inject = 3; --h->inject_nowait; }
    p.stamps = stamps && r0 == 0 ? stamps : nullptr;
    p.clk = h->prof_on && r0 == 0 ? h->clk : nullptr;
    if (stamps) { const char* e = getenv("BSG_STAMP_MODE"); p.stamp_mode = e ? atoi(e) : 0; }
    if (h2 && h->stack_parts) {
      BSG_REQUIRE(!tail && nb == B, "part launch: whole batch, no fused tail");
      p.apack1s = h->apack1s; p.apack2s = h->apack2s; p.h2_scale = h->h2_scale;
      const size_t need32 = (size_t)p.n_tiles * nct;   // slots scale with the tile width: size the buffers in 32-frame tile equivalents
      if (need32 > h->part_cap) {
        hipStreamCaptureStatus capst = hipStreamCaptureStatusNone;
        if (st) (void)hipStreamIsCapturing(st, &capst);
        BSG_REQUIRE(capst == hipStreamCaptureStatusNone, "part launch: the exchange buffers of this shape are allocated by its first eager call; run it once before capturing");
        BSG_HIP(hipStreamSynchronize(st));
        if (h->part_zx) (void)hipFree(h->part_zx);
        if (h->part_ix) (void)hipFree(h->part_ix);
        if (h->part_flags) (void)hipFree(h->part_flags);
        h->part_zx = nullptr; h->part_ix = nullptr; h->part_flags = nullptr; h->part_cap = 0;
        const size_t cap = (size_t)h->num_cus > need32 ? (size_t)h->num_cus : need32;
        const size_t tile_bytes = (size_t)2 * 32 * C * sizeof(unsigned short);   // all parts of a 32-frame tile: 2 planes x 32 frames x C fp16 = 32 KB
        BSG_HIP(hipMalloc((void**)&h->part_zx, cap * tile_bytes));
        BSG_HIP(hipMalloc((void**)&h->part_ix, 2 * cap * tile_bytes));
        BSG_HIP(hipMalloc((void**)&h->part_flags, 2 * cap * 4 * sizeof(unsigned)));
        BSG_HIP(hipMemsetAsync(h->part_flags, 0, 2 * cap * 4 * sizeof(unsigned), st));   // flag values are launch epoch x 64 + layer: monotonic
        h->part_cap = cap;
      }
      p.zx = h->part_zx; p.ix = h->part_ix; p.pflags = h->part_flags;
      p.pflag_words = (int)(2 * h->part_cap * 4);
      p.wrap_pflags = h->part_flags; p.wrap_pflag_words = p.pflag_words;   // (also under BSG_DEBUG_WRAP_R04: a part launch always zeroed its own)
      p.apack1q = h->apack1q; p.apack2q = h->apack2q;
      p.condterm_q = h->cond_q_valid ? h->condterm_q + row * 2 * C * T : nullptr;
      TRY(launch_residual_part_h2(p, st, h->stack_parts, nct));
    } else if (h2) {
      p.apack1s = h->apack1s; p.apack2s = h->apack2s; p.h2_scale = h->h2_scale;
      p.apack1q = h->apack1q; p.apack2q = h->apack2q;
      p.condterm_q = h->stack_q && h->cond_q_valid ? h->condterm_q + row * 2 * C * T : nullptr;
      if (tail) {   // the step tail in the same launch: its tensors start at this launch group's first row
        TailArgs a = *tail;
        const size_t mo = (size_t)r0 * h->M * T;
        a.x += mo; a.xa_next += (size_t)r0 * C * T; a.quad_row0 += mo;
        if (a.noise) a.noise += mo;
        if (a.e_new) a.e_new += mo;
        if (a.h1) a.h1 += mo;
        if (a.h2) a.h2 += mo;
        if (a.h3) a.h3 += mo;
        TRY(h->stack_q ? launch_residual_stack_h2q(p, &a, st, nct) : launch_residual_stack_h2(p, &a, st, nct));
      } else {
        TRY(h->stack_q ? launch_residual_stack_h2q(p, nullptr, st, nct) : launch_residual_stack_h2(p, nullptr, st, nct));
      }
    } else if (f43) {
      p.apackw43 = h->apackw43;
      TRY(launch_residual_stack_f43(p, st));
    } else {
      set_error("stack launch: no launch form selected");
      return BSG_ESTATE;
    }
  }
  BSG_REQUIRE(!tail || h2, "stack launch: a fused tail needs the split-fp16 form");
  h->last_path = h2 ? (h->stack_parts == 2 ? "stack_h2_pair64" : h->stack_parts ? (h->stack_nct == 2 ? "stack_h2_quad64" : "stack_h2_quad") : h->stack_q ? (tail ? "stack_h2q_tail" : "stack_h2q") : tail ? "stack_h2_tail" : "stack_h2") : "stack_f43";
  return BSG_OK;
}

// bf16-operand configuration: the stack launch (the default; BSG_STACK_BF16=0 selects per-layer launches).  64-frame tiles, one
// workgroup per CU (256 registers per wave), whole rows per launch group.  Measured on one box, ms per 100-step pass at T=1000,
// stack / per-layer: B=1 43.2 / 56.4, B=16 58.2 / 87.8, B=32 109.7 / 133.3, B=64 213.8 / 237.3 (tools/bench_small.py with
// BSG_DTYPE=bf16).  Per layer and tile ~21.5 us, of which the two GEMMs' MFMAs are ~7.
static int stack_rows_bf16(bsg_diffnet* h, int B, int T, hipStream_t st) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("BSG_STACK_BF16"); env = e ? atoi(e) : 1; }
  if (!env || h->compute != BSG_COMPUTE_BF16 || h->split_off || !h->num_cus || !h->hx || !h->epoch_dev) return 0;
  (void)st;
  if (h->occ_stack_h < 0) h->occ_stack_h = stack_bf16_occupancy() >= 1 ? 1 : 0;
  const int tpr = cdiv(T, 64);
  const long long slots = (long long)h->occ_stack_h * h->num_cus;
  if (h->occ_stack_h < 1 || tpr > slots) return 0;
  int rows = (int)(slots / tpr);
  if (rows > B) rows = B;
  return rows;
}

static int launch_stack_bf16(bsg_diffnet* h, const long long* t_dev, int t_uniform, int B, int T, int rows_per_launch, hipStream_t st,
                             unsigned long long* stamps = nullptr) {
  const int tpr = cdiv(T, 64);
  const size_t bt = (size_t)h->B * T;
  for (int r0 = 0; r0 < B; r0 += rows_per_launch) {
    const int nb = B - r0 < rows_per_launch ? B - r0 : rows_per_launch;
    const size_t row = (size_t)h->row_off + r0;
    StackArgs p{};
    p.x_in = h->xa + row * C * T;
    p.skip = h->skip + row * C * T;
    p.skip_h = h->skip_h + row * C * T;
    p.condterm_h = h->condterm_h + row * 2 * C * T;
    p.dproj = h->dproj; p.t_dev = t_dev ? t_dev + r0 : nullptr; p.t_uniform = t_uniform;
    p.apack1h = h->apack1h; p.apack2h = h->apack2h; p.bias_out = h->b_out;
    p.T = T; p.L = h->L; p.tiles_per_row = tpr;
    p.ct_stride = (long long)2 * C * (long long)bt;
    p.n_tiles = nb * tpr; p.cycle = h->cfg.dilation_cycle_length;
    BSG_REQUIRE((size_t)p.n_tiles <= h->flags_cap && h->L < 64, "bf16 stack launch: %d tiles exceed the exchange array (%zu)", p.n_tiles, h->flags_cap);
    p.hx = h->hx; p.flags = h->flags; p.status = h->flags + h->flags_cap;
    TRY(next_stack_epoch(h, p));
    if (h->inject_giveup > 0) { p.inject = 1; --h->inject_giveup; }
    p.stamps = stamps && r0 == 0 ? stamps : nullptr;
    p.clk = h->prof_on && r0 == 0 ? h->clk : nullptr;
    TRY(launch_residual_stack_bf16(p, st));
  }
  h->last_path = "stack_bf16";
  return BSG_OK;
}

static int check_bound(bsg_diffnet* h, int B, int T, const char* who) {
  if (!h) { set_error("%s: null handle", who); return BSG_EINVAL; }
  if (h->cap_bt == 0 || h->B != B || h->T != T) {
    set_error("%s: (B=%d,T=%d) does not match the condition bound by bsg_diffnet_prepare (B=%d,T=%d)", who, B, T, h->B, h->T);
    return BSG_ESTATE;
  }
  if (h->compute != h->prepared_compute) {
    // (either way: a BF16 prepare writes the bf16 quads only — since round 5 no fp32 copy — and an F32 prepare no bf16 quads)
    set_error("%s: the condition was bound under another bsg_diffnet_set_compute mode (%d) than the current one (%d); call bsg_diffnet_prepare again", who,
              h->prepared_compute, h->compute);
    return BSG_ESTATE;
  }
  return BSG_OK;
}

// Does the launch this shape takes read the ROW layout of the conditioner term?  (The 16-row stack launch and the part forms read the quads,
// the bf16 launches the bf16 quads; everything else — 32-row launch, F(4,3), per-layer and channel-split kernels — the rows.)
static bool cond_rows_needed(bsg_diffnet* h, int B, int T, hipStream_t st) {
  if (h->compute == BSG_COMPUTE_BF16) return false;
  const int srows = h->no_split ? 0 : stack_rows(h, B, T, st);
  return !(srows && h->stack_is_h2 && h->cond_q_valid && (h->stack_parts || h->stack_q));
}
static int cond_layout_for(bsg_diffnet* h, int B, int T, hipStream_t st) {
  return cond_rows_needed(h, B, T, st) ? ensure_cond_rows(h, st) : BSG_OK;
}

// eps = DiffNet(x, t); t either per-row on the device or uniform
static int forward_impl(bsg_diffnet* h, const float* x, const long long* t_dev, int t_uniform, float* eps, int B, int T,
                        hipStream_t st) {
  TRY(conv1x1(h->w_in, h->b_in, x, h->xa, C, h->M, B, T, ACT_RELU, st));  // net.py:116-118
  float* cur = h->xa;
  float* nxt = h->xb;
  const bool prof = h->prof_on && h->prof_used + 2 <= h->prof_ev.size();
  if (prof) BSG_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
  const int srows = stack_rows(h, B, T, st);
  const int hrows = stack_rows_bf16(h, B, T, st);
  if (hrows) {
    TRY(launch_stack_bf16(h, t_dev, t_uniform, B, T, hrows, st));
  } else if (srows) {
    TRY(launch_stack(h, t_dev, t_uniform, B, T, srows, st));
  } else {
    for (int l = 0; l < h->L; ++l) {
      TRY(launch_layer(h, l, cur, t_dev, t_uniform, nxt, h->skip, B, T, st));
      float* tmp = cur; cur = nxt; nxt = tmp;
    }
  }
  if (prof) {
    BSG_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
    h->prof_used += 2;
    h->prof_launches += h->L;
  }
  if (h->compute == BSG_COMPUTE_BF16) TRY(quad_bf16_to_f32(h->skip_h, h->skip, B, C, T, st));
  TRY(conv1x1(h->w_skip, h->b_skip, h->skip, h->hid, C, C, B, T, ACT_RELU, st));   // net.py:127-128
  TRY(conv1x1(h->w_fin, h->b_fin, h->hid, eps, h->M, C, B, T, ACT_NONE, st));      // net.py:129
  return BSG_OK;
}

extern "C" int bsg_diffnet_forward(bsg_diffnet* h, const float* x, const int64_t* t, float* eps, int32_t B, int32_t T,
                                   void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  TRY(check_bound(h, B, T, "diffnet_forward"));
  BSG_REQUIRE(x && t && eps, "diffnet_forward: null argument");
  TRY(cond_layout_for(h, B, T, (hipStream_t)stream));
  return forward_impl(h, x, (const long long*)t, 0, eps, B, T, (hipStream_t)stream);
}

extern "C" int bsg_diffnet_residual_layer(bsg_diffnet* h, int32_t layer, const float* x_in, const int64_t* t, float* x_out,
                                          float* skip, int32_t B, int32_t T, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  TRY(check_bound(h, B, T, "diffnet_residual_layer"));
  BSG_REQUIRE(x_in && t && x_out && skip && x_in != x_out, "diffnet_residual_layer: null or aliased argument");
  BSG_REQUIRE(layer >= 0 && layer < h->L, "diffnet_residual_layer: layer %d out of range", layer);
  if (h->compute == BSG_COMPUTE_F32) TRY(ensure_cond_rows(h, (hipStream_t)stream));
  return launch_layer(h, layer, x_in, (const long long*)t, 0, x_out, skip, B, T, (hipStream_t)stream);
}

static int check_schedule(const bsg_schedule* s, const char* who, bool plms) {
  if (!s || s->num_timesteps <= 0) { set_error("%s: bad schedule", who); return BSG_EINVAL; }
  if (plms ? !s->alphas_cumprod
           : !(s->sqrt_recip_alphas_cumprod && s->sqrt_recipm1_alphas_cumprod && s->posterior_mean_coef1 &&
               s->posterior_mean_coef2 && s->sigma)) {
    set_error("%s: schedule array missing", who);
    return BSG_EINVAL;
  }
  return BSG_OK;
}

// the 20 residual layers of one evaluation, input h->xa (the in-projection of x), output = the skip sum in h->skip(_h)
static int layers_from_xa(bsg_diffnet* h, int t_uniform, int B, int T, hipStream_t st) {
  const size_t off = (size_t)h->row_off * C * T;
  float* cur = h->xa + off;
  float* nxt = h->xb + off;
  const bool prof = h->prof_on && h->prof_used + 2 <= h->prof_ev.size();
  if (prof) BSG_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
  const int srows = h->no_split ? 0 : stack_rows(h, B, T, st);
  const int hrows = stack_rows_bf16(h, B, T, st);
  if (hrows) {
    TRY(launch_stack_bf16(h, nullptr, t_uniform, B, T, hrows, st));
  } else if (srows) {
    TRY(launch_stack(h, nullptr, t_uniform, B, T, srows, st));
  } else {
    for (int l = 0; l < h->L; ++l) {
      TRY(launch_layer(h, l, cur, nullptr, t_uniform, nxt, h->skip + off, B, T, st));
      float* tmp = cur; cur = nxt; nxt = tmp;
    }
  }
  if (prof) {
    BSG_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
    h->prof_used += 2;
    h->prof_launches += h->L;   // layer-equivalents (the stack launch runs L layers in one kernel)
  }
  return BSG_OK;
}

static bool fused_tail_ok(const bsg_diffnet* h) {
  return h->ws_pack != nullptr && (h->MP == 80 || h->MP == 96) && !getenv("BSG_NO_FUSED_TAIL");
}

// step_tail_kernel: skip projection, output projection, sampler update of x (DDPM, or PLMS when a.plms_hist > 0) and the next
// evaluation's in-projection into h->xa; the caller fills the sampler-specific fields of `a`
static int launch_tail(bsg_diffnet* h, TailArgs& a, float* x, int B, int T, hipStream_t st) {
  static bool tail_attr = false;
  const size_t tail_lds = (size_t)(C * 32 + 96 * 32) * sizeof(float);
  if (!tail_attr) {
    BSG_HIP(hipFuncSetAttribute((const void*)step_tail_kernel<80, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
    BSG_HIP(hipFuncSetAttribute((const void*)step_tail_kernel<96, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
    BSG_HIP(hipFuncSetAttribute((const void*)step_tail_kernel<80, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
    BSG_HIP(hipFuncSetAttribute((const void*)step_tail_kernel<96, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds));
    tail_attr = true;
  }
  const size_t off = (size_t)h->row_off * C * T;
  a.skip = h->skip + off; a.skip_h = h->compute == BSG_COMPUTE_BF16 ? h->skip_h + off : nullptr;
  a.x = x; a.xa_next = h->xa + off;
  a.ws_pack = h->ws_pack; a.wo_pack = h->wo_pack; a.wi_pack = h->wi_pack; a.b_skip = h->b_skip; a.b_fin = h->b_fin96; a.b_in = h->b_in;
  a.B = B; a.T = T; a.M = h->M; a.tiles_per_row = cdiv(T, 32);
  const char* tail_env = getenv("BSG_TAIL_BF16");   // "0": the fp32 tail also in the bf16-operand configuration
  if (h->compute == BSG_COMPUTE_BF16 && !(tail_env && atoi(tail_env) == 0) && h->tail_h) {   // bf16-operand configuration: the projections on bf16 MFMAs too
    a.ws_h = h->tail_h; a.wo_h = h->tail_h + C * C; a.wi_h = h->tail_h + C * C + 96 * C;
    return launch_step_tail_bf16(a, st);
  }
  const dim3 grid(B * a.tiles_per_row), block(512);
  if (a.plms_hist) {
    if (h->MP == 80) hipLaunchKernelGGL((step_tail_kernel<80, true>), grid, block, tail_lds, st, a);
    else hipLaunchKernelGGL((step_tail_kernel<96, true>), grid, block, tail_lds, st, a);
  } else {
    if (h->MP == 80) hipLaunchKernelGGL((step_tail_kernel<80, false>), grid, block, tail_lds, st, a);
    else hipLaunchKernelGGL((step_tail_kernel<96, false>), grid, block, tail_lds, st, a);
  }
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// One sampler step from the in-projected x in h->xa: the residual stack, then the tail (`a` carries the sampler-specific fields).  When
// the stack runs as the split-fp16 launch, the tail runs inside it (one launch per step; BSG_H2_TAIL=0: two launches).
static int step_from_xa(bsg_diffnet* h, int t_uniform, TailArgs& a, float* x, int B, int T, hipStream_t st) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("BSG_H2_TAIL"); env = e ? atoi(e) : 1; }
  const int srows = (h->no_split || h->compute != BSG_COMPUTE_F32 || !env || !h->tail_s || h->M > 96) ? 0 : stack_rows(h, B, T, st);
  if (srows && h->stack_is_h2 && !h->stack_parts) {
    const size_t off = (size_t)h->row_off * C * T;
    a.x = x; a.xa_next = h->xa + off;
    a.b_skip = h->b_skip; a.b_fin = h->b_fin96; a.b_in = h->b_in;
    a.B = B; a.T = T; a.M = h->M; a.tiles_per_row = cdiv(T, 64);
    a.ws_s = h->tail_s; a.wo_s = h->tail_s + 2 * C * C; a.wi_s = h->tail_s + 2 * C * C + 2 * 96 * C; a.tail_scale = h->tail_scale;
    const bool prof = h->prof_on && h->prof_used + 2 <= h->prof_ev.size();
    if (prof) BSG_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
    TRY(launch_stack(h, nullptr, t_uniform, B, T, srows, st, nullptr, &a));
    if (prof) {
      BSG_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
      h->prof_used += 2;
      h->prof_launches += h->L;   // layer-equivalents; the launch also carries the step tail (~1 % of its FLOPs)
    }
    return BSG_OK;
  }
  TRY(layers_from_xa(h, t_uniform, B, T, st));
  if (srows && h->stack_is_h2 && h->stack_parts && strncmp(h->last_path, "stack_h2_", 9) == 0) {
    // behind a part launch: the tail on the 16-bit matrix pipe too (step_tail_h2_kernel; BSG_H2_TAIL=0: the fp32-pipe tail)
    const size_t off = (size_t)h->row_off * C * T;
    a.skip = h->skip + off; a.skip_h = nullptr;
    a.x = x; a.xa_next = h->xa + off;
    a.b_skip = h->b_skip; a.b_fin = h->b_fin96; a.b_in = h->b_in;
    a.B = B; a.T = T; a.M = h->M; a.tiles_per_row = cdiv(T, 32);
    a.ws_s = h->tail_s; a.wo_s = h->tail_s + 2 * C * C; a.wi_s = h->tail_s + 2 * C * C + 2 * 96 * C; a.tail_scale = h->tail_scale;
    a.status = h->flags + h->flags_cap;
    return launch_step_tail_h2(a, st);
  }
  return launch_tail(h, a, x, B, T, st);
}

struct SubBatch { int off, B; hipStream_t st; };

// Two half-batches on two streams.  One launch per layer puts all workgroups of the chip in the same phase (they stage, hit
// the gate and drain together, and the younger of the two workgroups of a CU finishes alone); two independent launch chains
// drift apart and fill each other's gaps: measured 239.6 -> 226.5 ms per 100 steps at B=16, T=1000 (+5.8 %), +4.0 % at B=32,
// +9.9 % at B=12, +1.2 % at B=64, +4 % for the bf16 form at B=64; four chains are worse (a CU only holds two of these
// workgroups).  On one of the boxes measured the two chains brought no gain (and no loss).  BSG_DUAL=0 disables.
// Returns the number of sub-batches (1 or 2) and, for 2, forks the second stream off `st`.
static int dual_fork(bsg_diffnet* h, int B, int T, hipStream_t st, SubBatch (&subs)[2]) {
  static int dual_env = -1;
  if (dual_env < 0) { const char* e = getenv("BSG_DUAL"); dual_env = e ? atoi(e) : 1; }
  subs[0] = SubBatch{0, B, st};
  subs[1] = SubBatch{0, 0, nullptr};
  const long long tiles = (long long)B * cdiv(T, 32);
  const bool big = tiles > h->num_cus;
  // 129..256 tiles (B = 5..8): two chains of channel-split launches (each workgroup half the matrix work, two per CU, one of each
  // chain) instead of one chain of 16-wave workgroups
  // (measured per 100 steps at T=1000: B=5 130.5 -> 105.9 ms, B=6 133.5 -> 120.4, B=8 135.9 -> 133.8; BSG_DUAL=2: big batches only)
  // 65..128 tiles (B = 3, 4): two chains of 4-way split launches: B=3 89.0 -> 83.8 ms, B=4 93.3 -> 90.6
  bool small = dual_env != 2 && !big && 4 * tiles > h->num_cus && h->compute == BSG_COMPUTE_F32 && split_env() && !h->split_off;
  if (small) {
    // two chains of split launches share CUs (un-padded LDS): all workgroups of BOTH launches must be resident at once, or a
    // polling workgroup could wait for a partner that cannot start.  Per half: parts x tiles workgroups of 1024/parts threads.
    if (h->occ2 < 0) query_split_occupancy(h);
    double cus = 0.0;   // CUs the workgroups of both launches occupy at the residency the runtime reports for each form
    bool ok = true;
    for (int half = 0; half < 2; ++half) {
      const long long th = (long long)(half ? B - B / 2 : B / 2) * cdiv(T, 32);
      const int parts = 4 * th <= h->num_cus && split_env() != 2 ? 4 : 2 * th <= h->num_cus ? 2 : 0;
      int occ = parts == 4 ? h->occ4s : h->occ2s;
      const int cap = parts == 4 ? 3 : 2;   // 48 KB of LDS each -> 3 per CU; 8 waves of 128 VGPRs -> 2 per CU
      if (occ > cap) occ = cap;
      if (!parts || occ < 1) { ok = false; break; }
      cus += (double)(parts * th) / occ;
    }
    small = ok && cus <= (double)h->num_cus;
  }
  const bool dual = dual_env && B >= 2 && use_wino() && !stack_rows(h, B, T, st) && !stack_rows_bf16(h, B, T, st) && (big || small);
  if (!dual) return 1;
  if (!h->st2) {
    if (hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)
      return 1;
  }
  if (hipEventRecord(h->ev_fork, st) != hipSuccess || hipStreamWaitEvent(h->st2, h->ev_fork, 0) != hipSuccess) return 1;
  subs[0] = SubBatch{0, B / 2, st};
  subs[1] = SubBatch{B / 2, B - B / 2, h->st2};
  h->no_split = big;
  h->split_small_lds = small;
  return 2;
}

// joins the second stream back into `st` (also after an error, so that the caller's stream stays ordered after everything
// that was enqueued on the second one) and restores the handle's launch state
static int dual_join(bsg_diffnet* h, int n_sub, hipStream_t st, int rc) {
  h->row_off = 0;
  h->no_split = false;
  h->split_small_lds = false;
  if (n_sub == 2) {
    hipError_t e1 = hipEventRecord(h->ev_join, h->st2);
    hipError_t e2 = hipStreamWaitEvent(st, h->ev_join, 0);
    if (rc == BSG_OK && (e1 != hipSuccess || e2 != hipSuccess)) { set_error("sampler: stream join failed"); rc = BSG_EHIP; }
  }
  return rc;
}

extern "C" int bsg_ddpm_sample(bsg_diffnet* h, const bsg_schedule* s, float* x, const float* noise, uint64_t seed,
                               int32_t t_start, int32_t n_steps, int32_t B, int32_t T, int32_t row0, int32_t B_total,
                               void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  TRY(check_bound(h, B, T, "ddpm_sample"));
  TRY(check_schedule(s, "ddpm_sample", false));
  BSG_REQUIRE(x, "ddpm_sample: null x");
  BSG_REQUIRE(t_start < s->num_timesteps && t_start < h->cfg.max_steps && n_steps >= 0 && t_start - n_steps + 1 >= 0,
              "ddpm_sample: steps [%d..%d] outside the schedule (%d) / step table (%d)", t_start - n_steps + 1, t_start,
              s->num_timesteps, h->cfg.max_steps);
  BSG_REQUIRE(row0 >= 0 && row0 + B <= B_total, "ddpm_sample: rows [%d,%d) outside the global batch %d", row0, row0 + B, B_total);
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * h->M * T;
  BSG_REQUIRE(n % 4 == 0, "ddpm_sample: B*M*T must be a multiple of 4");
  TRY(cond_layout_for(h, B, T, st));
  const long long n4 = n / 4;
  const unsigned long long quad0 = (unsigned long long)row0 * h->M * T / 4;
  const bool fused = fused_tail_ok(h);
  if (!fused) {
    for (int k = 0; k < n_steps; ++k) {
      const int i = t_start - k;
      TRY(forward_impl(h, x, nullptr, i, h->eps, B, T, st));
      StepCoef c{s->sqrt_recip_alphas_cumprod[i], s->sqrt_recipm1_alphas_cumprod[i], s->posterior_mean_coef1[i],
                 s->posterior_mean_coef2[i], s->sigma[i]};
      hipLaunchKernelGGL(ddpm_step_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, x, (const float*)h->eps,
                         noise ? noise + (long long)k * n : nullptr, c, n4, (unsigned long long)seed, (unsigned)(i + 1), quad0);
      BSG_LAUNCH_CHECK();
    }
    return BSG_OK;
  }
  // fused loop: [in-projection once] -> per step: 20 residual layers -> step_tail_kernel (skip projection, output
  // projection, sampler update, next step's in-projection)
  SubBatch subs[2];
  const int n_sub = n_steps > 0 ? dual_fork(h, B, T, st, subs) : 1;
  if (n_sub == 1) subs[0] = SubBatch{0, B, st};
  int rc = BSG_OK;
  for (int u = 0; u < n_sub && rc == BSG_OK; ++u)
    rc = conv1x1(h->w_in, h->b_in, x + (size_t)subs[u].off * h->M * T, h->xa + (size_t)subs[u].off * C * T, C, h->M, subs[u].B, T, ACT_RELU,
                 subs[u].st);
  for (int k = 0; k < n_steps && rc == BSG_OK; ++k) {
    const int i = t_start - k;
    for (int u = 0; u < n_sub && rc == BSG_OK; ++u) {
      h->row_off = subs[u].off;
      TailArgs a{};
      a.noise = noise ? noise + (long long)k * n + (long long)subs[u].off * h->M * T : nullptr;
      a.k = StepCoef{s->sqrt_recip_alphas_cumprod[i], s->sqrt_recipm1_alphas_cumprod[i], s->posterior_mean_coef1[i],
                     s->posterior_mean_coef2[i], s->sigma[i]};
      a.seed = seed; a.quad_row0 = (unsigned long long)(row0 + subs[u].off) * h->M * T; a.stream = (unsigned)(i + 1);
      a.do_head = k + 1 < n_steps;
      rc = step_from_xa(h, i, a, x + (size_t)subs[u].off * h->M * T, subs[u].B, T, subs[u].st);
    }
  }
  rc = dual_join(h, n_sub, st, rc);
  return rc;
}

extern "C" int bsg_diffnet_set_compute(bsg_diffnet* h, int32_t mode) {
  BSG_REQUIRE(h, "diffnet_set_compute: null handle");
  BSG_REQUIRE(mode == BSG_COMPUTE_F32 || mode == BSG_COMPUTE_BF16, "diffnet_set_compute: unknown mode %d", mode);
  h->compute = mode;
  return BSG_OK;
}

extern "C" int bsg_diffnet_debug_stamps(bsg_diffnet* h, int32_t layer, const float* x_in, const int64_t* t, float* x_out,
                                       float* skip, int32_t B, int32_t T, uint64_t* stamps, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  TRY(check_bound(h, B, T, "diffnet_debug_stamps"));
  BSG_REQUIRE(x_in && t && x_out && skip && stamps && x_in != x_out, "diffnet_debug_stamps: null or aliased argument");
  BSG_REQUIRE(layer >= 0 && layer < h->L, "diffnet_debug_stamps: layer %d out of range", layer);
  if (h->compute == BSG_COMPUTE_F32) TRY(ensure_cond_rows(h, (hipStream_t)stream));
  return launch_layer(h, layer, x_in, (const long long*)t, 0, x_out, skip, B, T, (hipStream_t)stream, (unsigned long long*)stamps);
}

extern "C" int bsg_diffnet_status(bsg_diffnet* h, int32_t* handoff_timeouts) {
  BSG_REQUIRE(h && handoff_timeouts, "diffnet_status: null argument");
  *handoff_timeouts = 0;
  if (h->flags) {
    unsigned v[2] = {0, 0};   // word 0: hand-off give-ups, word 1: values beyond the fp16 range of the split-fp16 launch
    BSG_HIP(hipMemcpy(v, h->flags + h->flags_cap, 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
    *handoff_timeouts = (int32_t)(v[0] + v[1]);
  }
  if (h->split_flags) {
    unsigned v = 0;
    BSG_HIP(hipMemcpy(&v, h->split_flags + 16 * h->split_cap, sizeof(unsigned), hipMemcpyDeviceToHost));
    *handoff_timeouts += (int32_t)v;
  }
  return BSG_OK;
}

extern "C" int bsg_diffnet_health_take(bsg_diffnet* h, int32_t* counts, void* stream) {
  BSG_REQUIRE(h && counts, "diffnet_health_take: null argument");
  hipStream_t st = (hipStream_t)stream;
  counts[0] = counts[1] = 0;
  unsigned v[3] = {0, 0, 0};
  if (h->flags) BSG_HIP(hipMemcpyAsync(&v[0], h->flags + h->flags_cap, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  if (h->split_flags) BSG_HIP(hipMemcpyAsync(&v[2], h->split_flags + 16 * h->split_cap, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  BSG_HIP(hipStreamSynchronize(st));
  if (v[0] | v[1]) BSG_HIP(hipMemsetAsync(h->flags + h->flags_cap, 0, 3 * sizeof(unsigned), st));   // (word 2: the flag base of the part launch that gave up)
  if (v[2]) BSG_HIP(hipMemsetAsync(h->split_flags + 16 * h->split_cap, 0, sizeof(unsigned), st));
  counts[0] = (int32_t)(v[0] + v[2]);
  counts[1] = (int32_t)v[1];
  return BSG_OK;
}

extern "C" int bsg_diffnet_handoff_take(bsg_diffnet* h, int32_t* handoff_timeouts, void* stream) {
  BSG_REQUIRE(h && handoff_timeouts, "diffnet_handoff_take: null argument");
  int32_t c[2] = {0, 0};
  TRY(bsg_diffnet_health_take(h, c, stream));
  *handoff_timeouts = c[0] + c[1];
  return BSG_OK;
}

extern "C" int bsg_diffnet_uses_handoffs(bsg_diffnet* h, int32_t B, int32_t T, int32_t* uses) {
  BSG_REQUIRE(h && uses && B > 0 && T > 0, "diffnet_uses_handoffs: bad argument");
  if (!h->num_cus) {   // normally set by prepare(); asked before any condition was bound
    int dev = 0;
    BSG_HIP(hipGetDevice(&dev));
    BSG_HIP(hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  // conservative: any launch shape for which a channel-split (pair / 4-way) or the stack launch may be chosen
  const long long tiles = (long long)B * cdiv(T, 32);
  const bool split = h->compute == BSG_COMPUTE_F32 && use_wino() && split_env() && !h->split_off && h->num_cus && tiles <= h->num_cus;
  *uses = (split || stack_rows(h, B, T, nullptr) > 0 || stack_rows_bf16(h, B, T, nullptr) > 0) ? 1 : 0;
  return BSG_OK;
}

extern "C" int bsg_diffnet_set_split(bsg_diffnet* h, int32_t enable) {
  BSG_REQUIRE(h, "diffnet_set_split: null handle");
  h->split_off = enable == 0;
  return BSG_OK;
}

extern "C" int bsg_diffnet_set_h2(bsg_diffnet* h, int32_t enable) {
  BSG_REQUIRE(h, "diffnet_set_h2: null handle");
  h->h2_off = enable == 0;
  return BSG_OK;
}

// middle tier of the range guard (ABI v7): 0 takes the 16-row stack launch (its conv image holds 16 x: |x| < 3750) off this handle; shapes
// that ran it take the 32-row launch (residual_stack_h2_kernel: |x + d| < 60000, about 8 % slower) — still the 16-bit matrix pipe
extern "C" int bsg_diffnet_set_h2q(bsg_diffnet* h, int32_t enable) {
  BSG_REQUIRE(h, "diffnet_set_h2q: null handle");
  h->q_off = enable == 0;
  return BSG_OK;
}

// test hook (ABI v6): the launch epoch of the handle's stack / part launches (device memory) := epoch, so that the wrap at 2^25 — two
// hours of single-utterance serving away — can be driven by a test
extern "C" int bsg_diffnet_debug_set_epoch(bsg_diffnet* h, uint32_t epoch, void* stream) {
  BSG_REQUIRE(h && h->epoch_dev && epoch >= 1u, "diffnet_debug_set_epoch: no launch epoch yet (prepare allocates it), or epoch 0");
  hipStream_t st = (hipStream_t)stream;
  BSG_HIP(hipStreamSynchronize(st));
  const unsigned v[2] = {epoch, 0u};
  BSG_HIP(hipMemcpy(h->epoch_dev, v, sizeof(v), hipMemcpyHostToDevice));
  // flag words hold epoch x 64 + layer of launches that ran under smaller epochs: still older than every launch from here on
  return BSG_OK;
}

// fault injection (ABI v6): in the next n_launches PART launches the odd parts of every tile report another XCC id than the one they run on —
// what the parts would see under a dispatch order other than workgroup i -> XCD i mod 8
extern "C" int bsg_diffnet_debug_inject_xcc(bsg_diffnet* h, int32_t n_launches) {
  BSG_REQUIRE(h && n_launches >= 0, "diffnet_debug_inject_xcc: bad argument");
  h->inject_xcc = n_launches;
  return BSG_OK;
}

// part forms on / off for this handle (ABI v6): the first tier of the self-heal after a hand-off give-up inside a part launch — the
// one-workgroup-per-tile stack launch needs no placement on one XCD and stays; bsg_diffnet_set_split(h, 0) is the second tier
extern "C" int bsg_diffnet_set_parts(bsg_diffnet* h, int32_t enable) {
  BSG_REQUIRE(h, "diffnet_set_parts: null handle");
  h->parts_off = !enable;
  return BSG_OK;
}

extern "C" int bsg_diffnet_debug_inject_giveup(bsg_diffnet* h, int32_t n_launches) {
  BSG_REQUIRE(h, "diffnet_debug_inject_giveup: bad argument");
  // n < 0 (timing experiment, tools/part_nowait.py): the next -n PART launches skip every hand-off wait without counting anything — wrong
  // results, the same instructions otherwise: what the waits themselves cost
  h->inject_giveup = n_launches > 0 ? n_launches : 0;
  h->inject_nowait = n_launches < 0 ? -n_launches : 0;
  return BSG_OK;
}

extern "C" int bsg_diffnet_status_async(bsg_diffnet* h, int32_t* host_counts, void* stream) {
  BSG_REQUIRE(h && host_counts, "diffnet_status_async: null argument");
  hipStream_t st = (hipStream_t)stream;
  host_counts[0] = host_counts[1] = host_counts[2] = 0;
  // words 0 and 1 of the stack launches' status are adjacent: one copy fills host_counts[0] (give-ups) and [1] (range events)
  if (h->flags) BSG_HIP(hipMemcpyAsync(&host_counts[0], h->flags + h->flags_cap, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  if (h->split_flags) BSG_HIP(hipMemcpyAsync(&host_counts[2], h->split_flags + 16 * h->split_cap, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  return BSG_OK;
}

extern "C" int bsg_diffnet_debug_stack_stamps(bsg_diffnet* h, int32_t t_uniform, int32_t B, int32_t T, uint64_t* stamps, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  TRY(check_bound(h, B, T, "diffnet_debug_stack_stamps"));
  BSG_REQUIRE(stamps, "diffnet_debug_stack_stamps: null stamps");
  TRY(cond_layout_for(h, B, T, (hipStream_t)stream));
  if (h->compute == BSG_COMPUTE_BF16) {
    const int hrows = stack_rows_bf16(h, B, T, (hipStream_t)stream);
    BSG_REQUIRE(hrows >= B, "diffnet_debug_stack_stamps: (B=%d,T=%d) does not run as one bf16 stack launch", B, T);
    return launch_stack_bf16(h, nullptr, t_uniform, B, T, hrows, (hipStream_t)stream, (unsigned long long*)stamps);
  }
  const int rows = stack_rows(h, B, T, (hipStream_t)stream);
  BSG_REQUIRE(rows >= B, "diffnet_debug_stack_stamps: (B=%d,T=%d) does not run as one stack launch", B, T);
  return launch_stack(h, nullptr, t_uniform, B, T, rows, (hipStream_t)stream, (unsigned long long*)stamps);
}

extern "C" const char* bsg_diffnet_last_path(bsg_diffnet* h) { return h ? h->last_path : "none"; }

extern "C" int bsg_diffnet_clock_read(bsg_diffnet* h, double* shader_mhz, double* span_us) {
  BSG_REQUIRE(h && shader_mhz && span_us, "diffnet_clock_read: null argument");
  *shader_mhz = 0.0;
  *span_us = 0.0;
  if (!h->clk) return BSG_OK;
  unsigned long long v[4] = {0, 0, 0, 0};
  BSG_HIP(hipMemcpy(v, h->clk, sizeof(v), hipMemcpyDeviceToHost));
  if (v[3] > v[1] && v[2] > v[0]) {
    *span_us = (double)(v[3] - v[1]) / 100.0;            // s_memrealtime counts at 100 MHz
    *shader_mhz = (double)(v[2] - v[0]) / *span_us;      // s_memtime counts shader clocks
  }
  return BSG_OK;
}

extern "C" int bsg_diffnet_profile(bsg_diffnet* h, int32_t enable) {
  BSG_REQUIRE(h, "diffnet_profile: null handle");
  if (enable && h->prof_ev.empty()) {
    h->prof_ev.resize(2 * 4096);
    for (hipEvent_t& e : h->prof_ev) BSG_HIP(hipEventCreate(&e));
  }
  h->prof_on = enable != 0;
  h->prof_used = 0;
  h->prof_launches = 0;
  return BSG_OK;
}

extern "C" int bsg_diffnet_profile_read(bsg_diffnet* h, double* layer_ms_total, int64_t* n_layer_launches) {
  BSG_REQUIRE(h && layer_ms_total && n_layer_launches, "diffnet_profile_read: null argument");
  double total = 0.0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    BSG_HIP(hipEventSynchronize(h->prof_ev[i + 1]));
    float ms = 0.f;
    BSG_HIP(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
    total += ms;
  }
  *layer_ms_total = total;
  *n_layer_launches = (int64_t)h->prof_launches;
  return BSG_OK;
}

extern "C" int bsg_mel_finish(const float* x, const float* spec_min, const float* spec_max, const int64_t* mel2ph,
                              float* mel_out, int32_t B, int32_t M, int32_t T, void* stream) {
  BSG_REQUIRE(x && spec_min && spec_max && mel_out && B > 0 && M > 0 && M <= 512 && T > 0, "mel_finish: bad argument");
  hipLaunchKernelGGL(mel_finish_kernel, dim3(cdiv(T, 64), B), dim3(256), (size_t)M * 65 * sizeof(float), (hipStream_t)stream, x,
                     spec_min, spec_max, (const long long*)mel2ph, mel_out, M, T);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" int bsg_mel_start(const float* fs2_mel, const float* spec_min, const float* spec_max, const float* noise,
                             float sqrt_ac, float sqrt_1m_ac, float* x, int32_t B, int32_t M, int32_t T, void* stream) {
  BSG_REQUIRE(fs2_mel && spec_min && spec_max && noise && x && B > 0 && M > 0 && M <= 512 && T > 0, "mel_start: bad argument");
  hipLaunchKernelGGL(mel_start_kernel, dim3(cdiv(T, 64), B), dim3(256), (size_t)64 * (M + 1) * sizeof(float), (hipStream_t)stream,
                     fs2_mel, spec_min, spec_max, noise, sqrt_ac, sqrt_1m_ac, x, M, T);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" int bsg_ddpm_step(float* x, const float* eps, const float* noise, const bsg_schedule* s, int32_t t, int64_t n,
                             uint64_t seed, uint64_t offset, void* stream) {
  TRY(check_schedule(s, "ddpm_step", false));
  BSG_REQUIRE(x && eps && n > 0 && n % 4 == 0 && offset % 4 == 0 && t >= 0 && t < s->num_timesteps, "ddpm_step: bad argument");
  StepCoef c{s->sqrt_recip_alphas_cumprod[t], s->sqrt_recipm1_alphas_cumprod[t], s->posterior_mean_coef1[t],
             s->posterior_mean_coef2[t], s->sigma[t]};
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, eps, noise, c, (long long)(n / 4),
                     (unsigned long long)seed, (unsigned)(t + 1), (unsigned long long)(offset / 4));
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" int bsg_philox_normal(float* x, int64_t n, uint64_t seed, uint32_t stream_id, uint64_t offset, void* stream) {
  BSG_REQUIRE(x && n > 0 && n % 4 == 0 && offset % 4 == 0, "philox_normal: n=%lld offset=%llu must be multiples of 4", (long long)n, (unsigned long long)offset);
  hipLaunchKernelGGL(philox_fill_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)(n / 4),
                     (unsigned long long)seed, (unsigned)stream_id, (unsigned long long)(offset / 4));
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" int bsg_plms_sample(bsg_diffnet* h, const bsg_schedule* s, float* x, int32_t K_step, int32_t interval, int32_t B,
                               int32_t T, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  TRY(check_bound(h, B, T, "plms_sample"));
  TRY(check_schedule(s, "plms_sample", true));
  BSG_REQUIRE(x && interval > 0 && K_step > 0 && K_step <= s->num_timesteps && K_step <= h->cfg.max_steps,
              "plms_sample: K_step=%d interval=%d schedule=%d", K_step, interval, s->num_timesteps);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)B * h->M * T;
  BSG_REQUIRE(h->xpred, "plms_sample: history buffers missing (bsg_diffnet_prepare allocates them)");
  TRY(cond_layout_for(h, B, T, st));
  const dim3 grid(cdiv((long long)n, 256)), block(256);
  const bool fused = fused_tail_ok(h);
  // history ring: hist[0] = newest
  float* hist[4] = {h->eps_hist[0], h->eps_hist[1], h->eps_hist[2], h->eps_hist[3]};
  int n_hist = 0;
  const int last = ((K_step - 1) / interval) * interval;
  SubBatch subs[2];
  int n_sub = 0, rc = BSG_OK;   // n_sub == 0: the half-batch chains have not been forked yet
  for (int i = last; i >= 0; i -= interval) {
    const int ip = i - interval > 0 ? i - interval : 0;
    PlmsCoef c{};
    c.a_t = s->alphas_cumprod[i];
    c.a_prev = s->alphas_cumprod[ip];
    float* e_new = hist[3];  // slot about to be recycled
    if (n_hist > 0 && fused) {
      // fused iteration: h->xa already holds the in-projection of x (left by the previous iteration); the tail projects the skip
      // sum to eps, stores it to the history slot, applies the multistep update to x and projects the new x for the next one
      if (n_sub == 0) n_sub = dual_fork(h, B, T, st, subs);   // the first fused iteration forks the two half-batch chains
      for (int u = 0; u < n_sub && rc == BSG_OK; ++u) {
        const size_t mo = (size_t)subs[u].off * h->M * T;
        h->row_off = subs[u].off;
        TailArgs a{};
        a.plms_hist = n_hist; a.pk = c; a.e_new = e_new + mo; a.h1 = hist[0] + mo; a.h2 = hist[1] + mo; a.h3 = hist[2] + mo;
        if (n_hist == 1) { a.pk.w0 = 3.f; a.pk.inv = 2.f; }
        else if (n_hist == 2) { a.pk.w0 = 23.f; a.pk.w1 = -16.f; a.pk.w2 = 5.f; a.pk.inv = 12.f; }
        else { a.pk.w0 = 55.f; a.pk.w1 = -59.f; a.pk.w2 = 37.f; a.pk.w3 = -9.f; a.pk.inv = 24.f; }
        a.do_head = i - interval >= 0;
        rc = step_from_xa(h, i, a, x + mo, subs[u].B, T, subs[u].st);
      }
      if (rc != BSG_OK) break;
      hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = e_new;
      if (n_hist < 3) ++n_hist;
      continue;
    }
    TRY(forward_impl(h, x, nullptr, i, e_new, B, T, st));
    if (n_hist == 0) {
      c.inv = 1.f;
      hipLaunchKernelGGL(plms_step_kernel, grid, block, 0, st, (const float*)x, h->xpred, (const float*)e_new,
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, c, (long long)n);
      TRY(forward_impl(h, h->xpred, nullptr, ip, h->eps, B, T, st));
      c.w0 = 1.f; c.inv = 2.f;
      hipLaunchKernelGGL(plms_step_kernel, grid, block, 0, st, (const float*)x, x, (const float*)e_new, (const float*)h->eps,
                         (const float*)nullptr, (const float*)nullptr, c, (long long)n);
    } else if (n_hist == 1) {
      c.w0 = 3.f; c.inv = 2.f;
      hipLaunchKernelGGL(plms_step_kernel, grid, block, 0, st, (const float*)x, x, (const float*)e_new, (const float*)hist[0],
                         (const float*)nullptr, (const float*)nullptr, c, (long long)n);
    } else if (n_hist == 2) {
      c.w0 = 23.f; c.w1 = -16.f; c.w2 = 5.f; c.inv = 12.f;
      hipLaunchKernelGGL(plms_step_kernel, grid, block, 0, st, (const float*)x, x, (const float*)e_new, (const float*)hist[0],
                         (const float*)hist[1], (const float*)nullptr, c, (long long)n);
    } else {
      c.w0 = 55.f; c.w1 = -59.f; c.w2 = 37.f; c.w3 = -9.f; c.inv = 24.f;
      hipLaunchKernelGGL(plms_step_kernel, grid, block, 0, st, (const float*)x, x, (const float*)e_new, (const float*)hist[0],
                         (const float*)hist[1], (const float*)hist[2], c, (long long)n);
    }
    BSG_LAUNCH_CHECK();
    if (fused && i - interval >= 0) TRY(conv1x1(h->w_in, h->b_in, x, h->xa, C, h->M, B, T, ACT_RELU, st));   // for the fused iterations
    hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = e_new;
    if (n_hist < 3) ++n_hist;
  }
  return dual_join(h, n_sub, st, rc);
}

// One PLMS update given the noise predictions (p_sample_plms :168-201 without the denoiser calls), for denoisers that are not a
// bsg_diffnet (ABI v5): x_out = x + x_delta(eps') with eps' the multistep blend of e0 (newest) and the n_hist <= 3 older predictions
// e1..e3 (NULL beyond n_hist):  n_hist = 0, avg = 0: eps' = e0 (the predictor half-step of the first iteration);  n_hist = 1, avg = 1:
// (e0 + e1) / 2 (its corrector);  n_hist = 1: (3 e0 - e1) / 2;  2: (23 e0 - 16 e1 + 5 e2) / 12;  3: (55 e0 - 59 e1 + 37 e2 - 9 e3) / 24.
extern "C" int bsg_plms_step(const float* x, float* x_out, const float* e0, const float* e1, const float* e2, const float* e3, int32_t n_hist,
                             int32_t avg, const bsg_schedule* s, int32_t t, int32_t t_prev, int64_t n, void* stream) {
  TRY(check_schedule(s, "plms_step", true));
  BSG_REQUIRE(x && x_out && e0 && n > 0 && n_hist >= 0 && n_hist <= 3 && t >= 0 && t < s->num_timesteps && t_prev >= 0 && t_prev <= t,
              "plms_step: bad argument");
  BSG_REQUIRE((n_hist < 1 || e1) && (n_hist < 2 || e2) && (n_hist < 3 || e3), "plms_step: %d history tensors expected", n_hist);
  PlmsCoef c{};
  c.a_t = s->alphas_cumprod[t];
  c.a_prev = s->alphas_cumprod[t_prev];
  c.inv = 1.f;
  if (n_hist == 1 && avg) { c.w0 = 1.f; c.inv = 2.f; }
  else if (n_hist == 1) { c.w0 = 3.f; c.inv = 2.f; }
  else if (n_hist == 2) { c.w0 = 23.f; c.w1 = -16.f; c.w2 = 5.f; c.inv = 12.f; }
  else if (n_hist == 3) { c.w0 = 55.f; c.w1 = -59.f; c.w2 = 37.f; c.w3 = -9.f; c.inv = 24.f; }
  hipLaunchKernelGGL(plms_step_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_out, e0, n_hist >= 1 ? e1 : nullptr,
                     n_hist >= 2 ? e2 : nullptr, n_hist >= 3 ? e3 : nullptr, c, (long long)n);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// ---- per-handle range guard of the split-fp16 GEMMs (ABI v7): one implementation for the five handle kinds
namespace bsg {
Guard* guard_of_diffnet(void* h) { return &static_cast<bsg_diffnet*>(h)->guard; }
}
