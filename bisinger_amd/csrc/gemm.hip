// Generic batched fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bit-for-bit a k-ordered fmaf chain; 256 FLOP/clk/CU = the fp32 roofline of the chip).
//
// Used for every dense contraction of the path that is not the fused DiffNet residual block:
// 1x1 convolutions in [B,C,T] layout (B operand [K,N], N contiguous), Linear layers and attention
// products in [tokens,C] layout (B operand [N,K], K contiguous), and the k=9 / k=3 Conv1d of the
// FFN / duration predictor as K-segmented GEMMs over shifted A rows ("taps").
//
// Two kernels:
//  * gemm_fast_kernel<BM, TRANS_B> — every 16-byte-aligned problem (all of the path's shapes).  Tile BM x 128 x 32 per 256-thread
//    workgroup (BM = 128, or 64 when 128-row tiles would leave CUs without a second workgroup), double-buffered LDS with ONE barrier
//    per k-tile; the next-but-one tile is in flight from global memory while the current one is multiplied.  K-contiguous operands
//    keep their global layout in LDS ([row][32 + 4] floats): staging is a 16-byte load and a 16-byte ds_write per 4 elements, and an
//    MFMA operand fragment is ONE ds_read_b128 per lane = 4 k-steps (lane half lh takes k = 8g + 4 lh + 0..3; A and B use the same
//    map, so the chain is a permutation of the k order — exact fp32 products, different summation order).  The 36-float row stride
//    puts the 16 rows of a b128 lane group on 16 different bank quads.
//  * gemm_f32_kernel<TRANS_B> — round 1's kernel, kept for unaligned operands (K, lda or ldb not a multiple of 4, odd base
//    addresses): 128x128x32, [row][33] images read with ds_read_b32, two barriers per k-tile.
#include <stdlib.h>

#include "bsg_common.h"

namespace bsg {

// range guard of gemm_split_kernel: number of waves that staged an operand whose hi term leaves the fp16 range (bsg_gemm_range_events)
__device__ unsigned g_gemm_range_events = 0;

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDP = BK + 1;

// GELU / Mish of the epilogues as ONE out-of-line function: inlined at each of the 64 accumulator registers of a lane, erff / tanhf /
// log1pf / expf made every GEMM kernel 57-113 KB of code — more than the instruction cache — and each launch paid for streaming it
// (round 4: the same epilogue cost gemm_h2w_kernel ~10 us per workgroup)
__device__ __attribute__((noinline)) float act_slow(float v, int act) {
  if (act == ACT_GELU) return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  return v * tanhf(v > 20.f ? v : log1pf(expf(v)));
}

template <bool TRANS_B>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[BM * LDP];
  __shared__ float Bs[TRANS_B ? BN * LDP : BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN, bz = blockIdx.z;

  const int b2 = g.batch2 > 1 ? g.batch2 : 1;
  const int zo = bz / b2, zi = bz - zo * b2;
  const float* __restrict__ A = g.A + (long long)zo * g.sA + (long long)zi * g.sA2;
  const float* __restrict__ Bp = g.B + (long long)zo * g.sB + (long long)zi * g.sB2;

  const bool vecA = ((g.lda | g.K) & 3) == 0 && ((((uintptr_t)A) & 15) == 0);   // A, Bp include the batch offsets
  const bool vecB = TRANS_B ? (((g.ldb | g.K) & 3) == 0 && ((((uintptr_t)Bp) & 15) == 0) && ((g.sTapB & 3) == 0))
                            : (((g.ldb | g.N) & 3) == 0 && ((((uintptr_t)Bp) & 15) == 0) && ((g.sTapB & 3) == 0));

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kTiles = (g.K + BK - 1) / BK;
  const int nIter = kTiles * g.taps;

  f32x4 ra[4], rb[4];

  auto load_tiles = [&](int it) {
    const int tap = it / kTiles, k0 = (it - tap * kTiles) * BK;
    const int shift = g.tap_shift0 + tap;
    // A tile: 128 rows x 32 k
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx >> 3, c4 = (idx & 7) << 2;
      const int gr = bm + row + shift, gk = k0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gr >= 0 && gr < g.M && (bm + row) < g.M) {
        const float* p = A + (long long)gr * g.lda + gk;
        if (vecA && gk + 3 < g.K) {
          v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gk + e < g.K) v[e] = p[e];
        }
      }
      ra[j] = v;
    }
    const float* Bt = Bp + (long long)tap * g.sTapB;
    if (TRANS_B) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        const int row = idx >> 3, c4 = (idx & 7) << 2;
        const int gn = bn + row, gk = k0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gn < g.N) {
          const float* p = Bt + (long long)gn * g.ldb + gk;
          if (vecB && gk + 3 < g.K) {
            v = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gk + e < g.K) v[e] = p[e];
          }
        }
        rb[j] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        const int row = idx >> 5, c4 = (idx & 31) << 2;
        const int gk = k0 + row, gn = bn + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gk < g.K) {
          const float* p = Bt + (long long)gk * g.ldb + gn;
          if (vecB && gn + 3 < g.N) {
            v = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gn + e < g.N) v[e] = p[e];
          }
        }
        rb[j] = v;
      }
    }
  };

  auto store_tiles = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx >> 3, c4 = (idx & 7) << 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) As[row * LDP + c4 + e] = ra[j][e];
    }
    if (TRANS_B) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        const int row = idx >> 3, c4 = (idx & 7) << 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[row * LDP + c4 + e] = rb[j][e];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        const int row = idx >> 5, c4 = (idx & 31) << 2;
        *reinterpret_cast<f32x4*>(&Bs[row * BN + c4]) = rb[j];
      }
    }
  };

  load_tiles(0);
  for (int it = 0; it < nIter; ++it) {
    __syncthreads();   // previous tile fully consumed
    store_tiles();
    __syncthreads();
    if (it + 1 < nIter) load_tiles(it + 1);   // in flight while the MFMAs below run
    const float* a0p = &As[(wm * 64 + l31) * LDP + lh];
    const float* a1p = a0p + 32 * LDP;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float a0 = a0p[2 * s], a1 = a1p[2 * s];
      float b0, b1;
      if (TRANS_B) {
        b0 = Bs[(wn * 64 + l31) * LDP + 2 * s + lh];
        b1 = Bs[(wn * 64 + 32 + l31) * LDP + 2 * s + lh];
      } else {
        b0 = Bs[(2 * s + lh) * BN + wn * 64 + l31];
        b1 = Bs[(2 * s + lh) * BN + wn * 64 + 32 + l31];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // epilogue: lanes 0..31 of a register hold 32 consecutive columns of one row -> 128-B stores
  float* __restrict__ C = g.C + (long long)zo * g.sC + (long long)zi * g.sC2;
  const float* __restrict__ R = g.R ? g.R + (long long)zo * g.sR : nullptr;
  const float* __restrict__ RS = g.rowscale ? g.rowscale + (long long)zo * g.sRS : nullptr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = bn + wn * 64 + ni * 32 + l31;
      const float bn_v = (g.bias_n && col < g.N) ? g.bias_n[(long long)zo * g.sBiasN + col] : 0.f;
      const float ps_v = (g.post_scale_n && col < g.N) ? g.post_scale_n[col] : 1.f;
      const float pb_v = (g.post_scale_n && col < g.N) ? g.post_shift_n[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 64 + mi * 32 + acc_row(r, lh);
        if (row < g.M && col < g.N) {
          float v = acc[mi][ni][r] + bn_v;
          if (g.bias_m) v += g.bias_m[row];
          if (g.alpha_ncols == 0 || col < g.alpha_ncols) v *= g.alpha;
          if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (g.act >= ACT_GELU) v = act_slow(v, g.act);
          if (g.post_scale_n) v = v * ps_v + pb_v;
          if (R) v += R[(long long)row * g.ldr + col];
          if (RS) v *= RS[row];
          C[(long long)row * g.ldc + col] = v;
        }
      }
    }
}


// ------------------------------------------------------------------------------------------------
// gemm_fast_kernel
// ------------------------------------------------------------------------------------------------
constexpr int FBN = 128;
constexpr int FLDN = FBN + 8;                       // !TRANS_B image [k][n]: the two lane halves (k and k + 4) land 32 banks apart

template <int BM, int FBK, bool TRANS_B>   // FBK = 16 or 32: k extent of an LDS stage
__global__ __launch_bounds__(256) void gemm_fast_kernel(GemmArgs g) {
  constexpr int FLD = FBK + 4;   // row stride 20 / 36 floats: the 16 rows of a b128 lane group fall on 16 different bank quads
  constexpr int A_FLOATS = BM * FLD;
  constexpr int B_FLOATS = TRANS_B ? FBN * FLD : FBK * FLDN;
  constexpr int NI = BM == 128 ? 2 : 1;             // column tiles of 32 per wave (2 row tiles always)
  constexpr int A_LD4 = BM * FBK / 4 / 256;         // float4 loads per thread and k-tile
  constexpr int B_LD4 = FBN * FBK / 4 / 256;
  constexpr int KQ = FBK / 4, KQS = FBK == 32 ? 3 : 2;   // float4 per row of a K-contiguous tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][BM][FLD]
  float* Bs = lds + 2 * A_FLOATS;       // [2][FBN][FLD] or [2][FBK][FLDN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BM == 128 ? wave >> 1 : 0, wn = BM == 128 ? wave & 1 : wave;
  const int l31 = lane & 31, lh = lane >> 5;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * FBN, bz = blockIdx.z;
  const int b2 = g.batch2 > 1 ? g.batch2 : 1;
  const int zo = bz / b2, zi = bz - zo * b2;
  const float* __restrict__ A = g.A + (long long)zo * g.sA + (long long)zi * g.sA2;
  const float* __restrict__ Bp = g.B + (long long)zo * g.sB + (long long)zi * g.sB2;

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kTiles = (g.K + FBK - 1) / FBK;
  const int nIter = kTiles * g.taps;
  f32x4 ra[A_LD4], rb[B_LD4];

  auto load_tiles = [&](int it) {
    const int tap = it / kTiles, k0 = (it - tap * kTiles) * FBK;
    const int shift = g.tap_shift0 + tap;
#pragma unroll
    for (int j = 0; j < A_LD4; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx >> KQS, gk = k0 + ((idx & (KQ - 1)) << 2);
      const int gr = bm + row + shift;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gr >= 0 && gr < g.M && (bm + row) < g.M && gk < g.K) v = *reinterpret_cast<const f32x4*>(A + (long long)gr * g.lda + gk);
      ra[j] = v;
    }
    const float* Bt = Bp + (long long)tap * g.sTapB;
    if (TRANS_B) {
#pragma unroll
      for (int j = 0; j < B_LD4; ++j) {
        const int idx = tid + 256 * j;
        const int gn = bn + (idx >> KQS), gk = k0 + ((idx & (KQ - 1)) << 2);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gn < g.N && gk < g.K) v = *reinterpret_cast<const f32x4*>(Bt + (long long)gn * g.ldb + gk);
        rb[j] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < B_LD4; ++j) {
        const int idx = tid + 256 * j;
        const int gk = k0 + (idx >> 5), gn = bn + ((idx & 31) << 2);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gk < g.K && gn < g.N) v = *reinterpret_cast<const f32x4*>(Bt + (long long)gk * g.ldb + gn);   // N % 4 == 0: all in or all out
        rb[j] = v;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float* as = As + buf * A_FLOATS;
    float* bs = Bs + buf * B_FLOATS;
#pragma unroll
    for (int j = 0; j < A_LD4; ++j) {
      const int idx = tid + 256 * j;
      *reinterpret_cast<f32x4*>(as + (idx >> KQS) * FLD + ((idx & (KQ - 1)) << 2)) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_LD4; ++j) {
      const int idx = tid + 256 * j;
      if (TRANS_B) *reinterpret_cast<f32x4*>(bs + (idx >> KQS) * FLD + ((idx & (KQ - 1)) << 2)) = rb[j];
      else *reinterpret_cast<f32x4*>(bs + (idx >> 5) * FLDN + ((idx & 31) << 2)) = rb[j];
    }
  };

  load_tiles(0);
  store_tiles(0);
  if (nIter > 1) load_tiles(1);
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < nIter; ++it) {
    const int cur = it & 1;
    // tile it+1 (in registers since the previous iteration) goes to the other buffer — every wave finished reading that buffer
    // before the barrier that ended the previous iteration — and tile it+2 is requested; both overlap the MFMAs below
    if (it + 1 < nIter) store_tiles(cur ^ 1);
    if (it + 2 < nIter) load_tiles(it + 2);
    const float* as = As + cur * A_FLOATS + (wm * 64 + l31) * FLD + 4 * lh;
    const float* bs = TRANS_B ? Bs + cur * B_FLOATS + (wn * 32 * NI + l31) * FLD + 4 * lh
                              : Bs + cur * B_FLOATS + (4 * lh) * FLDN + wn * 32 * NI + l31;
#pragma unroll
    for (int g8 = 0; g8 < FBK / 8; ++g8) {
      f32x4 a[2], b[NI];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(as + mi * 32 * FLD + 8 * g8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (TRANS_B) {
          b[ni] = *reinterpret_cast<const f32x4*>(bs + ni * 32 * FLD + 8 * g8);
        } else {
          const float* p = bs + (8 * g8) * FLDN + ni * 32;
          b[ni] = f32x4{p[0], p[FLDN], p[2 * FLDN], p[3 * FLDN]};
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue (as gemm_f32_kernel): lanes 0..31 of a register hold 32 consecutive columns of one row -> 128-B stores
  float* __restrict__ C = g.C + (long long)zo * g.sC + (long long)zi * g.sC2;
  const float* __restrict__ R = g.R ? g.R + (long long)zo * g.sR : nullptr;
  const float* __restrict__ RS = g.rowscale ? g.rowscale + (long long)zo * g.sRS : nullptr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = bn + wn * 32 * NI + ni * 32 + l31;
      const float bn_v = (g.bias_n && col < g.N) ? g.bias_n[(long long)zo * g.sBiasN + col] : 0.f;
      const float ps_v = (g.post_scale_n && col < g.N) ? g.post_scale_n[col] : 1.f;
      const float pb_v = (g.post_scale_n && col < g.N) ? g.post_shift_n[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 64 + mi * 32 + acc_row(r, lh);
        if (row < g.M && col < g.N) {
          float v = acc[mi][ni][r] + bn_v;
          if (g.bias_m) v += g.bias_m[row];
          if (g.alpha_ncols == 0 || col < g.alpha_ncols) v *= g.alpha;
          if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (g.act >= ACT_GELU) v = act_slow(v, g.act);
          if (g.post_scale_n) v = v * ps_v + pb_v;
          if (R) v += R[(long long)row * g.ldr + col];
          if (RS) v *= RS[row];
          C[(long long)row * g.ldc + col] = v;
        }
      }
    }
}


// ------------------------------------------------------------------------------------------------
// gemm_split_kernel: the same problems on the 16-bit matrix pipe, fp32-grade.  Both operands are split EXACTLY into hi + lo fp16 terms
// while they are staged (a = ah + al to one fp32 ulp; see diffnet_h2.hip for the argument), and every fp32 product is formed as
// ah bh + ah bl + al bh by three v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3/16 of the matrix cycles of the fp32 MFMA form.
// Operands are scaled by 2^4 on the way in (products by 2^8, removed from the accumulator: exact) so that the lo terms of values down
// to 2^-6 are normal fp16 numbers; below that they carry an absolute error <= 2^-29.  |operand| must stay below 4062 (the guard trips at 65000 / 16).
// Tile BM x 128 x 16 per 256-thread workgroup, double-buffered LDS: per stage and operand two planes of [rows][16 fp16 + 8 pad] (48-byte
// rows: the 16 rows of a ds_read_b128 lane group fall on 16 different bank quads); an MFMA fragment = 8 consecutive k = one 16-byte
// read.  !TRANS_B ([K][N] operand, N contiguous): a thread gathers 8 consecutive k of one column with 8 coalesced dword loads.
// ------------------------------------------------------------------------------------------------
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
constexpr int SBK = 16, SROW = 48;        // k extent of a stage; bytes per LDS row of a plane
constexpr float SPLIT_IN = 16.0f, SPLIT_OUT = 1.0f / 256.0f;

__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo, bool& bad) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e] * SPLIT_IN;
    bad |= !(fabsf(x) < 65000.0f);   // also true for NaN / inf
    hi[e] = (_Float16)x;
    lo[e] = (_Float16)(x - (float)hi[e]);
  }
}

template <int BM, bool TRANS_B>
__global__ __launch_bounds__(256) void gemm_split_kernel(GemmArgs g) {
  constexpr int A_BYTES = BM * SROW, B_BYTES = FBN * SROW;       // one plane of one stage
  constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;               // hi A, lo A, hi B, lo B
  constexpr int NI = BM == 128 ? 2 : 1;
  constexpr int A_LD4 = BM * SBK / 4 / 256;                      // float4 per thread and stage: 2 (BM = 128) or 1
  extern __shared__ __attribute__((aligned(16))) char slds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BM == 128 ? wave >> 1 : 0, wn = BM == 128 ? wave & 1 : wave;
  const int l31 = lane & 31, lh = lane >> 5;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * FBN, bz = blockIdx.z;
  const int b2 = g.batch2 > 1 ? g.batch2 : 1;
  const int zo = bz / b2, zi = bz - zo * b2;
  const float* __restrict__ A = g.A + (long long)zo * g.sA + (long long)zi * g.sA2;
  const float* __restrict__ Bp = g.B + (long long)zo * g.sB + (long long)zi * g.sB2;

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kTiles = (g.K + SBK - 1) / SBK;
  const int nIter = kTiles * g.taps;
  bool bad = false;           // an operand outside the fp16 range was staged (range guard)
  f32x4 ra[A_LD4], rb[2];     // TRANS_B: 2 float4 (4 consecutive k of a row); else rb[0], rb[1] = 8 consecutive k of one column

  auto load_tiles = [&](int it) {
    const int tap = it / kTiles, k0 = (it - tap * kTiles) * SBK;
    const int shift = g.tap_shift0 + tap;
#pragma unroll
    for (int j = 0; j < A_LD4; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx >> 2, gk = k0 + ((idx & 3) << 2);
      const int gr = bm + row + shift;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gr >= 0 && gr < g.M && (bm + row) < g.M && gk < g.K) v = *reinterpret_cast<const f32x4*>(A + (long long)gr * g.lda + gk);
      ra[j] = v;
    }
    const float* Bt = Bp + (long long)tap * g.sTapB;
    if (TRANS_B) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        const int gn = bn + (idx >> 2), gk = k0 + ((idx & 3) << 2);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gn < g.N && gk < g.K) v = *reinterpret_cast<const f32x4*>(Bt + (long long)gn * g.ldb + gk);
        rb[j] = v;
      }
    } else {
      // item = (column n = tid & 127, k block kb = tid >> 7): 8 dword loads down the column; lanes = consecutive n (coalesced)
      const int gn = bn + (tid & 127), kb = k0 + 8 * (tid >> 7);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int gk = kb + e;
        const float v = (gn < g.N && gk < g.K) ? Bt[(long long)gk * g.ldb + gn] : 0.f;
        if (e < 4) rb[0][e] = v; else rb[1][e - 4] = v;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    char* st = slds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < A_LD4; ++j) {
      const int idx = tid + 256 * j;
      f16x4 hi, lo;
      split4(ra[j], hi, lo, bad);
      char* d = st + (idx >> 2) * SROW + ((idx & 3) << 3);
      *reinterpret_cast<f16x4*>(d) = hi;
      *reinterpret_cast<f16x4*>(d + A_BYTES) = lo;
    }
    char* sb = st + 2 * A_BYTES;
    if (TRANS_B) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        f16x4 hi, lo;
        split4(rb[j], hi, lo, bad);
        char* d = sb + (idx >> 2) * SROW + ((idx & 3) << 3);
        *reinterpret_cast<f16x4*>(d) = hi;
        *reinterpret_cast<f16x4*>(d + B_BYTES) = lo;
      }
    } else {
      f16x4 h0, l0, h1, l1;
      split4(rb[0], h0, l0, bad);
      split4(rb[1], h1, l1, bad);
      char* d = sb + (tid & 127) * SROW + ((tid >> 7) << 4);
      *reinterpret_cast<f16x8*>(d) = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      *reinterpret_cast<f16x8*>(d + B_BYTES) = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
  };

  load_tiles(0);
  store_tiles(0);
  if (nIter > 1) load_tiles(1);
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < nIter; ++it) {
    const int cur = it & 1;
    if (it + 1 < nIter) store_tiles(cur ^ 1);
    if (it + 2 < nIter) load_tiles(it + 2);
    const char* as = slds + cur * STAGE + (wm * 64 + l31) * SROW + lh * 16;
    const char* bs = slds + cur * STAGE + 2 * A_BYTES + (wn * 32 * NI + l31) * SROW + lh * 16;
    f16x8 ah[2], al[2], bh[NI], bl[NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const f16x8*>(as + mi * 32 * SROW);
      al[mi] = *reinterpret_cast<const f16x8*>(as + mi * 32 * SROW + A_BYTES);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bh[ni] = *reinterpret_cast<const f16x8*>(bs + ni * 32 * SROW);
      bl[ni] = *reinterpret_cast<const f16x8*>(bs + ni * 32 * SROW + B_BYTES);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
    __syncthreads();
  }
  if (g.range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(g.range_events, 1u);

  float* __restrict__ C = g.C + (long long)zo * g.sC + (long long)zi * g.sC2;
  const float* __restrict__ R = g.R ? g.R + (long long)zo * g.sR : nullptr;
  const float* __restrict__ RS = g.rowscale ? g.rowscale + (long long)zo * g.sRS : nullptr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = bn + wn * 32 * NI + ni * 32 + l31;
      const float bn_v = (g.bias_n && col < g.N) ? g.bias_n[(long long)zo * g.sBiasN + col] : 0.f;
      const float ps_v = (g.post_scale_n && col < g.N) ? g.post_scale_n[col] : 1.f;
      const float pb_v = (g.post_scale_n && col < g.N) ? g.post_shift_n[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 64 + mi * 32 + acc_row(r, lh);
        if (row < g.M && col < g.N) {
          float v = acc[mi][ni][r] * SPLIT_OUT + bn_v;
          if (g.bias_m) v += g.bias_m[row];
          if (g.alpha_ncols == 0 || col < g.alpha_ncols) v *= g.alpha;
          if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (g.act >= ACT_GELU) v = act_slow(v, g.act);
          if (g.post_scale_n) v = v * ps_v + pb_v;
          if (R) v += R[(long long)row * g.ldr + col];
          if (RS) v *= RS[row];
          C[(long long)row * g.ldc + col] = v;
        }
      }
    }
}

template <int BM, bool TRANS_B>
int launch_split(const GemmArgs& g, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (2 * BM * SROW + 2 * FBN * SROW);
  hipLaunchKernelGGL((gemm_split_kernel<BM, TRANS_B>), dim3(cdiv(g.N, FBN), cdiv(g.M, BM), g.batch), dim3(256), lds, st, g);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

template <int BM, int FBK, bool TRANS_B>
int launch_fast(const GemmArgs& g, hipStream_t st) {
  constexpr size_t lds = (size_t)(2 * BM * (FBK + 4) + 2 * (TRANS_B ? FBN * (FBK + 4) : FBK * FLDN)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    BSG_HIP(hipFuncSetAttribute((const void*)gemm_fast_kernel<BM, FBK, TRANS_B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_fast_kernel<BM, FBK, TRANS_B>), dim3(cdiv(g.N, FBN), cdiv(g.M, BM), g.batch), dim3(256), lds, st, g);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

}  // namespace

static int g_split = -1;   // -1: not read yet (environment); set by bsg_gemm_set_split (test hook: AND-ed into every handle's switch)
static thread_local Guard* tl_guard = nullptr;   // the handle whose compute entry this thread is inside (GuardScope)

GuardScope::GuardScope(Guard* g) : prev(tl_guard) { tl_guard = g; }
GuardScope::~GuardScope() { tl_guard = prev; }

int guard_init(Guard* g, hipStream_t st) {
  BSG_HIP(hipMalloc((void**)&g->counter, 4 * sizeof(unsigned)));
  BSG_HIP(hipMemsetAsync(g->counter, 0, 4 * sizeof(unsigned), st));
  g->split = 1;
  return BSG_OK;
}
void guard_free(Guard* g) {
  if (g->counter) (void)hipFree(g->counter);
  g->counter = nullptr;
}
int guard_events(Guard* g, int32_t* events, int reset, hipStream_t st) {
  unsigned v = 0;
  BSG_HIP(hipMemcpyAsync(&v, g->counter, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  BSG_HIP(hipStreamSynchronize(st));
  if (v && reset) {
    BSG_HIP(hipMemsetAsync(g->counter, 0, sizeof(unsigned), st));
    BSG_HIP(hipStreamSynchronize(st));
  }
  *events = (int32_t)v;
  return BSG_OK;
}
int guard_events_async(Guard* g, int32_t* host_word, hipStream_t st) {
  BSG_HIP(hipMemcpyAsync(host_word, g->counter, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  return BSG_OK;
}

unsigned* gemm_range_counter() {
  if (tl_guard && tl_guard->counter) return tl_guard->counter;   // inside a handle's call: that handle's own word
  // the symbol lives once per device: cache its address per device id (the Python wrappers switch devices per call)
  static unsigned* p[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!p[dev] && hipGetSymbolAddress((void**)&p[dev], HIP_SYMBOL(g_gemm_range_events)) != hipSuccess) p[dev] = nullptr;
  return p[dev];
}

bool gemm_split_enabled() {
  if (g_split < 0) { const char* e = getenv("BSG_GEMM_SPLIT"); g_split = e ? atoi(e) : 1; }
  return g_split != 0 && (!tl_guard || tl_guard->split != 0);
}

int launch_gemm(const GemmArgs& g_in, hipStream_t st) {
  GemmArgs g = g_in;
  g.range_events = gemm_range_counter();
  BSG_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0 && g.taps > 0, "gemm: empty problem M=%d N=%d K=%d batch=%d", g.M, g.N, g.K, g.batch);
  BSG_REQUIRE(g.batch <= 65535, "gemm: batch %d > 65535", g.batch);
  // the fast kernel moves 16-byte pieces: every operand row and every batch / tap offset must keep 16-byte alignment
  auto al4 = [](long long v) { return (v & 3) == 0; };
  const bool aligned = al4(g.K) && al4(g.lda) && al4(g.ldb) && al4(g.sA) && al4(g.sA2) && al4(g.sB) && al4(g.sB2) && al4(g.sTapB) &&
                       (g.trans_b || al4(g.N)) && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0;
  if (aligned) {
    // 64-row tiles when 128-row tiles would not give every CU two workgroups (e.g. [16000 x 256] outputs: 250 -> 500 workgroups)
    const long long wg128 = (long long)cdiv(g.N, FBN) * cdiv(g.M, 128) * g.batch;
    const bool small = wg128 < 2 * 256;
    // BSG_GEMM_SPLIT=0 / bsg_gemm_set_split(0): multiply on the fp32 matrix pipe (gemm_fast_kernel) instead of the split-fp16 form
    if (gemm_split_enabled()) {
      const long long wgs = (long long)cdiv(g.N, FBN) * cdiv(g.M, 128) * g.batch;
      const bool sm = wgs < 3 * 256;
      if (g.trans_b) return sm ? launch_split<64, true>(g, st) : launch_split<128, true>(g, st);
      return sm ? launch_split<64, false>(g, st) : launch_split<128, false>(g, st);
    }
    if (g.trans_b) return small ? launch_fast<64, 16, true>(g, st) : launch_fast<128, 16, true>(g, st);
    return small ? launch_fast<64, 16, false>(g, st) : launch_fast<128, 16, false>(g, st);
  }
  dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.batch);
  if (g.trans_b)
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, st, g);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

}  // namespace bsg

extern "C" int bsg_gemm_f32(const float* A, const float* Bm, float* C, const float* bias_m, const float* bias_n,
                            int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t trans_b,
                            int32_t batch, int64_t strideA, int64_t strideB, int64_t strideC, int32_t relu,
                            void* stream) {
  bsg::GemmArgs g{};
  g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.trans_b = trans_b; g.taps = 1; g.tap_shift0 = 0; g.sTapB = 0;
  g.bias_m = bias_m; g.bias_n = bias_n; g.alpha = 1.f; g.act = relu ? bsg::ACT_RELU : bsg::ACT_NONE;
  g.batch = batch;
  return bsg::launch_gemm(g, (hipStream_t)stream);
}

extern "C" int bsg_gemm_set_split(int32_t enable) {
  bsg::g_split = enable ? 1 : 0;
  return BSG_OK;
}

extern "C" int bsg_gemm_range_events(int32_t* events, int32_t reset, void* stream) {
  BSG_REQUIRE(events, "gemm_range_events: null argument");
  hipStream_t st = (hipStream_t)stream;
  unsigned v = 0;
  BSG_HIP(hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(bsg::g_gemm_range_events), sizeof(unsigned), 0, hipMemcpyDeviceToHost, st));
  BSG_HIP(hipStreamSynchronize(st));
  if (v && reset) {
    const unsigned z = 0;
    BSG_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(bsg::g_gemm_range_events), &z, sizeof(unsigned), 0, hipMemcpyHostToDevice, st));
    BSG_HIP(hipStreamSynchronize(st));
  }
  *events = (int32_t)v;
  return BSG_OK;
}

// Non-blocking read of the range-event counter (ABI v5): enqueues a copy of the current device's counter into *host_word (pinned host
// memory, caller-owned) on `stream`; valid once the stream has passed that point.  Nothing is reset.
extern "C" int bsg_gemm_range_events_async(int32_t* host_word, void* stream) {
  BSG_REQUIRE(host_word, "gemm_range_events_async: null argument");
  BSG_HIP(hipMemcpyFromSymbolAsync(host_word, HIP_SYMBOL(bsg::g_gemm_range_events), sizeof(unsigned), 0, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return BSG_OK;
}

// ---- per-handle range guard (ABI v7): every handle kind owns a Guard; these three entries reach it through (kind, handle) --------------
namespace bsg {
Guard* guard_of_diffnet(void* h);
Guard* guard_of_fs2midi(void* h);
Guard* guard_of_hifigan(void* h);
Guard* guard_of_pitchext(void* h);
Guard* guard_of_fftden(void* h);
static Guard* guard_of(int kind, void* h) {
  if (!h) return nullptr;
  switch (kind) {
    case BSG_HANDLE_DIFFNET: return guard_of_diffnet(h);
    case BSG_HANDLE_FS2MIDI: return guard_of_fs2midi(h);
    case BSG_HANDLE_HIFIGAN: return guard_of_hifigan(h);
    case BSG_HANDLE_PITCHEXT: return guard_of_pitchext(h);
    case BSG_HANDLE_FFTDEN: return guard_of_fftden(h);
  }
  return nullptr;
}
}  // namespace bsg

extern "C" int bsg_handle_range_events(int32_t kind, void* handle, int32_t* events, int32_t reset, void* stream) {
  bsg::Guard* g = bsg::guard_of(kind, handle);
  BSG_REQUIRE(g && g->counter && events, "handle_range_events: bad handle (kind %d) or null argument", kind);
  return bsg::guard_events(g, events, reset, (hipStream_t)stream);
}

extern "C" int bsg_handle_range_events_async(int32_t kind, void* handle, int32_t* host_word, void* stream) {
  bsg::Guard* g = bsg::guard_of(kind, handle);
  BSG_REQUIRE(g && g->counter && host_word, "handle_range_events_async: bad handle (kind %d) or null argument", kind);
  return bsg::guard_events_async(g, host_word, (hipStream_t)stream);
}

extern "C" int bsg_handle_set_gemm_split(int32_t kind, void* handle, int32_t enable) {
  bsg::Guard* g = bsg::guard_of(kind, handle);
  BSG_REQUIRE(g, "handle_set_gemm_split: bad handle (kind %d)", kind);
  g->split = enable ? 1 : 0;
  return BSG_OK;
}

extern "C" int bsg_handle_get_gemm_split(int32_t kind, void* handle, int32_t* enabled) {
  bsg::Guard* g = bsg::guard_of(kind, handle);
  BSG_REQUIRE(g && enabled, "handle_get_gemm_split: bad handle (kind %d) or null argument", kind);
  bsg::GuardScope scope(g);
  *enabled = bsg::gemm_split_enabled() ? 1 : 0;   // the handle's switch AND the process-wide one (BSG_GEMM_SPLIT / bsg_gemm_set_split)
  return BSG_OK;
}
