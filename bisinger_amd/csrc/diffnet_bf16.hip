// bf16-operand form of the fused DiffNet residual block (BASELINE config "B=64, T=1000, bf16").
//
// Same contract and the same HBM tensors as residual_layer_kernel (diffnet.hip; reference semantics
// /root/reference/train_bisinger/usr/diff/net.py:66-78): x, the hoisted conditioner term and the running skip
// residual stream x stays fp32 in HBM (it is the accumulator of the 20-layer chain); bf16 are every MFMA OPERAND — the
// packed weights (rounded once at create), the staged x + d tile, the gated z tile — and the two largest HBM streams: the
// hoisted conditioner term (rounded once per utterance) and the running skip sum, both in channel-quad order
// [rows/4][T][4] so that a lane's 4 consecutive accumulator rows are one 8-byte access.  fp32 accumulation
// (v_mfma_f32_32x32x16_bf16), fp32 gate math, fp32 epilogue.  Measured (round 1, B=64): the layer runs at the fabric's
// byte rate, not the matrix pipe's — every byte removed from HBM traffic is time.
//
// With 16x the fp32 MFMA rate the layer is no longer bound by the matrix pipe but by bytes: 6 KB / frame of HBM
// tensors and the weight stream every workgroup pulls from L2 (1 MB of bf16 per workgroup).  Hence 64-frame
// tiles (half the weight bytes per frame of the 32-frame fp32 tile) at 2 workgroups per CU:
//
//   stage  xs[f][c] = bf16(x[c][t0-8+f] + d[c])  (zero outside [0,T))  -> LDS, CHANNELS-LAST [80][256] bf16,
//          528-B rows.  The MFMA B operand wants 8 consecutive k (channels) per lane, and a tap shift is then a
//          row offset; lanes = consecutive frames, so global loads are 256-B coalesced dword rows and the
//          16-B LDS stores are conflict-free (row stride = 4 banks mod 32).
//   GEMM1  y[2C x 64] = W_dil * im2col(xs): 48 k-steps of 16 channels (3 taps x 16), per wave 2 row tiles
//          (gate 32w.., filter 256+32w..) x 2 column tiles; A fragments stream from L2 in fragment order
//          (16 B / lane), B fragments are one ds_read_b128 per column tile (conflict-free: consecutive frames
//          are 4 banks apart, mod 64).
//   gate   z = sigmoid(y_g + cond_g) * tanh(y_f + cond_f) in registers -> bf16 -> LDS [64][256] (own region)
//   GEMM2  o = W_out * z, 16 k-steps; residual rows start from x + b_out, skip rows from b_out
//   out    x_out = (x + res)/sqrt(2), skip += o_skip   (fp32, 128-B coalesced rows as in the fp32 kernel)
#include "diffnet_res.h"
#include "diffnet_tail.h"

namespace bsg {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int NT = 64;                    // frames per workgroup
constexpr int ROWB = 2 * C + 16;          // LDS image row: 256 bf16 + 16 B pad = 528 B (132 dwords = 4 mod 64)
constexpr int XROWS = NT + 2 * HALO;      // 80
constexpr int XS_BYTES = XROWS * ROWB;    // 42,240
constexpr int ZS_BYTES = NT * ROWB;       // 33,792
constexpr int NS = 4;                     // A-fragment ring depth (k-steps)
constexpr int KSB = 16 * 1024;            // bytes per k-step slab of a packed weight (16 row tiles x 1 KB)

// out[((ks*(M/32) + rt)*64 + lane)*8 + j] = bf16( W(m = 32*rt + (lane&31), k = 16*ks + 8*(lane>>5) + j) )
// k-step major: the 16 row tiles the 8 waves of a workgroup read in one k-step are one contiguous 16-KB slab, so the
// requests of workgroups that run in step spread over every L2 channel (row-tile major put them 48 KB apart — the same
// few channels for every wave — and ran at a quarter of the L2 rate)
// with W(m,k) at src[m*sm + (k % Kc)*sc + (k / Kc)*st]   (dilated conv: k = tap*C + ci, src [2C][C][3])
__global__ void pack_a_frag_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ out, int M, int K, int Kc,
                                        long long sm, long long sc, long long st) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * K) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  const long long rest = i >> 9;
  const int RT = M / 32;
  const int rt = (int)(rest % RT), ks = (int)(rest / RT);
  const int m = 32 * rt + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
  out[i] = (__bf16)src[(long long)m * sm + (long long)(k % Kc) * sc + (long long)(k / Kc) * st];
}

__device__ __forceinline__ bf16x8 lda8(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  return __builtin_bit_cast(unsigned, bf16x2{(__bf16)lo, (__bf16)hi});
}

#define BSG_MFMA_BF(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC, 0, 0, 0)

// k-step pipeline over two row tiles x two column tiles.  A fragments (global, L2) live in a ring of NS k-steps and are
// refilled right after use, i.e. NS-1 k-steps (4 MFMAs = 128 matrix-pipe cycles each, x4 waves per SIMD) ahead; the B
// fragments (LDS) of the next k-step are read while the current one multiplies.  sched_barrier pins that order.
template <typename LDB>
__device__ __forceinline__ void mfma_pipe_bf(f32x16& c00, f32x16& c10, f32x16& c01, f32x16& c11, bf16x8 (&A0)[NS], bf16x8 (&A1)[NS],
                                             rsrc_t rs, int vfrag, int sa0, int sa1, int n_ks, LDB ldb) {
  bf16x8 B0[2], B1[2];
  B0[0] = ldb(0, 0);
  B1[0] = ldb(0, 1);
  const int last = n_ks - 1;
#pragma unroll 1
  for (int ks = 0; ks < n_ks; ks += 2 * NS) {
#pragma unroll
    for (int s = 0; s < 2 * NS; ++s) {
      if (ks + s < n_ks) {   // wave-uniform; n_ks is 48 or 16, the unrolled body covers 6 k-steps
        const int kn = ks + s + 1 <= last ? ks + s + 1 : last;
        B0[(s + 1) & 1] = ldb(kn, 0);
        B1[(s + 1) & 1] = ldb(kn, 1);
        __builtin_amdgcn_sched_barrier(0);
        BSG_MFMA_BF(c00, A0[s % NS], B0[s & 1]);
        BSG_MFMA_BF(c10, A1[s % NS], B0[s & 1]);
        BSG_MFMA_BF(c01, A0[s % NS], B1[s & 1]);
        BSG_MFMA_BF(c11, A1[s % NS], B1[s & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int kr = ks + s + NS <= last ? ks + s + NS : last;
        A0[s % NS] = lda8(rs, vfrag, sa0 + kr * KSB);
        A1[s % NS] = lda8(rs, vfrag, sa1 + kr * KSB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

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

template <bool STAMP>   // STAMP: diagnostic build with s_memtime stamps at the phase boundaries (tools/stamp_layer.py)
__global__ __launch_bounds__(512, 4) void residual_layer_bf16_kernel(ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;
  char* zs = lds_raw + XS_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  // XCD-aware tile order: workgroup i runs on XCD i % 8, so give the workgroups of one XCD a contiguous run of tiles —
  // the two halo lines of a tile are its neighbours' core lines, and neighbours that share an L2 fetch them once
  const int per_xcd = (a.B * a.tiles_per_row + 7) >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile_id >= a.B * a.tiles_per_row) return;
  const int b = tile_id / a.tiles_per_row;
  const int t0 = (tile_id - b * a.tiles_per_row) * NT;
  const int T = a.T;
  const int tb = a.t_dev ? (int)a.t_dev[b] : a.t_uniform;

  const unsigned plane = (unsigned)C * T * 4;
  const rsrc_t rs_x = mk_rsrc(a.x_in + (long long)b * C * T, plane);
  const rsrc_t rs_xo = mk_rsrc(a.x_out + (long long)b * C * T, plane);
  const rsrc_t rs_sk = mk_rsrc(a.skip_h + (long long)b * C * T, plane / 2);
  const rsrc_t rs_ct = mk_rsrc(a.condterm_h + (long long)b * 2 * C * T, plane);
  const rsrc_t rs_a1 = mk_rsrc(a.apack1h, 2 * C * 3 * C * 2);
  const rsrc_t rs_a2 = mk_rsrc(a.apack2h, 2 * C * C * 2);
  const rsrc_t rs_bo = mk_rsrc(a.bias_out, 2 * C * 4);
  const float* dp = a.dproj + ((long long)tb * a.L + a.layer) * C;
  const rsrc_t rs_dp = mk_rsrc(dp, C * 4);
  const int rowT = T * 4;
  const int vfrag = lane * 16;

  BSG_STAMP(0);
  if (STAMP && lane == 0) a.stamps[((long long)tile_id * 8 + wave) * 10 + 8] = __builtin_amdgcn_s_memrealtime();
  // ---- (1) the first A fragments fly while the x tile is staged ------------------------------------
  bf16x8 Ag[NS], Af[NS];
  const int sa_g = wave * 1024, sa_f = (8 + wave) * 1024;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    Ag[k] = lda8(rs_a1, vfrag, sa_g + k * KSB);
    Af[k] = lda8(rs_a1, vfrag, sa_f + k * KSB);
  }

  // accumulator columns of this lane: frame t0 + 32*ct + l31
  int vcol[2], vst[2], vq[2], vqs[2];   // vq: byte offset in a channel-quad bf16 plane ((quad lh)*T + frame)*8
  bool col_ok[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = t0 + 32 * ct + l31;
    col_ok[ct] = col < T;
    vcol[ct] = (lh * 4 * T + (col_ok[ct] ? col : T - 1)) * 4;
    vst[ct] = (lh * 4 * T + col) * 4;
    vq[ct] = (lh * T + (col_ok[ct] ? col : T - 1)) * 8;
    vqs[ct] = (lh * T + col) * 8;
  }

  // ---- (2) stage xs[f][c] = bf16(x + d), zero outside [0,T) ---------------------------------------
  u32x2 cq[2][2][4];   // raw conditioner quads [column tile][gate / filter][g]
  {
    // core 64 frames: lane = frame, the wave walks its 4 chunks of 8 channels (wave-uniform rows -> SGPR offsets)
    const int t = t0 + lane;
    const bool ok = t < T;
    const int vt = (ok ? t : 0) * 4;
    float v[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = ldf(rs_x, vt, (8 * (4 * wave + i) + j) * rowT);
    // halo 2 x 8 frames: 16 lanes = 16 halo frames of one 8-channel chunk
    const int hf = tid & 15, hc = tid >> 4;
    const int th = hf < 8 ? t0 - HALO + hf : t0 + NT - 8 + hf;
    const int hrow = hf < 8 ? hf : NT + hf;
    const bool hok = th >= 0 && th < T;
    float hv[8], hd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hv[j] = ldf(rs_x, hok ? ((8 * hc + j) * T + th) * 4 : 0, 0);
      hd[j] = ldf(rs_dp, (8 * hc + j) * 4, 0);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        cq[ct][0][g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_ct, vq[ct], (8 * wave + 2 * g) * T * 8, 0));
        cq[ct][1][g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_ct, vq[ct], (C / 4 + 8 * wave + 2 * g) * T * 8, 0));
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c8 = 4 * wave + i;
      u32x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = v[i][2 * j] + dp[8 * c8 + 2 * j], hi = v[i][2 * j + 1] + dp[8 * c8 + 2 * j + 1];
        w[j] = ok ? pack2(lo, hi) : 0u;
      }
      *reinterpret_cast<u32x4*>(xs + (HALO + lane) * ROWB + c8 * 16) = w;
    }
    u32x4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = hok ? pack2(hv[2 * j] + hd[2 * j], hv[2 * j + 1] + hd[2 * j + 1]) : 0u;
    *reinterpret_cast<u32x4*>(xs + hrow * ROWB + hc * 16) = w;
  }
  // the hoisted conditioner term becomes the GEMM1 accumulators' initial value (registers 4g..4g+3 = channels 32w + 8g + 4*lh + (0..3) =
  // quad 8w + 2g + lh): requested above, with the staging loads, so that all three HBM streams of the tile's first half are in
  // flight at once instead of one after the other (the gate phase used to wait a full HBM round trip for them)
  f32x16 yg0, yf0, yg1, yf1;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    yg0[4 * g] = bf16_lo(cq[0][0][g][0]); yg0[4 * g + 1] = bf16_hi(cq[0][0][g][0]); yg0[4 * g + 2] = bf16_lo(cq[0][0][g][1]); yg0[4 * g + 3] = bf16_hi(cq[0][0][g][1]);
    yf0[4 * g] = bf16_lo(cq[0][1][g][0]); yf0[4 * g + 1] = bf16_hi(cq[0][1][g][0]); yf0[4 * g + 2] = bf16_lo(cq[0][1][g][1]); yf0[4 * g + 3] = bf16_hi(cq[0][1][g][1]);
    yg1[4 * g] = bf16_lo(cq[1][0][g][0]); yg1[4 * g + 1] = bf16_hi(cq[1][0][g][0]); yg1[4 * g + 2] = bf16_lo(cq[1][0][g][1]); yg1[4 * g + 3] = bf16_hi(cq[1][0][g][1]);
    yf1[4 * g] = bf16_lo(cq[1][1][g][0]); yf1[4 * g + 1] = bf16_hi(cq[1][1][g][0]); yf1[4 * g + 2] = bf16_lo(cq[1][1][g][1]); yf1[4 * g + 3] = bf16_hi(cq[1][1][g][1]);
  }
  __syncthreads();
  BSG_STAMP(1);
  BSG_STAMP(2);

  // ---- (3) GEMM1: 48 k-steps (tap-major) ----------------------------------------------------------
  {
    const char* xb = xs + (HALO + l31) * ROWB + lh * 16;
    const int dil = a.dil;
    auto ldb = [&](int ks, int ct) {
      const int tap = ks >> 4, kc = ks & 15;
      return *reinterpret_cast<const bf16x8*>(xb + ((tap - 1) * dil + 32 * ct) * ROWB + kc * 32);
    };
    mfma_pipe_bf(yg0, yf0, yg1, yf1, Ag, Af, rs_a1, vfrag, sa_g, sa_f, 48, ldb);
  }
  BSG_STAMP(3);

  // ---- (4) gate -> zs (the conditioner term is already in the accumulators) -----------------------------
  const int sb_r = wave * 1024, sb_s = (8 + wave) * 1024;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    Ag[k] = lda8(rs_a2, vfrag, sb_r + k * KSB);
    Af[k] = lda8(rs_a2, vfrag, sb_s + k * KSB);
  }

#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const f32x16& yg = ct ? yg1 : yg0;
    const f32x16& yf = ct ? yf1 : yf0;
    float z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = fast_sigmoid(yg[r]) * fast_tanh(yf[r]);
    // registers 4g..4g+3 are channels 32w + 8g + 4*lh + (0..3) of frame 32*ct + l31
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<u32x2*>(zs + (32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2) =
          u32x2{pack2(z[4 * g], z[4 * g + 1]), pack2(z[4 * g + 2], z[4 * g + 3])};
    }
  }
  BSG_STAMP(4);
  // residual rows start from x + b_out, skip rows from b_out.  Measured slower (the kernel sits at its 128-register budget, every
  // prefetch spills): requesting x before the gate math (99.2 vs 96.4 us per launch); requesting the skip quads before GEMM2 (102.5);
  // staging x in accumulator layout and keeping it for the residual, paid for by running GEMM1 per column tile so that its weights
  // stream from L2 twice (118.5 us: the weight stream, not the x re-read, is what the layer waits for).
  f32x16 or0, os0, or1, os1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float br = ldf(rs_bo, lh * 16, (32 * wave + acc_row0(r)) * 4);
    const float bs = ldf(rs_bo, lh * 16, (C + 32 * wave + acc_row0(r)) * 4);
    or0[r] = ldf(rs_x, vcol[0], (32 * wave + acc_row0(r)) * rowT) + br;
    or1[r] = ldf(rs_x, vcol[1], (32 * wave + acc_row0(r)) * rowT) + br;
    os0[r] = bs;
    os1[r] = bs;
  }
  __syncthreads();
  BSG_STAMP(5);

  // ---- (5) GEMM2: 16 k-steps ------------------------------------------------------------------------
  {
    const char* zb = zs + l31 * ROWB + lh * 16;
    auto ldb = [&](int ks, int ct) { return *reinterpret_cast<const bf16x8*>(zb + 32 * ct * ROWB + ks * 32); };
    mfma_pipe_bf(or0, os0, or1, os1, Ag, Af, rs_a2, vfrag, sb_r, sb_s, 16, ldb);
  }
  BSG_STAMP(6);

  // ---- (6) epilogue ---------------------------------------------------------------------------------
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const f32x16& orr = ct ? or1 : or0;
    const f32x16& oss = ct ? os1 : os0;
    u32x2 pq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      pq[g] = a.first ? u32x2{0u, 0u}
                      : __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_sk, vq[ct], (8 * wave + 2 * g) * T * 8, 0));
    if (col_ok[ct]) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stf(orr[r] / 1.41421356237309504880f, rs_xo, vst[ct], (32 * wave + acc_row0(r)) * rowT);   // (x + residual) / sqrt(2), net.py:78
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // running skip sum (/ sqrt(L) last, :126), stored as bf16
        const float s0 = (bf16_lo(pq[g][0]) + oss[4 * g]) / a.skip_div, s1 = (bf16_hi(pq[g][0]) + oss[4 * g + 1]) / a.skip_div;
        const float s2 = (bf16_lo(pq[g][1]) + oss[4 * g + 2]) / a.skip_div, s3 = (bf16_hi(pq[g][1]) + oss[4 * g + 3]) / a.skip_div;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(s0, s1), pack2(s2, s3)},
                                              rs_sk, vqs[ct], (8 * wave + 2 * g) * T * 8, 0);
      }
    }
  }
  BSG_STAMP(7);
  if (STAMP && lane == 0) a.stamps[((long long)tile_id * 8 + wave) * 10 + 9] = __builtin_amdgcn_s_memrealtime();
}
#undef BSG_STAMP

// ------------------------------------------------------------------------------------------------
// All L layers of the bf16-operand configuration in ONE launch, residual stream on chip (the bf16 sibling of
// the fp32 stack launches — same hand-off protocol, described in diffnet.hip).  The per-layer kernel above is bound by bytes and by
// the latency of its weight stream; here
//   * x (fp32) lives in 32 registers per lane in accumulator layout — it IS the initial value of GEMM2's residual rows — and as the
//     bf16 image xs = bf16(x + d_l) in LDS; it is read from HBM once (layer 0) and never written back;
//   * the running skip sum lives in 32 registers per lane in fp32 (the per-layer kernel rounds it to bf16 in HBM after every
//     layer) and is stored once;
//   * the only HBM stream per layer is the conditioner term (bf16, 1 KB per frame), requested into the GEMM1 accumulators while the
//     workgroup waits for its neighbours; neighbours exchange the two 8-frame edges of the new bf16 image (2 x 4 KB per tile);
//   * one workgroup per CU with 256 registers per wave and a weight ring of its own depth (8 k-steps until round 5; 4 since the conditioner
//     term is loaded non-temporally and the 1-MB-per-tile weight stream hits L2: BSG_BF_NSS).
// HBM bytes per frame and layer: 1.25 KB instead of 4.3 KB.  Arithmetic: as the per-layer bf16 kernel, except that the skip sum is
// never rounded to bf16 and the conditioner term is the accumulators' initial value.
// ------------------------------------------------------------------------------------------------
#ifndef BSG_BF_NSS
#define BSG_BF_NSS 4   // (round 5, with the weights staying in L2: 4 k-steps 405 k against 8 k-steps 401 k mel-frames/s, profiles/r05_bf16_variants_ab.log)
#endif
constexpr int NSS = BSG_BF_NSS;   // weight ring of the stack kernel (k-steps)

// i-th executed k-step -> k-step index: GEMM1 (ROT = 16, 48 k-steps, tap-major) starts with the CENTRE tap, whose B operand is
// the tile's own 64 frames, and visits the two outer taps (which read the neighbours' halo frames) afterwards
template <int ROT>
__device__ __forceinline__ int kmap(int i, int n_ks) {
  if (ROT == 0) return i;
  return i < ROT ? i + ROT : (i < 2 * ROT ? i - ROT : i);
}
// `mid()` runs after the first ROT k-steps have been issued (ROT = 0: never): the hand-off with the neighbours sits there, under
// the centre tap's MFMAs; the ring keeps prefetching across it.
template <int ROT, bool FAIRB, typename LDB, typename MID>
__device__ __forceinline__ void mfma_pipe_bf8(f32x16& c00, f32x16& c10, f32x16& c01, f32x16& c11, bf16x8 (&A0)[NSS], bf16x8 (&A1)[NSS],
                                              rsrc_t rs, int vfrag, int sa0, int sa1, int n_ks, LDB ldb, MID mid, int half) {
  bf16x8 B0[2], B1[2];
  B0[0] = ldb(kmap<ROT>(0, n_ks), 0);
  B1[0] = ldb(kmap<ROT>(0, n_ks), 1);
  const int last = n_ks - 1;
#pragma unroll 1
  for (int ks = 0; ks < n_ks; ks += NSS) {
    if (FAIRB) {   // the two waves of a SIMD take turns at issue priority (2048-cycle slices): see f43_gemm1, diffnet_f43.hip
      const unsigned tnow = (unsigned)__builtin_amdgcn_s_memtime();
      if (((tnow >> 11) & 1u) == (unsigned)half) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
    if (ROT > 0 && ks == ROT) {
      mid();
      // the B operand of the next k-step was read before the halo rows arrived: read it again
      B0[0] = ldb(kmap<ROT>(ks, n_ks), 0);
      B1[0] = ldb(kmap<ROT>(ks, n_ks), 1);
    }
#pragma unroll
    for (int s = 0; s < NSS; ++s) {
      const int in = ks + s + 1 <= last ? ks + s + 1 : last;
      // never read across the hand-off: the k-step after the centre tap is re-read above
      B0[(s + 1) & 1] = ldb(kmap<ROT>(in, n_ks), 0);
      B1[(s + 1) & 1] = ldb(kmap<ROT>(in, n_ks), 1);
      // the LDS reads and the ring's reloads inside the MFMA sequence (row tile 0 first: its fragment is reloaded right behind its two MFMAs,
      // row tile 1's one k-step later), as in diffnet_h2.hip
      BSG_MFMA_BF(c00, A0[s], B0[s & 1]);
      BSG_MFMA_BF(c01, A0[s], B1[s & 1]);
      {
        const int ir = ks + s + NSS <= last ? ks + s + NSS : last;
        A0[s] = lda8(rs, vfrag, sa0 + kmap<ROT>(ir, n_ks) * KSB);
        const int sp = (s + NSS - 1) % NSS;
        const int ip = ks + s - 1 + NSS <= last ? ks + s - 1 + NSS : last;
        A1[sp] = lda8(rs, vfrag, sa1 + kmap<ROT>(ip, n_ks) * KSB);
      }
      BSG_MFMA_BF(c10, A1[s], B0[s & 1]);
      BSG_MFMA_BF(c11, A1[s], B1[s & 1]);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <bool FAIRB>
__global__ __launch_bounds__(512, 2) void residual_stack_bf16_kernel(StackArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;              // [80 frames][528 B]: bf16(x + d_l), frames t0-8 .. t0+71
  char* zs = lds_raw + XS_BYTES;   // [64 frames][528 B]: gated activation
  float* dtab = reinterpret_cast<float*>(lds_raw + XS_BYTES + ZS_BYTES);   // [256]: d_{l+1} per channel, fetched a layer ahead
  float* btab = dtab + C;                                                     // [512]: output-projection bias of the current layer

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
  int vcol[2], vst[2], vq[2];
  bool col_ok[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = t0 + 32 * ct + l31;
    col_ok[ct] = col < T;
    vcol[ct] = (lh * 4 * T + (col_ok[ct] ? col : T - 1)) * 4;
    vst[ct] = (lh * 4 * T + col) * 4;
    vq[ct] = (lh * T + (col_ok[ct] ? col : T - 1)) * 8;
  }
  const int sa_g = wave * 1024, sa_f = (8 + wave) * 1024;   // gate / filter row tile inside a k-step slab
  const int sb_r = wave * 1024, sb_s = (8 + wave) * 1024;   // residual / skip row tile

  float xr[2][16];        // x, accumulator layout: registers 4g..4g+3 = channels 32w + 8g + 4 lh + (0..3) of frame 32 ct + l31
  float sk[2][16];        // running skip sum (fp32), same layout (skip rows C + 32w + ..)
  f32x16 yg0, yf0, yg1, yf1;   // GEMM1 accumulators; they start from the conditioner term

  // the conditioner term of a layer: requested (16 x 8 B per lane) a phase before it is needed, unpacked into the accumulators at the
  // top of the layer — the unpack is the first use, so no wait for HBM sits between the request and the barriers that follow it
  u32x2 cr[16];
#ifndef BSG_BQ_AUX
#define BSG_BQ_AUX 3   // nt + sc0 on the conditioner term's loads, as in the split-fp16 launches (profiles/r05_cq_aux_ab.log)
#endif
  auto cond_request = [&](int l) {
    const rsrc_t rs_ct = mk_rsrc(p.condterm_h + (long long)l * p.ct_stride + (long long)b * 2 * C * T, plane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      cr[4 * g + 0] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_ct, vq[0], (8 * wave + 2 * g) * T * 8, BSG_BQ_AUX));
      cr[4 * g + 1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_ct, vq[0], (C / 4 + 8 * wave + 2 * g) * T * 8, BSG_BQ_AUX));
      cr[4 * g + 2] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_ct, vq[1], (8 * wave + 2 * g) * T * 8, BSG_BQ_AUX));
      cr[4 * g + 3] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_ct, vq[1], (C / 4 + 8 * wave + 2 * g) * T * 8, BSG_BQ_AUX));
    }
  };
  auto cond_unpack = [&]() {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 g0 = cr[4 * g], f0 = cr[4 * g + 1], g1 = cr[4 * g + 2], f1 = cr[4 * g + 3];
      yg0[4 * g] = bf16_lo(g0[0]); yg0[4 * g + 1] = bf16_hi(g0[0]); yg0[4 * g + 2] = bf16_lo(g0[1]); yg0[4 * g + 3] = bf16_hi(g0[1]);
      yf0[4 * g] = bf16_lo(f0[0]); yf0[4 * g + 1] = bf16_hi(f0[0]); yf0[4 * g + 2] = bf16_lo(f0[1]); yf0[4 * g + 3] = bf16_hi(f0[1]);
      yg1[4 * g] = bf16_lo(g1[0]); yg1[4 * g + 1] = bf16_hi(g1[0]); yg1[4 * g + 2] = bf16_lo(g1[1]); yg1[4 * g + 3] = bf16_hi(g1[1]);
      yf1[4 * g] = bf16_lo(f1[0]); yf1[4 * g + 1] = bf16_hi(f1[0]); yf1[4 * g + 2] = bf16_lo(f1[1]); yf1[4 * g + 3] = bf16_hi(f1[1]);
    }
  };
  // xs core (frames t0 .. t0+63, this wave's 32 channels) = bf16(x + d_l), zero beyond T (the conv pads x + d)
  auto write_core = [&]() {   // d of the layer being prepared is in dtab (written a phase earlier, behind a barrier)
    float dv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = dtab[32 * wave + acc_row(r, lh)];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w2 = u32x2{pack2(xr[ct][4 * g] + dv[4 * g], xr[ct][4 * g + 1] + dv[4 * g + 1]),
                         pack2(xr[ct][4 * g + 2] + dv[4 * g + 2], xr[ct][4 * g + 3] + dv[4 * g + 3])};
        if (!col_ok[ct]) w2 = u32x2{0u, 0u};
        *reinterpret_cast<u32x2*>(xs + (HALO + 32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2) = w2;
      }
  };

  // ---- layer 0: x from HBM (the whole input exists, halo included) ------------------------------------------------------
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
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
    float hv[8], hd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      hv[k] = ldf(rs_x, hok ? ((8 * hc + k) * T + th) * 4 : 0, 0);
      hd[k] = ldf(rs_dp, (8 * hc + k) * 4, 0);
    }
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = hok ? pack2(hv[2 * k] + hd[2 * k], hv[2 * k + 1] + hd[2 * k + 1]) : 0u;
    *reinterpret_cast<u32x4*>(xs + hrow * ROWB + hc * 16) = w;
  }
  if (tid < C) dtab[tid] = p.dproj[((long long)tb * L + 0) * C + tid];
  btab[tid] = p.bias_out[tid];
  cond_request(0);
  __syncthreads();
  write_core();
  // weight ring, shared by both GEMMs.  GEMM1's first k-steps (it starts with the centre tap: kmap) are requested a phase ahead —
  // right behind the previous layer's GEMM2 — so that the L2 latency of the weight stream is never on the layer's critical path
  bf16x8 Ag[NSS], Af[NSS];
  auto prefetch_a1 = [&](int l) {
    const rsrc_t rs = mk_rsrc(p.apack1h + (long long)l * (2 * C * 3 * C), 2 * C * 3 * C * 2);
#pragma unroll
    for (int k = 0; k < NSS; ++k) {
      Ag[k] = lda8(rs, vfrag, sa_g + kmap<16>(k, 48) * KSB);
      Af[k] = lda8(rs, vfrag, sa_f + kmap<16>(k, 48) * KSB);
    }
  };
  prefetch_a1(0);

#define STK_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (p.stamps && tid == 0) p.stamps[((long long)tile_id * L + l) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
  if (p.clk && tile_id == 0 && tid == 0) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int dil = 1 << (l % p.cycle);
    const rsrc_t rs_a1 = mk_rsrc(p.apack1h + (long long)l * (2 * C * 3 * C), 2 * C * 3 * C * 2);
    const rsrc_t rs_a2 = mk_rsrc(p.apack2h + (long long)l * (2 * C * C), 2 * C * C * 2);
    const float dnext = (tid < C && l + 1 < L) ? p.dproj[((long long)tb * L + l + 1) * C + tid] : 0.f;   // lands during GEMM1
    const float bnext = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + tid] : 0.f;
    cond_unpack();
    if (l == 0) __syncthreads();   // layer 0: the staged image (core + halo rows); later layers: barrier (C) below covers the core rows
    STK_STAMP(0);
    // ---- GEMM1: 48 k-steps.  The centre tap (16 k-steps) reads the tile's own frames only, so it runs while the neighbours'
    // edges of this layer are still in flight; the wait for them, and the copy of the halo rows, sit behind it (mid) -------------
    {
      const char* xb = xs + (HALO + l31) * ROWB + lh * 16;
      auto ldb = [&](int ks, int ct) {
        const int tap = ks >> 4, kc = ks & 15;
        return *reinterpret_cast<const bf16x8*>(xb + ((tap - 1) * dil + 32 * ct) * ROWB + kc * 32);
      };
      auto mid = [&]() {
        if (l == 0) return;   // layer 0 staged its halo rows from HBM
        if (tid == 0) {
          const unsigned want = p.fbase + (unsigned)l;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            if (side == 0 ? !has_left : !has_right) continue;
            const unsigned* fl = p.flags + (side == 0 ? tile_id - 1 : tile_id + 1);
            if (p.inject == 1) { atomicAdd(p.status, 1u); continue; }
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
              __builtin_amdgcn_s_sleep(2);
              // ~ seconds: never reached unless a workgroup is not resident.  Once ANY wait of this handle has given up (status != 0: the host
              // repeats the call without hand-offs anyway) the others stop waiting within a thousand polls instead of seconds each
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
          // halo rows of this layer: rows 0..7 = the left neighbour's last 8 frames, rows 72..79 = the right neighbour's first 8
          const int side = tid >> 8, f = (tid >> 5) & 7, c16 = tid & 31;
          const bool have = side == 0 ? has_left : has_right;
          u32x4 v = u32x4{0u, 0u, 0u, 0u};
          if (have) {
            // write-through (sc1) stores, drained before the flag, one workgroup per CU, and EVERY load of the handed-off bytes an
            // sc1 buffer load to registers: the hand-off form that needs no agent-scope acquire (MI355X_MICROARCH.md, "Valid
            // forms") — an acquire would invalidate the CU's L1 (the weight stream's) every layer
            const unsigned short* src = reinterpret_cast<const unsigned short*>(p.hx) +
                                        ((long long)(l & 1) * n_tiles + (side == 0 ? tile_id - 1 : tile_id + 1)) * (2 * 8 * C);
            const rsrc_t rs_h = mk_rsrc(src, 2 * 8 * C * 2);
            v = __builtin_amdgcn_raw_buffer_load_b128(rs_h, (((side == 0 ? 8 : 0) + f) * C + c16 * 8) * 2, 0, 16);   // sc1
          }
          *reinterpret_cast<u32x4*>(xs + ((side ? HALO + NT : 0) + f) * ROWB + c16 * 16) = v;
        }
        __syncthreads();   // (A) halo rows in place
        STK_STAMP(2);
      };
      mfma_pipe_bf8<16, FAIRB>(yg0, yf0, yg1, yf1, Ag, Af, rs_a1, vfrag, sa_g, sa_f, 48, ldb, mid, wave >> 2);
      if (FAIRB) __builtin_amdgcn_s_setprio(0);
    }
    STK_STAMP(3);
    // ---- gate -> zs; GEMM2's first weights fly meanwhile -------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NSS; ++k) {
      Ag[k] = lda8(rs_a2, vfrag, sb_r + k * KSB);
      Af[k] = lda8(rs_a2, vfrag, sb_s + k * KSB);
    }
    if (tid < C) dtab[tid] = dnext;   // read by write_core() behind barrier (B)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const f32x16& yg = ct ? yg1 : yg0;
      const f32x16& yf = ct ? yf1 : yf0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x2 z01 = gate2_scaled(f32x2{yg[4 * g], yg[4 * g + 1]}, f32x2{yf[4 * g], yf[4 * g + 1]}, -1.44269504088896340736f, -2.88539008177792681472f, 15.0f, 1.0f);
        const f32x2 z23 = gate2_scaled(f32x2{yg[4 * g + 2], yg[4 * g + 3]}, f32x2{yf[4 * g + 2], yf[4 * g + 3]}, -1.44269504088896340736f, -2.88539008177792681472f, 15.0f, 1.0f);
        *reinterpret_cast<u32x2*>(zs + (32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2) = u32x2{pack2(z01[0], z01[1]), pack2(z23[0], z23[1])};
      }
    }
    // residual rows start from x + b_out, skip rows from b_out (the accumulators of GEMM1 are free now)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float br = btab[32 * wave + acc_row(r, lh)], bs = btab[C + 32 * wave + acc_row(r, lh)];
      yg0[r] = xr[0][r] + br;
      yg1[r] = xr[1][r] + br;
      yf0[r] = bs;
      yf1[r] = bs;
    }
    __syncthreads();   // (B) zs complete; every wave is done reading xs and this layer's biases
    btab[tid] = bnext;
    STK_STAMP(4);
    // ---- GEMM2: 16 k-steps; yg = residual rows, yf = skip rows -----------------------------------------------------------
    {
      const char* zb = zs + l31 * ROWB + lh * 16;
      auto ldb = [&](int ks, int ct) { return *reinterpret_cast<const bf16x8*>(zb + 32 * ct * ROWB + ks * 32); };
      mfma_pipe_bf8<0, FAIRB>(yg0, yf0, yg1, yf1, Ag, Af, rs_a2, vfrag, sb_r, sb_s, 16, ldb, [] {}, wave >> 2);
      if (FAIRB) __builtin_amdgcn_s_setprio(0);
    }
    if (l + 1 < L) prefetch_a1(l + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xr[0][r] = yg0[r] * 0.70710678118654752440f;   // (x + residual) / sqrt(2), net.py:78, as a product (the IEEE division is ~10 instructions
      xr[1][r] = yg1[r] * 0.70710678118654752440f;   // per element)
      sk[0][r] += yf0[r];
      sk[1][r] += yf1[r];
    }
    STK_STAMP(5);
    if (l + 1 == L) break;

    // ---- next layer: its conditioner term (64 KB per tile, the only big HBM stream) is requested into the free accumulators NOW, so
    // that it lands under the image / publish phase; then the image, the edges for the neighbours, the flag ------------------------
    cond_request(l + 1);
    write_core();
    __syncthreads();   // (C1) the core rows are complete (every wave wrote its 32 channels of every frame)
    STK_STAMP(6);
    {
      // publish the first and the last 8 frames (rows HALO .. HALO+7 and HALO+56 .. HALO+63): 2 x 8 x 512 B = 512 x 16 B, write-through
      unsigned short* hx_t = reinterpret_cast<unsigned short*>(p.hx) + ((long long)((l + 1) & 1) * n_tiles + tile_id) * (2 * 8 * C);
      const int side = tid >> 8, f = (tid >> 5) & 7, c16 = tid & 31;
      const u32x4 v = *reinterpret_cast<const u32x4*>(xs + (HALO + (side ? NT - 8 : 0) + f) * ROWB + c16 * 16);
      if (!(p.inject == 1 && (tile_id & 1))) {
        const rsrc_t rs_hx = mk_rsrc(hx_t, 2 * 8 * C * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_hx, ((side * 8 + f) * C + c16 * 8) * 2, 0, 16);   // sc1
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores (and has its conditioner term)
    __syncthreads();   // (C)
    if (tid == 0) __hip_atomic_store(p.flags + tile_id, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    STK_STAMP(7);
  }
#undef STK_STAMP
  if (tid == 0) stack_epoch_done(p, p.fbase, n_tiles);
  if (p.clk && tile_id == 0 && tid == 0) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); }
  // ---- the skip sum / sqrt(L) (net.py:126): rounded to bf16 ONCE (the per-layer kernel rounds the running sum after every layer) and
  // stored in channel-quad order, the layout the bf16 step tail stages with 8-byte loads -----------------------------------------
  {
    const rsrc_t rs_sk = mk_rsrc(p.skip_h + (long long)b * C * T, plane / 2);
    const float div = sqrtf((float)L);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
      if (col_ok[ct]) {
        const int vqs = (lh * T + t0 + 32 * ct + l31) * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(sk[ct][4 * g] / div, sk[ct][4 * g + 1] / div), pack2(sk[ct][4 * g + 2] / div, sk[ct][4 * g + 3] / div)},
                                                rs_sk, vqs, (8 * wave + 2 * g) * T * 8, 0);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Step tail of the bf16-operand configuration, one 64-frame tile per workgroup (the fp32 tail, diffnet.hip step_tail_kernel, is the
// model; same sampler arithmetic, shared through diffnet_tail.h):
//   h    = relu(W_skip s + b)     s = skip sum / sqrt(L) as bf16 channel quads (written by the last layer)      (net.py:126-128)
//   eps  = W_out h + b                                                                                          (net.py:129)
//   x   <- p_sample(x, eps, noise) or p_sample_plms(x, eps, history), fp32                   (shallow_diffusion_tts.py:149-201)
//   xa   = relu(W_in x + b)       the next evaluation's input projection, fp32 in HBM                           (net.py:116-118)
// The three projections run on v_mfma_f32_32x32x16_bf16 with bf16 operands (weights rounded once at create; s, h and the updated
// x rounded when they are staged) and fp32 accumulation; the sampler state x itself stays fp32.  At B=64, T=1000 the fp32 tail took
// 214 us per step (64 TFLOP/s of fp32 MFMA, 10 % of the configuration's device time); this form is bound by its 2.2 KB per frame.
// ------------------------------------------------------------------------------------------------
constexpr int XIN_ROWB = 2 * 96 + 16;   // updated-x image row: 96 bf16 + 16 B pad = 208 B (52 dwords: conflict-free ds_read_b128)
constexpr int TAIL_LDS = NT * ROWB + NT * XIN_ROWB;

template <bool PLMS>
__global__ __launch_bounds__(512, 4) void step_tail_bf16_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* ss = lds_raw;               // [64 frames][528 B]: s, then h
  char* xin = lds_raw + NT * ROWB;  // [64 frames][208 B]: updated x (rows >= M zero)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / a.tiles_per_row;
  const int t0 = (blockIdx.x - b * a.tiles_per_row) * NT;
  const int T = a.T, M = a.M;
  const rsrc_t rs_ws = mk_rsrc(a.ws_h, C * C * 2);
  const rsrc_t rs_wo = mk_rsrc(a.wo_h, 96 * C * 2);
  const rsrc_t rs_wi = mk_rsrc(a.wi_h, C * 96 * 2);
  const rsrc_t rs_bs = mk_rsrc(a.b_skip, C * 4);
  const rsrc_t rs_bf = mk_rsrc(a.b_fin, 96 * 4);
  const rsrc_t rs_bi = mk_rsrc(a.b_in, C * 4);
  const int vfrag = lane * 16, rowT = T * 4;

  // ---- the skip projection's first weights fly while the skip tile is staged ---------------------------------------------------
  bf16x8 A[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) A[k] = lda8(rs_ws, vfrag, wave * 1024 + k * 8 * 1024);
  {
    const rsrc_t rs_sh = mk_rsrc(a.skip_h + (long long)b * C * T, (unsigned)C * T * 2);
    u32x2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // 64 quads x 64 frames, one 8-byte load per item, lanes = consecutive frames
      const int q = 8 * k + wave, t = t0 + lane;
      v[k] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_sh, t < T ? t * 8 : 0, q * T * 8, 0));
      if (t >= T) v[k] = u32x2{0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<u32x2*>(ss + lane * ROWB + (8 * k + wave) * 8) = v[k];
  }
  f32x16 h0, h1;
#pragma unroll
  for (int r = 0; r < 16; ++r) h0[r] = h1[r] = ldf(rs_bs, lh * 16, (32 * wave + acc_row0(r)) * 4);
  __syncthreads();
  // ---- h = relu(W_skip s + b): 16 k-steps, this wave's 32 rows x 64 frames -------------------------------------------------------
  {
    const char* sb = ss + l31 * ROWB + lh * 16;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 B0 = *reinterpret_cast<const bf16x8*>(sb + ks * 32);
      const bf16x8 B1 = *reinterpret_cast<const bf16x8*>(sb + 32 * ROWB + ks * 32);
      BSG_MFMA_BF(h0, A[ks & 7], B0);
      BSG_MFMA_BF(h1, A[ks & 7], B1);
      if (ks + 8 < 16) A[ks & 7] = lda8(rs_ws, vfrag, wave * 1024 + (ks + 8) * 8 * 1024);
    }
  }
  // the other two projections' weights: requested now, used after the barriers below
  const int rt = wave % 3, ct2 = wave / 3;   // output projection: 3 row tiles x 2 column tiles on waves 0..5
  bf16x8 Ao[8];
  if (wave < 6) {
#pragma unroll
    for (int k = 0; k < 8; ++k) Ao[k] = lda8(rs_wo, vfrag, rt * 1024 + k * 3 * 1024);
  }
  __syncthreads();   // every wave is done reading s
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const f32x16& hh = ct ? h1 : h0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<u32x2*>(ss + (32 * ct + l31) * ROWB + (32 * wave + 8 * g + 4 * lh) * 2) =
          u32x2{pack2(fmaxf(hh[4 * g], 0.f), fmaxf(hh[4 * g + 1], 0.f)), pack2(fmaxf(hh[4 * g + 2], 0.f), fmaxf(hh[4 * g + 3], 0.f))};
  }
  __syncthreads();
  // ---- eps = W_out h + b and the sampler update, fp32, on the 3 row tiles that cover the M mel bins -----------------------------
  if (wave < 6) {
    const int col = t0 + 32 * ct2 + l31;
    const bool col_ok = col < T;
    const int vcol = (lh * 4 * T + (col_ok ? col : T - 1)) * 4, vst = (lh * 4 * T + col) * 4;
    const rsrc_t rs_x = mk_rsrc(a.x + (long long)b * M * T, (unsigned)M * T * 4);
    const rsrc_t rs_n = mk_rsrc(a.noise ? a.noise + (long long)b * M * T : a.x, a.noise ? (unsigned)M * T * 4 : 0u);
    f32x16 e;
    float xv[16], nv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // the lane's row is m0 + 4 lh; rows >= M fall outside the descriptor's range and read as 0, and are never stored
      const int m0 = 32 * rt + acc_row0(r);
      e[r] = ldf(rs_bf, lh * 16, m0 * 4);
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
        const int so = (32 * rt + acc_row0(r)) * rowT;
        h1v[r] = ldf(rs_h1, vcol, so);
        h2v[r] = ldf(rs_h2, vcol, so);   // zero-size descriptors read as 0
        h3v[r] = ldf(rs_h3, vcol, so);
      }
    }
    const char* hb_ = ss + (32 * ct2 + l31) * ROWB + lh * 16;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 B0 = *reinterpret_cast<const bf16x8*>(hb_ + ks * 32);
      BSG_MFMA_BF(e, Ao[ks & 7], B0);
      if (ks + 8 < 16) Ao[ks & 7] = lda8(rs_wo, vfrag, rt * 1024 + (ks + 8) * 3 * 1024);
    }
    const rsrc_t rs_en = mk_rsrc(PLMS ? a.e_new + (long long)b * M * T : a.x, PLMS ? (unsigned)M * T * 4 : 0u);
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * rt + acc_row(r, lh);
      o[r] = 0.f;
      if (m < M) {
        if constexpr (PLMS) {
          o[r] = plms_update(xv[r], e[r], h1v[r], h2v[r], h3v[r], a.plms_hist, a.pk, nullptr);
          if (col_ok) stf(e[r], rs_en, vst, (32 * rt + acc_row0(r)) * rowT);
        } else {
          float nz = nv[r];
          if (!a.noise && a.k.sigma != 0.f)
            nz = philox_normal1(a.seed, a.stream, a.quad_row0 + ((unsigned long long)b * M + m) * T + (col_ok ? col : T - 1));
          float x0 = __fsub_rn(__fmul_rn(a.k.recip, xv[r]), __fmul_rn(a.k.recipm1, e[r]));
          x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
          const float mean = __fadd_rn(__fmul_rn(a.k.pc1, x0), __fmul_rn(a.k.pc2, xv[r]));
          o[r] = __fadd_rn(mean, __fmul_rn(a.k.sigma, nz));
        }
        if (col_ok) stf(o[r], rs_x, vst, (32 * rt + acc_row0(r)) * rowT);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<u32x2*>(xin + (32 * ct2 + l31) * XIN_ROWB + (32 * rt + 8 * g + 4 * lh) * 2) =
          u32x2{pack2(o[4 * g], o[4 * g + 1]), pack2(o[4 * g + 2], o[4 * g + 3])};
  }
  if (!a.do_head) return;
  // ---- next evaluation's input projection: xa = relu(W_in x + b), K = 96 (in_dims zero-padded) --------------------------------
  bf16x8 Ai[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) Ai[k] = lda8(rs_wi, vfrag, wave * 1024 + k * 8 * 1024);
#pragma unroll
  for (int r = 0; r < 16; ++r) h0[r] = h1[r] = ldf(rs_bi, lh * 16, (32 * wave + acc_row0(r)) * 4);
  __syncthreads();
  {
    const char* xb = xin + l31 * XIN_ROWB + lh * 16;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const bf16x8 B0 = *reinterpret_cast<const bf16x8*>(xb + ks * 32);
      const bf16x8 B1 = *reinterpret_cast<const bf16x8*>(xb + 32 * XIN_ROWB + ks * 32);
      BSG_MFMA_BF(h0, Ai[ks], B0);
      BSG_MFMA_BF(h1, Ai[ks], B1);
    }
  }
  const rsrc_t rs_xa = mk_rsrc(a.xa_next + (long long)b * C * T, (unsigned)C * T * 4);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = t0 + 32 * ct + l31;
    if (col < T) {
      const f32x16& hh = ct ? h1 : h0;
      const int vst = (lh * 4 * T + col) * 4;
#pragma unroll
      for (int r = 0; r < 16; ++r) stf(fmaxf(hh[r], 0.f), rs_xa, vst, (32 * wave + acc_row0(r)) * rowT);
    }
  }
}

__global__ void f32_to_quad_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* s = src + ((long long)b * rows + 4 * q) * T + t;
  *reinterpret_cast<u32x2*>(dst + (((long long)b * (rows / 4) + q) * T + t) * 4) = u32x2{pack2(s[0], s[T]), pack2(s[2 * (long long)T], s[3 * (long long)T])};
}
__global__ void quad_bf16_to_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int rows, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const u32x2 v = *reinterpret_cast<const u32x2*>(src + (((long long)b * (rows / 4) + q) * T + t) * 4);
  float* d = dst + ((long long)b * rows + 4 * q) * T + t;
  d[0] = bf16_lo(v[0]); d[T] = bf16_hi(v[0]); d[2 * (long long)T] = bf16_lo(v[1]); d[3 * (long long)T] = bf16_hi(v[1]);
}

}  // namespace

int f32_to_quad_bf16(const float* src, unsigned short* dst, int B, int rows, int T, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_quad_bf16_kernel, dim3(cdiv(T, 256), rows / 4, B), dim3(256), 0, st, src, dst, rows, T);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}
int quad_bf16_to_f32(const unsigned short* src, float* dst, int B, int rows, int T, hipStream_t st) {
  hipLaunchKernelGGL(quad_bf16_to_f32_kernel, dim3(cdiv(T, 256), rows / 4, B), dim3(256), 0, st, src, dst, rows, T);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int pack_a_frag_bf16(const float* src, unsigned short* out, int M, int K, int Kc, long long sm, long long sc, long long stp,
                     hipStream_t st) {
  const long long total = (long long)M * K;
  hipLaunchKernelGGL(pack_a_frag_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, src, reinterpret_cast<__bf16*>(out), M, K,
                     Kc, sm, sc, stp);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int launch_step_tail_bf16(const TailArgs& a_in, hipStream_t st) {
  TailArgs a = a_in;
  a.tiles_per_row = cdiv(a.T, NT);
  const dim3 grid(a.B * a.tiles_per_row), block(512);
  if (a.plms_hist) hipLaunchKernelGGL(step_tail_bf16_kernel<true>, grid, block, TAIL_LDS, st, a);
  else hipLaunchKernelGGL(step_tail_bf16_kernel<false>, grid, block, TAIL_LDS, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int stack_bf16_occupancy() {
  const size_t lds = XS_BYTES + ZS_BYTES + 3 * C * 4;
  int o = 0;
  if (hipFuncSetAttribute((const void*)residual_stack_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)residual_stack_bf16_kernel<true>, 512, lds) != hipSuccess)
    return 0;
  return o;
}

int launch_residual_stack_bf16(const StackArgs& p, hipStream_t st) {
  const size_t lds = XS_BYTES + ZS_BYTES + 3 * C * 4;
  hipLaunchKernelGGL(residual_stack_bf16_kernel<true>, dim3(8 * cdiv(p.n_tiles, 8)), dim3(512), lds, st, p);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// `a.tiles_per_row` is set here: this form tiles an utterance in 64-frame pieces
int launch_residual_layer_bf16(const ResArgs& a_in, hipStream_t st) {
  ResArgs a = a_in;
  a.tiles_per_row = cdiv(a.T, NT);
  const size_t lds = XS_BYTES + ZS_BYTES;
  const int grid = 8 * cdiv(a.B * a.tiles_per_row, 8);   // see the tile order in the kernel
  static bool attr_set = false;
  if (!attr_set) {
    BSG_HIP(hipFuncSetAttribute((const void*)residual_layer_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BSG_HIP(hipFuncSetAttribute((const void*)residual_layer_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (a.stamps) hipLaunchKernelGGL(residual_layer_bf16_kernel<true>, dim3(grid), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(residual_layer_bf16_kernel<false>, dim3(grid), dim3(512), lds, st, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

}  // namespace bsg
