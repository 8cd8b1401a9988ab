// Split-fp16 GEMM with PRE-SPLIT operands (round 4): fp32-grade products on the 16-bit matrix pipe for the dense contractions outside the
// residual stack — FS2's Linear layers and Conv1d FFN (common_layers.py:625-644,706-730), the hoisted conditioner projections (net.py:68).
//
// gemm_split_kernel (gemm.hip) splits BOTH operands into hi + lo fp16 while it stages them: every tile of the activation is split once per
// column tile and per tap that reads it (72 times per element for the k = 9 FFN convolution) and every weight once per row tile, which
// makes that kernel VALU- and LDS-bound at 10-22 % of the fp16 pipe.  Here
//   * the WEIGHTS are split once, at create, into hi / lo fp16 fragments in the order the kernel executes them (h2w_pack_kernel: one
//     16-byte load per lane = an MFMA operand, straight from L2 into registers through a ring of 4 k-steps — the residual stack's form);
//   * the ACTIVATION is split once by whoever produces it (h2w_split_rows_kernel, the LayerNorm / GEMM / attention epilogues) into two
//     fp16 planes [rows][K]; a workgroup stages a 32-deep slice of its (128 + taps - 1)-row window into LDS ONCE and every tap of a
//     convolution reads it at a shifted row (one barrier per slice = per 24 x taps MFMAs and wave instead of one per 12).
// Arithmetic is gemm_split_kernel's: operands x 2^4 (so that the lo terms of values down to 2^-6 are normal fp16 numbers), products
// a b = ah bh + ah bl + al bh exact in the fp32 accumulator, accumulator x 2^-8; |operand| < 4062, counted otherwise
// (bsg_gemm_range_events), never clipped.
// Either operand can be the matrix A (rows of the output): ACT_IS_A = the activation — outputs [token][feature], FS2's layout — or the
// weights — outputs [feature][frame], the [B, C, T] layout of the conditioner term; in both the lanes of a store run along the
// contiguous axis.
#include "diffnet_res.h"

namespace bsg {
namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr float H2W_IN = 16.0f, H2W_OUT = 1.0f / 256.0f;
constexpr int H2W_BK = 32;          // k extent of a slice in the packed weights' step order (two MFMA steps of 16 per tap)
// LDS bytes per activation row and slice: 32 (or 64) fp16 + 16 pad = 80 / 144 B = 20 / 36 dwords: the 16 rows of a b128 lane group cover the 64 banks once
// weight ring, in k-steps of 16: template argument NS = 4 for 128-row activation tiles (12 MFMAs per k-step: 1 536 matrix clocks ahead, an L2 round
// trip is ~1 000), 8 for 64-row tiles (6 per k-step), 16 for 32-row tiles (3 per k-step) — where it divides the number of k-steps
constexpr int H2W_MAXTAPS = 17;

// out[((step * 2 + plane) * WT + wt) * 512 + lane * 8 + j] = plane ? lo : hi of 16 x W(tap, n = 32 wt + (lane & 31), k = 32 slice + 16 kk + 8 (lane >> 5) + j)
// with step = (slice * taps + tap) * 2 + kk — the order gemm_h2w_kernel executes — and W(tap, n, k) at src[tap * ts + n * rs + k * ks]; rows n >= Wn are zero
__global__ void h2w_pack_kernel(const float* __restrict__ src, _Float16* __restrict__ out, int Wn, int WnPad, int K, int taps, long long ts,
                                long long rs, long long ks, unsigned* __restrict__ bad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)WnPad * K * taps) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  const long long rest = i >> 9;
  const int WT = WnPad / 32;
  const int wt = (int)(rest % WT), step = (int)(rest / WT);
  const int kk = step & 1, t = step >> 1, tap = t % taps, slice = t / taps;
  const int n = 32 * wt + (lane & 31), k = H2W_BK * slice + 16 * kk + 8 * (lane >> 5) + j;
  const float v = n < Wn ? src[tap * ts + n * rs + k * ks] * H2W_IN : 0.f;
  if (!(fabsf(v) < 65000.0f)) atomicAdd(bad, 1u);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  const long long base = ((long long)(step * 2) * WT + wt) * 512 + lane * 8 + j;
  out[base] = hi;
  out[base + (long long)WT * 512] = lo;
}

// fp32 [rows][K] (row stride ld) -> hi / lo fp16 planes [rows][K] of 16 x value; 8 values per thread
__global__ void h2w_split_rows_kernel(const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo, long long rows,
                                      int K, long long ld, unsigned* __restrict__ range_events) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one 8-value piece
  const int kp = K >> 3;
  bool bad = false;
  if (i < rows * kp) {
    const long long r = i / kp;
    const int c = (int)(i - r * kp) << 3;
    const float* p = src + r * ld + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (e < 4 ? a[e] : b[e - 4]) * H2W_IN;
      bad |= !(fabsf(x) < 65000.0f);
      h[e] = (_Float16)x;
      l[e] = (_Float16)(x - (float)h[e]);
    }
    *reinterpret_cast<f16x8*>(hi + r * K + c) = h;
    *reinterpret_cast<f16x8*>(lo + r * K + c) = l;
  }
  if (range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(range_events, 1u);
}

// fp32 [B][K][T] (the [B, C, T] layout of the reference's cond) -> planes [B][TR][KP] (TR >= T rows per item, KP >= K columns; rows T.. and
// columns K.. zero) of 16 x lrelu_slope(value): a 32 x 32 tile through LDS per 256-thread workgroup
__global__ __launch_bounds__(256) void h2w_split_transposed_kernel(const float* __restrict__ src, _Float16* __restrict__ hi,
                                                                   _Float16* __restrict__ lo, int K, int KP, int T, int TR, float slope,
                                                                   unsigned* __restrict__ range_events) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, k0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int k = k0 + j, t = t0 + tx;
    const float v = (k < K && t < T) ? src[((long long)b * K + k) * T + t] : 0.f;
    tile[j][tx] = fmaxf(v, v * slope);   // slope in [0, 1]; 1: the identity
  }
  __syncthreads();
  bool bad = false;
  for (int j = ty; j < 32; j += 8) {
    const int t = t0 + j, k = k0 + tx;
    if (t < TR && k < KP) {
      const float x = tile[tx][j] * H2W_IN;
      bad |= !(fabsf(x) < 65000.0f);
      const _Float16 h = (_Float16)x;
      hi[((long long)b * TR + t) * KP + k] = h;
      lo[((long long)b * TR + t) * KP + k] = (_Float16)(x - (float)h);
    }
  }
  if (range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(range_events, 1u);
}

#define BSG_MFMA_W(ACC, W_, F_)                                                                  \
  do {                                                                                           \
    if (ACT_IS_A) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(F_, W_, ACC, 0, 0, 0);            \
    else ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(W_, F_, ACC, 0, 0, 0);                     \
  } while (0)

// Workgroup = 4 waves = (32 MI activation rows) x (128 weight rows); every wave owns ALL MI activation tiles of 32 x ONE weight tile of 32.
// What a k-step moves into registers: the activation fragments through LDS (2 MI KB per wave; the LDS delivers 128 B/clk per CU), the
// weight fragments through the vector L1 (2 KB per wave; 64 B/clk per CU, which the staging loads share) — with the square 64 x 64 wave
// tile of the first version the L1 side ran at its limit for every product that is not a convolution (4 + 2 KB per wave and k-step against
// 4 KB through LDS: QKV projection 87 TFLOP/s; this form: see DESIGN.md).
// k-steps are numbered over (slice, tap, kk); step i reads the activation fragments of LDS rows (tap + tile rows), bytes 32 kk .. of the
// slice's buffer, and ring slot i % 4 of the weights.  At the last step of a slice the next slice's staged registers go to the other LDS
// buffer, the slice after that is requested from global memory, and ONE barrier follows.
// KK = MFMA steps of 16 per slice and tap: 2 (32-deep slices) for convolutions, whose slices are long (2 x taps steps); 4 (64-deep slices)
// for plain products (taps = 1): with 32-deep slices the staging loads of a slice were requested 768 matrix cycles before they are
// written to LDS — about the L2 latency, so every slice ended in a stall — and a barrier stood behind every 24 MFMAs.
template <bool ACT_IS_A, int MI, int KK, int H2W_NS>
__global__ __launch_bounds__(256, 2) void gemm_h2w_kernel(H2wArgs g) {
  constexpr int BMA = 32 * MI;
  constexpr int BK = 16 * KK, H2W_ROWB = 2 * BK + 16, PPR = 2 * KK;   // pieces of 16 bytes per row and plane
  constexpr int NPH = ((BMA + (KK >= 4 ? 0 : H2W_MAXTAPS - 1)) * PPR + 255) / 256;   // 16-byte pieces per thread, plane and slice
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int taps = g.taps;
  const int WR = BMA + taps - 1, PLANE = WR * H2W_ROWB, STAGE = 2 * PLANE;

  // workgroup -> tile, XCD-aware: workgroup i runs on XCD i mod 8; every XCD takes a contiguous run of tiles, weight tiles fastest, so that the
  // workgroups that share an activation window share an L2 (bijective form of the remap: cdna_hip_programming.md, "XCD swizzle")
  const int AT = (g.rows + BMA - 1) / BMA, WT128 = g.Wn / 128;
  const int nwg = (int)gridDim.x;
  int tile;
  {
    const int orig = (int)blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int wt128 = tile % WT128, at = (tile / WT128) % AT, z = tile / (WT128 * AT);
  if (taps > 1 && (WT128 & 7) == 0) {
    // a convolution's packed weights (9.4 MB for the k = 9 FFN) do not fit an XCD's 4-MB L2: in the order above every XCD streams all of
    // them once per 8 activation tiles.  Weight-stationary instead: XCD x takes the weight tiles x, x + 8, ... (1.2 MB each: L2-resident)
    // for every activation tile; what is fetched once per XCD is then the activation (16 MB at B = 16)
    const int j = (int)blockIdx.x >> 3, wg = WT128 >> 3;
    wt128 = (j % wg) * 8 + ((int)blockIdx.x & 7);
    at = (j / wg) % AT;
    z = j / (wg * AT);
  }
  const int zw = z / g.zdiv, za = z - zw * g.zdiv;
  const int arow0 = at * BMA;

  const unsigned act_bytes = (unsigned)((long long)g.rows * g.lda * 2);
  const rsrc_t rs_hi = mk_rsrc(g.act + (long long)za * g.sAct, act_bytes);
  const rsrc_t rs_lo = mk_rsrc(g.act + g.act_plane + (long long)za * g.sAct, act_bytes);   // (pointer arithmetic in halfs)
  const int WTall = g.Wn / 32;
  const int STEPB = 2 * WTall * 1024, PLB = WTall * 1024;
  const int n_slices = g.K / BK;
  const int spl = KK * taps;                   // steps per slice
  const int total = n_slices * spl;
  const rsrc_t rs_w = mk_rsrc(g.wpack + (long long)zw * g.sW, (unsigned)((long long)total * STEPB));
  const int vfrag = lane * 16;
  const int sw0 = (wt128 * 4 + wave) * 1024;

  // staging: piece p = tid + 256 j of a plane -> window row p / PPR, 16-byte piece p % PPR of the slice's 2 BK bytes
  int goff[NPH], loff[NPH];
#pragma unroll
  for (int j = 0; j < NPH; ++j) {
    const int p = tid + 256 * j, row = p / PPR, qd = p % PPR;
    const bool ok = row < WR;
    const int grow = arow0 + row + g.tap_shift0;          // rows outside [0, rows) read as zero: the descriptor's range check (negative offsets wrap)
    goff[j] = ok && grow >= 0 ? (grow * g.lda + qd * 8) * 2 : 0x7ffffff0;
    loff[j] = ok ? row * H2W_ROWB + qd * 16 : -1;
  }
  u32x4 sh[NPH], sl[NPH];
  auto load_stage = [&](int slice) {
#pragma unroll
    for (int j = 0; j < NPH; ++j) {
      sh[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_hi, goff[j], slice * (BK * 2), 0);
      sl[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, goff[j], slice * (BK * 2), 0);
    }
  };
  auto store_stage = [&](int buf) {
    char* st = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < NPH; ++j)
      if (loff[j] >= 0) {
        *reinterpret_cast<u32x4*>(st + loff[j]) = sh[j];
        *reinterpret_cast<u32x4*>(st + PLANE + loff[j]) = sl[j];
      }
  };

  f32x16 c[MI];   // [activation tile]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[mi][r] = 0.f;

  // weight ring: W[s] = {hi, lo} fragment of k-step s (mod 4)
  f16x8 W[H2W_NS][2];
#pragma unroll
  for (int s = 0; s < H2W_NS; ++s) {
    W[s][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vfrag, s * STEPB + sw0, 0));
    W[s][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vfrag, s * STEPB + PLB + sw0, 0));
  }
  load_stage(0);
  store_stage(0);
  if (n_slices > 1) load_stage(1);
  __syncthreads();

  // activation fragments: F[..][2 mi] = hi, [2 mi + 1] = lo of tile mi
  const char* fb = lds + l31 * H2W_ROWB + lh * 16;
  f16x8 F[2][2 * MI];
  auto ldf_ = [&](int off, f16x8 (&Ff)[2 * MI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      Ff[2 * mi] = *reinterpret_cast<const f16x8*>(fb + off + mi * 32 * H2W_ROWB);
      Ff[2 * mi + 1] = *reinterpret_cast<const f16x8*>(fb + off + mi * 32 * H2W_ROWB + PLANE);
    }
  };
  int cur = 0, sis = 0, slice = 0;   // LDS buffer of the running slice, step inside the slice, slice index
  int off = 0;                       // LDS byte offset of the current step's fragments relative to fb
  ldf_(0, F[0]);
#pragma unroll 1
  for (int i = 0; i < total; i += H2W_NS) {
#pragma unroll
    for (int s = 0; s < H2W_NS; ++s) {
      const bool last = sis == spl - 1;          // the slice ends with this step
      // the next step's fragments: the same tap's second half, or the next tap's first (one row down)
      const int noff = (sis % KK) == KK - 1 ? off - 32 * (KK - 1) + H2W_ROWB : off + 32;
      ldf_(last ? off : noff, F[(s + 1) & 1]);   // (at a slice's last step: read again behind the barrier below)
      const f16x8(&Fc)[2 * MI] = F[s & 1];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) BSG_MFMA_W(c[mi], W[s][1], Fc[2 * mi]);   // lo weights x hi activation
      {
        const int so = (i + s + H2W_NS) * STEPB;     // beyond the last step: outside the descriptor, reads zero
        W[s][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vfrag, so + PLB + sw0, 0));
        const int sp = (s + H2W_NS - 1) % H2W_NS;    // the previous step's hi weights
        const int po = (i + s - 1 + H2W_NS) * STEPB;
        W[sp][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vfrag, po + sw0, 0));   // (step 0 loads slot 3 a second time)
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) BSG_MFMA_W(c[mi], W[s][0], Fc[2 * mi]);       // hi x hi
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) BSG_MFMA_W(c[mi], W[s][0], Fc[2 * mi + 1]);   // hi weights x lo activation
      // issue order: the 2 MI fragment reads one behind each of the first MFMAs, the ring's two reloads inside the second group
#pragma unroll
      for (int k = 0; k < 2 * MI; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * MI - 2 * MI - 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (last) {
        // slice boundary: the staged registers (slice + 1) -> the other buffer, slice + 2 requested, one barrier; then the first fragments of
        // the next slice (their LDS latency sits under this step's MFMAs, still in the pipe)
        if (slice + 1 < n_slices) store_stage(cur ^ 1);
        if (slice + 2 < n_slices) load_stage(slice + 2);
        __syncthreads();
        ++slice;
        cur ^= 1;
        sis = 0;
        off = cur * STAGE;
        if (slice < n_slices) ldf_(off, F[(s + 1) & 1]);
      } else {
        ++sis;
        off = noff;
      }
    }
  }

  // ---- epilogue through LDS: the accumulators of the whole workgroup tile go to an fp32 image (the stages are dead), and ONE rolled loop
  // applies the epilogue to 4 consecutive outputs per thread and stores them as 16 bytes.  (Unrolled over the 16 MI accumulator registers
  // of a lane, with erff inline per value, the epilogue alone was 45-90 KB of code — more than the instruction cache — and every workgroup
  // paid ~10 us for it.)
  float* __restrict__ Cp = g.C ? g.C + (long long)z * g.sC : nullptr;
  const float* __restrict__ Rp = g.R ? g.R + (long long)z * g.sR : nullptr;
  const float* __restrict__ RS = g.rowscale ? g.rowscale + (long long)za * g.sRS : nullptr;
  const float* __restrict__ bias = g.bias ? g.bias + (long long)zw * g.sBias : nullptr;
  _Float16* __restrict__ oh = g.out ? reinterpret_cast<_Float16*>(g.out) + (long long)z * g.sO : nullptr;
  constexpr int ER = ACT_IS_A ? BMA : 128, EC = ACT_IS_A ? 128 : BMA, EP = EC + 4;   // image rows x columns (columns = the output's contiguous axis), pitch
  float* et = reinterpret_cast<float*>(lds);
  if (!ACT_IS_A && !g.up_u && !Rp && !RS && g.act_fn == ACT_NONE && (!g.no_direct || g.Cq || g.Ch)) {
    // [feature][frame] output with nothing but bias / alpha in the epilogue (the conditioner hoist: 655 MB of fp32 per pass at B = 16): straight from
    // the accumulators — a register is 32 consecutive frames of one feature row per lane half (two 128-byte runs per store) — without the LDS
    // image, its two barriers and the rolled loop
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[wt128 * 128 + wave * 32 + acc_row(r, lh)] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int arow = arow0 + mi * 32 + l31;
      if (arow >= g.rows) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int wrow = wt128 * 128 + wave * 32 + acc_row(r, lh);
        float v = c[mi][r] * H2W_OUT + bv[r];
        if (g.alpha_ncols == 0 || wrow < g.alpha_ncols) v *= g.alpha;
        if (Cp) __builtin_nontemporal_store(v, &Cp[(long long)wrow * g.ldc + arow]);   // (a result far larger than the caches: keep the operands' lines)
        c[mi][r] = v;
      }
      if (g.Ch) {
        // the same quads rounded to bf16 (round to nearest even, as f32_to_quad_bf16_kernel rounds the fp32 term): 8 bytes per quad
        using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
        using u32x2g = __attribute__((ext_vector_type(2))) unsigned;
        unsigned short* __restrict__ Ch = g.Ch + (long long)z * g.sC;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int wrow = wt128 * 128 + wave * 32 + acc_row(4 * gq, lh);
          const u32x2g pk = u32x2g{__builtin_bit_cast(unsigned, bf16x2{(__bf16)c[mi][4 * gq], (__bf16)c[mi][4 * gq + 1]}),
                                   __builtin_bit_cast(unsigned, bf16x2{(__bf16)c[mi][4 * gq + 2], (__bf16)c[mi][4 * gq + 3]})};
          __builtin_nontemporal_store(pk, reinterpret_cast<u32x2g*>(Ch + ((long long)(wrow >> 2) * g.rows + arow) * 4));
        }
      }
      if (g.Cq) {
        // channel-quad order: the 4 registers of a group are 4 consecutive feature rows of one frame = one 16-byte store, 512 B contiguous per
        // half-wave (the 16-row stack launch loads its conditioner term as such quads, diffnet_h2q.hip)
        float* __restrict__ Cq = g.Cq + (long long)z * g.sC;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int wrow = wt128 * 128 + wave * 32 + acc_row(4 * gq, lh);
          __builtin_nontemporal_store(f32x4{c[mi][4 * gq], c[mi][4 * gq + 1], c[mi][4 * gq + 2], c[mi][4 * gq + 3]},
                                      reinterpret_cast<f32x4*>(Cq + ((long long)(wrow >> 2) * g.rows + arow) * 4));
        }
      }
    }
    return;
  }
  __syncthreads();   // every wave is done reading the stages
  if (!ACT_IS_A && g.up_u) {
    // polyphase output (H2wArgs::up_u = 8): the image as [activation row q][weight row n = 8 co + r] (pitch 132), so that the 4 phases r0 .. r0 + 3
    // of one (co, q) are one 16-byte LDS read and one 16-byte store at C[co][8 q + r0 - p]; a wave's stores cover (q, r0) in address order:
    // 1 KB contiguous per co
    constexpr int UP = 132;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) et[(mi * 32 + l31) * UP + wave * 32 + acc_row(r, lh)] = c[mi][r];
    __syncthreads();
    const int wrow0u = wt128 * 128;
#pragma unroll 1
    for (int it = tid; it < BMA * 32; it += 256) {
      const int cg = it / (2 * BMA), rem = it - cg * (2 * BMA), q = rem >> 1, c4 = cg * 8 + 4 * (rem & 1);
      const int arow = arow0 + q;
      if (arow >= g.rows) continue;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(et + q * UP + c4);
      const int wrow = wrow0u + c4, co = wrow >> 3;
      const float bv = bias ? bias[co] : 0.f;
      const int o0 = arow * 8 + (wrow & 7) - g.up_p;
      float* __restrict__ dst = Cp + (long long)co * g.ldc;
      f32x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = a4[e] * H2W_OUT + bv;
      if (o0 >= 0 && o0 + 3 < g.up_lout && ((((uintptr_t)(dst + o0)) & 15) == 0)) {
        *reinterpret_cast<f32x4*>(dst + o0) = o4;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (o0 + e >= 0 && o0 + e < g.up_lout) dst[o0 + e] = o4[e];
      }
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (ACT_IS_A) et[(mi * 32 + acc_row(r, lh)) * EP + wave * 32 + l31] = c[mi][r];
      else et[(wave * 32 + acc_row(r, lh)) * EP + mi * 32 + l31] = c[mi][r];
    }
  __syncthreads();
  bool bad = false;
  const bool vec = (g.ldc & 3) == 0 && (((uintptr_t)Cp) & 15) == 0;
  const int wrow0 = wt128 * 128;
  if (ACT_IS_A && g.qkv_T && wrow0 >= 2 * g.qkv_H) {
    // V tile of the QKV projection -> V^T planes (what qkv_split_kernel did in a launch of its own): item = 4 consecutive keys of one (head, d)
    // column; lanes run along the keys (8-byte stores, contiguous within a V^T row)
    using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
    const int T = g.qkv_T, Tp = g.qkv_Tp;
    _Float16* __restrict__ vth = reinterpret_cast<_Float16*>(g.vt);
    constexpr int NRG = BMA / 4;
    auto stored = [](int t) { const int g4 = (t >> 2) & 3; return (t & ~15) + ((((g4 & 1) << 1) | (g4 >> 1)) << 2) + (t & 3); };   // fragment order inside 16 keys
#pragma unroll 1
    for (int it = tid; it < 128 * NRG; it += 256) {
      const int cc = it / NRG, rg = it - cc * NRG;
      const int row0 = arow0 + 4 * rg;
      if (row0 >= g.rows) continue;
      const int d = wrow0 + cc - 2 * g.qkv_H;
      const float bv = bias ? bias[wrow0 + cc] : 0.f;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = et[(4 * rg + e) * EP + cc] * H2W_OUT + bv;
        if (g.alpha_ncols == 0 || wrow0 + cc < g.alpha_ncols) x *= g.alpha;
        v[e] = x * H2W_IN;
        bad |= !(fabsf(v[e]) < 65000.0f) && row0 + e < g.rows;
      }
      const int b0 = row0 / T, t0 = row0 - b0 * T;
      if ((T & 3) == 0 && row0 + 3 < g.rows) {   // the 4 keys belong to one utterance and stay consecutive in the stored order
        f16x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[e] = (_Float16)v[e]; lv[e] = (_Float16)(v[e] - (float)hv[e]); }
        _Float16* dst = vth + ((long long)b0 * g.qkv_H + d) * Tp;
        *reinterpret_cast<f16x4*>(dst + stored(t0)) = hv;
        *reinterpret_cast<f16x4*>(dst + g.vt_plane + stored(t0)) = lv;
        if (t0 + 4 == T) {   // the utterance's last keys: zero the padding T .. Tp - 1
          const f16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          for (int t = T; t < Tp; t += 4) {
            *reinterpret_cast<f16x4*>(dst + stored(t)) = z;
            *reinterpret_cast<f16x4*>(dst + g.vt_plane + stored(t)) = z;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = row0 + e;
          if (row >= g.rows) break;
          const int b = row / T, t = row - b * T;
          _Float16* dst = vth + ((long long)b * g.qkv_H + d) * Tp;
          const _Float16 h = (_Float16)v[e];
          dst[stored(t)] = h;
          dst[g.vt_plane + stored(t)] = (_Float16)(v[e] - (float)h);
          if (t + 1 == T)
            for (int tz = T; tz < Tp; ++tz) { dst[stored(tz)] = (_Float16)0.f; dst[g.vt_plane + stored(tz)] = (_Float16)0.f; }
        }
      }
    }
    if (g.range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(g.range_events, 1u);
    return;
  }
#pragma unroll 1
  for (int it = tid; it < ER * EC / 4; it += 256) {
    const int er = it / (EC / 4), ec = (it - er * (EC / 4)) * 4;
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(et + er * EP + ec);
    const int arow_b = arow0 + (ACT_IS_A ? er : ec), wrow_b = wrow0 + (ACT_IS_A ? ec : er);
    if (arow_b >= g.rows) continue;
    f32x4 o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int arow = arow_b + (ACT_IS_A ? 0 : e), wrow = wrow_b + (ACT_IS_A ? e : 0);
      float v = a4[e] * H2W_OUT;
      if (bias) v += bias[wrow];
      if (g.alpha_ncols == 0 || wrow < g.alpha_ncols) v *= g.alpha;
      if (g.act_fn == ACT_RELU) v = fmaxf(v, 0.f);
      else if (g.act_fn == ACT_GELU) v = v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
      if (arow < g.rows) {
        if (Rp) v += Rp[ACT_IS_A ? (long long)arow * g.ldr + wrow : (long long)wrow * g.ldr + arow];
        if (RS) v *= RS[arow];
      }
      o4[e] = v;
    }
    if (Cp) {
      const long long o = ACT_IS_A ? (long long)arow_b * g.ldc + wrow_b : (long long)wrow_b * g.ldc + arow_b;
      if (vec && (ACT_IS_A || arow_b + 3 < g.rows)) {
        *reinterpret_cast<f32x4*>(Cp + o) = o4;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ACT_IS_A || arow_b + e < g.rows) Cp[o + e] = o4[e];
      }
    }
    if (ACT_IS_A && oh) {   // the result as an operand of the next GEMM: hi / lo planes of 16 v, 8 bytes per plane
      using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
      f16x4 hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = o4[e] * H2W_IN;
        bad |= !(fabsf(x) < 65000.0f);
        hv[e] = (_Float16)x;
        lv[e] = (_Float16)(x - (float)hv[e]);
      }
      const long long oo = (long long)arow_b * g.ldo + wrow_b;
      *reinterpret_cast<f16x4*>(oh + oo) = hv;
      *reinterpret_cast<f16x4*>(oh + g.out_plane + oo) = lv;
    }
  }
  if (g.range_events && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(g.range_events, 1u);
}
#undef BSG_MFMA_W

template <bool ACT_IS_A, int MI, int KK, int NS>
int h2w_launch(const H2wArgs& g, hipStream_t st) {
  constexpr int BMA = 32 * MI;
  // the two stages; the epilogue's fp32 image of the workgroup tile aliases them (pitch = contiguous extent + 4 floats)
  constexpr size_t epi_a = (size_t)BMA * 132 * 4, epi_w = (size_t)128 * (BMA + 4) * 4;   // (the polyphase output of the weights-as-A form uses the [row][132] image too)
  constexpr size_t epi = ACT_IS_A ? epi_a : (epi_a > epi_w ? epi_a : epi_w);
  const size_t stages = (size_t)2 * 2 * (BMA + g.taps - 1) * (32 * KK + 16);
  const size_t lds = stages > epi ? stages : epi;
  const int nwg = cdiv(g.rows, BMA) * (g.Wn / 128) * g.batch;
  static bool attr_set = false;
  if (!attr_set) {
    constexpr size_t smax = (size_t)2 * 2 * (BMA + H2W_MAXTAPS - 1) * (32 * KK + 16);
    BSG_HIP(hipFuncSetAttribute((const void*)gemm_h2w_kernel<ACT_IS_A, MI, KK, NS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(smax > epi ? smax : epi)));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_h2w_kernel<ACT_IS_A, MI, KK, NS>), dim3(nwg), dim3(256), lds, st, g);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

}  // namespace

bool h2w_supports(int rows, int Wn, int K, int taps, int lda) {
  return rows > 0 && Wn > 0 && Wn % 128 == 0 && K >= 64 && K % 64 == 0 && taps >= 1 && taps <= H2W_MAXTAPS && lda % 8 == 0 &&
         (long long)rows * lda * 2 < (1LL << 31) && (long long)(K / 32) * taps * 2 * (Wn / 32) * 2048 < (1LL << 31);
}

int h2w_pack(H2wWeights* w, const float* src, int Wn, int K, int taps, long long ts, long long rs, long long ks, unsigned* bad_dev, hipStream_t st) {
  BSG_REQUIRE(w && src && Wn > 0 && K % 64 == 0 && taps >= 1 && taps <= H2W_MAXTAPS, "h2w_pack: Wn=%d K=%d taps=%d", Wn, K, taps);
  const int WnPad = (Wn + 127) / 128 * 128;
  const long long halfs = (long long)2 * WnPad * K * taps;
  if (!w->pack) BSG_HIP(hipMalloc((void**)&w->pack, (size_t)halfs * sizeof(unsigned short)));
  w->Wn = WnPad; w->K = K; w->taps = taps; w->halfs = halfs;
  const long long n = (long long)WnPad * K * taps;
  hipLaunchKernelGGL(h2w_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, src, reinterpret_cast<_Float16*>(w->pack), Wn, WnPad, K, taps, ts,
                     rs, ks, bad_dev);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int h2w_pack_into(unsigned short* dst, const float* src, int Wn, int K, int taps, long long ts, long long rs, long long ks, unsigned* bad_dev,
                  hipStream_t st) {
  BSG_REQUIRE(dst && src && Wn % 128 == 0 && K % 64 == 0 && taps >= 1 && taps <= H2W_MAXTAPS, "h2w_pack_into: Wn=%d K=%d taps=%d", Wn, K, taps);
  const long long n = (long long)Wn * K * taps;
  hipLaunchKernelGGL(h2w_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, src, reinterpret_cast<_Float16*>(dst), Wn, Wn, K, taps, ts, rs, ks,
                     bad_dev);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

void h2w_free(H2wWeights* w) {
  if (w && w->pack) (void)hipFree(w->pack);
  if (w) { w->pack = nullptr; w->ok = false; }
}

int h2w_split_rows(const float* src, unsigned short* hi, unsigned short* lo, long long rows, int K, long long ld, hipStream_t st) {
  BSG_REQUIRE(src && hi && lo && rows > 0 && K % 8 == 0 && ld % 4 == 0, "h2w_split_rows: rows=%lld K=%d ld=%lld", rows, K, ld);
  hipLaunchKernelGGL(h2w_split_rows_kernel, dim3(cdiv(rows * (K / 8), 256)), dim3(256), 0, st, src, reinterpret_cast<_Float16*>(hi),
                     reinterpret_cast<_Float16*>(lo), rows, K, ld, gemm_range_counter());
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int h2w_split_transposed_lrelu(const float* src, unsigned short* hi, unsigned short* lo, int B, int K, int Kp, int T, int rows_per_b, float slope,
                               hipStream_t st) {
  BSG_REQUIRE(src && hi && lo && B > 0 && K > 0 && Kp >= K && T > 0 && B <= 65535 && rows_per_b >= T && slope >= 0.f && slope <= 1.f,
              "h2w_split_transposed: B=%d K=%d Kp=%d T=%d rows=%d slope=%g", B, K, Kp, T, rows_per_b, (double)slope);
  hipLaunchKernelGGL(h2w_split_transposed_kernel, dim3(cdiv(rows_per_b, 32), cdiv(Kp, 32), B), dim3(256), 0, st, src, reinterpret_cast<_Float16*>(hi),
                     reinterpret_cast<_Float16*>(lo), K, Kp, T, rows_per_b, slope, gemm_range_counter());
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int h2w_split_transposed(const float* src, unsigned short* hi, unsigned short* lo, int B, int K, int T, hipStream_t st) {
  return h2w_split_transposed_lrelu(src, hi, lo, B, K, K, T, T, 1.0f, st);
}

int launch_gemm_h2w(const H2wArgs& g0, hipStream_t st) {
  H2wArgs g = g0;
  BSG_REQUIRE(g.act && g.wpack && (g.C || g.out || g.Ch || g.Cq) && g.batch > 0, "gemm_h2w: null operand");
  BSG_REQUIRE(g.C || !(g.Ch || g.Cq) || (!g.act_is_a && !g.up_u && !g.R && !g.rowscale && g.act_fn == ACT_NONE), "gemm_h2w: a quad-order output without the fp32 row-order one needs the direct-store epilogue");
  BSG_REQUIRE(h2w_supports(g.rows, g.Wn, g.K, g.taps, g.lda), "gemm_h2w: unsupported shape rows=%d Wn=%d K=%d taps=%d lda=%d", g.rows, g.Wn, g.K,
              g.taps, g.lda);
  BSG_REQUIRE(!g.out || g.act_is_a, "gemm_h2w: plane output needs the [token][feature] form");
  BSG_REQUIRE(g.act_fn == ACT_NONE || g.act_fn == ACT_RELU || g.act_fn == ACT_GELU, "gemm_h2w: activation %d not built", g.act_fn);
  BSG_REQUIRE(!g.out || (g.ldo % 4 == 0 && g.out_plane % 4 == 0 && g.sO % 4 == 0), "gemm_h2w: plane output needs 8-byte aligned rows");
  BSG_REQUIRE(g.up_u == 0 || (g.up_u == 8 && !g.act_is_a && g.C && g.up_p % 4 == 0 && g.up_lout > 0 && g.act_fn == ACT_NONE && !g.R && !g.rowscale &&
                              g.alpha == 1.f && g.sBias == 0),
              "gemm_h2w: polyphase output: up_u=%d act_is_a=%d", g.up_u, g.act_is_a);
  BSG_REQUIRE(g.qkv_T == 0 || (g.act_is_a && g.out && g.vt && g.batch == 1 && g.qkv_H % 128 == 0 && g.Wn == 3 * g.qkv_H && g.ldo == 2 * g.qkv_H &&
                               g.qkv_Tp % 32 == 0 && g.qkv_Tp >= g.qkv_T && g.rows % g.qkv_T == 0 && !g.C && g.act_fn == ACT_NONE && !g.R && !g.rowscale),
              "gemm_h2w: QKV output: T=%d Tp=%d H=%d Wn=%d", g.qkv_T, g.qkv_Tp, g.qkv_H, g.Wn);
  if (g.zdiv <= 0) g.zdiv = g.batch;
  g.range_events = gemm_range_counter();
  {
    static int direct_env = -1;
    if (direct_env < 0) { const char* e = getenv("BSG_H2W_DIRECT"); direct_env = e ? atoi(e) : 1; }
    g.no_direct = direct_env ? 0 : 1;
  }
  // 64-row activation tiles when 128-row tiles would leave CUs without a second workgroup
  const long long wg128 = (long long)cdiv(g.rows, 128) * (g.Wn / 128) * g.batch;
  const bool small = wg128 < 2 * 256;
  // 32-row tiles when even 64-row tiles leave most CUs without a workgroup (a single utterance: the product is then one workgroup's serial chain
  // of k-steps deep, and half the MFMAs per step shorten that chain)
  static int tiny_env = -1;
  if (tiny_env < 0) { const char* e = getenv("BSG_H2W_TINY"); tiny_env = e ? atoi(e) : 256; }   // (workgroups of 64-row tiles below which 32-row tiles are used; 0: never)
  const bool tiny = g.act_is_a && (long long)cdiv(g.rows, 64) * (g.Wn / 128) * g.batch < tiny_env;
  static int ring_env = -1;   // BSG_H2W_RING=4: the 4-step weight ring at every tile size (rounds 4's first form)
  if (ring_env < 0) { const char* e = getenv("BSG_H2W_RING"); ring_env = e ? atoi(e) : 16; }
  const int steps = g.K / 16 * g.taps;
  const bool r16 = ring_env >= 16 && steps % 16 == 0, r8 = ring_env >= 8 && steps % 8 == 0;
  if (tiny) {
    // plain products of a handful of workgroups: 256-deep slices (K = 256: the whole contraction staged once, no barrier inside the k-loop)
    static int deep_env = -1;
    if (deep_env < 0) { const char* e = getenv("BSG_H2W_DEEP"); deep_env = e ? atoi(e) : 1; }
    if (g.taps == 1 && deep_env && g.K % 256 == 0 && ring_env >= 16) return h2w_launch<true, 1, 16, 16>(g, st);
    if (g.taps == 1) return r16 ? h2w_launch<true, 1, 4, 16>(g, st) : r8 ? h2w_launch<true, 1, 4, 8>(g, st) : h2w_launch<true, 1, 4, 4>(g, st);
    return r16 ? h2w_launch<true, 1, 2, 16>(g, st) : r8 ? h2w_launch<true, 1, 2, 8>(g, st) : h2w_launch<true, 1, 2, 4>(g, st);
  }
  if (g.taps == 1) {   // plain products: 64-deep slices
    if (g.act_is_a) return !small ? h2w_launch<true, 4, 4, 4>(g, st) : r8 ? h2w_launch<true, 2, 4, 8>(g, st) : h2w_launch<true, 2, 4, 4>(g, st);
    return !small ? h2w_launch<false, 4, 4, 4>(g, st) : r8 ? h2w_launch<false, 2, 4, 8>(g, st) : h2w_launch<false, 2, 4, 4>(g, st);
  }
  if (g.act_is_a) return !small ? h2w_launch<true, 4, 2, 4>(g, st) : r8 ? h2w_launch<true, 2, 2, 8>(g, st) : h2w_launch<true, 2, 2, 4>(g, st);
  return !small ? h2w_launch<false, 4, 2, 4>(g, st) : r8 ? h2w_launch<false, 2, 2, 8>(g, st) : h2w_launch<false, 2, 2, 4>(g, st);
}

}  // namespace bsg

// unit-test / micro-benchmark hook (include/bisinger_hip.h): fp32 in, fp32 out; packs the weights and splits the activation on every call
extern "C" int bsg_gemm_presplit_f32(const float* act, const float* w, float* out, const float* bias, int32_t rows, int32_t Wn, int32_t K,
                                     int32_t taps, int32_t act_is_a, int32_t batch, int32_t relu, int32_t reps, void* stream) {
  using namespace bsg;
  BSG_REQUIRE(act && w && out && batch > 0 && reps > 0, "gemm_presplit_f32: null argument");
  BSG_REQUIRE(h2w_supports(rows, Wn, K, taps, K), "gemm_presplit_f32: unsupported shape rows=%d Wn=%d K=%d taps=%d (Wn %% 128, K %% 64, taps <= 17)", rows, Wn,
              K, taps);
  hipStream_t st = (hipStream_t)stream;
  H2wWeights W{};
  unsigned short* planes = nullptr;
  unsigned* bad = nullptr;
  const long long n_act = (long long)batch * rows * K;
  int rc = BSG_OK;
  if (hipMalloc((void**)&planes, (size_t)2 * n_act * sizeof(unsigned short)) != hipSuccess || hipMalloc((void**)&bad, sizeof(unsigned)) != hipSuccess ||
      hipMemsetAsync(bad, 0, sizeof(unsigned), st) != hipSuccess) {
    set_error("gemm_presplit_f32: out of device memory");
    rc = BSG_ENOMEM;
  }
  if (rc == BSG_OK) rc = h2w_pack(&W, w, Wn, K, taps, (long long)Wn * K, K, 1, bad, st);
  if (rc == BSG_OK) rc = h2w_split_rows(act, planes, planes + n_act, (long long)batch * rows, K, K, st);
  H2wArgs g{};
  g.act = planes; g.act_plane = n_act; g.lda = K; g.sAct = (long long)rows * K; g.wpack = W.pack; g.rows = rows; g.K = K; g.Wn = Wn; g.taps = taps;
  g.tap_shift0 = -(taps / 2); g.act_is_a = act_is_a; g.C = out; g.ldc = act_is_a ? Wn : rows; g.sC = (long long)rows * Wn; g.bias = bias; g.alpha = 1.f;
  g.act_fn = relu ? ACT_RELU : ACT_NONE; g.batch = batch;
  for (int i = 0; i < reps && rc == BSG_OK; ++i) rc = launch_gemm_h2w(g, st);
  unsigned nbad = 0;
  if (rc == BSG_OK && (hipMemcpyAsync(&nbad, bad, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
    set_error("gemm_presplit_f32: device error");
    rc = BSG_EHIP;
  } else {
    (void)hipStreamSynchronize(st);
  }
  h2w_free(&W);
  if (planes) (void)hipFree(planes);
  if (bad) (void)hipFree(bad);
  if (rc == BSG_OK && nbad) { set_error("gemm_presplit_f32: %u weights beyond the fp16 range of the split", nbad); rc = BSG_EINVAL; }
  return rc;
}
