// Fused DiffNet residual block with GEMM1 as Winograd F(4,3) over the dilated taps (fp32).
//
// Same contract and tensors as residual_layer_kernel (diffnet.hip; reference semantics
// /root/reference/train_bisinger/usr/diff/net.py:66-78).  The k=3 dilated conv of a QUAD of output frames (t, t+d, t+2d, t+3d)
// needs 6 products per input channel instead of 12 (direct) or 8 (two F(2,3) pairs): GEMM1 is 6 GEMMs of K = 256 over 16 quads
// = 393 k MACs per frame-row-tile instead of 524 k, and the whole layer issues 5/6 of the F(2,3) form's MFMA work (62.5 % of the
// direct form's).  The fp32 layer is bound by the matrix pipe, so MFMAs not issued are time.
//
//   d_i = x[t + (i-1) d], i = 0..5 (x = the staged, zero-padded x + d_l tile), g = the 3 taps
//   V = B^T d:  V0 = 4 d0 - 5 d2 + d4          U = G g:  U0 = g0 / 4
//               V1 = (d4 - 4 d2) + (d3 - 4 d1)           U1 = -(g0 + g1 + g2) / 6
//               V2 = (d4 - 4 d2) - (d3 - 4 d1)           U2 = -(g0 - g1 + g2) / 6
//               V3 = (d4 - d2) + 2 (d3 - d1)             U3 = g0 / 24 + g1 / 12 + g2 / 6
//               V4 = (d4 - d2) - 2 (d3 - d1)             U4 = g0 / 24 - g1 / 12 + g2 / 6
//               V5 = 4 d1 - 5 d3 + d5                    U5 = g2
//   M_c = sum over channels of U_c V_c (the MFMAs);  y0 = M0 + M1 + M2 + M3 + M4,  y1 = (M1 - M2) + 2 (M3 - M4),
//   y2 = (M1 + M2) + 4 (M3 + M4),  y3 = (M1 - M2) + 8 (M3 - M4) + M5        (Lavin & Gray 2015, F(4,3), points 0, +-1, +-2, inf)
//
// 16x16x4 MFMAs want N = 16 quads, i.e. 64-frame tiles: one workgroup of 8 waves per CU with 256 registers per wave (the 64
// accumulator registers of the 4 outputs x 4 row tiles do not fit the 128-register budget of two workgroups per CU).  Measured
// with the F(2,3) kernel: one workgroup alone on a CU runs a tile in 55.2 us, two co-resident ones in 105.8 us — co-residency is
// worth 4 %, the matrix work saved here 17 %.  Components are processed in the pairs (0,5), (1,2), (3,4): a pair shares its raw
// LDS reads (14 instead of 22 per channel) and the partial sums d4 - 4 d2, d3 - 4 d1, d4 - d2, d3 - d1.
// Rounding: the transforms multiply by 4, 5, 2, 8 and the weights by 1/4 .. 1/24 — ~10x the rounding error of the direct form,
// ~3e-6 absolute on the pre-activations; the mel bar is 1e-3 (tests/test_gpu_configs.py asserts it over the full 100-step sampler at B=16, T=1000: 6.7e-6).
#include <type_traits>

#include "diffnet_res.h"

namespace bsg {

namespace {

constexpr int NT6 = 64;                  // frames per workgroup = 16 quads
constexpr int FS6 = 88;                  // frames per xs row (80 used; 88 = 8 mod 16 keeps the two k-rows of a ds_read_b128 lane group 8 slots apart)
constexpr int LDZ6 = NT6;
constexpr int SLAB = 32 * 16 * 1024;     // bytes per component of the packed weights: [32 row tiles of 16][16 channel groups][1 KB]

// out[((comp*32 + mt)*16 + q)*256 + lane*4 + jj] = U_comp(m = 16 mt + (lane & 15), k = 16 q + 4 jj + (lane >> 4)),  w = [2C][C][3]
__global__ void pack_wino43_kernel(const float* __restrict__ w, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 6 * 2 * C * C) return;
  const int jj = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
  const int q = rest & 15, mt = (rest >> 4) & 31, comp = rest >> 9;
  const int m = 16 * mt + (lane & 15), k = 16 * q + 4 * jj + (lane >> 4);
  const float* p = w + ((long long)m * C + k) * 3;
  const double g0 = p[0], g1 = p[1], g2 = p[2];
  double u;
  switch (comp) {
    case 0: u = g0 / 4.0; break;
    case 1: u = -(g0 + g1 + g2) / 6.0; break;
    case 2: u = -(g0 - g1 + g2) / 6.0; break;
    case 3: u = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
    case 4: u = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
    default: u = g2; break;
  }
  out[i] = (float)u;
}

#define BSG_MFMA16(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, ACC, 0, 0, 0)
#define BSG_MFMA32(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A_, B_, ACC, 0, 0, 0)

// GEMM1 of one layer: y[j][i] += (output j of this lane's quad, row tile i) over the 6 components x 256 channels.  On entry AW holds
// the weights of step 0 and component a of step 1 (f43_prefetch); xo = this lane's float4 offset of (row lq, frame of d1).
// FAIR: the two waves of a SIMD take turns at issue priority in slices of 8192 shader cycles (s_memtime bit 13 against the wave's
// half of the workgroup).  The SIMD's arbiter favours the older wave: left alone, waves 0-3 finish GEMM1 ~30 % before waves 4-7, which
// then run alone with nobody to fill the issue bubbles of their loads (a lone wave multiplies at ~75 %).
struct F43NoOp {
  __device__ __forceinline__ void operator()() const {}
};
// `pre_last()` runs before the last pair of components (a third of GEMM1 still to go): loads requested there land under its MFMAs
template <int DBG, int FAIR = 0, typename PRE = F43NoOp>
__device__ __forceinline__ void f43_gemm1(f32x4 (&y)[4][4], f32x4 (&AW)[2][2][4], const float* xs, const int xo, const int dil,
                                          const rsrc_t rs_aw, const int vfrag, const int (&sw)[4], const int half = 0, PRE pre_last = PRE()) {
  f32x4 Ma[4], Mb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) Ma[i] = Mb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // running LDS element offsets of d1 .. d4 (every pair reads them) and d0, d5 (pair (0,5) only), channel group 0, k-row lq
  // B fragments of one channel group for the pair PASS (0: comps 0,5; 1: comps 1,2; 2: comps 3,4) from the raw reads
  auto raw_read = [&](int pass, int o, f32x4 (&r)[6]) {   // r[i] = d_i of the 4 channels (jj) of this lane's k-row
    const f32x4* p = reinterpret_cast<const f32x4*>(xs) + o;
    r[1] = p[0]; r[2] = p[dil]; r[3] = p[2 * dil]; r[4] = p[3 * dil];
    if (pass == 0) { r[0] = p[-dil]; r[5] = p[4 * dil]; }
  };
  auto transform = [&](int pass, const f32x4 (&r)[6], f32x4& Ba, f32x4& Bb) {
    if (pass == 0) {
      Ba = 4.0f * r[0] + (r[4] - 5.0f * r[2]);
      Bb = 4.0f * r[1] + (r[5] - 5.0f * r[3]);
    } else if (pass == 1) {
      const f32x4 sv = r[4] - 4.0f * r[2], tv = r[3] - 4.0f * r[1];
      Ba = sv + tv;
      Bb = sv - tv;
    } else {
      const f32x4 sv = r[4] - r[2], tv = r[3] - r[1];
      Ba = sv + 2.0f * tv;
      Bb = sv - 2.0f * tv;
    }
  };
  // component slabs of the pairs
  auto slab_a = [](int pass) { return pass == 0 ? 0 : pass == 1 ? SLAB : 3 * SLAB; };
  auto slab_b = [](int pass) { return pass == 0 ? 5 * SLAB : pass == 1 ? 2 * SLAB : 4 * SLAB; };

  // Software pipeline over the 48 steps (pair, channel group): step n multiplies with B(n) and the weights of ring slot n & 1 while
  //   block a (16 MFMAs, component a): the transform of the raw reads of step n+1 and the weight loads b(n+1) are interleaved,
  //   block b (16 MFMAs, component b): the raw LDS reads of step n+2 and the weight loads a(n+2)
  // so that every non-MFMA instruction issues in the shadow of an MFMA of the same wave (sched_group_barrier pins the
  // interleave; a wave then keeps the matrix pipe busy on its own, whatever the other wave of its SIMD is doing).
  auto step_off = [&](int sidx, int& o, int& wa, int& wb, int& ps) {   // step index -> LDS offset of its raw reads, weight offsets, pair
    sidx = sidx < 47 ? sidx : 47;
    ps = sidx >> 4;
    const int q = sidx & 15;
    o = xo + 4 * q * FS6;
    wa = slab_a(ps) + q * 1024;
    wb = slab_b(ps) + q * 1024;
  };
  f32x4 Ba, Bb, r[6];
  {
    raw_read(0, xo, r);
    transform(0, r, Ba, Bb);
    raw_read(0, xo + 4 * FS6, r);   // step 1
  }
  // slot = n & 1 (weight ring of 2 steps); tp = pair of step n+1 (its transform), rp = pair of step n+2 (its raw reads); o2 = LDS
  // offset of step n+2, wb1 = weights b(n+1), wa2 = weights a(n+2).  (A ring of 3 steps, 32 more registers, measured the same: the
  // GEMM1 phase is not waiting for L2 — with every weight load hitting L1 it takes as long.)
  auto step = [&](auto slot_c, auto tp_c, auto rp_c, int o2, int wb1, int wa2) {
    constexpr int sl = decltype(slot_c)::value, tpp = decltype(tp_c)::value, rpp = decltype(rp_c)::value;
    f32x4 nBa, nBb;
    if (FAIR) {
      const unsigned tnow = (unsigned)__builtin_amdgcn_s_memtime();
      if (((tnow >> 13) & 1u) == (unsigned)half) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
    // ---- block a.  The raw values are re-defined here by empty volatile asm statements: instruction selection otherwise places
    // the (chain-free) transform FMAs right behind the LDS reads of the previous block b, where they wait for the LDS latency
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = (tpp == 0 ? 0 : 1); i < (tpp == 0 ? 6 : 5); ++i)
        if (!(DBG & 2)) asm volatile("" : "+v"(r[i][jj]));
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) BSG_MFMA16(Ma[i], AW[sl][0][i][jj], Ba[jj]);
    if (DBG & 2) { nBa = Ba; nBb = Bb; } else transform(tpp, r, nBa, nBb);
    if (!(DBG & 1)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) AW[sl ^ 1][1][i] = ldf4(rs_aw, vfrag, sw[i] + wb1);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- block b
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) BSG_MFMA16(Mb[i], AW[sl][1][i][jj], Bb[jj]);
    if (!(DBG & 2)) raw_read(rpp, o2, r);
    if (!(DBG & 1)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) AW[sl][0][i] = ldf4(rs_aw, vfrag, sw[i] + wa2);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    Ba = nBa; Bb = nBb;
  };
  // one step with its offsets: n = global step index
  auto do_step = [&](auto slot_c, auto tp_c, auto rp_c, int n) {
    int o1, wa1, wb1, o2, wa2, wb2, ps;
    step_off(n + 1, o1, wa1, wb1, ps);
    step_off(n + 2, o2, wa2, wb2, ps);
    if (DBG & 4) { wb1 = 0; wa2 = 0; }   // timing experiment: every weight load hits the same 8 KB (L1)
    step(slot_c, tp_c, rp_c, o2, wb1, wa2);
  };
  auto run_pass = [&](auto pass_c, auto np_c) {
    constexpr int pass = decltype(pass_c)::value;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int s0 = 16 * pass;
#pragma unroll 1
    for (int q = 0; q < 14; q += 2) {   // steps whose successors n+1, n+2 are in the same pair
      do_step(S0{}, pass_c, pass_c, s0 + q);
      do_step(S1{}, pass_c, pass_c, s0 + q + 1);
    }
    do_step(S0{}, pass_c, np_c, s0 + 14);   // its n+2 is group 0 of the next pair
    do_step(S1{}, np_c, np_c, s0 + 15);     // (after the last pair: clamped repeats, unused)
    // the pair is complete: output transform A^T
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float ma = Ma[i][r4], mb = Mb[i][r4];
        if (pass == 0) {
          y[0][i][r4] += ma;
          y[3][i][r4] += mb;
        } else {
          const float sm = ma + mb, df = ma - mb;
          if (pass == 1) {
            y[0][i][r4] += sm; y[1][i][r4] += df; y[2][i][r4] += sm; y[3][i][r4] += df;
          } else {
            y[0][i][r4] += sm;
            y[1][i][r4] = __builtin_fmaf(2.0f, df, y[1][i][r4]);
            y[2][i][r4] = __builtin_fmaf(4.0f, sm, y[2][i][r4]);
            y[3][i][r4] = __builtin_fmaf(8.0f, df, y[3][i][r4]);
          }
        }
        Ma[i][r4] = 0.f; Mb[i][r4] = 0.f;
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  run_pass(I0{}, I1{});
  run_pass(I1{}, I2{});
  pre_last();
  run_pass(I2{}, I2{});
}

__device__ __forceinline__ void f43_prefetch(f32x4 (&AW)[2][2][4], const rsrc_t rs_aw, const int vfrag, const int (&sw)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    AW[0][0][i] = ldf4(rs_aw, vfrag, sw[i] + 0 * SLAB);          // step 0: components 0 and 5, channel group 0
    AW[0][1][i] = ldf4(rs_aw, vfrag, sw[i] + 5 * SLAB);
    AW[1][0][i] = ldf4(rs_aw, vfrag, sw[i] + 0 * SLAB + 1024);   // a(1); b(1) is requested in block a of step 0
  }
}

// ------------------------------------------------------------------------------------------------
// All L layers of one 64-frame tile in ONE launch, residual stream on chip (the fp32-matrix-pipe sibling of diffnet_h2.hip;
// same hand-off protocol — write-through stores, drain, barrier, flag with the launch epoch, bounded poll, sc1 loads — see there).
// One workgroup per CU; every tile of the launch must be resident (the host checks).
//   * the conv input image x + d_l (fp32) stays in LDS in GEMM1's layout and is rewritten in place by the residual rows of GEMM2
//     (x itself is recovered as image - d_l: one rounding of x + d, 6e-8 relative, per layer);
//   * neighbours exchange the two 8-frame edges of the new image (2 x 8 KB per tile) through L2; GEMM2 runs its residual rows first,
//     publishes, and only then its skip rows, so that the edges fly under the skip rows' MFMAs;
//   * the running skip sum stays in HBM (read-modify-write per layer): registers are what this kernel is out of.
// HBM bytes per frame and layer: conditioner term 2 KB + skip 2 KB (+ edges 0.5 KB through L2) instead of 6 KB.
// (Round 2 also had a per-layer kernel with this GEMM1; it was neither a default nor a fallback and was removed in round 3.)
// ------------------------------------------------------------------------------------------------
constexpr int XS4_FLOATS = 64 * FS6 * 4;                        // 22,528
constexpr size_t STACK43_LDS = (size_t)(XS4_FLOATS + C * LDZ6 + 2 * C) * sizeof(float);   // 88 KB + 64 KB + 2 KB = 157,696 B

template <int FAIR>
__global__ __launch_bounds__(512, 2) void residual_stack_f43_kernel(StackArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                       // image x + d_l, layout [g][lq][frame 0..79 (+8)][jj]
  float* zs = lds + XS4_FLOATS;          // [C][64] gated activation
  float* dcur = zs + C * LDZ6;           // [C] d_l
  float* dnxt = dcur + C;                // [C] d_{l+1}

  const int n_tiles = p.n_tiles, per_xcd = (n_tiles + 7) >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile_id >= n_tiles) return;
  p.fbase = stack_epoch_take(p);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5, p16 = lane & 15, lq = lane >> 4;
  const int tpr = p.tiles_per_row, L = p.L, T = p.T;
  const int b = tile_id / tpr, j = tile_id - b * tpr;
  const int t0 = j * NT6;
  const int tb = p.t_dev ? (int)p.t_dev[b] : p.t_uniform;
  const bool has_left = j > 0, has_right = j + 1 < tpr;

  const unsigned plane = (unsigned)C * T * 4;
  const rsrc_t rs_x = mk_rsrc(p.x_in + (long long)b * C * T, plane);
  const rsrc_t rs_sk = mk_rsrc(p.skip + (long long)b * C * T, plane);
  const int rowT = T * 4, vfrag = lane * 16;
  int sw[4];
  sw[0] = (2 * wave) * 16 * 1024; sw[1] = (2 * wave + 1) * 16 * 1024;
  sw[2] = (16 + 2 * wave) * 16 * 1024; sw[3] = (16 + 2 * wave + 1) * 16 * 1024;
  const int sb_r = wave * 32 * 1024, sb_s = (8 + wave) * 32 * 1024;   // residual / skip row tile of the packed output projection
  int vcol[2], vst[2];
  bool col_ok[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = t0 + 32 * ct + l31;
    col_ok[ct] = col < T;
    vcol[ct] = (lh * 4 * T + (col_ok[ct] ? col : T - 1)) * 4;
    vst[ct] = (lh * 4 * T + col) * 4;
  }
  // accumulator register r of GEMM2 = channel c = 32 w + (r & 3) + 8 (r >> 2) + 4 lh: row (c / 16) 4 + c % 4 = (2 w + (r >> 3)) 4 + (r & 3),
  // jj = (c / 4) % 4 = (2 (r >> 2)) % 4 + lh.  Element of the image, column tile 0 (column tile 1: + 32 frames): one per-lane base
  // plus a compile-time offset per register
  const int ibase = ((8 * wave) * FS6 + HALO + l31) * 4 + lh;
  auto img_off = [&](int r) { return ibase + (((r >> 3) * 4 + (r & 3)) * FS6) * 4 + ((2 * (r >> 2)) & 3); };
  f32x4 AW[2][2][4];
  f43_prefetch(AW, mk_rsrc(p.apackw43, 6 * 2 * C * C * 4), vfrag, sw);
  // ---- layer 0: stage x + d_0 from HBM (the whole input exists, halo included) ------------------------------------------------------
  {
    const rsrc_t rs_dp = mk_rsrc(p.dproj + ((long long)tb * L + 0) * C, C * 4);
    if ((T & 3) == 0) {
#pragma unroll 1
      for (int it = tid; it < 64 * 20; it += 512) {
        const int row = it / 20, j4 = it - row * 20;
        const int c0 = 16 * (row >> 2) + (row & 3);
        const int t = t0 - HALO + 4 * j4;
        const bool ok = t >= 0 && t < T;
        f32x4 v[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          v[jj] = ldf4(rs_x, ok ? ((c0 + 4 * jj) * T + t) * 4 : 0, 0);
          v[jj] += ldf(rs_dp, (c0 + 4 * jj) * 4, 0);
          if (!ok) v[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          *reinterpret_cast<f32x4*>(xs + ((row * FS6 + 4 * j4 + k) << 2)) = f32x4{v[0][k], v[1][k], v[2][k], v[3][k]};
      }
    } else {
#pragma unroll 4
      for (int idx = tid; idx < C * 80; idx += 512) {
        const int c = idx / 80, jf = idx - c * 80;
        const int t = t0 - HALO + jf;
        const bool ok = t >= 0 && t < T;
        const float v = ldf(rs_x, ok ? (c * T + t) * 4 : 0, 0) + ldf(rs_dp, c * 4, 0);
        xs[(((((c >> 4) << 2) + (c & 3)) * FS6 + jf) << 2) + ((c >> 2) & 3)] = ok ? v : 0.f;
      }
    }
    if (tid < C) {
      dcur[tid] = p.dproj[((long long)tb * L + 0) * C + tid];
      dnxt[tid] = L > 1 ? p.dproj[((long long)tb * L + 1) * C + tid] : 0.f;
    }
  }

#define STK_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (p.stamps && tid == 0) p.stamps[((long long)tile_id * L + l) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
  const unsigned long long clk0 = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;   // shader clock, to price the phases in cycles
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int dil = 1 << (l % p.cycle);
    const int ld = dil == 1 ? 0 : dil == 2 ? 1 : dil == 4 ? 2 : 3;
    const int tp = ((p16 >> ld) << (ld + 2)) + (p16 & (dil - 1));
    const rsrc_t rs_aw = mk_rsrc(p.apackw43 + (long long)l * (6 * 2 * C * C), 6 * 2 * C * C * 4);
    const rsrc_t rs_a2 = mk_rsrc(p.apack2 + (long long)l * (2 * C * C), 2 * C * C * 4);
    const rsrc_t rs_bo = mk_rsrc(p.bias_out + (long long)l * (2 * C), 2 * C * 4);
    const rsrc_t rs_ct = mk_rsrc(p.condterm + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
    __syncthreads();   // (A) the image of this layer is complete: core (own update / staging) and halo rows
    STK_STAMP(0);
    f32x4 y[4][4];
#pragma unroll
    for (int jo = 0; jo < 4; ++jo)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[jo][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the gate rows' conditioner term (this wave's 32 rows x 64 frames) goes straight into the z tile by LDS-DMA while GEMM1 runs: the
    // gate then reads it where it writes z, and only the filter rows' term is loaded from HBM behind GEMM1 (T % 4 == 0: 16-byte pieces)
    const bool dma = (T & 3) == 0;
    if (dma) {
      const int voff = (((lane >> 4) * T) + t0 + 4 * (lane & 15)) * 4;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_ct, (__attribute__((address_space(3))) void*)(zs + (32 * wave + 4 * k) * LDZ6), 16, voff,
                                                 (32 * wave + 4 * k) * rowT, 0, 0);
    }
    // the filter rows' conditioner term: requested before the last third of GEMM1 (32 registers), so that the gate does not wait for HBM
    float cfv[4][8];
    auto cond_f = [&](int i) {
#pragma unroll
      for (int jo = 0; jo < 4; ++jo) {
        const int f = t0 + tp + jo * dil;
        const int vc = (lq * 4 * T + (f < T ? f : T - 1)) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) cfv[jo][4 * i + r] = ldf(rs_ct, vc, (C + 32 * wave + 16 * i + r) * rowT);
      }
    };
    f43_gemm1<0, FAIR>(y, AW, xs, lq * FS6 + HALO + tp, dil, rs_aw, vfrag, sw, wave >> 2, [&]() { cond_f(0); });   // half of them: registers
    cond_f(1);   // the other half: 16 loads that fly across the barrier and the first half of the gate
    if (dma) {   // LDS-DMA data is ordered for a ds_read by the issuing wave's vmcnt followed by a barrier the reader has passed
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // everything older than those 16 loads, the LDS-DMA pieces included
      __builtin_amdgcn_s_barrier();
    }
    STK_STAMP(1);
    // ---- gate -> zs (its own region: no barrier before the stores); GEMM2's first weights fly meanwhile ---------------------------
    f32x4 Ag[2], Af[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) Ag[k] = ldf4(rs_a2, vfrag, sb_r + k * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jo = 0; jo < 4; ++jo) {
        const int f = t0 + tp + jo * dil;
        const int vc = (lq * 4 * T + (f < T ? f : T - 1)) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* zp = zs + (32 * wave + 16 * i + 4 * lq + r) * LDZ6 + tp + jo * dil;
          const float cg = dma ? *zp : ldf(rs_ct, vc, (32 * wave + 16 * i + r) * rowT);
          *zp = gate1(y[jo][i][r] + cg, y[jo][2 + i][r] + cfv[jo][4 * i + r]);
        }
      }
    // residual rows start from x + b_out with x = image - d_l (this wave's own 32 channels)
    f32x16 r0, r1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = img_off(r);
      const float dl = dcur[32 * wave + acc_row(r, lh)], br = ldf(rs_bo, lh * 16, (32 * wave + acc_row0(r)) * 4);
      r0[r] = (xs[o] - dl) + br;
      r1[r] = (xs[o + 32 * 4] - dl) + br;
    }
    __syncthreads();   // (B) z complete; every wave is done reading the image
    STK_STAMP(2);
    // ---- GEMM2, residual rows: 32 groups of 8 channels, two column tiles share the A fragments -----------------------------------
    const float* zrow = zs + lh * LDZ6 + l31;
    auto ldb = [&](int q, int ct) {
      const float* pz = zrow + 8 * q * LDZ6 + 32 * ct;
      return f32x4{pz[0], pz[2 * LDZ6], pz[4 * LDZ6], pz[6 * LDZ6]};
    };
    auto gemm2 = [&](f32x16& c0, f32x16& c1, f32x4 (&A)[2], int sbase) {
      f32x4 B0[2], B1[2];
      B0[0] = ldb(0, 0);
      B1[0] = ldb(0, 1);
#pragma unroll 1
      for (int q = 0; q < 32; q += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int qn = q + s + 1 <= 31 ? q + s + 1 : 31;
          B0[(s + 1) & 1] = ldb(qn, 0);
          B1[(s + 1) & 1] = ldb(qn, 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            BSG_MFMA32(c0, A[s][k], B0[s & 1][k]);
            BSG_MFMA32(c1, A[s][k], B1[s & 1][k]);
          }
          __builtin_amdgcn_sched_barrier(0);
          const int qr = q + s + 2 <= 31 ? q + s + 2 : 31;
          A[s] = ldf4(rs_a2, vfrag, sbase + qr * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    gemm2(r0, r1, Ag, sb_r);
    STK_STAMP(3);
#pragma unroll
    for (int k = 0; k < 2; ++k) Af[k] = ldf4(rs_a2, vfrag, sb_s + k * 1024);   // skip rows' first weights
    // running skip sum of this wave's rows: requested now, added after the skip rows' chain
    float prev0[16], prev1[16];
    if (l > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int so = (32 * wave + acc_row0(r)) * rowT;
        prev0[r] = ldf(rs_sk, vcol[0], so);
        prev1[r] = ldf(rs_sk, vcol[1], so);
      }
    }
    if (l + 1 < L) {
      // ---- the next layer's image: (x + residual) / sqrt(2) + d_{l+1}, zero beyond T (the conv pads x + d) ---------------------------
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = img_off(r);
        const float dn = dnxt[32 * wave + acc_row(r, lh)];
        xs[o] = col_ok[0] ? r0[r] / 1.41421356237309504880f + dn : 0.f;   // net.py:78
        xs[o + 32 * 4] = col_ok[1] ? r1[r] / 1.41421356237309504880f + dn : 0.f;
      }
      f43_prefetch(AW, mk_rsrc(p.apackw43 + (long long)(l + 1) * (6 * 2 * C * C), 6 * 2 * C * C * 4), vfrag, sw);
      __syncthreads();   // (C1) core image complete
      {
        // publish the first and the last 8 frames: [side][row 64][frame 8] float4 = 2 x 8 KB, write-through
        float* hx_t = p.hx + ((long long)((l + 1) & 1) * n_tiles + tile_id) * (2 * C * 8);
        const rsrc_t rs_hx = mk_rsrc(hx_t, 2 * C * 8 * 4);
        const int row = tid >> 3, f = tid & 7;
        if (!(p.inject == 1 && (tile_id & 1))) {
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xs + ((row * FS6 + HALO + (side ? NT6 - 8 : 0) + f) << 2));
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_hx, ((side * 64 + row) * 8 + f) * 16, 0, 16);   // sc1
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
      __syncthreads();   // (C)
      if (tid == 0) __hip_atomic_store(p.flags + tile_id, p.fbase + (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STK_STAMP(4);
    // ---- GEMM2, skip rows (the edges fly meanwhile) ----------------------------------------------------------------------------------
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = s1[r] = ldf(rs_bo, lh * 16, (C + 32 * wave + acc_row0(r)) * 4);
    gemm2(s0, s1, Af, sb_s);
    {
      const float div = l + 1 == L ? sqrtf((float)L) : 1.0f;   // skip sum / sqrt(L) (net.py:126)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        if (col_ok[ct]) {
          const f32x16& ss = ct ? s1 : s0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = l > 0 ? (ct ? prev1[r] : prev0[r]) : 0.f;
            stf((pv + ss[r]) / div, rs_sk, vst[ct], (32 * wave + acc_row0(r)) * rowT);
          }
        }
    }
    STK_STAMP(5);
    if (l + 1 == L) {
      if (p.stamps && tid == 0) {   // slots 6, 7 of the last layer: shader cycles from the first layer's start to here
        p.stamps[((long long)tile_id * L + l) * 8 + 6] = clk0;
        p.stamps[((long long)tile_id * L + l) * 8 + 7] = __builtin_amdgcn_s_memtime();
      }
      break;
    }
    // ---- wait for the neighbours' edges of layer l+1, copy them into the halo rows ------------------------------------------------------
    if (tid < C) {   // d tables: every reader of d_l / d_{l+1} is behind barrier (C)
      dcur[tid] = dnxt[tid];
      dnxt[tid] = l + 2 < L ? p.dproj[((long long)tb * L + l + 2) * C + tid] : 0.f;
    }
    if (tid == 0) {
      const unsigned want = p.fbase + (unsigned)(l + 1);
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
    STK_STAMP(6);
    {
      // halo frames 0..7 = the left neighbour's last 8 frames (its side 1), frames 72..79 = the right neighbour's first 8 (its side 0).
      // Every load of handed-off bytes is an sc1 buffer load to registers (MI355X_MICROARCH.md, hand-off "Valid forms").
      const int row = tid >> 3, f = tid & 7;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const bool have = side == 0 ? has_left : has_right;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (have) {
          const float* src = p.hx + ((long long)((l + 1) & 1) * n_tiles + (side == 0 ? tile_id - 1 : tile_id + 1)) * (2 * C * 8);
          const rsrc_t rs_h = mk_rsrc(src, 2 * C * 8 * 4);
          v = __builtin_amdgcn_raw_buffer_load_b128(rs_h, (((side == 0 ? 1 : 0) * 64 + row) * 8 + f) * 16, 0, 16);   // sc1
        }
        *reinterpret_cast<u32x4*>(xs + ((row * FS6 + (side ? HALO + NT6 : 0) + f) << 2)) = v;
      }
    }
    STK_STAMP(7);
  }
#undef STK_STAMP
  if (tid == 0) stack_epoch_done(p, p.fbase, n_tiles);
}
#undef BSG_MFMA16
#undef BSG_MFMA32

}  // namespace

int stack_f43_occupancy() {
  int o = 0;
  if (hipFuncSetAttribute((const void*)residual_stack_f43_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)STACK43_LDS) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)residual_stack_f43_kernel<1>, 512, STACK43_LDS) != hipSuccess)
    return 0;
  return o;
}

int launch_residual_stack_f43(const StackArgs& p, hipStream_t st) {
  hipLaunchKernelGGL(residual_stack_f43_kernel<1>, dim3(8 * cdiv(p.n_tiles, 8)), dim3(512), STACK43_LDS, st, p);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int pack_wino43(const float* w, float* out, hipStream_t st) {
  hipLaunchKernelGGL(pack_wino43_kernel, dim3(cdiv(6LL * 2 * C * C, 256)), dim3(256), 0, st, w, out);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

}  // namespace bsg
