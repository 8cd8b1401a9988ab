// fp32 DiffNet residual stack on the 16-bit matrix pipe, on 16-ROW MATRIX TILES: residual_stack_h2_kernel (diffnet_h2.hip) with every
// product issued as v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16.
//
// Same contract, tensors, LDS images, hand-off protocol, split (hi + lo fp16 per fp32 operand; lo hi + hi hi + hi lo per product), scales, range
// guard and fused step tail as that launch — reference semantics /root/reference/train_bisinger/usr/diff/net.py:66-78,107-130, with TAIL also
// net.py:126-129 and usr/diff/shallow_diffusion_tts.py:149-201 — and the same matrix cycles per layer (a 16x16x32 MFMA is 16 clocks for half the
// FLOP of a 32-clock 32x32x16).  What differs is what the chip holds its clock at under it (MI355X_MICROARCH.md "DVFS give-back" item 7: the
// clock a dense MFMA loop sustains depends on the MFMA shape): per matrix cycle the 16-row shape reads and writes 0.25 accumulator registers
// per lane where the 32-row shape moves 0.5, and tools/gemm_shape_ab.hip — this launch's GEMM1 loop alone, both shapes, the same output tile
// per wave, weights streamed from L2, operand fragments from LDS, random hi / lo planes — holds 1.57-1.60 GHz with it against 1.37-1.39 GHz:
// 26.4-27.4 us per GEMM1 of a 64-frame tile on all 256 CUs against 28.8-29.7 us (profiles/r05_gemm_shape_ab.txt).
//
// Layout.  A wave owns 32 channels, as before, as 2 + 2 row tiles of 16 (gate + filter rows of GEMM1, residual + skip rows of GEMM2) over
// NQ = 2 NCT column tiles of 16 frames: lane l holds A[row l & 15][k = 8 (l >> 4) + j], B[k][column l & 15] and, of a 16 x 16 result, column
// l & 15, rows 4 (l >> 4) + r.  So a lane's 4 registers of a tile are 4 CONSECUTIVE channels of one frame — the 8-byte pieces of the
// channels-last LDS images as before — and element (ct, rt, i) of x / the skip sum / an accumulator is channel 32 w + 16 rt + 4 (l >> 4) + i of
// frame 16 ct + (l & 15).  (The 16-lane groups of ds_read_b128 — {0-3, 12-15, 20-27}, ... — meet one 2-way bank conflict each on 528-byte rows.
// Two conflict-free layouts were built and measured, profiles/r05_q_layouts.txt: rows two apart per lane (frame 2 n + ..) made the global loads of
// x and the conditioner term strided dwords, 15k cycles to issue instead of 2.6k; a permutation of a tile's 16 frames plus units g, g + 1 of a
// fragment half a bank row apart kept those loads contiguous and changed nothing measurable: a conflict costs one LDS cycle in five.)
// Weights: the 16-row fragments the part forms stream (pack_a_frag_q_kernel, diffnet_h2.hip): per 32-deep k-step a hi
// and a lo slab of 32 row tiles x 1 KB.  A k-step is 48 MFMAs per wave on 8 weight fragments (ring of 2 k-steps = 64 registers, as the 4 x 16
// deep of the 32-row form) and, per column tile, 2 operand fragments from LDS, read two column tiles ahead into four buffers (32 registers, as there).
#include "diffnet_h2_shared.h"
#include <type_traits>

// (float)hi + (float)lo of one half of two packed f16 pairs in ONE instruction (round 5)
__device__ __forceinline__ float mix_add(unsigned hp, unsigned lp, int half) {
  float r;
#if defined(__HIP_DEVICE_COMPILE__)
  if (half == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hp), "v"(lp));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hp), "v"(lp));
#else
  r = 0.f;
#endif
  return r;
}

#ifndef BSG_CQ_AUX
#define BSG_CQ_AUX 3   // cache-policy bits of the conditioner term's loads: nt + sc0 — a stream read once per step that should not displace the
                       // weight fragments in L2 (same-box A/B, profiles/r05_cq_aux_ab.log: 43.11 -> 42.8 us per layer; 0 = default policy, 16 = sc1)
#endif

namespace bsg {

namespace {

constexpr int QPLB = 32 * 1024;            // bytes per plane of a k-step slab: 32 row tiles of 16 x 1 KB
constexpr int QKSB = 2 * QPLB;             // bytes per k-step (32 deep): hi slab, lo slab
using f32x4q = __attribute__((ext_vector_type(4))) float;
// the conv image holds x x 2^4 (x alone, not x + d: see the kernel): with x of order 1 unscaled, most lo terms were subnormal fp16 (an ABSOLUTE
// error of 2^-25 each — harmless as a GEMM operand, but the residual path takes x back from these planes: eps rms error 4.4e-7 against float64
// instead of 7.5e-8, tests/test_gpu_h2.py); x 16 keeps lo normal down to |x| = 0.03 and the range guard at |x| < 3750
constexpr float XSCALE = 16.0f, XINV = 1.0f / 16.0f;
#define BSG_MFMA_Q(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_, B_, ACC, 0, 0, 0)

// i-th executed k-step -> k-step index: GEMM1 (ROT = 8: 24 k-steps of 32, tap-major) starts with the CENTRE tap, whose B operand is the tile's
// own frames, and visits the two outer taps (which read the neighbours' halo frames) afterwards
template <int ROT>
__device__ __forceinline__ int kmapq(int i) {
  if (ROT == 0) return i;
  return i < ROT ? i + ROT : (i < 2 * ROT ? i - ROT : i);
}

// k-step pipeline over 4 row tiles (c[0], c[1]: gate / residual; c[2], c[3]: filter / skip) x NQ column tiles of 16 frames.  Per column tile 12
// MFMAs — lo hi, hi hi, hi lo for each of the 4 row tiles; an accumulator is revisited every 4th MFMA — on the k-step's 8 weight fragments
// A[s][2 rt] = hi, A[s][2 rt + 1] = lo and the tile's two operand fragments; the fragments of the column tile after the next are read from LDS
// inside the first MFMA group.
//   The weight ring holds NS k-steps (slot = executed k-step mod NS) and runs THROUGH the GEMMs: a slot is refilled during the LAST column tile
// of its k-step (lo fragments behind the first MFMA group, hi fragments behind the last) with the k-step NS ahead — and the last NS k-steps of
// a GEMM fetch the FIRST NS of the GEMM that follows (`rs_next`, in that GEMM's order NEXT_ROT), each into its own slot: k-step j of the next
// GEMM behind the k-step of this one with (executed index mod NS) = j, so that every GEMM starts with its k-step 0 in slot 0 whatever N_KS mod
// NS is.  (Clamped to the GEMM's last k-step instead, the reloads re-read it — 12 % more weight bytes per layer — and every GEMM began with a
// burst of fragment requests of its own.)  N_KS need not be a multiple of NS: the remainder is a shorter, separately unrolled pass.
//   `mid()` runs in front of executed k-step MIDK (0: never): GEMM1's hand-off with the neighbours sits there, under the centre tap's MFMAs.
// FAIRB: the two waves of a SIMD take turns at issue priority.
template <int ROT, int NEXT_ROT, int N_KS, int MIDK, bool FAIRB, int NQ, int NS, int DIAG, typename LDB, typename MID, typename STAMP>
__device__ __forceinline__ void mfma_pipe_q(f32x4q (&c)[4][NQ], f16x8 (&A)[NS][8], rsrc_t rs, rsrc_t rs_next, int vfrag, const int (&sa)[4], LDB ldb, MID mid,
                                            int half, bool diag_l1, STAMP stamp) {
  static_assert((NQ == 2 || NQ == 4) && (NS * NQ) % 4 == 0 && N_KS >= NS, "four operand buffers, indexed by the item's position in a pass of NS k-steps");
  // operand fragments: item (k-step, column tile) -> buffer (item index in the pass) & 3, read from LDS TWO column tiles = 24 MFMAs ahead (one
  // tile ahead, its latency — 8 waves' 16-byte reads of 528-byte rows — showed between the column tiles: GEMM2 8.05 us against 6.14 for the 32-row form)
  f16x8 B[4][2];
  constexpr int last = N_KS - 1;
  ldb(kmapq<ROT>(0), 0, B[0]);
  ldb(kmapq<ROT>(0), 1, B[1]);
  // one k-step: e0 = executed index of the pass's first k-step (a multiple of NS), s = position in the pass = ring slot
  auto one_step = [&](int e0, auto s_) {
    constexpr int s = decltype(s_)::value;
    const int e = e0 + s;
    if (MIDK > 0 && e == MIDK) {
      mid();
      ldb(kmapq<ROT>(e), 0, B[(s * NQ) & 3]);   // the B operands of the next two column tiles were read before the halo rows arrived: read them again
      ldb(kmapq<ROT>(e), 1, B[(s * NQ + 1) & 3]);
    }
    const bool over = e + NS > last;   // this slot's next occupant belongs to the GEMM that follows: its k-step s
    const int kr = diag_l1 ? 0 : (over ? kmapq<NEXT_ROT>(s) : kmapq<ROT>(e + NS)) * QKSB;   // (diag_l1: every reload from ONE L1-resident k-step — wrong results)
    const rsrc_t rsr = over ? rs_next : rs;
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct) {
      const int item = s * NQ + ct;
      if (DIAG != 2) {   // (DIAG: timing experiments, wrong results: 1 = no weight reloads, 2 = no operand reads either)
        const int en = e + (ct + 2) / NQ;
        ldb(kmapq<ROT>(en <= last ? en : last), (ct + 2) % NQ, B[(item + 2) & 3]);
      }
      const f16x8(&Bc)[2] = B[item & 3];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) BSG_MFMA_Q(c[rt][ct], A[s][2 * rt + 1], Bc[0]);   // lo hi
      if (ct == NQ - 1 && DIAG != 1 && DIAG != 2) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) A[s][2 * rt + 1] = lda8(rsr, vfrag, sa[rt] + kr + QPLB);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) BSG_MFMA_Q(c[rt][ct], A[s][2 * rt], Bc[0]);       // hi hi
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) BSG_MFMA_Q(c[rt][ct], A[s][2 * rt], Bc[1]);       // hi lo
      if (ct == NQ - 1 && DIAG != 1 && DIAG != 2) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) A[s][2 * rt] = lda8(rsr, vfrag, sa[rt] + kr);
      }
      if (DIAG != 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (ct == NQ - 1 && DIAG != 1 && DIAG != 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      } else if (DIAG != 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto pass_top = [&](int e0) {
    if (DIAG == 3) stamp(e0 / NS);   // (diagnostic instantiation: one stamp per pass)
    if (FAIRB) {
      const unsigned tnow = (unsigned)__builtin_amdgcn_s_memtime();
      if (((tnow >> 12) & 1u) == (unsigned)half) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
  };
  constexpr int FULL = N_KS / NS, REM = N_KS % NS;
#pragma unroll 1
  for (int e0 = 0; e0 < FULL * NS; e0 += NS) {
    pass_top(e0);
    one_step(e0, std::integral_constant<int, 0>{});
    if constexpr (NS > 1) one_step(e0, std::integral_constant<int, 1>{});
    if constexpr (NS > 2) one_step(e0, std::integral_constant<int, 2>{});
    if constexpr (NS > 3) one_step(e0, std::integral_constant<int, 3>{});
  }
  if constexpr (REM > 0) {
    pass_top(FULL * NS);
    one_step(FULL * NS, std::integral_constant<int, 0>{});
    if constexpr (REM > 1) one_step(FULL * NS, std::integral_constant<int, 1>{});
    if constexpr (REM > 2) one_step(FULL * NS, std::integral_constant<int, 2>{});
  }
}

template <bool FAIRB, bool TAIL, int NCT, int DIAG = 0, int NS = 2>
__global__ __launch_bounds__(512, 2) void residual_stack_q_kernel(StackArgs p, TailArgs a) {
  constexpr int NT = 32 * NCT, NQ = 2 * NCT, XP = h2_xp(NCT), ZP = h2_zp(NCT);   // frames / column tiles of 16 per workgroup; bytes per plane
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  char* xs = lds_raw;                  // [2 planes][NT + 16 frames][528 B]: hi / lo of x + d_l, frames t0-8 .. t0+NT+7
  char* zs = lds_raw + 2 * XP;         // [2 planes][NT frames][528 B]: hi / lo of 2^10 x gated activation
  float* btab = reinterpret_cast<float*>(lds_raw + 2 * XP + 2 * ZP);   // [512]: output-projection bias of the current layer

  const int n_tiles = p.n_tiles, per_xcd = (n_tiles + 7) >> 3;
  const int tile_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile_id >= n_tiles) return;
  p.fbase = stack_epoch_take(p);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, q4 = lane >> 4;
  auto fo = [](int ct) { return 16 * ct; };   // frame of column tile ct, column n16: fo(ct) + n16
  const int tpr = p.tiles_per_row, L = p.L, T = p.T;
  const int b = tile_id / tpr, j = tile_id - b * tpr;
  const int t0 = j * NT;
  const int tb = p.t_dev ? (int)p.t_dev[b] : p.t_uniform;
  const bool has_left = j > 0, has_right = j + 1 < tpr;

  const unsigned plane = (unsigned)C * T * 4;
  const rsrc_t rs_x = mk_rsrc(p.x_in + (long long)b * C * T, plane);
  const int rowT = T * 4, vfrag = lane * 16;
  // global addressing: ONE per-lane byte offset per layout; column tile ct adds a constant (16 frames) that the compiler folds into the
  // instruction's immediate offset.  Frames beyond T are NOT clamped (a clamp made the offsets of the four column tiles four live registers,
  // three times over): such a lane reads the next row's first elements, or 0 behind the buffer — finite either way — and its image rows
  // are written as zeros (write_core), its results never stored
  const int col0 = t0 + n16;
  const int vbase = (q4 * 4 * T + col0) * 4;       // fp32 [rows][T]: row q4 * 4 (+ the uniform row offset), frame col0
  const int vqbase = (q4 * T + col0) * 16;         // channel-quad order [rows / 4][T][4]: quad q4, frame col0
  auto col_ok = [&](int ct) { return col0 + 16 * ct < T; };
  auto vcol = [&](int ct) { return vbase + 64 * ct; };
  auto vquad = [&](int ct) { return vqbase + 256 * ct; };
  // row tiles (of 16) inside a plane of a k-step slab: gate / residual rows 2w, 2w + 1; filter / skip rows 16 + 2w, 17 + 2w
  const int sa[4] = {(2 * wave) * 1024, (2 * wave + 1) * 1024, (16 + 2 * wave) * 1024, (17 + 2 * wave) * 1024};
  const int cw = 32 * wave + 4 * q4;   // the lane's first channel; + 16 rt + i

  // x is NOT kept in registers through a layer (round 5): 32 registers that were live through both GEMM loops of a kernel that spilled 30-40
  // registers per layer into the memory pipeline its weight stream needs.  The residual path takes x back from the conv image in LDS — and
  // so that this costs no accuracy the image is the hi / lo split of x ITSELF, not of x + d_l (hi + lo = x to one ulp, mostly exactly): the
  // diffusion-step term d_l is constant over the frames, so its share of the dilated conv is a vector per (step, layer, tap), D_tap = W_tap d_l,
  // tabulated at create (StackArgs::dconv, fp64 accumulation) and added where the accumulators are initialised: y = (cond + D_0 [f >= dil] + D_1 +
  // D_2 [f + dil < T]) s1 — a tap that falls on the reference's zero padding of x + d (net.py:72-74) contributes nothing.  (The first
  // version took x = hi + lo - d from the image of x + d: correct to an ulp of x + d, which is not an ulp of x when |d| >> |x| —
  // tests/test_gpu_h2.py 'big_act', rms error 6.5e-6 against 1.3-1.8e-6 for the fp32-pipe forms; now 0.6e-6.)
  float sk[NQ][2][4];     // running skip sum (fp32), same layout
  f32x4q y[4][NQ];        // accumulators: y[rt] gate / residual rows, y[2 + rt] filter / skip rows; GEMM1's start from the conditioner term x s1
  int range_flag = 0;     // see residual_stack_h2_kernel: a split value beyond 60000 (or not finite) raises the launch's status word 1
  auto range_check = [&](unsigned worst) {
    if (__builtin_amdgcn_ballot_w64(worst >= 0x476A6000u) != 0ull) range_flag = 1;   // 60000.0f
  };
  auto absbits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };

  // the conditioner term of a layer (fp32 [2C][T] rows of this utterance): 32 NCT dword loads per lane, 64 B contiguous per 16 lanes,
  // requested straight into the accumulators a phase before they are used
#define BSG_CQ_LD(R, V, S) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(R, V, S, BSG_CQ_AUX))
  auto cond_request = [&](int l) {
    if (DIAG == 4) {
      // timing experiment (wrong results): the term costs NOTHING — a per-lane pseudo-value of its magnitude instead of the 128 KB per tile
      // and layer from HBM: what is left is the layer without its only HBM stream
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) y[k][ct][i] = (float)((lane * 37 + l * 13 + 29 * k + 11 * ct + 5 * i) & 31) * 0.0625f - 1.0f;
      return;
    }
    if (p.condterm_q) {
      // channel-quad order [2C/4][T][4]: the 4 registers of an accumulator tile are ONE 16-byte load, 256 B contiguous per 16 lanes: 8 NCT
      // requests per lane instead of 32 NCT (as dwords every 128-byte line of the term was looked up by two requests of 64 B, and the
      // requests alone kept the trailing wave 2.6k cycles)
      const rsrc_t rs_cq = mk_rsrc(p.condterm_q + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int so = (8 * wave + 4 * rt) * T * 16;
#pragma unroll
        for (int ct = 0; ct < NQ; ++ct) {
          y[rt][ct] = BSG_CQ_LD(rs_cq, vquad(ct), so);
          y[2 + rt][ct] = BSG_CQ_LD(rs_cq, vquad(ct), so + (C / 4) * T * 16);
        }
      }
      return;
    }
    const rsrc_t rs_ct = mk_rsrc(p.condterm + (long long)l * p.ct_stride + (long long)b * 2 * C * T, 2 * plane);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int so = (32 * wave + 16 * rt + i) * rowT;
#pragma unroll
        for (int ct = 0; ct < NQ; ++ct) {
          y[rt][ct][i] = ldf(rs_ct, vcol(ct), so);
          y[2 + rt][ct][i] = ldf(rs_ct, vcol(ct), so + C * rowT);
        }
      }
  };
  // the diffusion-step term's share of the dilated conv, rows of this lane (gate: cw + 16 rt + i, filter: C + ..): dA = D_0 + D_1 + D_2 for every
  // frame, requested with the conditioner term.  On the first / last tile of a row the frames within a dilation of the sequence's start / end
  // lack a tap (the reference pads x + d with zeros): its D_0 / D_2 is taken out again where the accumulators are initialised, fetched THERE —
  // one exposed L2 round trip per layer on two tiles of a row; requested early like dA, its 16 registers were live on every tile from the
  // image phase to the next layer's top and put 30 spilled registers back into the layer loop (164.5 k against 175.8 k mel-frames/s)
  const bool first_tile = !has_left, last_tile = t0 + NT + HALO > T;   // (a tile whose right neighbour holds fewer frames than a dilation lacks right taps too)
  f32x4 dA[4];   // [2 half + rt]
  auto dconv_request = [&](int l) {
    const rsrc_t rs_d = mk_rsrc(p.dconv + ((long long)tb * L + l) * (4 * 2 * C), 4 * 2 * C * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) dA[2 * h + rt] = ldf4(rs_d, cw * 4, (3 * 2 * C + h * C + 16 * rt) * 4);
  };
  // y = (cond + D) x s1 on arrival of the requested terms (their first use)
  auto acc_init = [&](int l, float s1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 ds = dA[k] * s1;
#pragma unroll
      for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) y[k][ct][i] = __builtin_fmaf(y[k][ct][i], s1, ds[i]);
    }
    if (first_tile || last_tile) {   // (uniform)
      const int dil = 1 << (l % p.cycle);
      const rsrc_t rs_d = mk_rsrc(p.dconv + ((long long)tb * L + l) * (4 * 2 * C), 4 * 2 * C * 4);
#pragma unroll 1
      for (int side = 0; side < 2; ++side) {   // 0: the left end (tap 0), 1: the right end (tap 2); a row of one tile has both
        if (side == 0 ? !first_tile : !last_tile) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            const f32x4 de = ldf4(rs_d, cw * 4, ((side ? 2 : 0) * 2 * C + h * C + 16 * rt) * 4);
#pragma unroll
            for (int ct = 0; ct < NQ; ++ct) {
              const int f = col0 + 16 * ct;
              const float m = (side == 0 ? f < dil : f + dil >= T) ? -s1 : 0.f;
#pragma unroll
              for (int i = 0; i < 4; ++i) y[2 * h + rt][ct][i] = __builtin_fmaf(m, de[i], y[2 * h + rt][ct][i]);
            }
          }
      }
    }
  };
  // image core (frames t0 .. t0+NT-1, this wave's 32 channels) = hi / lo of x, zero beyond T (the conv pads with zeros)
  // xr16 = 16 x (the caller folds the factor into the product that forms x); a tile whose 64 frames all lie inside the utterance — all but the
  // last of a row — skips the masks of the frames beyond T
  const bool all_cols = t0 + NT <= T;   // (wave-uniform)
  auto write_core = [&](const float (&xr16)[NQ][2][4]) {
    unsigned worst = 0;
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float v0 = xr16[ct][rt][0], v1 = xr16[ct][rt][1], v2 = xr16[ct][rt][2], v3 = xr16[ct][rt][3];
        // (columns beyond T hold other rows' values — they are not clamped, only masked below: they must not raise a range event either)
        const unsigned w4 = max(max(absbits(v0), absbits(v1)), max(absbits(v2), absbits(v3)));
        worst = max(worst, all_cols || col_ok(ct) ? w4 : 0u);
        const HiLo s0 = split2(v0, v1);
        const HiLo s1_ = split2(v2, v3);
        u32x2 wh = u32x2{s0.hi, s1_.hi}, wl = u32x2{s0.lo, s1_.lo};
        if (!all_cols && !col_ok(ct)) { wh = u32x2{0u, 0u}; wl = u32x2{0u, 0u}; }
        char* dst = xs + (HALO + n16 + fo(ct)) * ROWB + (cw + 16 * rt) * 2;
        *reinterpret_cast<u32x2*>(dst) = wh;
        *reinterpret_cast<u32x2*>(dst + XP) = wl;
      }
    range_check(worst);
  };

  // ---- layer 0: x from HBM (the whole input exists, halo included) ------------------------------------------------------
  float x0[NQ][2][4];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x0[ct][rt][i] = XSCALE * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vcol(ct), (32 * wave + 16 * rt + i) * rowT, BSG_CQ_AUX));   // read once per step
        sk[ct][rt][i] = 0.f;
      }
  {
    const int hf = tid & 15, hc = tid >> 4;   // 16 halo frames x 32 chunks of 8 channels
    const int th = hf < 8 ? t0 - HALO + hf : t0 + NT - 8 + hf;
    const int hrow = hf < 8 ? hf : NT + hf;
    const bool hok = th >= 0 && th < T;
    float hv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) hv[k] = XSCALE * ldf(rs_x, hok ? ((8 * hc + k) * T + th) * 4 : 0, 0);
    range_check(max(max(max(absbits(hv[0]), absbits(hv[1])), max(absbits(hv[2]), absbits(hv[3]))),
                    max(max(absbits(hv[4]), absbits(hv[5])), max(absbits(hv[6]), absbits(hv[7])))));
    const HiLo h0 = split2(hv[0], hv[1]), h1 = split2(hv[2], hv[3]), h2 = split2(hv[4], hv[5]), h3 = split2(hv[6], hv[7]);
    u32x4 wh = u32x4{h0.hi, h1.hi, h2.hi, h3.hi}, wl = u32x4{h0.lo, h1.lo, h2.lo, h3.lo};
    if (!hok) { wh = u32x4{0u, 0u, 0u, 0u}; wl = u32x4{0u, 0u, 0u, 0u}; }
    *reinterpret_cast<u32x4*>(xs + hrow * ROWB + hc * 16) = wh;
    *reinterpret_cast<u32x4*>(xs + XP + hrow * ROWB + hc * 16) = wl;
  }
  btab[tid] = p.bias_out[tid];
  cond_request(0);
  dconv_request(0);
  __syncthreads();
  write_core(x0);
  // weight ring, shared by both GEMMs and running through them (mfma_pipe_q); layer 0's first two k-steps (GEMM1 starts with the centre
  // tap: kmapq) are requested here
  f16x8 A[NS][8];
  auto prefetch_a1 = [&](int l) {
    const rsrc_t rs = mk_rsrc(p.apack1q + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int kr = kmapq<8>(k) * QKSB;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        A[k][2 * rt] = lda8(rs, vfrag, sa[rt] + kr);
        A[k][2 * rt + 1] = lda8(rs, vfrag, sa[rt] + kr + QPLB);
      }
    }
  };
  prefetch_a1(0);

#define STK_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (p.stamps && lane == 0 && (wave == 0 || p.stamp_mode >= 4)) {                                              \
      unsigned long long sv_ = __builtin_amdgcn_s_memrealtime();                                                  \
      if (p.stamp_mode >= 2) sv_ = (sv_ & 0xffffffffull) | ((unsigned long long)__builtin_amdgcn_s_memtime() << 32); /* + shader cycles */ \
      /* modes 4, 5: every wave stamps, [tile][layer][wave][8] */                                                 \
      p.stamps[DIAG == 3 ? (((long long)tile_id * L + l) * 8 + wave) * 32 + (i)                                    \
               : p.stamp_mode >= 4 ? (((long long)tile_id * L + l) * 8 + wave) * 8 + (i) : ((long long)tile_id * L + l) * 8 + (i)] = sv_; \
    }                                                                                                             \
  } while (0)
  if (p.stamps && tid == 0 && p.stamp_mode < 4) p.stamps[((long long)tile_id * L + L - 1) * 8 + 6] = __builtin_amdgcn_s_memtime();   // ... and start of the first (tools/stack_stamps.py)
  if (p.clk && tile_id == 0 && tid == 0) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int dil = 1 << (l % p.cycle);
    const rsrc_t rs_a1 = mk_rsrc(p.apack1q + (long long)l * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);
    const rsrc_t rs_a2 = mk_rsrc(p.apack2q + (long long)l * (2 * 2 * C * C), 2 * 2 * C * C * 2);
    const rsrc_t rs_a1n = mk_rsrc(p.apack1q + (long long)(l + 1 < L ? l + 1 : 0) * (2 * 2 * C * 3 * C), 2 * 2 * C * 3 * C * 2);   // the next layer's GEMM1
    // GEMM1's accumulators carry weights x s1 and x x 2^4: s1 / inv1 below are the products (powers of two: exact)
    const float s1 = p.h2_scale[4 * l] * XSCALE, inv1 = p.h2_scale[4 * l + 1] * XINV, s2 = p.h2_scale[4 * l + 2], inv2 = p.h2_scale[4 * l + 3];
    const float bnext = l + 1 < L ? p.bias_out[(long long)(l + 1) * (2 * C) + tid] : 0.f;
    // GEMM1 accumulates (conditioner term + D + W x) x s1: the requested terms are combined and scaled on arrival (their first use)
    acc_init(l, s1);
    if (l == 0) __syncthreads();   // layer 0: the staged image (core + halo rows); later layers: barrier (C) below covers the core rows
    STK_STAMP(0);
    // ---- GEMM1: 24 k-steps of 32.  The centre tap (8 k-steps) reads the tile's own frames only, so it runs while the neighbours' edges of
    // this layer are still in flight; the wait for them, and the copy of the halo rows, sit behind it (mid) ------------------------------
    {
      const char* xb = xs + (HALO + n16) * ROWB + q4 * 16;
      auto ldb = [&](int ks, int ct, f16x8 (&Bf)[2]) {
        const int tap = ks >> 3, kc = ks & 7;
        const char* q = xb + ((tap - 1) * dil + fo(ct)) * ROWB + kc * 64;
        Bf[0] = *reinterpret_cast<const f16x8*>(q);
        Bf[1] = *reinterpret_cast<const f16x8*>(q + XP);
      };
      auto mid = [&]() {
        if (l == 0) return;   // layer 0 staged its halo rows from HBM
        if (DIAG == 6) return;   // timing experiment (wrong results): no hand-off — no flag poll, no halo copy (the halo rows keep layer 0's), no barriers (D), (A)
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
        __syncthreads();   // (D) the polling wave has seen both flags
        if (p.stamp_mode != 3 && p.stamp_mode < 5) STK_STAMP(1);
        {
          // halo rows of this layer, both planes: rows 0..7 = the left neighbour's last 8 frames, rows NT+8..NT+15 = the right neighbour's
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
      mfma_pipe_q<8, 0, 24, 8, FAIRB, NQ, NS, DIAG>(y, A, rs_a1, rs_a2, vfrag, sa, ldb, mid, wave >> 2, p.stamp_mode >= 6, [&](int it) { STK_STAMP(16 + it); });
      if (FAIRB) __builtin_amdgcn_s_setprio(0);
    }
    STK_STAMP(3);
    // ---- gate -> zs (hi / lo of 2^10 z); GEMM2's first weights are on their way since GEMM1's last two k-steps -------------------------
    const float rs2x = inv2 * 0.70710678118654752440f * XSCALE;   // (XSCALE is a power of two: the product's rounding is the unscaled one's)
    const float gcg = -1.44269504088896340736f * inv1, gcf = -2.88539008177792681472f * inv1, glim = 15.0f * s1;
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x2 z01, z23;
        if (DIAG == 5) {
          // timing experiment (wrong results): sigmoid x tanh costs NOTHING — a clamped product of the two halves (2 instructions per value
          // instead of 2 exponentials, a reciprocal and ~12 others), values of the gate's magnitude
          auto cheap = [&](float a, float c) { return __builtin_amdgcn_fmed3f(a * inv1 * (c * inv1) * 0.25f, -1.0f, 1.0f) * ZSCALE; };
          z01 = f32x2{cheap(y[rt][ct][0], y[2 + rt][ct][0]), cheap(y[rt][ct][1], y[2 + rt][ct][1])};
          z23 = f32x2{cheap(y[rt][ct][2], y[2 + rt][ct][2]), cheap(y[rt][ct][3], y[2 + rt][ct][3])};
        } else {
          z01 = gate2_scaled(f32x2{y[rt][ct][0], y[rt][ct][1]}, f32x2{y[2 + rt][ct][0], y[2 + rt][ct][1]}, gcg, gcf, glim, ZSCALE);   // 2^10 z from the raw (scaled) accumulators
          z23 = gate2_scaled(f32x2{y[rt][ct][2], y[rt][ct][3]}, f32x2{y[2 + rt][ct][2], y[2 + rt][ct][3]}, gcg, gcf, glim, ZSCALE);
        }
        const HiLo s0 = split2(z01[0], z01[1]), s1_ = split2(z23[0], z23[1]);
        char* dst = zs + (n16 + fo(ct)) * ROWB + (cw + 16 * rt) * 2;
        *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
        *reinterpret_cast<u32x2*>(dst + ZP) = u32x2{s0.lo, s1_.lo};
      }
    }
    // residual rows start from (x + b_out) x s2', skip rows from b_out x s2' (the accumulators of GEMM1 are free now); x = hi + lo from the
    // image rows this lane wrote (the sum is exact in fp32)
    {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const f32x4 br = *reinterpret_cast<const f32x4*>(btab + cw + 16 * rt), bs = *reinterpret_cast<const f32x4*>(btab + C + cw + 16 * rt);
#pragma unroll
        for (int ct = 0; ct < NQ; ++ct) {
          const char* src = xs + (HALO + n16 + fo(ct)) * ROWB + (cw + 16 * rt) * 2;
          const u32x2 hp = *reinterpret_cast<const u32x2*>(src), lp = *reinterpret_cast<const u32x2*>(src + XP);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // (hi + lo) / 16 + b, times s2: v_fma_mix_f32 adds the two halves straight from the packed pairs (exact, as the two conversions and
            // the addition it replaces), and s2 — a power of two — folds into the second fused multiply-add: 2 instructions per value instead of 5
            y[rt][ct][i] = __builtin_fmaf(mix_add(hp[i >> 1], lp[i >> 1], i & 1), XINV * s2, br[i] * s2);
            y[2 + rt][ct][i] = bs[i] * s2;
          }
        }
      }
    }
    __syncthreads();   // (B) zs complete; every wave is done reading xs and this layer's biases
    btab[tid] = bnext;
    STK_STAMP(4);
    // ---- GEMM2: 8 k-steps of 32; y[0..1] = residual rows, y[2..3] = skip rows.  (Measured and not kept, profiles/r05_q_not_kept.txt: waves
    // 0..3 — the older wave of each SIMD pair, which finish GEMM1 thousands of cycles before their partners — running GEMM2's k-steps over
    // their own z before barrier (B), under the partners' gate: the partners' gate then takes 6.6k cycles instead of 3.3-4.3k, because the
    // matrix instructions of the older wave win the vector issue; 164.4 k against 166.8 k mel-frames/s) ----------------------------------
    {
      const char* zb = zs + n16 * ROWB + q4 * 16;
      auto ldb = [&](int ks, int ct, f16x8 (&Bf)[2]) {
        const char* q = zb + fo(ct) * ROWB + ks * 64;
        Bf[0] = *reinterpret_cast<const f16x8*>(q);
        Bf[1] = *reinterpret_cast<const f16x8*>(q + ZP);
      };
      mfma_pipe_q<0, 8, 8, 0, FAIRB, NQ, NS, DIAG>(y, A, rs_a2, rs_a1n, vfrag, sa, ldb, [] {}, wave >> 2, p.stamp_mode >= 6, [&](int it) { STK_STAMP(8 + it); });
      if (FAIRB) __builtin_amdgcn_s_setprio(0);
    }
    if (DIAG == 3) STK_STAMP(12);
    float xn[NQ][2][4];   // the new x: lives from here to write_core() only (in the registers the operand fragments occupy inside the GEMM loops)
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xn[ct][rt][i] = y[rt][ct][i] * rs2x;         // 16 (x + residual) / sqrt(2), net.py:78: un-scaling, 1 / sqrt(2) and the image's 2^4 in one factor
          sk[ct][rt][i] += y[2 + rt][ct][i] * inv2;
        }
    STK_STAMP(5);
    if (l + 1 == L) break;

    // ---- next layer: its conditioner term (128 KB per tile, the only HBM stream) is requested into the free accumulators NOW, so
    // that it lands under the image / publish phase; then the image, the edges for the neighbours, the flag ------------------------
    // (Requested behind barrier (C1) or behind the flag instead — so that it does not share the CU's memory pipeline with the partner wave's
    // weight stream while that wave is still in GEMM2 — the trailing wave's GEMM2 is 5k cycles shorter and the wait for the term as much
    // longer: a CU takes ~12k cycles for its 128 KB wherever they are requested, profiles/r05_q_not_kept.txt)
    cond_request(l + 1);
    if (p.stamp_mode == 3 || p.stamp_mode >= 5) STK_STAMP(1);   // diagnostics: the image phase's inner boundaries instead of GEMM1's
    write_core(xn);
    dconv_request(l + 1);   // (L2-resident: 8 KB per step and layer; lands with the conditioner term)
    if (p.stamp_mode == 3 || p.stamp_mode >= 5) STK_STAMP(2);
    __syncthreads();   // (C1) the core rows are complete (every wave wrote its 32 channels of every frame)
    STK_STAMP(6);
    if (DIAG == 6) {   // (no publish either: no edge stores, no drain, no flag, no barrier (C))
      STK_STAMP(7);
      continue;
    }
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
    // every storing wave drains its write-through stores before the flag goes up.  vmcnt counts in order: the conditioner loads of this
    // wave are older than its edge stores, so this also waits for them (they have had the image phase to land)
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
    for (int ct = 0; ct < NQ; ++ct)
      if (col_ok(ct)) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int i = 0; i < 4; ++i) stf(sk[ct][rt][i] * rdiv, rs_sk, vcol(ct), (32 * wave + 16 * rt + i) * rowT);
      }
  } else {
    // ================= fused step tail: s = skip sum / sqrt(L) -> hi / lo image rows (the conv image is dead: every wave is behind barrier
    // (B) of the last layer), then the tail of residual_stack_h2_kernel on 32-row tiles (h2_fused_tail, diffnet_h2_shared.h) ================
    {
      const float rdiv = 1.0f / sqrtf((float)L);
      unsigned worst = 0;
#pragma unroll
      for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int i = 0; i < 4; ++i) worst = max(worst, all_cols || col_ok(ct) ? absbits(sk[ct][rt][i]) : 0u);   // |s| <= |skip sum|; real frames only
      range_check(worst);
#pragma unroll
      for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const HiLo s0 = split2(sk[ct][rt][0] * rdiv, sk[ct][rt][1] * rdiv), s1_ = split2(sk[ct][rt][2] * rdiv, sk[ct][rt][3] * rdiv);
          char* dst = xs + (HALO + n16 + fo(ct)) * ROWB + (cw + 16 * rt) * 2;
          *reinterpret_cast<u32x2*>(dst) = u32x2{s0.hi, s1_.hi};
          *reinterpret_cast<u32x2*>(dst + XP) = u32x2{s0.lo, s1_.lo};
        }
    }
    h2_fused_tail<NCT>(a, p.status, xs, zs, b, t0, T, L, tid, wave, range_flag);
  }
}
#undef BSG_MFMA_Q

}  // namespace

// depth of the weight ring in k-steps of 32.  Deeper rings were built (the pipe takes any NS) and measured slower, profiles/r05_q_not_kept.txt:
// 3 k-steps on 64-frame tiles 157.5 k against 175.7 k mel-frames/s (65 spilled registers), 4 k-steps on 32-frame tiles — which have the
// registers — 63.9 against 60.4 ms per pass at B = 8: the weight stream is not waiting for its latency
constexpr int NS_DEEP_64 = 2, NS_DEEP_32 = 2;
static int ring_depth(int) { return 2; }

template <int NCT, int NS>
static int h2q_occupancy_ns() {
  int o = 0;
  const int lds = (int)h2_lds(NCT);
  if (hipFuncSetAttribute((const void*)residual_stack_q_kernel<true, false, NCT, 0, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
      hipFuncSetAttribute((const void*)residual_stack_q_kernel<true, true, NCT, 0, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)residual_stack_q_kernel<true, true, NCT, 0, NS>, 512, h2_lds(NCT)) != hipSuccess)
    return 0;
  return o;
}
// resident workgroups per CU (0 on error) of the form with `nct` units of 32 frames per workgroup (1 or 2)
int stack_h2q_occupancy(int nct) {
  const int ns = ring_depth(nct);
  if (nct == 1) return ns == 2 ? h2q_occupancy_ns<1, 2>() : h2q_occupancy_ns<1, NS_DEEP_32>();
  return ns == 2 ? h2q_occupancy_ns<2, 2>() : h2q_occupancy_ns<2, NS_DEEP_64>();
}

template <int NCT, int NS>
static int h2q_launch(const StackArgs& p, const TailArgs* tail, hipStream_t st) {
  const dim3 grid(8 * cdiv(p.n_tiles, 8)), block(512);
  const TailArgs a = tail ? *tail : TailArgs{};
  const size_t lds = h2_lds(NCT);
  static int diag = -1;   // BSG_H2Q_DIAG=1 / 2 / 3 / 4 / 5 / 6: timing experiments on the launches of 64-frame tiles (1: no weight reloads, 2: no operand reads either; 3: a stamp per pass;
                          // round 6, the buckets outside the GEMM phases: 4: no conditioner-term loads, 5: a two-instruction gate, 6: no hand-off and no publish — all but 3: wrong results)
  if (diag < 0) { const char* e = getenv("BSG_H2Q_DIAG"); diag = e ? atoi(e) : 0; }
  if constexpr (NCT == 2) {
    if (diag) {
      auto go = [&](auto kt, auto kn) {
        (void)hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)kn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (tail) hipLaunchKernelGGL(kt, grid, block, lds, st, p, a);
        else hipLaunchKernelGGL(kn, grid, block, lds, st, p, a);
      };
      if (diag == 1) go(residual_stack_q_kernel<true, true, 2, 1, NS>, residual_stack_q_kernel<true, false, 2, 1, NS>);
      else if (diag == 2) go(residual_stack_q_kernel<true, true, 2, 2, NS>, residual_stack_q_kernel<true, false, 2, 2, NS>);
      else if (diag == 4) go(residual_stack_q_kernel<true, true, 2, 4, NS>, residual_stack_q_kernel<true, false, 2, 4, NS>);
      else if (diag == 5) go(residual_stack_q_kernel<true, true, 2, 5, NS>, residual_stack_q_kernel<true, false, 2, 5, NS>);
      else if (diag == 6) go(residual_stack_q_kernel<true, true, 2, 6, NS>, residual_stack_q_kernel<true, false, 2, 6, NS>);
      else go(residual_stack_q_kernel<true, true, 2, 3, NS>, residual_stack_q_kernel<true, false, 2, 3, NS>);
      BSG_LAUNCH_CHECK();
      return BSG_OK;
    }
  }
  static int fair = -1;   // BSG_H2Q_FAIR=0: no time-sliced issue priority between the two waves of a SIMD
  if (fair < 0) { const char* e = getenv("BSG_H2Q_FAIR"); fair = e ? atoi(e) : 1; }
  if (!fair) {
    static bool attr = false;
    if (!attr) {
      BSG_HIP(hipFuncSetAttribute((const void*)residual_stack_q_kernel<false, false, NCT, 0, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      BSG_HIP(hipFuncSetAttribute((const void*)residual_stack_q_kernel<false, true, NCT, 0, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr = true;
    }
    if (tail) hipLaunchKernelGGL((residual_stack_q_kernel<false, true, NCT, 0, NS>), grid, block, lds, st, p, a);
    else hipLaunchKernelGGL((residual_stack_q_kernel<false, false, NCT, 0, NS>), grid, block, lds, st, p, a);
  } else if (tail) hipLaunchKernelGGL((residual_stack_q_kernel<true, true, NCT, 0, NS>), grid, block, lds, st, p, a);
  else hipLaunchKernelGGL((residual_stack_q_kernel<true, false, NCT, 0, NS>), grid, block, lds, st, p, a);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

// the arguments of launch_residual_stack_h2 (diffnet_h2.hip); p.apack1q / p.apack2q must hold the 16-row weight fragments
int launch_residual_stack_h2q(const StackArgs& p, const TailArgs* tail, hipStream_t st, int nct) {
  BSG_REQUIRE(p.apack1q && p.apack2q && p.dconv, "16-row stack launch: the 16-row weight fragments / the step-term table are missing");
  const int ns = ring_depth(nct);
  if (nct == 1) return ns == 2 ? h2q_launch<1, 2>(p, tail, st) : h2q_launch<1, NS_DEEP_32>(p, tail, st);
  return ns == 2 ? h2q_launch<2, 2>(p, tail, st) : h2q_launch<2, NS_DEEP_64>(p, tail, st);
}

}  // namespace bsg
