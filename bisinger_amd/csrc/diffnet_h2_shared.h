// Shared by the split-fp16 stack launches on 32-row matrix tiles (diffnet_h2.hip) and on 16-row matrix tiles (diffnet_h2q.hip): fragment
// types, the LDS image geometry, the hi / lo split, and the small GEMMs of the fused step tail.
#pragma once
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

__device__ __forceinline__ f16x8 lda8(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// hi / lo split of two values into two packed dwords
struct HiLo { unsigned hi, lo; };
// Round 5: 3 instructions per pair.  The compiler's form of `hi = (f16)x; lo = (f16)(x - (float)hi)` is pack-convert, two converts back, two
// subtractions, pack-convert; v_fma_mix{lo,hi}_f16 takes the f16 hi straight from the packed pair, forms x - hi in fp32 (exact) and rounds it to
// f16 into one half of the destination — the same two roundings, bit for bit (tools/split_asm_check.hip: 0 of 8.4 M values differ, subnormal
// lo terms, the range edge and signed zeros included).  BSG_ASM_SPLIT=0 at build time keeps the plain form.
#ifndef BSG_ASM_SPLIT
#define BSG_ASM_SPLIT 1
#endif
__device__ __forceinline__ HiLo split2(float a, float b) {
#if BSG_ASM_SPLIT && defined(__HIP_DEVICE_COMPILE__)
  HiLo r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.hi) : "v"(a), "v"(b));
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r.lo) : "v"(a), "v"(b), "v"(r.hi));
  return r;
#else
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  return HiLo{__builtin_bit_cast(unsigned, f16x2{ha, hb}),
              __builtin_bit_cast(unsigned, f16x2{(_Float16)(a - (float)ha), (_Float16)(b - (float)hb)})};
#endif
}

#define BSG_MFMA_H(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, ACC, 0, 0, 0)

// Small split-fp16 GEMM of the fused step tail: one row tile of 32 x NC column tiles of 32, N_KS k-steps of 16, fully unrolled; A fragments
// (hi at sa + ks*ksb, lo at + plb) in a ring of up to 8 k-steps, B fragments from `ldb(ks, Bf)`: Bf[2 nc] = hi, Bf[2 nc + 1] = lo of column
// tile nc.
template <int N_KS>
struct TailRing {
  static constexpr int R = N_KS < 8 ? N_KS : 8;
  f16x8 a[R][2];
};
// the first R k-steps' fragments: can be requested long before the GEMM runs (step_tail_h2_kernel asks for all three projections' first
// fragments while the skip sum is still on its way from HBM: each GEMM used to start with a ring fill of its own, an exposed L2 round trip)
template <int N_KS>
__device__ __forceinline__ void tail_ring_fill(TailRing<N_KS>& q, rsrc_t rs, int vfrag, int sa, int ksb, int plb) {
#pragma unroll
  for (int k = 0; k < TailRing<N_KS>::R; ++k) {
    q.a[k][0] = lda8(rs, vfrag, sa + k * ksb);
    q.a[k][1] = lda8(rs, vfrag, sa + k * ksb + plb);
  }
}
template <int NC, int N_KS, typename LDB>
__device__ __forceinline__ void tail_gemm_h2_run(f32x16 (&c)[NC], TailRing<N_KS>& q, rsrc_t rs, int vfrag, int sa, int ksb, int plb, LDB ldb) {
  constexpr int R = TailRing<N_KS>::R;
  f16x8 Bf[2][2 * NC];
  ldb(0, Bf[0]);
#pragma unroll
  for (int ks = 0; ks < N_KS; ++ks) {
    if (ks + 1 < N_KS) ldb(ks + 1, Bf[(ks + 1) & 1]);
    const f16x8(&Bc)[2 * NC] = Bf[ks & 1];
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) BSG_MFMA_H(c[nc], q.a[ks % R][0], Bc[2 * nc]);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) BSG_MFMA_H(c[nc], q.a[ks % R][0], Bc[2 * nc + 1]);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) BSG_MFMA_H(c[nc], q.a[ks % R][1], Bc[2 * nc]);
    if (ks + R < N_KS) {
      q.a[ks % R][0] = lda8(rs, vfrag, sa + (ks + R) * ksb);
      q.a[ks % R][1] = lda8(rs, vfrag, sa + (ks + R) * ksb + plb);
    }
  }
}
template <int NC, int N_KS, typename LDB>
__device__ __forceinline__ void tail_gemm_h2(f32x16 (&c)[NC], rsrc_t rs, int vfrag, int sa, int ksb, int plb, LDB ldb) {
  TailRing<N_KS> q;
  tail_ring_fill<N_KS>(q, rs, vfrag, sa, ksb, plb);
  tail_gemm_h2_run<NC, N_KS>(c, q, rs, vfrag, sa, ksb, plb, ldb);
}

// The fused step tail of the stack launches, from the point where s = skip sum / sqrt(L) sits in the image rows (hi / lo planes, core frames)
// and every wave has passed that write: skip projection + ReLU -> zs, the step's Philox normals, output projection and the sampler update
// of x (DDPM or PLMS), and the next evaluation's input projection (net.py:126-129, shallow_diffusion_tts.py:149-201).  32-row matrix tiles
// over NCT column tiles of 32 frames whatever the layout of the launch's layer loop was: everything it reads is in LDS or HBM.
template <int NCT>
__device__ __forceinline__ void h2_fused_tail(const TailArgs& a, unsigned* status, char* xs, char* zs, int b, int t0, int T, int L, int tid, int wave,
                                              int& range_flag) {
  constexpr int NT = 32 * NCT, XP = h2_xp(NCT), ZP = h2_zp(NCT);
  const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int rowT = T * 4, vfrag = lane * 16;
  const unsigned plane = (unsigned)C * T * 4;
  int vst[NCT];
  bool col_ok[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int col = t0 + 32 * ct + l31;
    col_ok[ct] = col < T;
    vst[ct] = (lh * 4 * T + col) * 4;
  }
  auto range_check = [&](unsigned worst) {
    if (__builtin_amdgcn_ballot_w64(worst >= 0x476A6000u) != 0ull) range_flag = 1;   // 60000.0f
  };
  auto absbits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };
  const int M = a.M;
  const float* tsc = a.tail_scale;   // [3][2]: scale, 1 / scale of the skip / output / input projection
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
  if (range_flag && lane == 0) atomicAdd(status + 1, 1u);   // word 1: range events (word 0: hand-off give-ups)
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
        for (int r = 0; r < 16; ++r)   // 16 MB per step at B = 16, read once by the next launch: non-temporal, like the conditioner term's loads
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(hc[ct][r] * inv, 0.f)), rs_xa, vst[ct], (32 * wave + acc_row0(r)) * rowT, 2);
      }
  }
}

}  // namespace

}  // namespace bsg
