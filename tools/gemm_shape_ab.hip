// Which MFMA shape holds the higher clock under the stack launch's GEMM1 loop?  (MI355X_MICROARCH.md "DVFS give-back" item 7: in bare bf16
// loops on random data the 16x16x32 shape delivered ~1.15 x the FLOP/s of 32x32x16 at equal cycles per FLOP.)
// The loop of csrc/diffnet_h2.hip mfma_pipe_h2 — split-fp16 products (lo hi, hi hi, hi lo), weight fragments streamed from L2 through a
// register ring, operand fragments re-read from the LDS image by every wave — on the same output tile per wave (64 rows x 64 frames), once
// with v_mfma_f32_32x32x16_f16 (2 x 2 tiles, 12 MFMAs of 32 clocks per 16-deep k-step) and once with v_mfma_f32_16x16x32_f16 (4 x 4 tiles,
// 48 MFMAs of 16 clocks per 32-deep k-step), 256 workgroups of 8 waves, random operands with realistic hi / lo planes.  Switches take the
// weight stream / the LDS reads out (operands stay in registers) to rank what the clock pays for.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_shape_ab tools/gemm_shape_ab.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using rsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int C = 256, HALO = 8, ROWB = 2 * C + 16, NT = 64;
constexpr int XP = (NT + 2 * HALO) * ROWB;   // bytes per plane of the image
constexpr int KTOT = 3 * C;                  // GEMM1: K = 768
constexpr int LAYB = 2 * 2 * C * KTOT * 2;   // bytes of packed weights per layer (hi + lo): 1.57 MB

__device__ __forceinline__ rsrc_t mk_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f16x8 lda8(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int SHAPE, bool STREAM_A, bool READ_B, bool ROTATE, bool PERM>
__global__ __launch_bounds__(512, 2) void gemm1_loop(const char* __restrict__ w, const char* __restrict__ img, float* __restrict__ out,
                                                     unsigned long long* __restrict__ clk, int layers, int wlayers) {
  extern __shared__ __attribute__((aligned(16))) char xs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * XP / 16; i += 512) reinterpret_cast<u32x4*>(xs)[i] = reinterpret_cast<const u32x4*>(img)[i];
  __syncthreads();
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  const int vfrag = lane * 16;
  float sum = 0.f;
  if constexpr (SHAPE == 32) {
    constexpr int NSH = 4, PLB = 16 * 1024, KSB2 = 2 * PLB, NKS = KTOT / 16;
    const int l31 = lane & 31, lh = lane >> 5;
    const int sa0 = wave * 1024, sa1 = (8 + wave) * 1024;
    f32x16 c0[2], c1[2];
    for (int ct = 0; ct < 2; ++ct)
      for (int r = 0; r < 16; ++r) c0[ct][r] = c1[ct][r] = 0.f;
    const char* xb = xs + (HALO + l31) * ROWB + lh * 16;
    const int rot = ROTATE ? (int)((blockIdx.x * 7u) % 48u) : 0;   // workgroups stream the layer's k-steps out of phase, as drifting tiles do
    auto kro = [&](int k) { const int r = k + rot; return r >= 48 ? r - 48 : r; };
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
      const rsrc_t rs = mk_rsrc(w + (long long)(l % wlayers) * LAYB, LAYB);
      const int dil = 1 << (l & 3);
      auto ldb = [&](int ks, f16x8(&Bf)[4]) {
        const int kk = kro(ks);
        const int tap = kk >> 4, kc = kk & 15;
        const char* q = xb + ((tap - 1) * dil) * ROWB + kc * 32;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          Bf[2 * ct] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB);
          Bf[2 * ct + 1] = *reinterpret_cast<const f16x8*>(q + 32 * ct * ROWB + XP);
        }
      };
      f16x8 A[NSH][4];
#pragma unroll
      for (int k = 0; k < NSH; ++k) {
        A[k][0] = lda8(rs, vfrag, sa0 + kro(k) * KSB2);
        A[k][1] = lda8(rs, vfrag, sa0 + kro(k) * KSB2 + PLB);
        A[k][2] = lda8(rs, vfrag, sa1 + kro(k) * KSB2);
        A[k][3] = lda8(rs, vfrag, sa1 + kro(k) * KSB2 + PLB);
      }
      f16x8 B[2][4];
      ldb(0, B[0]);
      const int last = NKS - 1;
#pragma unroll 1
      for (int ks = 0; ks < NKS; ks += NSH) {
#pragma unroll
        for (int s = 0; s < NSH; ++s) {
          const int in = ks + s + 1 <= last ? ks + s + 1 : last;
          if (READ_B) ldb(in, B[(s + 1) & 1]);
          const f16x8(&Bc)[4] = B[READ_B ? (s & 1) : 0];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            c0[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][1], Bc[2 * ct], c0[ct], 0, 0, 0);
            c1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][3], Bc[2 * ct], c1[ct], 0, 0, 0);
          }
          if (STREAM_A) {
            const int ir = kro(ks + s + NSH <= last ? ks + s + NSH : last);
            A[s][1] = lda8(rs, vfrag, sa0 + ir * KSB2 + PLB);
            A[s][3] = lda8(rs, vfrag, sa1 + ir * KSB2 + PLB);
            const int sp = (s + NSH - 1) % NSH;
            const int ip = kro(ks + s - 1 + NSH <= last ? ks + s - 1 + NSH : last);
            A[sp][0] = lda8(rs, vfrag, sa0 + ip * KSB2);
            A[sp][2] = lda8(rs, vfrag, sa1 + ip * KSB2);
          }
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            c0[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][0], Bc[2 * ct], c0[ct], 0, 0, 0);
            c1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][2], Bc[2 * ct], c1[ct], 0, 0, 0);
          }
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            c0[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][0], Bc[2 * ct + 1], c0[ct], 0, 0, 0);
            c1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][2], Bc[2 * ct + 1], c1[ct], 0, 0, 0);
          }
          if (STREAM_A && READ_B) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[ct][r] *= 1e-4f; c1[ct][r] *= 1e-4f; }
    }
    for (int ct = 0; ct < 2; ++ct)
      for (int r = 0; r < 16; ++r) sum += c0[ct][r] + c1[ct][r];
  } else if constexpr (SHAPE == 16) {
    // 16x16x32: lane l holds A[row l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][column l & 15]; D column l & 15, rows 4 (l >> 4) + r.
    // Weights: per 32-deep k-step a hi and a lo slab of 32 row tiles x 1 KB.  A wave owns row tiles 2w, 2w+1 (gate) and 16+2w, 16+2w+1.
    // LDS reads are conflict-free with column n of a tile reading frame pi(n) = n ^ (n < 8 ? 4 : 0) and the four 16-byte chunks g of a
    // k-step placed at 16-byte units (g & 1) * 8 + (g >> 1) + 2 (ks & 3) + 16 (ks >> 2) of the row (ds_read_b128 lane groups, MI355X_MICROARCH.md)
    constexpr int NKS = KTOT / 32, PLB = 32 * 1024, KSB2 = 2 * PLB;   // 24 k-steps of 64 KB
    const int n = lane & 15, g = lane >> 4;
    const int pin = PERM ? n ^ ((n < 8) ? 4 : 0) : n;
    const int rot = ROTATE ? (int)((blockIdx.x * 7u) % 24u) : 0;
    auto kro = [&](int k) { const int r = k + rot; return r >= 24 ? r - 24 : r; };
    int sa[4];
    sa[0] = (2 * wave) * 1024; sa[1] = (2 * wave + 1) * 1024; sa[2] = (16 + 2 * wave) * 1024; sa[3] = (17 + 2 * wave) * 1024;
    f32x4 acc[4][4];
    for (int rt = 0; rt < 4; ++rt)
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xb = xs + (HALO + pin) * ROWB + (PERM ? (g & 1) * 8 + (g >> 1) : g) * 16;
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
      const rsrc_t rs = mk_rsrc(w + (long long)(l % wlayers) * LAYB, LAYB);
      const int dil = 1 << (l & 3);
      // item i = 4 ks + ct: the hi and lo fragment of column tile ct at k-step ks
      auto ldb = [&](int item, f16x8(&Bf)[2]) {
        const int ks = kro(item >> 2), ct = item & 3;
        const int tap = ks >> 3, kc = ks & 7;
        const char* q = xb + ((tap - 1) * dil + 16 * ct) * ROWB + (PERM ? 2 * (kc & 3) + 16 * (kc >> 2) : 4 * kc) * 16;
        Bf[0] = *reinterpret_cast<const f16x8*>(q);
        Bf[1] = *reinterpret_cast<const f16x8*>(q + XP);
      };
      f16x8 A[2][8];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          A[k][2 * rt] = lda8(rs, vfrag, sa[rt] + kro(k) * KSB2);
          A[k][2 * rt + 1] = lda8(rs, vfrag, sa[rt] + kro(k) * KSB2 + PLB);
        }
      f16x8 B[2][2];
      ldb(0, B[0]);
      const int last_item = 4 * NKS - 1, last = NKS - 1;
#pragma unroll 1
      for (int ks = 0; ks < NKS; ks += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const int item = 4 * (ks + s) + ct;
            const int nx = item + 1 <= last_item ? item + 1 : last_item;
            if (READ_B) ldb(nx, B[(ct + 1) & 1]);
            const f16x8(&Bc)[2] = B[READ_B ? (ct & 1) : 0];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * rt + 1], Bc[0], acc[rt][ct], 0, 0, 0);
            if (STREAM_A && ct == 3) {
              const int ir = kro(ks + s + 2 <= last ? ks + s + 2 : last);
#pragma unroll
              for (int rt = 0; rt < 4; ++rt) A[s][2 * rt + 1] = lda8(rs, vfrag, sa[rt] + ir * KSB2 + PLB);
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * rt], Bc[0], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * rt], Bc[1], acc[rt][ct], 0, 0, 0);
            if (STREAM_A && ct == 3) {
              const int ir = kro(ks + s + 2 <= last ? ks + s + 2 : last);
#pragma unroll
              for (int rt = 0; rt < 4; ++rt) A[s][2 * rt] = lda8(rs, vfrag, sa[rt] + ir * KSB2);
            }
            if (READ_B) {
              // the next item's two LDS reads inside the first MFMA group
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (STREAM_A && ct == 3) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] *= 1e-4f;
    }
    for (int rt = 0; rt < 4; ++rt)
      for (int ct = 0; ct < 4; ++ct) sum += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
  } else {
    // SHAPE 17: 16x16x32 in HALF steps.  A k-step runs the gate row tiles over all column tiles, then the filter row tiles: a half's four weight
    // fragments are dead after its 24 MFMAs and are reloaded there (1.5 k-steps ahead of their next use instead of 1.0); the 8 operand
    // fragments of a k-step stay in registers for both halves and are replaced one column-tile pair at a time during the second
    constexpr int NKS = KTOT / 32, PLB = 32 * 1024, KSB2 = 2 * PLB;
    const int n = lane & 15, g = lane >> 4;
    const int rot = ROTATE ? (int)((blockIdx.x * 7u) % 24u) : 0;
    auto kro = [&](int k) { const int r = k + rot; return r >= 24 ? r - 24 : r; };
    int sa[4];
    sa[0] = (2 * wave) * 1024; sa[1] = (2 * wave + 1) * 1024; sa[2] = (16 + 2 * wave) * 1024; sa[3] = (17 + 2 * wave) * 1024;
    f32x4 acc[4][4];
    for (int rt = 0; rt < 4; ++rt)
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xb = xs + (HALO + n) * ROWB + g * 16;
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
      const rsrc_t rs = mk_rsrc(w + (long long)(l % wlayers) * LAYB, LAYB);
      const int dil = 1 << (l & 3);
      f16x8 Bh[4], Bl[4];
      auto ldb = [&](int ksx, int ct) {
        const int ks = kro(ksx);
        const int tap = ks >> 3, kc = ks & 7;
        const char* q = xb + ((tap - 1) * dil + 16 * ct) * ROWB + kc * 64;
        Bh[ct] = *reinterpret_cast<const f16x8*>(q);
        Bl[ct] = *reinterpret_cast<const f16x8*>(q + XP);
      };
      f16x8 A[2][8];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          A[k][2 * rt] = lda8(rs, vfrag, sa[rt] + kro(k) * KSB2);
          A[k][2 * rt + 1] = lda8(rs, vfrag, sa[rt] + kro(k) * KSB2 + PLB);
        }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) ldb(0, ct);
      const int last = NKS - 1;
#pragma unroll 1
      for (int ks = 0; ks < NKS; ks += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int ir = kro(ks + s + 2 <= last ? ks + s + 2 : last);
          const int kn = ks + s + 1 <= last ? ks + s + 1 : last;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
              const int c0 = 2 * cp, c1 = 2 * cp + 1, r0 = 2 * h, r1 = 2 * h + 1;
              acc[r0][c0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r0 + 1], Bh[c0], acc[r0][c0], 0, 0, 0);
              acc[r1][c0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r1 + 1], Bh[c0], acc[r1][c0], 0, 0, 0);
              acc[r0][c1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r0 + 1], Bh[c1], acc[r0][c1], 0, 0, 0);
              acc[r1][c1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r1 + 1], Bh[c1], acc[r1][c1], 0, 0, 0);
              acc[r0][c0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r0], Bh[c0], acc[r0][c0], 0, 0, 0);
              acc[r1][c0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r1], Bh[c0], acc[r1][c0], 0, 0, 0);
              acc[r0][c1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r0], Bh[c1], acc[r0][c1], 0, 0, 0);
              acc[r1][c1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r1], Bh[c1], acc[r1][c1], 0, 0, 0);
              acc[r0][c0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r0], Bl[c0], acc[r0][c0], 0, 0, 0);
              acc[r1][c0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r1], Bl[c0], acc[r1][c0], 0, 0, 0);
              acc[r0][c1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r0], Bl[c1], acc[r0][c1], 0, 0, 0);
              acc[r1][c1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][2 * r1], Bl[c1], acc[r1][c1], 0, 0, 0);
              if (h == 1 && READ_B) { ldb(kn, c0); ldb(kn, c1); }
              if (cp == 1 && STREAM_A) {
                A[s][2 * r0 + 1] = lda8(rs, vfrag, sa[r0] + ir * KSB2 + PLB);
                A[s][2 * r1 + 1] = lda8(rs, vfrag, sa[r1] + ir * KSB2 + PLB);
                A[s][2 * r0] = lda8(rs, vfrag, sa[r0] + ir * KSB2);
                A[s][2 * r1] = lda8(rs, vfrag, sa[r1] + ir * KSB2);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] *= 1e-4f;
    }
    for (int rt = 0; rt < 4; ++rt)
      for (int ct = 0; ct < 4; ++ct) sum += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
  }
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

static unsigned short f2h(float v) {
  _Float16 h = (_Float16)v;
  unsigned short u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}
static float h2f(unsigned short u) {
  _Float16 h;
  __builtin_memcpy(&h, &u, 2);
  return (float)h;
}

template <int SHAPE, bool SA, bool RB, bool RO = true, bool PE = true>
static void run(const char* label, const char* w, const char* img, float* out, unsigned long long* clk, int wlayers, int layers, int launches) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm1_loop<SHAPE, SA, RB, RO, PE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * XP);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm1_loop<SHAPE, SA, RB, RO, PE>), dim3(256), dim3(512), 2 * XP, 0, w, img, out, clk, layers, wlayers);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((gemm1_loop<SHAPE, SA, RB, RO, PE>), dim3(256), dim3(512), 2 * XP, 0, w, img, out, clk, layers, wlayers);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int i = 0; i < 256; ++i) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  std::sort(mhz.begin(), mhz.end());
  const double us_layer = ms * 1e3 / ((double)launches * layers);
  const double mfma_clk = 36864.0;   // matrix cycles per SIMD and layer of GEMM1, either shape
  printf("%-34s %8.2f us / layer   clock %5.0f MHz (p10 %5.0f p90 %5.0f)   pipe busy %.3f   %6.1f TFLOP/s executed\n", label, us_layer, mhz[128],
         mhz[25], mhz[230], mfma_clk / (us_layer * mhz[128]), 3.0 * 2.0 * 512 * 768 * 64 * 256 / us_layer * 1e-6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int wlayers = 20;
  const int layers = argc > 1 ? atoi(argv[1]) : 1000, launches = argc > 2 ? atoi(argv[2]) : 40;
  // weights: normal(0, 1/16) x 2^13-ish scale as in the product (max |w| in [2^13, 2^14)); hi = fp16(v), lo = fp16(v - hi); the fragment
  // order does not matter for timing, the value distribution does
  // two weight buffers with the same values: [k-step of 16][hi slab, lo slab of 16 KB] for the 32-shape, [k-step of 32][hi, lo slab of 32 KB]
  // for the 16-shape (the fragment order inside a slab does not matter for timing, the hi / lo value distribution per operand does)
  std::vector<unsigned short> wh((size_t)wlayers * LAYB / 2), wq((size_t)wlayers * LAYB / 2);
  srand(1);
  auto rnd = [] { return (rand() + 0.5f) / ((float)RAND_MAX + 1.0f); };
  auto gauss = [&] { return sqrtf(-2.0f * logf(rnd())) * cosf(6.2831853f * rnd()); };
  for (int l = 0; l < wlayers; ++l)
    for (int ks = 0; ks < KTOT / 16; ++ks)
      for (int i = 0; i < 16 * 512; ++i) {
        const float v = gauss() * 2500.0f;
        const unsigned short hi = f2h(v);
        const unsigned short lo = f2h(v - h2f(hi));
        const size_t lb = (size_t)l * LAYB / 2;
        wh[lb + ((size_t)(ks * 2) * 16 * 512) + i] = hi;
        wh[lb + ((size_t)(ks * 2 + 1) * 16 * 512) + i] = lo;
        const size_t q = lb + (size_t)(ks >> 1) * 2 * 32 * 512 + (size_t)(ks & 1) * 16 * 512 + i;
        wq[q] = hi;
        wq[q + 32 * 512] = lo;
      }
  std::vector<unsigned short> im((size_t)XP);   // 2 planes x (XP / 2) halfs
  for (int i = 0; i < XP / 2; ++i) {
    const float v = gauss() * 2.0f;
    const unsigned short hi = f2h(v);
    im[i] = hi;
    im[XP / 2 + i] = f2h(v - h2f(hi));
  }
  char *w, *w16, *img;
  float* out;
  unsigned long long* clk;
  hipMalloc(&w, wh.size() * 2);
  hipMalloc(&w16, wq.size() * 2);
  hipMalloc(&img, im.size() * 2);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&clk, 512 * 8);
  hipMemcpy(w, wh.data(), wh.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w16, wq.data(), wq.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(img, im.data(), im.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<32, true, true, false>("32x32x16  stream A, LDS B, lockstep", w, img, out, clk, wlayers, layers, launches);
    run<32, true, true>("32x32x16  stream A, LDS B", w, img, out, clk, wlayers, layers, launches);
    run<16, true, true>("16x16x32  stream A, LDS B", w16, img, out, clk, wlayers, layers, launches);
    run<16, true, true, true, false>("16x16x32  stream A, LDS B natural", w16, img, out, clk, wlayers, layers, launches);
    run<17, true, true, true, false>("16x16x32 half steps stream A, LDS B", w16, img, out, clk, wlayers, layers, launches);
    run<17, false, true, true, false>("16x16x32 half steps A regs, LDS B", w16, img, out, clk, wlayers, layers, launches);
    run<32, false, true>("32x32x16  A in regs, LDS B", w, img, out, clk, wlayers, layers, launches);
    run<16, false, true>("16x16x32  A in regs, LDS B", w16, img, out, clk, wlayers, layers, launches);
    run<16, false, true, true, false>("16x16x32  A in regs, LDS B natural", w16, img, out, clk, wlayers, layers, launches);
    run<32, false, false>("32x32x16  A, B in regs", w, img, out, clk, wlayers, layers, launches);
    run<16, false, false>("16x16x32  A, B in regs", w16, img, out, clk, wlayers, layers, launches);
  }
  return 0;
}
