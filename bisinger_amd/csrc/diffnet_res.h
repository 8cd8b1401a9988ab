// Shared by the fp32 (diffnet.hip) and bf16-operand (diffnet_bf16.hip) forms of the fused residual block.
#pragma once
#include "bsg_common.h"

namespace bsg {

constexpr int C = 256;     // residual channels == encoder hidden (checked at create)
constexpr int HALO = 8;    // max dilation 2^3

struct ResArgs {
  const float* x_in;     // [B][C][T]
  float* x_out;          // [B][C][T]
  float* skip;           // [B][C][T]
  const float* condterm; // this layer's [B][2C][T]: conditioner_projection(cond) + b_cond + b_dil
  const float* dproj;    // [S][L][C] table: diffusion_projection_l(mlp(emb(step)))
  const long long* t_dev;  // [B] or null
  int t_uniform;
  const float* apack1;   // dilated conv, packed  [16][96][64][4]
  const float* apack2;   // output projection     [16][32][64][4]
  const float* apackw;   // dilated conv in Winograd F(2,3) form, packed for 16x16x4 MFMAs [4][32][16][64][4]
  const float* apackw43; // dilated conv in Winograd F(4,3) form (diffnet_f43.hip) [6][32][16][64][4]
  const unsigned short* apack1h;  // bf16 operand path: dilated conv packed for 32x32x16 bf16 MFMAs [16][48][64][8]
  const unsigned short* apack2h;  // bf16 operand path: output projection                           [16][16][64][8]
  // bf16-operand path: conditioner term and running skip sum are STORED as bf16 in channel-quad order
  // [B][rows/4][T][4] (one 8-byte access per lane = 4 consecutive channels of one frame, 256-B coalesced)
  const unsigned short* condterm_h;  // this layer's [B][2C/4][T][4]
  unsigned short* skip_h;            // [B][C/4][T][4]
  const float* bias_out; // [2C]
  int B, T, L, layer, dil, tiles_per_row;
  int first;             // layer 0: skip is stored, not accumulated
  float skip_div;        // last layer: skip_sum / sqrt(L) (net.py:126); 1 otherwise
  unsigned long long* stamps;  // diagnostic build only (STAMP = true): [workgroup][wave][8] s_memtime values
};

// arguments of the stack launches (all L layers of a group of rows in one launch, residual stream on chip):
// residual_stack_h2_kernel (diffnet_h2.hip), residual_stack_f43_kernel (diffnet_f43.hip), residual_stack_bf16_kernel (diffnet_bf16.hip)
struct StackArgs {
  const float* x_in;      // [B][C][T] in-projected x of this launch's rows
  float* skip;            // [B][C][T] output: skip sum / sqrt(L)
  unsigned short* skip_h; // bf16 form: the same as bf16 channel quads [B][C/4][T][4]
  const float* condterm;  // layer 0, this launch's rows: [B][2C][T]; + l * ct_stride for layer l
  const float* condterm_q; // 16-row stack launch: the same in channel-quad order [B][2C/4][T][4] (null: `condterm`); + l * ct_stride
  const float* dproj;     // [S][L][C]
  const float* dconv;     // 16-row stack launch: [S][L][4][2C] = W_tap d for tap 0, 1, 2 and their sum (the step term's share of the dilated conv)
  const long long* t_dev; // [B] or null
  const float* apackw;    // layer 0; + l * aw_stride
  const float* apackw43;  // F(4,3) form (diffnet_f43.hip), layer 0; + l * 6*2C*C
  const unsigned short* apack1h;     // bf16 form: dilated conv fragments, layer 0; + l * 2C*3C
  const unsigned short* apack2h;     // bf16 form: output projection fragments, layer 0; + l * 2C*C
  const unsigned short* condterm_h;  // bf16 form: conditioner term in channel-quad order, layer 0 / this launch's rows; + l * ct_stride
  const float* apack2;    // layer 0; + l * a2_stride
  const unsigned short* apack1s;  // split-fp16 form (diffnet_h2.hip): hi / lo fp16 fragments of the dilated conv, layer 0; + l * 2*2C*3C
  const unsigned short* apack2s;  // split-fp16 form: output projection, layer 0; + l * 2*2C*C
  const float* h2_scale;          // split-fp16 form: [L][4] = s1, 1/s1, s2 * 2^10, 1/(s2 * 2^10)
  const unsigned short* apack1q;  // quad form: the same weights as 16-row fragments of v_mfma_f32_16x16x32_f16, layer 0; + l * 2*2C*3C
  const unsigned short* apack2q;  // quad form: output projection, layer 0; + l * 2*2C*C
  const float* bias_out;  // layer 0; + l * 2C
  long long ct_stride;
  float* hx;              // [2 parities][n_tiles][2 sides][C][8] edge exchange
  unsigned* flags;        // [n_tiles]
  unsigned* status;       // += 1 for every spin that gave up
  int t_uniform, T, L, tiles_per_row, n_tiles, cycle;
  unsigned fbase;         // launch epoch * 64: flag value = fbase + layers published (host-side epoch: diagnostics; otherwise `epoch`)
  // Round 4: the launch epoch lives in DEVICE memory, so that a captured launch can be replayed (a replay repeats its kernel arguments):
  // epoch[0] = epoch of the launch that runs next, epoch[1] = workgroups of the running launch that are through their layers.  Every
  // workgroup takes fbase = epoch[0] * 64 at entry; the LAST one through its layer loop (all others have read the epoch long ago)
  // advances it — and, before the 32-bit flag values could come round, zeroes every flag word and starts the epochs again.
  unsigned* epoch;        // null: p.fbase
  int flag_words;         // words of `flags` to zero at a wrap (the status words behind them are not touched)
  int pflag_words;        // the same for `pflags` (part forms)
  // zeroed at an epoch wrap by WHATEVER launch wraps (round 5): the handle's part-form flags even when this launch does not wait on
  // them (p.pflags null) — a wrap inside a whole-tile launch used to leave them near 2^31 while the epochs restarted at 1, and the next part
  // launch then saw every flag 'published'
  unsigned* wrap_pflags;
  int wrap_pflag_words;
  int stamp_mode;         // diagnostics (BSG_STAMP_MODE, part forms): 1 = stamp slots 1 / 2 mark the gate phase's inner boundaries instead of GEMM1's
  int inject;             // fault injection: consumers do not wait
  // part forms of the split-fp16 launch (residual_part_h2_kernel, small batches): exchange of the P channel parts of a tile
  unsigned short* zx;     // [n_tiles][P parts][2 planes][tile frames][C/P] fp16: gated activation parts
  unsigned short* ix;     // [2 parities][n_tiles][P parts][2 planes][tile frames][C/P] fp16: image parts (core frames; neighbours read the edges)
  unsigned* pflags;       // [2][n_tiles][P parts]: [0] image flags (layers prepared), [1] z flags
  unsigned long long* stamps;   // diagnostic (bsg_diffnet_debug_stack_stamps) or null: [n_tiles][L][8] s_memrealtime at the phase boundaries
  unsigned long long* clk;      // null, or [4]: tile 0 stores s_memtime / s_memrealtime at its start and end (sustained shader clock, bench.py)
};

// Launch epoch in device memory (StackArgs::epoch): taken at entry by every workgroup ...
__device__ __forceinline__ unsigned stack_epoch_take(const StackArgs& p) {
  if (!p.epoch) return p.fbase;
  return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) * 64u;
}
// ... and advanced by the last of the launch's `n_wgs` workgroups to leave its layer loop (one thread of each calls this once)
__device__ __forceinline__ void stack_epoch_done(const StackArgs& p, unsigned fbase, int n_wgs) {
  if (!p.epoch) return;
  const unsigned done = atomicAdd(p.epoch + 1, 1u);
  if (done + 1u != (unsigned)n_wgs) return;
  unsigned next = fbase / 64u + 1u;
  if (next >= (1u << 25)) {   // flag values = epoch * 64 + layer in 32 bits: start again from zeroed flags, as nothing else runs on them now
    for (int i = 0; i < p.flag_words; ++i) p.flags[i] = 0u;
    if (p.wrap_pflags)
      for (int i = 0; i < p.wrap_pflag_words; ++i) p.wrap_pflags[i] = 0u;
    next = 1u;
  }
  __hip_atomic_store(p.epoch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p.epoch, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bf16 stack launch: 64-frame tiles, one workgroup per CU (see diffnet_bf16.hip); grid = p.n_tiles rounded up to 8
int launch_residual_stack_bf16(const StackArgs& p, hipStream_t st);
int stack_bf16_occupancy();   // resident workgroups per CU of residual_stack_bf16_kernel (0 on error)

// fp32 stack launch on the 16-bit matrix pipe: operands split exactly into hi + lo fp16 terms (diffnet_h2.hip); 64-frame tiles, one
// workgroup per CU; grid = p.n_tiles rounded up to 8
int stack_h2_occupancy(int nct);   // nct = column tiles of 32 frames per workgroup (1 or 2)
int stack_h2q_occupancy(int nct);  // the 16-row-tile form of the same launch (diffnet_h2q.hip)
// part forms on 16-row matrix tiles: `parts` workgroups (on as many CUs of one XCD) per tile of 32 nct frames, each C / parts channels:
// (4, 1) quad of a 32-frame tile, (4, 2) quad of a 64-frame tile, (2, 2) pair of a 64-frame tile (8 waves); p.n_tiles tiles -> grid of 8 parts ceil(n_tiles / 8) workgroups, all resident
int launch_residual_part_h2(const StackArgs& p, hipStream_t st, int parts, int nct);
int part_h2_occupancy(int parts, int nct);
int pack_a_frag_q(const float* src, unsigned short* out, int M, int K, int Kc, long long sm, long long sc, long long stp, const float* tab,
                  int is_gemm2, hipStream_t st);
int h2_scales(const float* const* w1, const float* const* w2, int L, unsigned* maxbits, float* tab, hipStream_t st);
int pack_a_frag_h2(const float* src, unsigned short* out, int M, int K, int Kc, long long sm, long long sc, long long stp, const float* tab,
                   int is_gemm2, hipStream_t st);

int pack_wino43(const float* w, float* out, hipStream_t st);   // [2C][C][3] -> [6][32][16][64][4]
// F(4,3) stack launch: all L layers of 64-frame tiles, one workgroup per CU (diffnet_f43.hip); grid = p.n_tiles rounded up to 8
int launch_residual_stack_f43(const StackArgs& p, hipStream_t st);
int stack_f43_occupancy();

// bf16-operand form of the residual layer (diffnet_bf16.hip); same tensors, 64-frame tiles
int launch_residual_layer_bf16(const ResArgs& a, hipStream_t st);
// fp32 [M][K] weights -> bf16 A fragments of v_mfma_f32_32x32x16_bf16 (see diffnet_bf16.hip)
int pack_a_frag_bf16(const float* src, unsigned short* out, int M, int K, int Kc, long long sm, long long sc, long long stp,
                     hipStream_t st);

// fp32 [B][rows][T] <-> bf16 channel-quad order [B][rows/4][T][4] (rows % 4 == 0)
int f32_to_quad_bf16(const float* src, unsigned short* dst, int B, int rows, int T, hipStream_t st);
int quad_bf16_to_f32(const unsigned short* src, float* dst, int B, int rows, int T, hipStream_t st);

__device__ __forceinline__ float bf16_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

// fast gate math: v_exp_f32 / v_rcp_f32 based (abs error ~2e-7, far inside the 1e-3 mel budget); the libm
// versions cost ~60 VALU instructions per element and the gate phase runs with the MFMA pipe idle.
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); }
// sigmoid(g) * tanh(f) with two exponentials and ONE reciprocal: (1 - e^{-2f}) / ((1 + e^{-g}) (1 + e^{-2f})); f is clamped to +-15
// (tanh = +-1 to 1e-13 there) so that e^{-2f} stays finite.  Relative error ~2e-7 (two v_exp_f32, one v_rcp_f32): used where the result is rounded to bf16
// anyway, and by the F(4,3) stack launch, whose rounding differs from the other fp32 kernels' already.
__device__ __forceinline__ float gate1(float g, float f) {
  f = fminf(fmaxf(f, -15.0f), 15.0f);
  const float eg = __expf(-g), ef = __expf(-2.0f * f);
  return (1.0f - ef) * __frcp_rn((1.0f + eg) * (1.0f + ef));
}

// gate1 for two elements on packed fp32 math (v_pk_mul / v_pk_add / v_pk_fma: two elements per instruction; only the exponentials and the
// reciprocals are per element): `num` x sigmoid(g / s) tanh(f / s) from accumulators that carry a scale s — cg = -log2(e) / s,
// cf = -2 log2(e) / s, lim = 15 s; s = 1, num = 1 is gate1 itself
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 gate2_scaled(f32x2 g, f32x2 f, float cg, float cf, float lim, float num) {
  f[0] = __builtin_amdgcn_fmed3f(f[0], -lim, lim);   // (one instruction for the clamp; a NaN comes out as -lim in both forms)
  f[1] = __builtin_amdgcn_fmed3f(f[1], -lim, lim);
  const f32x2 a = g * cg, b = f * cf;
  const f32x2 eg = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const f32x2 ef = f32x2{__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
  const f32x2 den = (eg + 1.0f) * (ef + 1.0f);
  const f32x2 r = f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  return (num - num * ef) * r;
}

// Buffer (SRSRC) addressing: one wave-uniform 128-bit descriptor per tensor, a per-lane 32-bit byte offset that
// is computed once, and a wave-uniform SGPR offset per access — the 32 row-strided loads of an accumulator tile
// then need ONE address VGPR instead of 32 64-bit pairs (what keeps the kernel at <= 80 VGPRs = 3 workgroups/CU).
using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ldf(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 ldf4(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void stf(float v, rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
// uniform part of acc_row(): register r of a 32x32 accumulator covers row (r&3) + 8*(r>>2) (+ 4 for lanes >= 32)
__device__ __forceinline__ constexpr int acc_row0(int r) { return (r & 3) + 8 * (r >> 2); }


}  // namespace bsg
