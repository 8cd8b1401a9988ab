// PitchExtractor (mel -> f0) and the NSF harmonic source of NSF-HiFiGAN, on gfx950.
//
// Reference semantics (paths relative to /root/reference/train_bisinger):
//   modules/fastspeech/pe.py: Prenet :9-42, ConvBlock :45-78, ConvStacks :81-117, PitchExtractor :120-149
//   modules/fastspeech/tts_modules.py:194-247 (PitchPredictor), utils/pitch_utils.py:63-76 (denorm_f0)
//   modules/parallel_wavegan/models/source.py: SineGen :8-138, SourceModuleHnNSF :352-399
//   modules/hifigan/hifigan.py:111-132, :145-160 (noise_convs, source add)
//
// The CNN is three k=5 conv stacks over [B*T][256] rows: every conv is the K-segmented fp32 MFMA GEMM of gemm.hip
// (bias / ReLU / eval-BatchNorm affine / mask fused in its epilogue); LayerNorm, GroupNorm, the position
// embedding and f0 de-normalisation are small HBM-bound row kernels.  The sine source is a per-(utterance,
// harmonic) prefix sum over T*hop samples: a chunked two-level scan, then one merge kernel.
#include <math.h>

#include <vector>

#include "bsg_common.h"

namespace bsg {
namespace {

constexpr int H = 256;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// keep[row] = (sum_j |x[row][j]| != 0)                     (pe.py:29, :142)
__global__ void row_keep_kernel(const float* __restrict__ x, float* __restrict__ keep, long long rows, int width) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int j = lane; j < width; j += 64) s += fabsf(x[row * width + j]);
  s = wsum(s);
  if (lane == 0) keep[row] = s == 0.f ? 0.f : 1.f;
}

// eval-mode BatchNorm1d as a per-channel affine: y = x*scale + shift          (pe.py:18)
__global__ void bn_affine_kernel(const float* w, const float* b, const float* mean, const float* var, float eps, float* scale,
                                 float* shift, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = w[i] / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = b[i] - mean[i] * s;
}

// y = res + relu(GroupNorm_{16 ch/group}(h))   h, res, y: [B][T][H]; statistics over (T x 16 channels)   (pe.py:57,71-77,111)
__global__ __launch_bounds__(256) void groupnorm_res_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ res,
                                                            float* __restrict__ y, int T, float eps) {
  __shared__ float red[4];
  const int g = blockIdx.x, bb = blockIdx.y, tid = threadIdx.x;
  const float* hp = h + (long long)bb * T * H + g * 16;
  auto block_sum = [&](float v) {
    v = wsum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  // element e of the group = (t = e / 4, float4 e % 4)
  const long long n4 = (long long)T * 4;
  float s = 0.f;
  for (long long e = tid; e < n4; e += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(hp + (e >> 2) * H + (e & 3) * 4);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = block_sum(s) / (float)(T * 16);
  float q = 0.f;
  for (long long e = tid; e < n4; e += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(hp + (e >> 2) * H + (e & 3) * 4);
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = 1.0f / sqrtf(block_sum(q) / (float)(T * 16) + eps);
  for (long long e = tid; e < n4; e += 256) {
    const long long off = (long long)bb * T * H + (e >> 2) * H + g * 16 + (e & 3) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(h + off);
    const f32x4 r = *reinterpret_cast<const f32x4*>(res + off);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + g * 16 + (e & 3) * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + g * 16 + (e & 3) * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = r[k] + fmaxf((v[k] - mean) * rstd * wv[k] + bv[k], 0.f);
    *reinterpret_cast<f32x4*>(y + off) = o;
  }
}

// positions = cumsum(x[...,0] != 0) * (x[...,0] != 0)  (one wave per utterance), then x += alpha * table[pos]  (tts_modules.py:239-240)
__global__ void positions_kernel(const float* __restrict__ x, int* __restrict__ pos, int T) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int carry = 0;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    bool nz = false;
    if (t < T) nz = x[((long long)b * T + t) * H] != 0.f;
    const unsigned long long bal = __ballot(nz);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull)) + (nz ? 1 : 0);
    if (t < T) pos[(long long)b * T + t] = nz ? carry + pre : 0;
    carry += __popcll(bal);
  }
}
__global__ void add_positions_kernel(float* __restrict__ x, const int* __restrict__ pos, const float* __restrict__ table,
                                     const float* __restrict__ alpha, long long rows, int n_pos) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  int p = pos[row];
  p = p < n_pos ? p : n_pos - 1;
  f32x4 v = reinterpret_cast<f32x4*>(x + row * H)[lane];
  const f32x4 pe = reinterpret_cast<const f32x4*>(table + (long long)p * H)[lane];
  const float a = alpha[0];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], __fmul_rn(a, pe[e]));
  reinterpret_cast<f32x4*>(x + row * H)[lane] = v;
}

__global__ void layernorm256_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                    float* __restrict__ y, long long rows, float eps) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const f32x4 v = reinterpret_cast<const f32x4*>(x + row * H)[lane];
  const float mean = wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / H);
  const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
  const float var = wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / H);
  const float rstd = 1.0f / sqrtf(var + eps);
  const f32x4 wv = reinterpret_cast<const f32x4*>(w)[lane], bv = reinterpret_cast<const f32x4*>(b)[lane];
  f32x4 o = {d0 * rstd * wv[0] + bv[0], d1 * rstd * wv[1] + bv[1], d2 * rstd * wv[2] + bv[2], d3 * rstd * wv[3] + bv[3]};
  reinterpret_cast<f32x4*>(y + row * H)[lane] = o;
}

// f0 = 2^pred0, zeroed where unvoiced (pred1 > 0) or padded          (pitch_utils.py:63-76, pe.py:142-148)
__global__ void f0_denorm_kernel(const float* __restrict__ pred, const float* __restrict__ keep, float* __restrict__ f0,
                                 long long rows, int use_uv) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  float v = exp2f(pred[2 * i]);
  if (use_uv && pred[2 * i + 1] > 0.f) v = 0.f;
  if (keep[i] == 0.f) v = 0.f;
  f0[i] = v;
}

// ---- NSF sine source ------------------------------------------------------------------------------
struct SineArgs {
  const float* f0;        // [B][T]
  const float* rand_ini;  // [B][NH] (column 0 ignored: the fundamental starts at phase 0, source.py:56)
  const float* noise;     // [B][L][NH]
  float* sw;              // [B][NH][L] sine_waves * uv + noise
  int T, hop, NH;
  float sr, sine_amp, noise_std;
};

// one workgroup per (harmonic, utterance); thread k owns samples [k*S, (k+1)*S)
__global__ __launch_bounds__(256) void sine_source_kernel(SineArgs a) {
  __shared__ float sh[256];
  const int hh = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const long long L = (long long)a.T * a.hop;
  const long long S = (L + 255) / 256;
  const long long i0 = tid * S, i1 = (i0 + S < L) ? i0 + S : L;
  const float mult = (float)(hh + 1);
  const float ri = hh == 0 ? 0.f : a.rand_ini[(long long)b * a.NH + hh];
  const float* f0 = a.f0 + (long long)b * a.T;
  auto rad_at = [&](long long i) {
    const float f = f0[i / a.hop] * mult;                 // f0_buf = f0 * (idx + 2)          :112-116
    float r = fmodf(f / a.sr, 1.0f);                       // (f0 / sr) % 1                    :50
    if (i == 0) r += ri;                                   // initial phase noise              :57
    return r;
  };
  auto excl_scan = [&](float v) {                          // exclusive prefix over the 256 threads
    __syncthreads();
    sh[tid] = v;
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < tid; ++k) acc += sh[k];
    return acc;
  };
  // level 1: cumsum(rad) -> where it wraps past an integer                                        :67-71
  float s = 0.f;
  for (long long i = i0; i < i1; ++i) s += rad_at(i);
  float c = excl_scan(s);
  // level 2: cumsum(rad + shift), shift = -1 at every wrap; first the per-thread partial sums
  float prev = fmodf(c, 1.0f);          // tmp_over_one at i0 - 1 (c = cumsum up to i0-1)
  float run = c, part = 0.f;
  for (long long i = i0; i < i1; ++i) {
    const float r = rad_at(i);
    run += r;
    const float cur = fmodf(run, 1.0f);
    part += (i > 0 && cur - prev < 0.f) ? r - 1.0f : r;
    prev = cur;
  }
  float phase = excl_scan(part);
  prev = fmodf(c, 1.0f);
  run = c;
  float* out = a.sw + ((long long)b * a.NH + hh) * L;
  const float* nz = a.noise + (long long)b * L * a.NH + hh;
  for (long long i = i0; i < i1; ++i) {
    const float r = rad_at(i);
    run += r;
    const float cur = fmodf(run, 1.0f);
    phase += (i > 0 && cur - prev < 0.f) ? r - 1.0f : r;
    prev = cur;
    const float sine = sinf(phase * 2.0f * 3.14159265358979323846f) * a.sine_amp;      // :73-74, :119
    const float uv = f0[i / a.hop] > 0.f ? 1.f : 0.f;                                  // :42-43
    const float namp = uv * a.noise_std + (1.f - uv) * a.sine_amp / 3.f;               // :129
    out[i] = sine * uv + namp * nz[i * a.NH];                                          // :130-134
  }
}

// har[b][i] = tanh(sum_h w[h] * sw[b][h][i] + bias)                    (source.py:391)
__global__ void sine_merge_kernel(const float* __restrict__ sw, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ har, long long L, int NH) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= L) return;
  float acc = 0.f;
  for (int h = 0; h < NH; ++h) acc = fmaf(sw[((long long)b * NH + h) * L + i], w[h], acc);
  har[(long long)b * L + i] = tanhf(acc + bias[0]);
}

// x[b][c][t] += LayerNorm_c(relu(Conv1d(1 -> CC, k, stride, pad)(har)))      (hifigan.py:154-160)
template <int CC>
__global__ void nsf_source_add_kernel(float* __restrict__ x, const float* __restrict__ har, const float* __restrict__ w,
                                      const float* __restrict__ bias, int Lx, long long Lh, int k, int stride, int pad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= Lx) return;
  float v[CC];
#pragma unroll
  for (int c = 0; c < CC; ++c) v[c] = bias[c];
  const float* hp = har + (long long)b * Lh;
  for (int j = 0; j < k; ++j) {
    const long long i = (long long)t * stride - pad + j;
    if (i < 0 || i >= Lh) continue;
    const float hv = hp[i];
#pragma unroll
    for (int c = 0; c < CC; ++c) v[c] = fmaf(w[c * k + j], hv, v[c]);
  }
  float mean = 0.f;
#pragma unroll
  for (int c = 0; c < CC; ++c) { v[c] = fmaxf(v[c], 0.f); mean += v[c]; }
  mean /= (float)CC;
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < CC; ++c) { const float d = v[c] - mean; var += d * d; }
  const float rstd = 1.0f / sqrtf(var / (float)CC + 1e-5f);
#pragma unroll
  for (int c = 0; c < CC; ++c) x[((long long)b * CC + c) * Lx + t] += (v[c] - mean) * rstd;
}

__global__ void repack_conv5_kernel(const float* __restrict__ w, float* __restrict__ out, int M, int Cin, int k) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)M * Cin * k;
  if (i >= total) return;
  const int c = (int)(i % Cin);
  const int m = (int)((i / Cin) % M);
  const int tap = (int)(i / ((long long)Cin * M));
  out[i] = w[((long long)m * Cin + c) * k + tap];
}

}  // namespace

int nsf_launch_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w, const float* lin_b,
                      float* sw_tmp, float* har, int B, int T, int hop, int NH, float sr, hipStream_t st) {
  SineArgs a{f0, rand_ini, noise, sw_tmp, T, hop, NH, sr, 0.1f, 0.003f};
  hipLaunchKernelGGL(sine_source_kernel, dim3(NH, B), dim3(256), 0, st, a);
  BSG_LAUNCH_CHECK();
  const long long L = (long long)T * hop;
  hipLaunchKernelGGL(sine_merge_kernel, dim3(cdiv(L, 256), B), dim3(256), 0, st, (const float*)sw_tmp, lin_w, lin_b, har, L, NH);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

int nsf_launch_source_add(float* x, const float* har, const float* w, const float* bias, int B, int Cc, int Lx, long long Lh, int k,
                          int stride, int pad, hipStream_t st) {
  const dim3 grid(cdiv(Lx, 128), B), block(128);
  switch (Cc) {
    case 64: hipLaunchKernelGGL(nsf_source_add_kernel<64>, grid, block, 0, st, x, har, w, bias, Lx, Lh, k, stride, pad); break;
    case 32: hipLaunchKernelGGL(nsf_source_add_kernel<32>, grid, block, 0, st, x, har, w, bias, Lx, Lh, k, stride, pad); break;
    case 16: hipLaunchKernelGGL(nsf_source_add_kernel<16>, grid, block, 0, st, x, har, w, bias, Lx, Lh, k, stride, pad); break;
    case 8: hipLaunchKernelGGL(nsf_source_add_kernel<8>, grid, block, 0, st, x, har, w, bias, Lx, Lh, k, stride, pad); break;
    default: set_error("nsf: %d channels not built (64/32/16/8)", Cc); return BSG_EINVAL;
  }
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

}  // namespace bsg

// ================================================================================================
using namespace bsg;

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != BSG_OK) return _rc; \
  } while (0)

struct bsg_pitchext {
  bsg_pitchext_cfg cfg;
  Guard guard;   // this handle's range-event word and split-fp16 GEMM switch
  std::vector<float*> owned;
  float *pre_w[3], *pre_b[3], *pre_scale[3], *pre_shift[3], *pre_out_w, *pre_out_b;
  std::vector<float*> enc_w, enc_b, enc_gw, enc_gb;
  float *enc_in_w, *enc_in_b, *enc_out_w, *enc_out_b;
  float* alpha;
  std::vector<float*> pp_w, pp_b, pp_lw, pp_lb;
  float *pp_lin_w, *pp_lin_b, *table;
  size_t cap = 0;
  float *a = nullptr, *b = nullptr, *c = nullptr, *keep = nullptr, *pred = nullptr;
  int* pos = nullptr;
};

static int pe_alloc(bsg_pitchext* h, float** p, size_t n) {
  BSG_HIP(hipMalloc((void**)p, n * sizeof(float)));
  h->owned.push_back(*p);
  return BSG_OK;
}
static int pe_copy(bsg_pitchext* h, float** dst, const void* src, size_t n, hipStream_t st) {
  TRY(pe_alloc(h, dst, n));
  BSG_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  return BSG_OK;
}
static int pe_conv(bsg_pitchext* h, float** dst, const void* src, int M, int Cin, int k, hipStream_t st) {
  TRY(pe_alloc(h, dst, (size_t)M * Cin * k));
  const long long total = (long long)M * Cin * k;
  hipLaunchKernelGGL(repack_conv5_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)src, *dst, M, Cin, k);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

extern "C" void bsg_pitchext_destroy(bsg_pitchext* h) {
  if (!h) return;
  guard_free(&h->guard);
  for (float* p : h->owned) (void)hipFree(p);
  float* ws[] = {h->a, h->b, h->c, h->keep, h->pred};
  for (float* p : ws)
    if (p) (void)hipFree(p);
  if (h->pos) (void)hipFree(h->pos);
  delete h;
}

extern "C" int bsg_pitchext_n_weights(const bsg_pitchext_cfg* c) { return 7 * 3 + 2 + 4 * c->conv_layers + 4 + 1 + 4 * c->predictor_layers + 2 + 1; }

static int pe_create_impl(bsg_pitchext* h, const void* const* w, const float* table, hipStream_t st) {
  const bsg_pitchext_cfg& c = h->cfg;
  int i = 0;
  for (int l = 0; l < 3; ++l) {
    const int cin = l == 0 ? c.n_mel : H;
    TRY(pe_conv(h, &h->pre_w[l], w[i++], H, cin, 5, st));
    TRY(pe_copy(h, &h->pre_b[l], w[i++], H, st));
    const float *bw = (const float*)w[i++], *bb = (const float*)w[i++], *bm = (const float*)w[i++], *bv = (const float*)w[i++];
    i++;  // num_batches_tracked
    TRY(pe_alloc(h, &h->pre_scale[l], H));
    TRY(pe_alloc(h, &h->pre_shift[l], H));
    hipLaunchKernelGGL(bn_affine_kernel, dim3(1), dim3(256), 0, st, bw, bb, bm, bv, 1e-5f, h->pre_scale[l], h->pre_shift[l], H);
    BSG_LAUNCH_CHECK();
  }
  TRY(pe_copy(h, &h->pre_out_w, w[i++], (size_t)H * H, st));
  TRY(pe_copy(h, &h->pre_out_b, w[i++], H, st));
  h->enc_w.resize(c.conv_layers); h->enc_b.resize(c.conv_layers); h->enc_gw.resize(c.conv_layers); h->enc_gb.resize(c.conv_layers);
  for (int l = 0; l < c.conv_layers; ++l) {
    TRY(pe_conv(h, &h->enc_w[l], w[i++], H, H, 5, st));
    TRY(pe_copy(h, &h->enc_b[l], w[i++], H, st));
    TRY(pe_copy(h, &h->enc_gw[l], w[i++], H, st));
    TRY(pe_copy(h, &h->enc_gb[l], w[i++], H, st));
  }
  TRY(pe_copy(h, &h->enc_in_w, w[i++], (size_t)H * H, st));
  TRY(pe_copy(h, &h->enc_in_b, w[i++], H, st));
  TRY(pe_copy(h, &h->enc_out_w, w[i++], (size_t)H * H, st));
  TRY(pe_copy(h, &h->enc_out_b, w[i++], H, st));
  TRY(pe_copy(h, &h->alpha, w[i++], 1, st));
  h->pp_w.resize(c.predictor_layers); h->pp_b.resize(c.predictor_layers); h->pp_lw.resize(c.predictor_layers); h->pp_lb.resize(c.predictor_layers);
  for (int l = 0; l < c.predictor_layers; ++l) {
    TRY(pe_conv(h, &h->pp_w[l], w[i++], H, H, c.predictor_kernel, st));
    TRY(pe_copy(h, &h->pp_b[l], w[i++], H, st));
    TRY(pe_copy(h, &h->pp_lw[l], w[i++], H, st));
    TRY(pe_copy(h, &h->pp_lb[l], w[i++], H, st));
  }
  TRY(pe_copy(h, &h->pp_lin_w, w[i++], (size_t)2 * H, st));
  TRY(pe_copy(h, &h->pp_lin_b, w[i++], 2, st));
  i++;  // embed_positions._float_tensor
  TRY(pe_copy(h, &h->table, table, (size_t)c.n_pos * H, st));
  BSG_HIP(hipStreamSynchronize(st));
  return BSG_OK;
}

extern "C" int bsg_pitchext_create(bsg_pitchext** out, const bsg_pitchext_cfg* cfg, const void* const* dev_weights,
                                   int32_t n_weights, const float* pos_table, void* stream) {
  BSG_REQUIRE(out && cfg && dev_weights && pos_table, "pitchext_create: null argument");
  BSG_REQUIRE(cfg->hidden_size == H && cfg->n_mel > 0 && cfg->n_mel % 4 == 0, "pitchext_create: hidden_size=%d n_mel=%d unsupported", cfg->hidden_size, cfg->n_mel);
  BSG_REQUIRE(cfg->conv_layers >= 0 && cfg->conv_layers <= 16 && cfg->predictor_layers > 0 && cfg->predictor_layers <= 16 &&
                  cfg->predictor_kernel % 2 == 1 && cfg->n_pos > 1, "pitchext_create: bad config");
  BSG_REQUIRE(n_weights == bsg_pitchext_n_weights(cfg), "pitchext_create: expected %d weight tensors, got %d", bsg_pitchext_n_weights(cfg), n_weights);
  for (int i = 0; i < n_weights; ++i) BSG_REQUIRE(dev_weights[i] != nullptr, "pitchext_create: weight %d is null", i);
  bsg_pitchext* h = new bsg_pitchext();
  h->cfg = *cfg;
  int rc = guard_init(&h->guard, (hipStream_t)stream);
  if (rc == BSG_OK) rc = pe_create_impl(h, dev_weights, pos_table, (hipStream_t)stream);
  if (rc != BSG_OK) { bsg_pitchext_destroy(h); return rc; }
  *out = h;
  return BSG_OK;
}

static int pe_linear(const float* X, const float* W, const float* bias, float* Y, long long rows, int N, int K, const float* rowscale,
                     hipStream_t st) {
  GemmArgs g{};
  g.A = X; g.B = W; g.C = Y; g.M = (int)rows; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.trans_b = 1; g.taps = 1;
  g.bias_n = bias; g.alpha = 1.f; g.rowscale = rowscale; g.batch = 1;
  return launch_gemm(g, st);
}
static int pe_conv_gemm(const float* X, const float* Wt, const float* bias, float* Y, int B, int T, int Cin, int ks, int act,
                        const float* ps, const float* pb, const float* rowscale, hipStream_t st) {
  GemmArgs g{};
  g.A = X; g.B = Wt; g.C = Y; g.M = T; g.N = H; g.K = Cin; g.lda = Cin; g.ldb = Cin; g.ldc = H; g.trans_b = 1;
  g.taps = ks; g.tap_shift0 = -(ks / 2); g.sTapB = (long long)H * Cin; g.bias_n = bias; g.alpha = 1.f; g.act = act;
  g.post_scale_n = ps; g.post_shift_n = pb; g.rowscale = rowscale; g.sRS = T; g.batch = B;
  g.sA = (long long)T * Cin; g.sC = (long long)T * H;
  return launch_gemm(g, st);
}

extern "C" int bsg_pitchext_forward(bsg_pitchext* h, const float* mel, float* pitch_pred, float* f0, int32_t B, int32_t T, void* stream) {
  GuardScope guard_scope(h ? &h->guard : nullptr);
  BSG_REQUIRE(h && mel && f0 && B > 0 && T > 0 && T < h->cfg.n_pos, "pitchext_forward: bad argument (T=%d, table %d rows)", T, h ? h->cfg.n_pos : 0);
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)B * T;
  if ((size_t)rows > h->cap) {
    BSG_HIP(hipStreamSynchronize(st));
    float** bufs[] = {&h->a, &h->b, &h->c, &h->keep, &h->pred};
    for (float** p : bufs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (h->pos) { (void)hipFree(h->pos); h->pos = nullptr; }
    h->cap = 0;
    BSG_HIP(hipMalloc((void**)&h->a, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->b, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->c, rows * H * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->keep, rows * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->pred, rows * 2 * sizeof(float)));
    BSG_HIP(hipMalloc((void**)&h->pos, rows * sizeof(int)));
    h->cap = (size_t)rows;
  }
  const dim3 rg(cdiv(rows, 4)), rb(256);
  hipLaunchKernelGGL(row_keep_kernel, rg, rb, 0, st, mel, h->keep, rows, h->cfg.n_mel);
  BSG_LAUNCH_CHECK();
  // Prenet: 3 x [conv k5 -> ReLU -> BatchNorm(eval) -> * keep], out_proj * keep        (pe.py:24-42)
  const float* x = mel;
  float* bufs[2] = {h->a, h->b};
  for (int l = 0; l < 3; ++l) {
    TRY(pe_conv_gemm(x, h->pre_w[l], h->pre_b[l], bufs[l & 1], B, T, l == 0 ? h->cfg.n_mel : H, 5, ACT_RELU, h->pre_scale[l],
                     h->pre_shift[l], h->keep, st));
    x = bufs[l & 1];
  }
  TRY(pe_linear(x, h->pre_out_w, h->pre_out_b, h->b, rows, H, H, h->keep, st));      // x = a (l=2 -> bufs[0]); out -> b
  float* cur = h->b;
  if (h->cfg.conv_layers > 0) {
    // ConvStacks: in_proj, n x (x + relu(GroupNorm(conv(x)))), out_proj                   (pe.py:99-117)
    TRY(pe_linear(cur, h->enc_in_w, h->enc_in_b, h->a, rows, H, H, nullptr, st));
    float* xx = h->a;
    float* other = h->b;
    for (int l = 0; l < h->cfg.conv_layers; ++l) {
      TRY(pe_conv_gemm(xx, h->enc_w[l], h->enc_b[l], h->c, B, T, H, 5, ACT_NONE, nullptr, nullptr, nullptr, st));
      hipLaunchKernelGGL(groupnorm_res_kernel, dim3(H / 16, B), dim3(256), 0, st, (const float*)h->c, h->enc_gw[l], h->enc_gb[l],
                         (const float*)xx, other, T, 1e-5f);
      BSG_LAUNCH_CHECK();
      float* t = xx; xx = other; other = t;
    }
    TRY(pe_linear(xx, h->enc_out_w, h->enc_out_b, other, rows, H, H, nullptr, st));
    cur = other;
  }
  // PitchPredictor                                                                        (tts_modules.py:233-247)
  hipLaunchKernelGGL(positions_kernel, dim3(B), dim3(64), 0, st, (const float*)cur, h->pos, T);
  BSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(add_positions_kernel, rg, rb, 0, st, cur, (const int*)h->pos, h->table, h->alpha, rows, h->cfg.n_pos);
  BSG_LAUNCH_CHECK();
  float* nxt = cur == h->a ? h->b : h->a;
  for (int l = 0; l < h->cfg.predictor_layers; ++l) {
    TRY(pe_conv_gemm(cur, h->pp_w[l], h->pp_b[l], h->c, B, T, H, h->cfg.predictor_kernel, ACT_RELU, nullptr, nullptr, nullptr, st));
    hipLaunchKernelGGL(layernorm256_kernel, rg, rb, 0, st, (const float*)h->c, h->pp_lw[l], h->pp_lb[l], nxt, rows, 1e-12f);
    BSG_LAUNCH_CHECK();
    float* t = cur; cur = nxt; nxt = t;
  }
  float* pred = pitch_pred ? pitch_pred : h->pred;
  TRY(pe_linear(cur, h->pp_lin_w, h->pp_lin_b, pred, rows, 2, H, nullptr, st));
  hipLaunchKernelGGL(f0_denorm_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, (const float*)pred, h->keep, f0, rows, h->cfg.use_uv);
  BSG_LAUNCH_CHECK();
  return BSG_OK;
}

namespace bsg {
Guard* guard_of_pitchext(void* h) { return &static_cast<bsg_pitchext*>(h)->guard; }
}
