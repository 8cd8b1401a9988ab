/* bisinger_hip.h — C ABI of libbisinger_hip.so: the MI355X (gfx950) mel-generation hot path of BiSinger.
 *
 * The reference (BiSinger-SVS/BiSinger) is pure Python on PyTorch and has no FFI of its own; its
 * plug points for this path are Python callables/registries.  Each entry point below
 * names the reference interface it stands behind (paths relative to /root/reference/train_bisinger).
 * The Python drop-ins in bisinger_amd/ (same class names, constructor arguments, state_dict keys) bind
 * these with ctypes: see INTEGRATION.md.
 *
 * Conventions
 *   - return 0 on success, a negative BSG_E* code on failure; never throws; bsg_last_error() gives
 *     the thread-local message of the last failure on the calling thread.
 *   - every tensor argument is a caller-owned, contiguous DEVICE pointer unless it says "host";
 *     float = IEEE fp32, indices = int64 (torch.long), exactly the reference's dtypes.
 *   - the library owns packed copies of the weights and its workspaces (sized at create /
 *     prepare time; nothing is allocated by *_forward / *_sample, which only enqueue on `stream`
 *     and never synchronise the host, so they can be captured into a hipGraph).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - a handle is single-stream and not re-entrant; distinct handles are independent.
 */
#ifndef BISINGER_HIP_H
#define BISINGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSG_ABI_VERSION 7

#define BSG_OK 0
#define BSG_EINVAL (-22)  /* bad argument / shape the kernels do not support            */
#define BSG_ENOMEM (-12)  /* hipMalloc failed                                            */
#define BSG_EHIP (-5)     /* a HIP runtime call failed (message has hipGetErrorString)   */
#define BSG_ESTATE (-1)   /* call order violated (e.g. forward before prepare)           */

int bsg_abi_version(void);
const char* bsg_last_error(void);
/* Name of device 0's architecture ("gfx950"...), or NULL when no HIP device is usable. */
const char* bsg_device_arch(void);

/* ------------------------------------------------------------------------------------------------
 * DiffNet — the WaveNet noise predictor.
 * Stands behind: DIFF_DECODERS['wavenet'] = lambda hp: DiffNet(hp['audio_num_mel_bins'])
 *   (usr/diffsinger_task.py:24-29) with contract
 *   denoise_fn(spec [B,1,M,T] f32, t [B] i64, cond [B,H,T] f32) -> [B,1,M,T]  (usr/diff/net.py:107-130).
 * ---------------------------------------------------------------------------------------------- */
typedef struct bsg_diffnet bsg_diffnet;

typedef struct {
  int32_t in_dims;               /* M: mel bins, DiffNet(in_dims)                 net.py:82     */
  int32_t residual_channels;     /* C: hparams['residual_channels'] (must be 256) net.py:88     */
  int32_t encoder_hidden;        /* H: hparams['hidden_size']      (must be 256)  net.py:86     */
  int32_t residual_layers;       /* L: hparams['residual_layers']                 net.py:87     */
  int32_t dilation_cycle_length; /* dilation of layer i = 2^(i % cycle), <= 8     net.py:99     */
  int32_t max_steps;             /* rows of step_table (>= diffusion timesteps)                 */
} bsg_diffnet_cfg;

/* dev_weights: 10 + 8*L device pointers in DiffNet.state_dict() order (net.py:91-105):
 *   input_projection.{weight[C,M,1],bias[C]}, mlp.0.{weight[4C,C],bias}, mlp.2.{weight[C,4C],bias},
 *   residual_layers.i.{dilated_conv.weight[2C,C,3],bias, diffusion_projection.weight[C,C],bias,
 *                      conditioner_projection.weight[2C,H,1],bias, output_projection.weight[2C,C,1],bias},
 *   skip_projection.{weight[C,C,1],bias}, output_projection.{weight[M,C,1],bias}.
 * step_table: device [max_steps, C] = SinusoidalPosEmb(C)(arange(max_steps)) (net.py:32-44), built by
 *   the host with the same fp32 ops as the reference so the embedding is bit-identical.
 * The call packs the weights into MFMA fragment order and tabulates mlp(step_table) and every layer's
 * diffusion_projection of it (net.py:119-120, :67) — work that does not depend on the input.
 * Synchronises `stream` before returning (the caller may free/modify its weight tensors afterwards). */
int bsg_diffnet_create(bsg_diffnet** out, const bsg_diffnet_cfg* cfg, const void* const* dev_weights,
                       int32_t n_weights, const float* step_table, void* stream);
void bsg_diffnet_destroy(bsg_diffnet* h);

/* Bind a condition: cond [B,H,T] (= decoder_inp^T, shallow_diffusion_tts.py:235).  Computes every
 * layer's conditioner_projection(cond) + its bias + the dilated conv's bias once ([L,B,2C,T], the
 * step-invariant part of net.py:68-71) and sizes the workspaces for (B,T).  May allocate. */
int bsg_diffnet_prepare(bsg_diffnet* h, const float* cond, int32_t B, int32_t T, void* stream);

/* eps = DiffNet(x, t, cond bound by prepare).  x, eps: [B,M,T] (the reference's [B,1,M,T]); t: [B] i64. */
int bsg_diffnet_forward(bsg_diffnet* h, const float* x, const int64_t* t, float* eps, int32_t B,
                        int32_t T, void* stream);

/* Per-layer fused residual block alone (net.py:66-78), exported for unit tests and micro-benchmarks:
 * x_in [B,C,T], t [B] -> x_out [B,C,T]; skip [B,C,T] is read-modify-written unless layer == 0
 * (first layer stores); the last layer stores skip_sum / sqrt(L) (net.py:126). */
int bsg_diffnet_residual_layer(bsg_diffnet* h, int32_t layer, const float* x_in, const int64_t* t,
                               float* x_out, float* skip, int32_t B, int32_t T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Samplers.  Stand behind GaussianDiffusion.p_sample / p_sample_plms and the inference loops
 * (usr/diff/shallow_diffusion_tts.py:159-201, :258-267).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t num_timesteps;
  /* host arrays [num_timesteps] = the module's fp32 buffers (shallow_diffusion_tts.py:103-122) */
  const float* sqrt_recip_alphas_cumprod;
  const float* sqrt_recipm1_alphas_cumprod;
  const float* posterior_mean_coef1;
  const float* posterior_mean_coef2;
  const float* sigma;          /* exp(0.5*posterior_log_variance_clipped), sigma[0] = 0 (:165-166) */
  const float* alphas_cumprod; /* PLMS only (:175-176)                                            */
} bsg_schedule;

/* DDPM ancestral loop B (:265-267): for i = t_start, t_start-1, ..., t_start-n_steps+1:
 *     x <- p_sample(x, full((B,), i), cond)
 * x: [B,M,T] in (x_{t_start+1}) / out.  noise: [n_steps,B,M,T] N(0,1) draws, one per executed step
 * (parity mode), or NULL: draws come from the on-device Philox4x32-10 stream (seed, global row, step)
 * documented in bisinger_amd/synth.py (bench mode).  row0/B_total: global batch row of this shard's
 * row 0 and global batch size, so a sharded run reproduces the unsharded noise.
 * For batches with more 32-frame tiles than the device has CUs the loop runs as two concurrent launch chains over half the
 * rows each: a handle-owned second stream is forked off `stream` and joined back before the call returns (BSG_DUAL=0
 * disables); results are bit-identical either way. */
int bsg_ddpm_sample(bsg_diffnet* h, const bsg_schedule* s, float* x, const float* noise, uint64_t seed,
                    int32_t t_start, int32_t n_steps, int32_t B, int32_t T, int32_t row0,
                    int32_t B_total, void* stream);

/* One ancestral update given eps (p_sample :159-166 without the denoiser call), for denoisers that are not a
 * bsg_diffnet (the FFT candidate below): x, eps, noise (or NULL = Philox stream of timestep t) are n floats. */
int bsg_ddpm_step(float* x, const float* eps, const float* noise, const bsg_schedule* s, int32_t t, int64_t n,
                  uint64_t seed, uint64_t offset, void* stream);

/* One PLMS update given the noise predictions (p_sample_plms :168-201 without the denoiser calls; ABI v5), for denoisers that are not a
 * bsg_diffnet — DIFF_DECODERS['fft'] under pndm_speedup: x_out = x + x_delta(eps'), eps' = the multistep blend of e0 (the newest prediction)
 * with the n_hist <= 3 older ones e1..e3 (NULL beyond n_hist): n_hist 0: e0 (the predictor half-step of the first iteration); n_hist 1 with
 * avg: (e0 + e1) / 2 (its corrector); 1: (3 e0 - e1) / 2; 2: (23 e0 - 16 e1 + 5 e2) / 12; 3: (55 e0 - 59 e1 + 37 e2 - 9 e3) / 24.
 * alphas_cumprod at t and t_prev = max(t - interval, 0).  x_out may alias x. */
int bsg_plms_step(const float* x, float* x_out, const float* e0, const float* e1, const float* e2, const float* e3, int32_t n_hist,
                  int32_t avg, const bsg_schedule* s, int32_t t, int32_t t_prev, int64_t n, void* stream);

/* x[0..n) <- N(0,1) from the same Philox4x32-10 stream family: element i = lane i%4 of counter
 * ((offset+i)/4, stream_id, 0, 0), key = seed (x_T under gaussian_start uses stream_id 0, the step with
 * timestep i uses stream_id i+1).  n and offset must be multiples of 4. */
int bsg_philox_normal(float* x, int64_t n, uint64_t seed, uint32_t stream_id, uint64_t offset, void* stream);

/* Glue of GaussianDiffusion.forward around the loop (shallow_diffusion_tts.py:244-272), all [B,M,T] <-> [B,T,M]:
 *   bsg_mel_start : x = sqrt_ac * norm_spec(fs2_mel)^T + sqrt_1m_ac * noise      (q_sample at t = K_step-1, :249-252)
 *   bsg_mel_finish: mel_out = denorm_spec(x^T) * (mel2ph > 0)   (mel2ph NULL: no mask, the non-singing branch :271-272)
 * spec_min / spec_max: device [M] (the module's buffers). */
int bsg_mel_start(const float* fs2_mel, const float* spec_min, const float* spec_max, const float* noise,
                  float sqrt_ac, float sqrt_1m_ac, float* x, int32_t B, int32_t M, int32_t T, void* stream);
int bsg_mel_finish(const float* x, const float* spec_min, const float* spec_max, const int64_t* mel2ph,
                   float* mel_out, int32_t B, int32_t M, int32_t T, void* stream);

/* Live timing of the dominant kernel (bench.py's roofline): while enabled, every DiffNet evaluation records
 * a hipEvent pair around its L fused residual-layer launches on the launch stream.  profile_read waits for the
 * recorded events and returns the summed device time and the number of layer-equivalents they cover (L per evaluation,
 * also when the L layers run as one stack launch). */
int bsg_diffnet_profile(bsg_diffnet* h, int32_t enable);
int bsg_diffnet_profile_read(bsg_diffnet* h, double* layer_ms_total, int64_t* n_layer_launches);
/* Arithmetic of the fused residual layers (BASELINE config "bf16 diffusion mel-gen"):
 *   BSG_COMPUTE_F32  (default) fp32 operands, fp32 MFMA — the parity configuration (<= 1e-3 on the mel);
 *   BSG_COMPUTE_BF16 MFMA operands (weights, staged x + d, gated z) rounded to bf16 (RNE), fp32 accumulation, fp32
 *                    gate/epilogue, and x / conditioner term / skip kept fp32 in HBM — a throughput configuration
 *                    whose deviation from the fp32 path is reported by tests/test_gpu_bf16.py and bench.py.
 * Takes effect on the next forward / sample call on the handle. */
#define BSG_COMPUTE_F32 0
#define BSG_COMPUTE_BF16 1
int bsg_diffnet_set_compute(bsg_diffnet* h, int32_t mode);
/* Synchronous health check of the launches that hand data between workgroups (the on-chip stack launch, whose neighbouring
 * tiles exchange their edges every layer, and the channel-split launch used for small batches): number of inter-workgroup hand-off spins that
 * gave up since the handle was bound (must be 0; non-zero means a result is invalid). */
int bsg_diffnet_status(bsg_diffnet* h, int32_t* handoff_timeouts);
/* Asynchronous form (ABI v5: THREE words): enqueues copies of the handle's health words into host_counts[0..2] (pinned host memory,
 * caller-owned) on `stream` and returns; the values are valid once the stream has passed that point (e.g. an event recorded after the
 * call): [0] hand-off give-ups of the stack / part launches, [1] values beyond the fp16 range seen by the split-fp16 stack / part / tail
 * launches (16-row stack launch: |x| >= 3750; 32-row launch, part forms and tails: |x + d| >= 60000: the result is invalid, repeat with
 * bsg_diffnet_set_h2q(h, 0) after a 16-row launch, with bsg_diffnet_set_h2(h, 0) otherwise), [2] give-ups of the channel-split launches.
 * Nothing is reset (bsg_diffnet_health_take does that).  Lets a caller fail loudly — or repeat — one call later without adding a
 * synchronisation to the path: what a replayed capture of the sampler loop (round 4: the stack / part launches keep their launch epoch in
 * device memory and can be captured) and the `deferred` guard mode of the Python drop-ins use. */
int bsg_diffnet_status_async(bsg_diffnet* h, int32_t* host_counts, void* stream);
/* Same-call form (what the Python drop-ins use): waits for `stream`, returns the give-up count accumulated since the last
 * take and resets it.  bsg_diffnet_uses_handoffs says whether launches of shape (B,T) on this handle may hand data between
 * workgroups at all (small batches: B*ceil(T/32) <= CUs); when it says 0 no check — and no synchronisation — is needed.
 * bsg_diffnet_set_split(h, 0) makes the handle use one-workgroup-per-tile launches only (no hand-offs): the caller's
 * recovery after a non-zero take is set_split(0) + re-running the evaluation, which then cannot give up. */
int bsg_diffnet_handoff_take(bsg_diffnet* h, int32_t* handoff_timeouts, void* stream);
/* The same take with the two causes apart (ABI v4): counts[0] = hand-off spins that gave up (a partner workgroup was not resident:
 * recovery = bsg_diffnet_set_split(h, 0)); counts[1] = values beyond the fp16 range seen by the split-fp16 stack launch (data, not
 * residency: recovery = bsg_diffnet_set_h2(h, 0), the fp32-matrix-pipe kernels, for this input only).  Waits for `stream`; resets both.
 * bsg_diffnet_handoff_take returns their sum. */
int bsg_diffnet_health_take(bsg_diffnet* h, int32_t* counts, void* stream);
int bsg_diffnet_uses_handoffs(bsg_diffnet* h, int32_t B, int32_t T, int32_t* uses);
int bsg_diffnet_set_split(bsg_diffnet* h, int32_t enable);
/* Fault injection for tests: in the next n_launches channel-split (or stack) launches on the handle the consumers give up every
 * hand-off without waiting (counted exactly like a timed-out spin) and one producer per tile (pair) never publishes, so the
 * consumers deterministically read stale exchange data. */
int bsg_diffnet_debug_inject_giveup(bsg_diffnet* h, int32_t n_launches);
/* Fault injection (ABI v6): in the next n_launches PART launches (several workgroups per tile on CUs of one XCD) the odd parts report
 * another XCC id than the one they run on — a dispatch order other than workgroup i -> XCD i mod 8.  Counted as hand-off give-ups. */
int bsg_diffnet_debug_inject_xcc(bsg_diffnet* h, int32_t n_launches);
/* Part forms on (1, default) / off (0) for this handle (ABI v6): with them off small batches run the one-workgroup-per-tile stack launch,
 * which exchanges only tile edges with write-through stores and needs no placement on one XCD.  First tier of the recovery after a
 * give-up inside a part launch; bsg_diffnet_set_split(h, 0) (no hand-off launches at all) is the second. */
int bsg_diffnet_set_parts(bsg_diffnet* h, int32_t enable);
/* Test hook (ABI v6): sets the device-side launch epoch of the handle's stack / part launches (hand-off flag values are epoch x 64 + layer;
 * the epoch restarts at 1, with every flag array of the handle zeroed, once it reaches 2^25).  Waits for `stream`.  epoch >= 1. */
int bsg_diffnet_debug_set_epoch(bsg_diffnet* h, uint32_t epoch, void* stream);
/* Name of the form the last residual-layer launch on the handle took: "stack" (all L layers in one launch with the residual
 * stream on chip), "layer" (one launch per layer, one workgroup per tile), "split2" / "split4" (a tile as 2 / 4 workgroups),
 * "wide" (one 16-wave workgroup per tile), "bf16", or "none".  Static string. */
const char* bsg_diffnet_last_path(bsg_diffnet* h);
/* Shader clock the chip held over the last PROFILED stack launch (bsg_diffnet_profile on): tile 0 of the launch stores s_memtime (shader
 * clocks) and s_memrealtime (100 MHz) at its start and end; shader_mhz = their ratio, span_us = the launch's in-kernel span.  Synchronous
 * copy; 0 when no profiled stack launch has run (ABI v4). */
int bsg_diffnet_clock_read(bsg_diffnet* h, double* shader_mhz, double* span_us);

/* Diagnostic run of the stack launch on whatever h->xa holds (timing only): the L layers of the bound batch (which must fit one
 * launch) at timestep t_uniform, with s_memrealtime (100 MHz) stamps per tile and layer at
 * {0 xs complete, 1 GEMM1 done, 2 z in LDS, 3 residual half done, 4 skip half done, 5 edges drained, 6 neighbours' flags seen,
 *  7 halo barrier passed} into stamps [B*ceil(T/32)][L][8] (device, uint64).  Used by tools/stack_stamps.py. */
int bsg_diffnet_debug_stack_stamps(bsg_diffnet* h, int32_t t_uniform, int32_t B, int32_t T, uint64_t* stamps, void* stream);

/* Diagnostic build of the fused residual layer (separate kernel instantiation; the product kernel executes no
 * stamp): same computation, plus s_memtime stamps per wave at the phase boundaries
 * {0 start, 1 staged, 2 acc init, 3 GEMM1 done, 4 gate done, 5 z in LDS, 6 GEMM2 residual pass, 7 skip pass}
 * into stamps [B*ceil(T/32)][8 waves][10] (device, uint64; slots 8, 9 = s_memrealtime (100 MHz) at start / end, which
 * gives the shader clock the chip held).  Never timed; used by tools/stamp_layer.py. */
int bsg_diffnet_debug_stamps(bsg_diffnet* h, int32_t layer, const float* x_in, const int64_t* t, float* x_out,
                             float* skip, int32_t B, int32_t T, uint64_t* stamps, void* stream);

/* PLMS / PNDM loop A (:258-264, p_sample_plms :168-201): for i in reversed(range(0,K_step,interval)).
 * Batched semantics = element-wise clamp of t-interval (the reference raises for B>1, :189). */
int bsg_plms_sample(bsg_diffnet* h, const bsg_schedule* s, float* x, int32_t K_step, int32_t interval,
                    int32_t B, int32_t T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FastSpeech2-MIDI: phoneme/pitch encoder -> per-frame condition, and the FFT mel decoder.
 * Stands behind FastSpeech2MIDI.forward (modules/diffsinger_midi/fs2.py:94-197), split at the one
 * data-dependent point of the reference (the length regulator's output length, tts_modules.py:182):
 *   encode  = embeddings + ESM + FastspeechMIDIEncoder (+ DurationPredictor.inference)     fs2.py:111-165
 *   bsg_length_regulator = LengthRegulator.forward                                 tts_modules.py:161-191
 *   decode  = gather by mel2ph, +spk +style, mask -> decoder_inp; FastspeechDecoder + mel_out  fs2.py:166-195
 * ---------------------------------------------------------------------------------------------- */
typedef struct bsg_fs2midi bsg_fs2midi;

typedef struct {
  int32_t hidden_size;          /* must be 256                                  hparams['hidden_size']   */
  int32_t vocab;                /* len(phone_encoder)                           fastspeech/fs2.py:94     */
  int32_t enc_layers, dec_layers, num_heads;
  int32_t enc_ffn_kernel_size, dec_ffn_kernel_size;
  int32_t out_dims;             /* mel bins                                                              */
  int32_t dur_layers, dur_kernel;
  int32_t spk_rows;             /* num_spk + 1                                  fastspeech/fs2.py:40     */
  int32_t esm_heads;            /* 8                                            diffsinger_midi/fs2.py:83 */
  int32_t n_pos;                /* rows of dec_pos_table (frames T must be < n_pos)                      */
  int32_t n_rel;                /* rows of rel_pos_table (T_txt must be <= n_rel)                        */
} bsg_fs2midi_cfg;

/* Number of entries of FastSpeech2MIDI.state_dict() for this configuration (143 for BiSinger). */
int bsg_fs2midi_n_weights(const bsg_fs2midi_cfg* cfg);

/* dev_weights: device pointers in FastSpeech2MIDI.state_dict() order (tests/golden/state_dict_spec.json,
 * keys 'fs2.*'): encoder_embed_tokens, decoder.{pos_embed_alpha, embed_positions._float_tensor,
 * layers.i.op.{layer_norm1.w,b, self_attn.in_proj_weight, self_attn.out_proj.weight, layer_norm2.w,b,
 * ffn.ffn_1.w,b, ffn.ffn_2.w,b}, layer_norm.w,b}, mel_out.w,b, spk_embed_proj, dur_predictor.{conv.i.1.w,b,
 * conv.i.3.w,b, linear.w,b}, esm.{mh.in_proj_weight, in_proj_bias, out_proj.w,b, ffn.0.w,b, ffn.2.w,b,
 * ln1.w,b, ln2.w,b}, encoder.{layers..., layer_norm.w,b, embed_tokens (alias), esm.* (alias)},
 * midi_embed, midi_dur_layer.w,b, is_slur_embed, lang_embed, style_embed.
 * dec_pos_table [n_pos,H]: SinusoidalPositionalEmbedding table (common_layers.py:124-146, row 0 zero);
 * rel_pos_table [n_rel,H]: RelPositionalEncoding's reversed table rows 0..n_rel-1 of max_len 5000
 * (espnet_positional_embedding.py:25-46) — both built by the host with the reference's own fp32 ops.
 * The library keeps its own (re-packed) copies; synchronises `stream` before returning. */
int bsg_fs2midi_create(bsg_fs2midi** out, const bsg_fs2midi_cfg* cfg, const void* const* dev_weights,
                       int32_t n_weights, const float* dec_pos_table, const float* rel_pos_table, void* stream);
void bsg_fs2midi_destroy(bsg_fs2midi* h);

/* txt, pitch_midi, is_slur, lang: [B,T_txt] i64; midi_dur [B,T_txt] f32; spk_id [B] i64.
 * enc_out [B,T_txt,H].  dur_xs [B,T_txt] f32 (log-domain predictor output, ret['dur']) and dur [B,T_txt] i64
 * (ret['dur_choice']) are both NULL (mel2ph given) or both set (mel2ph predicted).  May grow workspaces. */
int bsg_fs2midi_encode(bsg_fs2midi* h, const int64_t* txt, const int64_t* pitch_midi, const float* midi_dur,
                       const int64_t* is_slur, const int64_t* lang, const int64_t* spk_id, int32_t B,
                       int32_t T_txt, float* enc_out, float* dur_xs, int64_t* dur, void* stream);

/* ABI v7 (one rank of a sharded batch): the same front for the batch rows [row0, row0 + n_rows) only.  The inputs are the
 * WHOLE batch's ([B,T_txt] / [B]); enc_out [n_rows,T_txt,H], dur_xs / dur [n_rows,T_txt].  Only the ESM couples the utterances of a batch
 * (it attends over the batch axis, common_layers.py:853) and only through K / V = projections of LN(lang_embed[lang]) (:850-853): K / V
 * are projected for all B rows; Q, the ESM's FFN, the embedding sum, the FFT encoder (tts_modules.py:312-328) and the duration predictor
 * run on the n_rows rows asked for.  Row for row the result equals bsg_fs2midi_encode's on the whole batch (same kernels, same order of
 * summation per row).  row0 = 0, n_rows = B is bsg_fs2midi_encode. */
int bsg_fs2midi_encode_rows(bsg_fs2midi* h, const int64_t* txt, const int64_t* pitch_midi, const float* midi_dur,
                            const int64_t* is_slur, const int64_t* lang, const int64_t* spk_id, int32_t B, int32_t T_txt,
                            int32_t row0, int32_t n_rows, float* enc_out, float* dur_xs, int64_t* dur, void* stream);

/* ABI v7, introspection (tests): token rows (utterances x T_txt) the last encode ran its ENCODER on, and rows of the last FFT stack
 * (encoder or decoder) — how a test sees that a rank's front did not encode the other ranks' utterances. */
int bsg_fs2midi_last_rows(const bsg_fs2midi* h, int32_t* token_rows, int32_t* stack_rows);

/* mel2ph [B,T] from dur [B,T_txt] (padded tokens, txt == 0, count 0 when txt != NULL); T = max_b sum(dur). */
int bsg_length_regulator(const int64_t* dur, const int64_t* txt, int64_t* mel2ph, int32_t B, int32_t T_txt,
                         int32_t T, void* stream);

/* decoder_inp [B,T,H] (ret['decoder_inp']); mel_out [B,T,out_dims] (ret['mel_out']) or NULL = skip_decoder. */
int bsg_fs2midi_decode(bsg_fs2midi* h, const float* enc_out, const int64_t* mel2ph, const int64_t* spk_id,
                       const int64_t* speechsing, int32_t B, int32_t T_txt, int32_t T, float* decoder_inp,
                       float* mel_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FFT candidate denoiser: DIFF_DECODERS['fft'] = FFT(hidden, dec_layers, dec_ffn_kernel_size,
 * num_heads) (usr/diffsinger_task.py:26-28, usr/diff/candidate_decoder.py:39-100); same denoise_fn contract as DiffNet.
 * dev_weights in FFT.state_dict() order: pos_embed_alpha, embed_positions._float_tensor, layers.i.op.* (10 per layer),
 * layer_norm.{w,b}, input_projection.{w,b}, mlp.0.{w,b}, mlp.2.{w,b}, get_mel_out.{w,b}, get_decode_inp.{w,b}.
 * step_table [max_steps,256] and pos_table [n_pos,256] as for bsg_diffnet_create / bsg_fs2midi_create.
 * ---------------------------------------------------------------------------------------------- */
typedef struct bsg_fftden bsg_fftden;
int bsg_fftden_n_weights(int32_t n_layers);
int bsg_fftden_create(bsg_fftden** out, int32_t in_dims, int32_t n_layers, int32_t num_heads, int32_t ffn_kernel,
                      int32_t max_steps, int32_t n_pos, const void* const* dev_weights, int32_t n_weights,
                      const float* step_table, const float* pos_table, void* stream);
void bsg_fftden_destroy(bsg_fftden* h);
int bsg_fftden_prepare(bsg_fftden* h, const float* cond, int32_t B, int32_t T, void* stream);
int bsg_fftden_forward(bsg_fftden* h, const float* x, const int64_t* t, float* eps, int32_t B, int32_t T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HiFi-GAN generator forward (mel -> waveform).
 * Stands behind HifiGanGenerator(h)(x [B,80,T]) -> [B,1,T*prod(upsample_rates)]  (modules/hifigan/hifigan.py:104-173)
 * as used by HifiGAN.spec2wav (vocoders/hifigan.py:55-69).  The NSF variant (use_pitch_embed) is bsg_hifigan_forward_nsf below.
 * ---------------------------------------------------------------------------------------------- */
typedef struct bsg_hifigan bsg_hifigan;

typedef struct {
  int32_t n_mel;                      /* 80 (conv_pre in-channels, hifigan.py:119)                       */
  int32_t upsample_initial_channel;   /* h['upsample_initial_channel']                                   */
  int32_t n_ups;                      /* len(h['upsample_rates'])                                        */
  int32_t upsample_rates[8];
  int32_t upsample_kernel_sizes[8];
  int32_t n_kernels;                  /* len(h['resblock_kernel_sizes']); kernels must be 3, 5, 7 or 11  */
  int32_t resblock_kernel_sizes[8];
  int32_t n_dil;                      /* dilations per ResBlock1 (3)                                     */
  int32_t resblock_dilations[8][4];
  int32_t weight_norm;                /* 1: weights come as (bias, weight_g, weight_v) triples           */
  int32_t use_nsf;                    /* h['use_pitch_embed']: NSF harmonic source (hifigan.py:111-132)   */
  int32_t sample_rate;                /* h['audio_sample_rate'] (NSF only)                               */
  int32_t harmonic_num;               /* 8 (hifigan.py:112)                                              */
  int32_t resblock;                   /* ABI v5: h['resblock']: 1 = ResBlock1 (hifigan.py:30-52), 2 = ResBlock2 (:70-91: per dilation
                                         ONE conv, x = conv_d(lrelu(x)) + x; 0 reads as 1)                   */
} bsg_hifigan_cfg;

int bsg_hifigan_n_weights(const bsg_hifigan_cfg* cfg);
/* dev_weights in HifiGanGenerator.state_dict() order: conv_pre, ups.i, resblocks.r.convs1.m (all m), then
 * resblocks.r.convs2.m (ResBlock2: resblocks.r.convs.m only), ..., conv_post; each conv contributes (bias, weight) — the layout after
 * remove_weight_norm() — or (bias, weight_g, weight_v) when cfg->weight_norm (checkpoint layout; folded here
 * as w = g * v / ||v||, norm over dims != 0).  Synchronises `stream` before returning. */
int bsg_hifigan_create(bsg_hifigan** out, const bsg_hifigan_cfg* cfg, const void* const* dev_weights,
                       int32_t n_weights, void* stream);
void bsg_hifigan_destroy(bsg_hifigan* h);
/* mel [B,n_mel,T] -> wav [B,1,T*prod(upsample_rates)].  May grow the workspace when B*T grows. */
int bsg_hifigan_forward(bsg_hifigan* h, const float* mel, float* wav, int32_t B, int32_t T, void* stream);
/* NSF-HiFiGAN: with cfg->use_nsf the weight list starts with m_source.l_linear.{weight,bias}
 * and noise_convs.i.{weight,bias} (state_dict order).  f0 [B,T] (Hz, 0 = unvoiced); the reference's two random draws
 * are supplied: rand_ini [B,harmonic_num+1] uniform[0,1) (torch.rand, source.py:53; column 0 is ignored) and
 * noise [B, T*hop, harmonic_num+1] N(0,1) (torch.randn_like, source.py:130). */
int bsg_hifigan_forward_nsf(bsg_hifigan* h, const float* mel, const float* f0, const float* rand_ini, const float* noise,
                            float* wav, int32_t B, int32_t T, void* stream);

/* PitchExtractor: mel [B,T,n_mel] -> pitch_pred [B,T,2] (may be NULL) and f0_denorm_pred [B,T]
 * (modules/fastspeech/pe.py:120-149; pitch_norm 'log', pitch_type 'frame').  dev_weights in PitchExtractor.state_dict()
 * order; pos_table [n_pos,256] = SinusoidalPositionalEmbedding table (row 0 zero), built by the host. */
typedef struct bsg_pitchext bsg_pitchext;
typedef struct {
  int32_t hidden_size;        /* 256 */
  int32_t n_mel;              /* 80  */
  int32_t conv_layers;        /* 2   (pe.py:121)                  */
  int32_t predictor_layers;   /* 5   (pe.py:134)                  */
  int32_t predictor_kernel;   /* hparams['predictor_kernel'] = 5  */
  int32_t use_uv;             /* hparams['pitch_type']=='frame' and hparams['use_uv'] */
  int32_t n_pos;
} bsg_pitchext_cfg;
int bsg_pitchext_n_weights(const bsg_pitchext_cfg* cfg);
int bsg_pitchext_create(bsg_pitchext** out, const bsg_pitchext_cfg* cfg, const void* const* dev_weights, int32_t n_weights,
                        const float* pos_table, void* stream);
void bsg_pitchext_destroy(bsg_pitchext* h);
int bsg_pitchext_forward(bsg_pitchext* h, const float* mel, float* pitch_pred, float* f0, int32_t B, int32_t T, void* stream);

/* w[d0,...] = g[d0] * v[d0,...] / ||v[d0,...]||   (remove_weight_norm, hifigan.py:175-182) */
int bsg_weight_norm_fold(const float* g, const float* v, float* w, int32_t dim0, int32_t inner, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Building block exported for unit tests: C[b] = op(A[b]) * B[b] (+bias)(+epilogue), fp32 MFMA.
 *   A: [M,K] row-major (lda);  B: trans_b ? [N,K] row-major : [K,N] row-major (ldb);  C: [M,N] (ldc).
 * ---------------------------------------------------------------------------------------------- */
int bsg_gemm_f32(const float* A, const float* Bm, float* C, const float* bias_m, const float* bias_n,
                 int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t trans_b,
                 int32_t batch, int64_t strideA, int64_t strideB, int64_t strideC, int32_t relu,
                 void* stream);

/* The GEMMs outside the residual stack form fp32 products on the 16-bit matrix pipe from hi + lo fp16 splits of both operands
 * (csrc/gemm.hip gemm_split_kernel; fp32-grade, |operand| < 4062).  A staged operand outside that range is counted instead of being
 * clipped silently: bsg_gemm_range_events waits for `stream`, returns the count (and resets it), and bsg_gemm_set_split(0) moves every
 * later GEMM to the fp32 matrix pipe — what the Python drop-ins do before they repeat the call (bisinger_amd/diffnet.py guarded). */
int bsg_gemm_set_split(int32_t enable);
int bsg_diffnet_set_h2(bsg_diffnet* h, int32_t enable);   /* 0: this handle's residual stack on the fp32 matrix pipe only (as BSG_H2=0) */
/* ABI v7, the tier in between: 0 = not the 16-row stack launch (residual_stack_q_kernel: its conv image holds 16 x, so its range guard
 * trips at |x| >= 3750) but the 32-row one (residual_stack_h2_kernel: |x + d| < 60000, about 8 % slower), as BSG_H2_Q=0 does for the
 * process.  Recovery order of a range event (status word 1) in the 16-row launch: bsg_diffnet_set_h2q(h, 0), repeat; only if that launch
 * trips too bsg_diffnet_set_h2(h, 0) (the fp32 matrix pipe, about 2x slower). */
int bsg_diffnet_set_h2q(bsg_diffnet* h, int32_t enable);
int bsg_gemm_range_events(int32_t* events, int32_t reset, void* stream);
/* ABI v5, non-blocking: enqueues a copy of the counter into *host_word (pinned host memory) on `stream`; nothing is reset. */
int bsg_gemm_range_events_async(int32_t* host_word, void* stream);

/* ABI v7: the range guard is PER HANDLE.  Every bsg_diffnet / bsg_fs2midi / bsg_hifigan / bsg_pitchext / bsg_fftden owns the device word
 * its split-fp16 kernels count out-of-range operands into, and its own switch between the split-fp16 products and the fp32 matrix pipe:
 * an out-of-range input to one model demotes THAT handle only ("distinct handles are independent", top of this file); inside a handle's
 * compute entries the state is looked up thread-locally, so two host threads on two handles do not see each other.  The two process-wide
 * functions above remain as the guard of the handle-less entries (bsg_gemm_f32, bsg_gemm_presplit_f32) and as a test hook:
 * bsg_gemm_set_split(0) (like BSG_GEMM_SPLIT=0 in the environment) is AND-ed into every handle's switch.
 *   kind: which handle type `handle` points to.
 *   bsg_handle_range_events        waits for `stream`; *events = waves of this handle's kernels that staged an operand beyond the fp16 range
 *                                  (|16 v| >= 65000, i.e. |v| >= 4062) since the last reset; reset != 0 zeroes the word when it is non-zero
 *   bsg_handle_range_events_async  no wait: enqueues the copy of the word into *host_word (pinned host memory) on `stream`
 *   bsg_handle_set_gemm_split      0: this handle's GEMMs / fused attention / ResBlock convolutions on the fp32 matrix pipe; 1: split-fp16 again
 *   bsg_handle_get_gemm_split      the EFFECTIVE switch (handle AND process-wide) */
enum { BSG_HANDLE_DIFFNET = 0, BSG_HANDLE_FS2MIDI = 1, BSG_HANDLE_HIFIGAN = 2, BSG_HANDLE_PITCHEXT = 3, BSG_HANDLE_FFTDEN = 4 };
int bsg_handle_range_events(int32_t kind, void* handle, int32_t* events, int32_t reset, void* stream);
int bsg_handle_range_events_async(int32_t kind, void* handle, int32_t* host_word, void* stream);
int bsg_handle_set_gemm_split(int32_t kind, void* handle, int32_t enable);
int bsg_handle_get_gemm_split(int32_t kind, void* handle, int32_t* enabled);

/* Round 4: the same products with PRE-SPLIT operands (csrc/gemm_h2w.hip gemm_h2w_kernel) — weights split once into hi / lo fp16 MFMA
 * fragments at create, activations written as hi / lo fp16 planes by their producers; FS2's Linear / Conv1d-FFN layers
 * (TB/modules/commons/common_layers.py:625-644,706-730) and the hoisted conditioner projections (TB/usr/diff/net.py:68) run on it
 * (BSG_GEMM_H2W=0: gemm_split_kernel).  Exported for unit tests and micro-benchmarks: out = epi(conv_taps(act, w) + bias), SAME padding,
 *   act [batch][rows][K], w [taps][Wn][K], out act_is_a ? [batch][rows][Wn] : [batch][Wn][rows];  Wn % 128 == 0, K % 64 == 0, taps <= 17;
 *   the launch is repeated `reps` times (timing); waits for `stream`. */
int bsg_gemm_presplit_f32(const float* act, const float* w, float* out, const float* bias, int32_t rows, int32_t Wn, int32_t K,
                          int32_t taps, int32_t act_is_a, int32_t batch, int32_t relu, int32_t reps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BISINGER_HIP_H */
