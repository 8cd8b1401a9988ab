#!/usr/bin/env python3
"""bench.py — mel-frames/s of the BiSinger mel-generation hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]          # N > 1 from a bare shell: starts its own N worker processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W               # or under torchrun (RANK / LOCAL_RANK / WORLD_SIZE from the env)

A "step" = one pass of the hot path over one batch of synthetic utterances already resident in HBM:
GaussianDiffusion.forward(infer=True) = FastSpeech2-MIDI encoder+decoder, 100 DDPM ancestral sampler steps of the 20-layer
DiffNet (on-device Philox noise), de-normalisation — and, for N > 1, the RCCL all-gather of the generated mels.

Workloads
  N = 1   BASELINE.json configs[1]: B=16, T=1000, 80 mel, fp32 (the configuration the metric is quoted on).
  N > 1   BASELINE.json configs[3]: B_total = 64 utterances sharded 64/N per GPU ("scaling": "strong"), one RCCL all-gather of
          the mels per pass; a weak-scaling figure (16 utterances per GPU) is measured after it and attached as `weak_scaling`.

The JSON line also carries
  roofline     : the dominant kernel (the residual stack launch: all 20 layers — and the step tail — per launch), timed live with
                 HIP events on the launch stream during the timed steps.  `achieved` prices ALGORITHMIC FLOPs of one layer over
                 the batch (the reference's direct convolution) against the dense peak of the matrix pipe the
                 products run on (fp16 for the default split-fp16 form, fp32 with BSG_H2=0); `frac_executed` prices the MFMAs
                 the form actually issues (3 fp16 products per fp32 product; 5/8 of the direct FLOPs for Winograd F(4,3)), i.e.
                 how busy the pipe is; `sustained_mhz` = shader clock over one whole timed launch of that kernel (s_memtime /
                 s_memrealtime, tile 0) and `frac_executed_at_sustained_clock` the same share against the peak at THAT clock.
                 `traffic` = HBM-side bytes from PMC counters (profiles/traffic*.json, looked up by the launch form that ran),
                 condition stated in `traffic_condition`; `traffic_build_matches` says whether those passes ran on the build
                 that was just timed;
  secondary    : (N = 1, untimed against the headline) BASELINE configs[2] — bf16 operands, B=64 — with its own roofline,
                 configs[4] — B=1, T=1000 mel generation + HiFi-GAN vocoder, real-time factor —, cfg3_rank: configs[3] as ONE of
                 its 8 ranks sees it (B_total = 64, front on 64 rows, 8 local rows, no collective) with the same 64 utterances on
                 one GPU beside it and their ratio = the strong scaling 8 GPUs can reach at most, captured_sampler: what a
                 stream-captured sampler loop runs and how fast, and f32_matrix_pipe: the headline workload with every product on
                 the fp32 matrix pipe (the default forms fp32 products on the 16-bit pipe from hi + lo fp16 splits: `arithmetic`);
  cpu_baseline : the oracle (PyTorch-CPU restatement of the reference, oracle/) timed on this host: ONE FULL pass (FS2 + all 100 sampler
                 steps, nothing extrapolated) at the thread count a short sweep finds fastest (`cores`); `survey_setting`: the same at
                 SURVEY §8d's setting (one socket's physical cores), median of 3 runs of a bounded sample (--cpu-steps steps, extrapolated).
                 The full pass replayed on the GPU with the same supplied noise must agree within 1e-3 on the de-normalised mel
                 (`parity`, asserted: a fast but wrong bench exits non-zero);
  range_headroom: how close the batch sits to the range contract of the split-fp16 launches (max |x_l| over layers and steps against
                 3750 / 60000, the observable GEMM operands against 4062) and every handle's event counters after the timed passes.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FRAME_LAYER = 2 * (512 * 768 + 512 * 256)   # dilated conv + output projection, algorithmic
FLOP_PER_FRAME_LAYER_EXEC = 2 * (4 * 256 * 256 + 512 * 256)   # Winograd F(2,3): 4 K=256 products on half the columns + out-proj
FLOP_PER_FRAME_LAYER_EXEC_F43 = 2 * (3 * 256 * 256 + 512 * 256)   # Winograd F(4,3): 6 K=256 products on a quarter of the columns + out-proj
PEAK_F32_MFMA_TFLOPS = 157.3                           # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0                                 # MI355X_MICROARCH.md, HBM3E spec (achievable ~6.3 TB/s)
HBM_BYTES_PER_FRAME_LAYER_BF16 = 4 * 256 * 4          # bf16 config: x in + x out fp32 (2 KB), conditioner term bf16 (1 KB), skip sum bf16 r+w (1 KB)
# the stack launch (all 20 layers in one kernel, x and skip on chip): conditioner term bf16 (1 KB) + x in and skip out fp32 once per 20
# layers (2 KB / 20) + the two 8-frame bf16 edges written and read per 64-frame tile (2 x 8 KB / 64)
HBM_BYTES_PER_FRAME_LAYER_BF16_STACK = 1024 + 2048 / 20 + 2 * 8192 / 64
PEAK_BF16_MFMA_TFLOPS = 2516.0                         # MI355X_MICROARCH.md, dense bf16
HIFIGAN_FLOP_PER_FRAME = 38.51e6
B_CFG1, B_CFG3_TOTAL, T_FRAMES, T_TXT, N_MEL, N_DIFF_STEPS = 16, 64, 1000, 100, 80, 100


class PhoneEncoder:
    """Stand-in for TokenTextEncoder: the model only needs len() and pad() (fastspeech/fs2.py:28,92)."""

    def __len__(self):
        return 65

    def pad(self):
        return 0


def build_model(device):
    import torch
    from bisinger_amd import synth
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    from bisinger_amd.hparams import hparams, set_hparams
    set_hparams(os.path.join(ROOT, 'bisinger_amd', 'configs', 'bisinger_diff100.yaml'), print_hparams=False)
    model = GaussianDiffusion(PhoneEncoder(), N_MEL, DIFF_DECODERS[hparams['diff_decoder_type']](hparams),
                              timesteps=hparams['timesteps'], K_step=hparams['K_step'],
                              spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    from collections import OrderedDict
    spec = OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items())
    w = synth.synth_state_dict(spec, 0, synth.DIFFNET_GAIN)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    return model.to(device).eval()


def build_vocoder(device):
    import torch
    import yaml
    from collections import OrderedDict
    from bisinger_amd import synth
    from bisinger_amd.hifigan import HifiGanGenerator
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'bisinger_amd', 'configs', 'hifigan.yaml')))
    voc = HifiGanGenerator(cfg)
    spec = OrderedDict((k, tuple(v.shape)) for k, v in voc.state_dict().items())
    voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 7).items()})
    voc = voc.to(device)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # (it prints 'Removing weight norm...' as the reference's does: stdout carries the ONE record)
        voc.remove_weight_norm()
    return voc, cfg


# ------------------------------------------------------------------------------------------------------------------
# host facts for the CPU baseline
# ------------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    """(model name, physical cores of ONE socket, sockets, logical CPUs) from /proc/cpuinfo."""
    model, phys, logical = 'unknown', {}, 0
    try:
        cur = {}
        for line in open('/proc/cpuinfo'):
            if ':' in line:
                k, v = [s.strip() for s in line.split(':', 1)]
                cur[k] = v
            elif not line.strip() and cur:
                logical += 1
                model = cur.get('model name', model)
                phys.setdefault(cur.get('physical id', '0'), set()).add(cur.get('core id', str(logical)))
                cur = {}
        if cur:
            logical += 1
            phys.setdefault(cur.get('physical id', '0'), set()).add(cur.get('core id', str(logical)))
    except OSError:
        pass
    sockets = max(1, len(phys))
    per_socket = max((len(v) for v in phys.values()), default=os.cpu_count() or 1)
    try:
        per_socket = min(per_socket, len(os.sched_getaffinity(0)))     # a container may own fewer CPUs than the host shows
    except (AttributeError, OSError):
        pass
    return model, per_socket, sockets, logical or (os.cpu_count() or 1)


def cpu_baseline_and_parity(model, inp_np, device, n_sample_steps, repeats=3):
    """The oracle on the host CPU, and the same pass replayed on the GPU with the same supplied noise.
      * `value`: ONE FULL pass — FS2 + all 100 sampler steps, nothing extrapolated — at the thread count a short sweep finds fastest
        (`cores` = that count): the strongest CPU figure this host gives, and the one the GPU / CPU ratio is quoted against;
      * `survey_setting`: SURVEY §8(d)'s definition — torch.set_num_threads(physical cores of one socket) — as the median of `repeats`
        runs of a bounded sample (FS2 + n_sample_steps sampler steps, extrapolated to 100): on a 64-core socket PyTorch-CPU is 2x slower
        there than at 16 threads, and three full passes would take five minutes."""
    import numpy as np
    import torch
    from bisinger_amd import synth
    from oracle import diffnet as odn, fs2 as ofs2, melgen as omg
    B = inp_np['txt_tokens'].shape[0]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    inp = {k: torch.from_numpy(v) for k, v in inp_np.items()}
    noise = torch.from_numpy(synth.synth_noise(N_DIFF_STEPS, B, N_MEL, T_FRAMES, seed=1))
    cpu_model, cores, sockets, logical = host_cpu_info()
    default_threads = torch.get_num_threads()
    xs_, cs_ = torch.randn(B, 1, N_MEL, T_FRAMES), torch.randn(B, 256, T_FRAMES)
    ts_ = torch.full((B,), 50, dtype=torch.long)
    sweep = {}
    for n in sorted({8, 16, 32, cores}):
        if n > logical:
            continue
        torch.set_num_threads(n)
        with torch.no_grad():
            odn.diffnet_forward(sd, xs_, ts_, cs_, 'denoise_fn.')
            t0 = time.perf_counter()
            odn.diffnet_forward(sd, xs_, ts_, cs_, 'denoise_fn.')
            sweep[n] = time.perf_counter() - t0
    best_n = min(sweep, key=sweep.get)
    # ---- the full pass at the fastest setting (measured end to end)
    torch.set_num_threads(best_n)
    with torch.no_grad():
        ofs2.fs2_forward(sd, inp)                                   # warm-up (allocator, oneDNN primitive cache)
        t0 = time.perf_counter()
        fs2_out = ofs2.fs2_forward(sd, inp)
        t1 = time.perf_counter()
        ret = omg.mel_gen(sd, inp, noise, fs2_out=fs2_out)
        t2 = time.perf_counter()
    full_s, full_fs2, full_steps = t2 - t0, t1 - t0, t2 - t1
    # ---- SURVEY's setting: bounded sample, median of `repeats`
    survey = None
    n_s = min(max(int(n_sample_steps), 1), N_DIFF_STEPS)
    if cores != best_n:
        torch.set_num_threads(cores)
        runs = []
        with torch.no_grad():
            ofs2.fs2_forward(sd, inp)
            for _ in range(repeats):
                t0 = time.perf_counter()
                fo = ofs2.fs2_forward(sd, inp)
                t1 = time.perf_counter()
                omg.mel_gen(sd, inp, noise[:n_s + 1], fs2_out=fo, **({} if n_s >= N_DIFF_STEPS else {'n_steps': n_s}))
                t2 = time.perf_counter()
                runs.append(((t1 - t0) + (t2 - t1) / n_s * N_DIFF_STEPS, t1 - t0, t2 - t1))
        runs.sort()
        est, t_fs2, t_steps = runs[len(runs) // 2]
        survey = {'value': B * T_FRAMES / est, 'unit': 'mel-frames/s', 'cores': cores, 'runs': len(runs), 'statistic': 'median',
                  'run_seconds_per_pass': [round(r[0], 2) for r in runs], 'est_seconds_per_pass': est,
                  'sample': f'torch.set_num_threads({cores}) = one socket\'s physical cores (SURVEY §8d): FS2-MIDI enc+dec ({t_fs2:.2f} s) + {n_s} of '
                            f'{N_DIFF_STEPS} sampler steps ({t_steps:.2f} s), steps extrapolated x{N_DIFF_STEPS}/{n_s}'}
    torch.set_num_threads(default_threads)
    # ---- the same pass on the GPU, same supplied noise
    d = {k: v.to(device) for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    g = model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True, noise=noise, **kw)
    torch.cuda.synchronize()
    dev = float((g['mel_out'].cpu() - ret['mel_out']).abs().max())
    parity = {
        'what': 'GPU vs fp32 oracle: de-normalised mel after FS2 + all 100 sampler steps, same supplied noise, full bench shape',
        'tolerance': 1e-3, 'max_abs': dev,
        'cond_max_abs': float((g['decoder_inp'].cpu() - ret['decoder_inp']).abs().max()),
        'fs2_mel_max_abs': float((g['fs2_mel'].cpu() - ret['fs2_mel']).abs().max()),
    }
    parity['ok'] = bool(np.isfinite(dev) and dev <= 1e-3 and parity['cond_max_abs'] <= 1e-3 and parity['fs2_mel_max_abs'] <= 1e-3)
    base = {
        'value': B * T_FRAMES / full_s, 'unit': 'mel-frames/s', 'cores': best_n, 'kind': 'port',
        'cpu_model': cpu_model, 'sockets': sockets, 'physical_cores_per_socket': cores, 'logical_cpus': logical,
        'runs': 1, 'statistic': 'one full pass', 'seconds_per_pass': round(full_s, 2),
        'sample': f'oracle (PyTorch-CPU restatement, fp32), ONE FULL pass at B={B}, T={T_FRAMES}: FS2-MIDI enc+dec ({full_fs2:.2f} s) + all '
                  f'{N_DIFF_STEPS} sampler steps ({full_steps:.2f} s), torch.set_num_threads({best_n}) = the fastest of the sweep '
                  f'{sorted(sweep)} (nothing extrapolated)',
        'thread_sweep_s_per_diffnet_call': {str(k): round(v, 3) for k, v in sweep.items()},
        'survey_setting': survey,
    }
    return base, parity


# ------------------------------------------------------------------------------------------------------------------
# rooflines
# ------------------------------------------------------------------------------------------------------------------
def build_digest():
    """Source digest of the library that is loaded (bisinger_amd/lib/build.sha256, written by bisinger_amd/build.py)."""
    try:
        return open(os.path.join(ROOT, 'bisinger_amd', 'lib', 'build.sha256')).read().strip()
    except OSError:
        return None


def traffic_from_profiles(bf16, frames_per_launch, path=None):
    """HBM-side bytes of the dominant kernel from the committed PMC summary (profiles/traffic*.json, written by
    tools/summarize_profiles.py) — only when that summary describes the launch form that just ran (`path`) at this size.
    -> (bytes or None, digest of the build the PMC passes ran on or None)."""
    tpath = os.path.join(ROOT, 'profiles', 'traffic_bf16.json' if bf16 else 'traffic.json')
    if not os.path.exists(tpath):
        return None, None
    tj = json.load(open(tpath))
    form = lambda q: q.replace('_tail', '')
    if form(tj.get('path', 'bf16' if bf16 else 'layer')) != form(path or ('bf16' if bf16 else 'layer')):
        return None, tj.get('build_sha256')
    if abs(tj.get('frames_per_launch', 16000) - frames_per_launch) < 1:
        return tj.get('residual_layer_kernel_hbm_bytes_per_launch'), tj.get('build_sha256')
    return None, tj.get('build_sha256')


def roofline(bf16, layer_ms, n_layer, steps, b_per_gpu, path=None, clock=None):
    """Roofline object of the dominant kernel from the live HIP-event timing of its launches.  `clock` = (MHz, span us) of
    DiffNet.clock_read(): the shader clock the chip held over the last timed stack launch."""
    if not n_layer:
        return None
    avg_ms = layer_ms / n_layer
    # what one launch covers, from the counts: the sampler runs a large batch as two concurrent launch chains over half the
    # rows each (bsg_ddpm_sample, BSG_DUAL), so a launch is B/2 x T frames and two are in flight
    frame_layers = steps * N_DIFF_STEPS * 20 * b_per_gpu * T_FRAMES
    frames_per_launch = frame_layers / n_layer
    concurrent = max(1, round(b_per_gpu * T_FRAMES / frames_per_launch))
    per_launch = FLOP_PER_FRAME_LAYER * frames_per_launch / (avg_ms * 1e-3) / 1e12
    achieved = per_launch * concurrent
    traffic, traffic_build = traffic_from_profiles(bf16, frames_per_launch, path)
    common = {'avg_launch_us': avg_ms * 1e3, 'launches_timed': n_layer, 'frames_per_launch': frames_per_launch,
              'concurrent_launches': concurrent, 'traffic': traffic,
              # the PMC passes behind `traffic` ran on the build with this source digest; false = the counters describe another build of
              # the library than the one that was just timed (re-run tools/run_profiles.sh)
              'traffic_build_sha256': traffic_build, 'traffic_build_matches': bool(traffic_build) and traffic_build == build_digest(),
              # s_memtime / s_memrealtime over one whole timed launch of the dominant kernel (tile 0): the clock the chip sustains under it
              'sustained_mhz': round(clock[0], 1) if clock and clock[0] else None,
              'sustained_clock_span_us': round(clock[1], 1) if clock and clock[0] else None,
              'traffic_condition': ('PMC passes of the same command (profiles/traffic*.json): HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, '
                                    'MI355X_MICROARCH.md) of the stack launch / 20 layers x launch groups; the stack path has one launch in '
                                    'flight, so the serialised PMC run is the timed condition') if (path or '').startswith('stack') else
                                   ('solo-launch PMC: rocprofv3 --pmc serialises kernels, so these are the HBM-side bytes (2 x FETCH_SIZE + '
                                    'WRITE_SIZE, MI355X_MICROARCH.md) of one launch running alone, not of two chains in flight')}
    if bf16 and (path or '').startswith('stack_bf16'):
        # one launch = all 20 layers of up to 256 64-frame tiles; the timed region is the launch group of one DiffNet evaluation and
        # n_layer counts its layers, so avg_ms is the time of one layer over all rows.  1,048,576 FLOP / 1.38 KB = 760 FLOP/B is
        # above the ridge (312 FLOP/B): matrix-pipe bound
        by = HBM_BYTES_PER_FRAME_LAYER_BF16_STACK * frames_per_launch
        return dict(common, kernel='residual_stack_bf16_kernel (20 fused DiffNet residual blocks per launch, bf16 MFMA operands; figures per layer)',
                    bound='mfma', achieved=achieved, peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s', frac=achieved / PEAK_BF16_MFMA_TFLOPS,
                    flop_per_layer=FLOP_PER_FRAME_LAYER * frames_per_launch, hbm_bytes_per_layer=by,
                    hbm_gbs=by * concurrent / (avg_ms * 1e-3) / 1e9, hbm_frac=by * concurrent / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    traffic_over_algorithmic=traffic / by if traffic else None,
                    note='achieved = algorithmic FLOPs of one layer over the batch / (launch-group duration / 20 layers), HIP events around '
                         'the launch group on its own stream')
    if bf16:
        # 16x the fp32 MFMA rate moves the layer under the HBM roof: AI = 1,048,576 FLOP / 4 KB = 256 FLOP/B against a
        # ridge of 2,500 TFLOP/s / 8 TB/s = 312 FLOP/B
        ach_gbs = HBM_BYTES_PER_FRAME_LAYER_BF16 * frames_per_launch * concurrent / (avg_ms * 1e-3) / 1e9
        return dict(common, kernel='residual_layer_bf16_kernel (fused DiffNet residual block, bf16 MFMA operands)', bound='hbm',
                    achieved=ach_gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=ach_gbs / PEAK_HBM_GBS,
                    bytes_per_launch=HBM_BYTES_PER_FRAME_LAYER_BF16 * frames_per_launch,
                    traffic_over_algorithmic=traffic / (HBM_BYTES_PER_FRAME_LAYER_BF16 * frames_per_launch) if traffic else None,
                    mfma_tflops=achieved, mfma_frac_of_bf16_peak=achieved / PEAK_BF16_MFMA_TFLOPS,
                    note='achieved = algorithmic bytes of one launch / its average duration (HIP events on its own stream) x launches in flight')
    if (path or '').startswith('stack_h2'):
        # split-fp16 stack launch (diffnet_h2.hip): every fp32 product as 3 fp16 MFMA products (hi hi, hi lo, lo hi), fp32 accumulate, so
        # the matrix pipe in use is the 16-bit one.  One launch = all 20 layers of up to 256 64-frame tiles; avg_ms = one layer over all rows
        executed = 3.0 * achieved
        part = {'stack_h2_pair64': 'residual_part_h2_kernel<2, 8, 4> (a 64-frame tile on 2 CUs of one XCD, each half of the channels)',
                'stack_h2_quad64': 'residual_part_h2_kernel<4, 4, 4> (a 64-frame tile on 4 CUs of one XCD, each a quarter of the channels)',
                'stack_h2_quad': 'residual_part_h2_kernel<4, 4, 2> (a 32-frame tile on 4 CUs of one XCD, each a quarter of the channels)'}.get(path or '')
        kname = part or ('residual_stack_q_kernel' if (path or '').startswith('stack_h2q') else 'residual_stack_h2_kernel')
        shape = 'v_mfma_f32_16x16x32_f16' if part or kname == 'residual_stack_q_kernel' else 'v_mfma_f32_32x32x16_f16'
        what = '20 fused DiffNet residual blocks per launch, the step tail in a launch of its own' if part else '20 fused DiffNet residual blocks + the step tail per launch'
        return dict(common, kernel=f'{kname} ({what}; fp32 operands split exactly into hi + lo '
                                   f'fp16 terms, 3 fp16 MFMAs ({shape}) per fp32 product, fp32 accumulate; x and the skip sum on chip; figures per layer)',
                    frac_executed_at_sustained_clock=(executed / (PEAK_BF16_MFMA_TFLOPS * clock[0] / 2400.0)) if clock and clock[0] else None,
                    bound='mfma', achieved=achieved, peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s', frac=achieved / PEAK_BF16_MFMA_TFLOPS,
                    executed_tflops=executed, frac_executed=executed / PEAK_BF16_MFMA_TFLOPS,
                    # (not a roofline fraction: how many times the fp32 MATRIX pipe's peak rate the algorithmic FLOPs run at — the products are
                    # formed on the 16-bit pipe, whose peak `frac` is priced against)
                    algorithmic_rate_over_fp32_matrix_pipe_peak=achieved / PEAK_F32_MFMA_TFLOPS,
                    flop_per_launch=FLOP_PER_FRAME_LAYER * frames_per_launch, achieved_per_launch=per_launch,
                    algorithmic_bytes_per_layer=6 * 256 * 4 * frames_per_launch,
                    traffic_over_algorithmic=traffic / (6 * 256 * 4 * frames_per_launch) if traffic else None,
                    note='achieved = algorithmic FLOPs (direct conv, SURVEY §8d) of one layer over the batch / (launch-group duration / 20 '
                         'layers), HIP events around the launch group on its own stream; peak = dense fp16 MFMA (= bf16, MI355X_MICROARCH.md): '
                         'the pipe the products run on.  frac_executed prices the 3 fp16 products issued per fp32 product = matrix-pipe busy '
                         'share.  Arithmetic is fp32-grade: product error a few 2^-24 (worst case 2^-21), fp32 accumulation; measured against float64 it equals the fp32-MFMA forms (tests/test_gpu_h2.py)')
    if path == 'stack_f43':
        # one launch = all 20 layers of up to 256 64-frame tiles, one workgroup per CU; the timed region is the launch group of one
        # DiffNet evaluation and n_layer counts its layers, so avg_ms is the time of one layer over all rows
        executed = achieved * FLOP_PER_FRAME_LAYER_EXEC_F43 / FLOP_PER_FRAME_LAYER
        # `frac` is a ROOFLINE fraction and never exceeds 1: the FLOPs the Winograd form executes (5/8 of the direct form) over the fp32 matrix
        # peak.  The algorithmic rate (SURVEY §8d prices the reference's direct convolution) can exceed that peak — the transforms remove
        # 3/8 of the multiplications — and is reported as `achieved` with its ratio under a name of its own.
        return dict(common, kernel='residual_stack_f43_kernel (20 fused DiffNet residual blocks per launch, Winograd F(4,3) GEMM1, x on chip; '
                                   'figures per layer)', bound='mfma',
                    achieved=achieved, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s', frac=executed / PEAK_F32_MFMA_TFLOPS,
                    algorithmic_rate_over_peak=achieved / PEAK_F32_MFMA_TFLOPS,
                    executed_tflops=executed, frac_executed=executed / PEAK_F32_MFMA_TFLOPS,
                    flop_per_launch=FLOP_PER_FRAME_LAYER * frames_per_launch, achieved_per_launch=per_launch,
                    algorithmic_bytes_per_layer=6 * 256 * 4 * frames_per_launch,
                    traffic_over_algorithmic=traffic / (6 * 256 * 4 * frames_per_launch) if traffic else None,
                    note='achieved = algorithmic FLOPs (direct conv, SURVEY §8d) of one layer over the batch / (launch-group duration / 20 '
                         'layers), HIP events around the launch group on its own stream; frac_executed prices the FLOPs the F(4,3) form '
                         'issues (5/8 of the direct form) = matrix-pipe busy share = `frac`.  The algorithmic rate (`algorithmic_rate_over_peak`) '
                         'exceeds the fp32 MFMA peak because the Winograd transforms remove 3/8 of the multiplications')
    wino = os.environ.get('BSG_WINO', '2') != '0'
    executed = achieved * (FLOP_PER_FRAME_LAYER_EXEC / FLOP_PER_FRAME_LAYER if wino else 1.0)
    return dict(common, kernel='residual_layer_kernel<false,true> (fused DiffNet residual block, Winograd GEMM1)', bound='mfma',
                achieved=achieved, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s', frac=achieved / PEAK_F32_MFMA_TFLOPS,
                executed_tflops=executed, frac_executed=executed / PEAK_F32_MFMA_TFLOPS,
                flop_per_launch=FLOP_PER_FRAME_LAYER * frames_per_launch, achieved_per_launch=per_launch,
                note='achieved = algorithmic FLOPs (direct conv, SURVEY §8d) of one launch / its average duration (HIP events on its own '
                     'stream) x launches in flight; frac_executed prices the FLOPs the Winograd form issues (3/4) = matrix-pipe busy share')


# ------------------------------------------------------------------------------------------------------------------
# the measured pass
# ------------------------------------------------------------------------------------------------------------------
class Workload:
    """Synthetic batch of B_total utterances resident in HBM; this rank generates `rows`."""

    def __init__(self, model, device, B_total, rank, world, emulate=False):
        """emulate: this process plays rank `rank` of `world` on its own — rows as that rank, no collective (secondary.cfg3_rank)."""
        import torch
        self.emulate = emulate
        from bisinger_amd import dist as bdist, synth
        self.model, self.B_total, self.rank, self.world = model, B_total, rank, world
        self.inp_np = synth.synth_inputs(B_total, T_TXT, T_FRAMES, seed=1)
        self.d = {k: torch.from_numpy(v).to(device) for k, v in self.inp_np.items()}
        self.kw = {k: self.d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
        self.rows = bdist.shard_rows(B_total, rank, world)
        self.b_local = self.rows.stop - self.rows.start

    def step(self, seed):
        from bisinger_amd import dist as bdist
        d = self.d
        out = self.model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True, seed=seed,
                         rows=self.rows if self.world > 1 else None, **self.kw)
        if self.emulate:
            return out['mel_out']
        if getattr(self, 'gather_on_host', False):      # --rehearse-one-gpu: gloo through host memory
            return bdist.all_gather_rows(out['mel_out'].cpu(), self.B_total, self.world, self.rank)
        return bdist.all_gather_rows(out['mel_out'], self.B_total, self.world, self.rank)


def timed(wl, steps, warmup, fence, profile=True):
    """W warm-up passes, then exactly K timed passes between fences; returns (seconds, layer_ms, n_layer_launches, last mel)."""
    net = wl.model.denoise_fn
    mel = None
    for i in range(warmup):
        mel = wl.step(1000 + i)
    fence()
    if profile:
        net.profile(True)
    t0 = time.perf_counter()
    for i in range(steps):
        mel = wl.step(i)
    fence()
    dt = time.perf_counter() - t0
    layer_ms, n_layer = net.profile_read() if profile else (0.0, 0)
    if profile:
        net.profile(False)
    return dt, layer_ms, n_layer, mel


def secondary_bf16(model, device, fence):
    """BASELINE configs[2]: B=64, T=1000, bf16 MFMA operands / fp32 accumulate.  1 warm-up + 3 timed passes."""
    import torch
    net = model.denoise_fn
    net.set_compute('bf16')
    try:
        wl = Workload(model, device, 64, 0, 1)
        dt, layer_ms, n_layer, mel = timed(wl, 3, 1, fence)
        ok = bool(torch.isfinite(mel).all())
        path = net.last_path()
        clock = net.clock_read()
    finally:
        net.set_compute('fp32')
    del wl
    torch.cuda.empty_cache()
    return {'config': {'workload': 'BASELINE.json configs[2]: B=64 x T=1000 x 80-mel, FS2-MIDI enc+dec + 100-step DDPM sampler, bf16 MFMA '
                                   'operands / fp32 accumulate in the residual layers (FS2, projections, sampler fp32)'},
            'dtype': 'bf16', 'metric': 'mel_frames_per_sec', 'value': 64 * T_FRAMES * 3 / dt, 'unit': 'mel-frames/s',
            'steps': 3, 'warmup': 1, 'ms_per_step': dt / 3 * 1e3, 'finite': ok, 'path': path,
            'roofline': roofline(True, layer_ms, n_layer, 3, 64, path, clock)}


def secondary_cfg3_rank(model, device, fence):
    """BASELINE configs[3] as ONE of its 8 ranks sees it, on one GPU and without the collective: B_total = 64, the token-level front on all
    64 rows, frames and sampler on rows 8..15.  Beside it the same 64 utterances on one GPU; their ratio is what 8-GPU strong scaling
    can reach at most (the all-gather of 8 x 2.56 MB over xGMI is not in it)."""
    import torch
    net = model.denoise_fn
    wl = Workload(model, device, B_CFG3_TOTAL, 1, 8, emulate=True)
    # both sides of the ratio are timed WITHOUT the per-launch HIP events of the profiling mode (round 5: the rank's passes carried them — two
    # launches per sampler step, 400 event records per pass — and the one-GPU passes did not); the per-layer figure comes from one more pass
    dt, _, _, mel = timed(wl, 3, 1, fence, profile=False)
    _, layer_ms, n_layer, _ = timed(wl, 1, 0, fence)
    path = net.last_path()
    front_rows = model.fs2.last_rows()[0]
    ok = bool(torch.isfinite(mel).all()) and tuple(mel.shape) == (8, T_FRAMES, N_MEL) and front_rows == 8 * T_TXT
    wl1 = Workload(model, device, B_CFG3_TOTAL, 0, 1)
    dt1, _, _, mel1 = timed(wl1, 2, 1, fence, profile=False)
    # Philox is indexed by the global row: with the same seed the shard reproduces the same rows of the unsharded pass
    same = float((wl1.step(77)[wl.rows] - wl.step(77)).abs().max())
    del wl, wl1
    torch.cuda.empty_cache()
    t_rank, t_one = dt / 3 * 1e3, dt1 / 2 * 1e3
    return {'config': {'workload': 'BASELINE.json configs[3], one rank emulated on one GPU: B=64 total; token front: ESM K / V from all 64 rows\' lang ids, '
                                   'Q / FFT encoder on rows 8..15 (bsg_fs2midi_encode_rows); frame-level FS2 + 100-step DDPM sampler on rows 8..15 '
                                   '(8 utterances x T=1000), fp32, no collective'},
            'dtype': 'f32', 'metric': 'mel_frames_per_sec', 'value': 8 * T_FRAMES * 3 / dt, 'unit': 'mel-frames/s (this rank\'s 8 utterances)',
            'steps': 3, 'warmup': 1, 'ms_per_step': t_rank, 'finite': ok, 'path': path,
            'ms_per_step_b64_one_gpu': t_one, 'value_b64_one_gpu': B_CFG3_TOTAL * T_FRAMES / (t_one * 1e-3),
            'predicted_strong_scaling_8': t_one / t_rank, 'target_strong_scaling_8': 6.5,
            # the absolute figure beside the ratio (the ratio falls when the ONE-GPU pass gets faster): 8 ranks x 8 utterances per rank pass,
            # collective (8 x 2.56 MB all-gather) and host contention not included
            'projected_8gpu_frames_per_sec': B_CFG3_TOTAL * T_FRAMES / (t_rank * 1e-3),
            'front_token_rows_encoded': front_rows,
            'shard_vs_unsharded_rows_max_abs': same,
            'avg_layer_us': layer_ms / n_layer * 1e3 if n_layer else None, 'handoff_timeouts': net.handoff_timeouts()}


def range_headroom(model, wl, device):
    """How close the workload sits to the range contract of the split-fp16 launches — measured, outside the timed region, on
    the bench's own batch: the residual stream x_l of all 20 layers (per-layer launches of the fp32 matrix pipe, bsg_diffnet_residual_layer,
    so that every x_l is visible in HBM) at four points of the sampler trajectory, the running skip sum, and the one GEMM operand that is
    observable from outside (the condition that enters the 20 hoisted conditioner projections); beside them the limits and the event
    counters of every handle after the timed passes.  VERDICT r05 item 4c."""
    import numpy as np
    import torch
    net = model.denoise_fn
    d = wl.d
    ret = model.fs2(d['txt_tokens'], d['mel2ph'], d['spk_embed'], None, None, None, None, skip_decoder=True, infer=True, **wl.kw)
    cond = ret['decoder_inp'].transpose(1, 2).contiguous()
    B, H, T = cond.shape
    w_in = net.input_projection.weight.detach().cpu().numpy()[:, :, 0].astype(np.float64)
    b_in = net.input_projection.bias.detach().cpu().numpy().astype(np.float64)
    x = model.philox_normal((B, 1, N_MEL, T), device, 4321, 0)
    k_step = model.K_step
    worst_x, worst_skip, worst_in = 0.0, 0.0, float(x.abs().max())
    try:
        net.set_split_fp16(False)          # per-layer launches (fp32 matrix pipe): x_l of every layer goes through HBM
        net._ensure_bound(cond)
        for t_hi, n in ((100, 33), (67, 33), (34, 33), (1, 0)):
            t_i = t_hi - 1
            xa = np.maximum(np.einsum('ck,bkt->bct', w_in, x[:, 0].cpu().numpy().astype(np.float64)) + b_in[None, :, None], 0.0)
            xl = torch.from_numpy(xa.astype(np.float32)).to(device)
            skip = torch.zeros_like(xl)
            tt = torch.full((B,), t_i, device=device, dtype=torch.long)
            worst_x = max(worst_x, float(xl.abs().max()))
            for layer in range(net.n_layers):
                xl = net.residual_layer(layer, xl, tt, skip)
                worst_x = max(worst_x, float(xl.abs().max()))
            worst_skip = max(worst_skip, float(skip.abs().max()))
            if n:
                model.K_step = t_hi
                x = model.sample(cond, x, seed=4321, n_steps=n)
                worst_in = max(worst_in, float(x.abs().max()))
    finally:
        model.K_step = k_step
        net.set_split_fp16(True)
    give, rng = net.take_health()
    return {'max_abs_x_residual_stream': worst_x, 'limit_x_16_row_stack_launch': 3750.0, 'limit_x_plus_d_32_row_and_part_launches': 60000.0,
            'max_abs_skip_sum': worst_skip, 'limit_skip_sum_step_tail': 60000.0,
            'max_abs_gemm_operand_observed': max(float(cond.abs().max()), worst_in), 'limit_gemm_operand': 4062.0,
            'gemm_operand_note': 'the operands visible at the boundary: the condition entering the 20 conditioner projections and x entering the input '
                                 'projection; FS2-internal operands are LayerNorm outputs and GELU / attention outputs of O(10)',
            'sampled_at': 'all 20 layers at sampler steps t = 99, 66, 33, 0 of a Philox trajectory on the bench batch',
            'range_events': {'diffnet_stack_launches': int(rng), 'diffnet_gemms': int(net.gemm_range_peek()), 'fs2_gemms': int(model.fs2.gemm_range_peek())},
            'range_strikes': {'diffnet_gemms': net.gemm_range_strikes, 'fs2_gemms': model.fs2.gemm_range_strikes,
                              'diffnet_stack_16_row': getattr(net, '_h2q_strikes', 0), 'diffnet_stack_split_fp16': getattr(net, '_h2_strikes', 0)},
            'handoff_give_ups': int(give)}


def secondary_captured(model, device, fence):
    """What a stream-captured sampler loop runs.  Round 4: the stack / part launches keep their launch epoch in device memory (every workgroup
    reads it at entry, the last one through its layers advances it), so a captured GaussianDiffusion.sample() runs the SAME launches as the
    eager call (rounds 2-3: per-layer launches on the fp32 matrix pipe, 2.2x slower).  Captured once, replayed 3 times, at the headline shape,
    with the eager loop timed beside it."""
    import torch
    net = model.denoise_fn
    B, T = B_CFG1, T_FRAMES
    cond = torch.randn(B, 256, T, device=device)
    x0 = torch.randn(B, 1, N_MEL, T, device=device)
    model.sample(cond, x0.clone(), seed=3)                   # eager: binds, warms up
    fence()
    eager_path = net.last_path()
    t0 = time.perf_counter()
    for _ in range(3):
        xe = model.sample(cond, x0.clone(), seed=3)
    fence()
    dt_eager = (time.perf_counter() - t0) / 3
    xg = x0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                model.sample(cond, xg, seed=3)
            cap_path = net.last_path()
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            xg.copy_(x0)                                     # the sampler updates x in place: the comparison replay starts from the eager run's x_T
            g.replay()
            torch.cuda.synchronize()
        torch.cuda.current_stream().wait_stream(side)
        return {'what': 'hipGraph capture of the 100-step sampler loop at B=16, T=1000 (sampler only: no FS2), replayed 3 times',
                'captured_path': cap_path, 'eager_path': eager_path, 'ms_per_replay': dt * 1e3, 'ms_per_eager_call': dt_eager * 1e3,
                'replay_over_eager': dt / dt_eager, 'replay_bit_identical_to_eager': bool(torch.equal(xg, xe)),
                'sampler_frames_per_sec_captured': B * T / dt, 'finite': bool(torch.isfinite(xg).all()),
                'handoff_timeouts_after_replays': net.handoff_timeouts()}
    except Exception as e:      # a secondary must never take the headline down
        return {'error': f'{type(e).__name__}: {e}'}


def under_profiler():
    """rocprofv3 preloads its tool library into the process: a child benchmark would inherit it (and its counter passes)."""
    return any(k.startswith(('ROCPROF', 'ROCP_', 'ROCPROFILER')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')


def secondary_fp32_pipe(steps, warmup):
    """The headline workload with every product on the fp32 matrix pipe (BSG_H2=0: F(4,3) stack launch, BSG_GEMM_SPLIT=0: fp32-MFMA GEMMs)
    — a child process, because the launch-form switches are read once per process.  Timed like the headline (the same --steps / --warmup),
    no secondaries, no CPU leg."""
    if under_profiler():
        return {'skipped': 'running under a profiler'}
    env = dict(os.environ, BSG_H2='0', BSG_GEMM_SPLIT='0')
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--no-secondary', '--cpu-steps', '0', '--steps', str(steps), '--warmup', str(warmup)],
                             env=env, capture_output=True, text=True, timeout=600)
        j = json.loads(out.stdout.strip().splitlines()[-1])
        r = j['roofline']
        return {'config': {'workload': j['config']['workload'] + '; BSG_H2=0 BSG_GEMM_SPLIT=0: every product on the fp32 matrix pipe'},
                'dtype': 'f32', 'metric': 'mel_frames_per_sec', 'value': j['value'], 'unit': 'mel-frames/s', 'steps': j['steps'],
                'warmup': j['warmup'], 'ms_per_step': j['ms_per_step'],
                'roofline': {k: r.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_executed', 'algorithmic_rate_over_peak',
                                                   'avg_launch_us')}}
    except Exception as e:      # the secondary must never take the headline down
        return {'error': f'{type(e).__name__}: {e}'}


def secondary_e2e(model, device, fence):
    """BASELINE configs[4]: B=1, T=1000 mel generation + HiFi-GAN vocoder, 22.05 kHz; real-time factor."""
    import torch
    voc, cfg = build_vocoder(device)
    sr, hop = cfg['audio_sample_rate'], 256
    wl = Workload(model, device, 1, 0, 1)

    def run(seed):
        mel = wl.step(seed)
        return mel, voc(mel.transpose(1, 2))
    mel, wav = run(1)
    fence()
    n = 3
    t0 = time.perf_counter()
    for i in range(n):
        mel, wav = run(2 + i)
    fence()
    dt = (time.perf_counter() - t0) / n
    for _ in range(3):                       # the vocoder alone: warm-up, then the median of 3 batches of 10 forwards (a single batch of 10 x 1 ms
        voc(mel.transpose(1, 2))             # read 46 % high on one box of round 4)
    fence()
    dvs = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            voc(mel.transpose(1, 2))
        fence()
        dvs.append((time.perf_counter() - t0) / 10)
    dv = sorted(dvs)[1]
    audio_s = T_FRAMES * hop / sr
    voc_tf = HIFIGAN_FLOP_PER_FRAME * T_FRAMES / dv / 1e12
    voc_traffic, voc_traffic_build = None, None
    try:      # HBM-side bytes of one vocoder forward at this shape from the committed PMC passes (tools/run_profiles.sh -> profiles/traffic_voc.json)
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic_voc.json')))
        if tj.get('B') == 1 and tj.get('T') == T_FRAMES:
            voc_traffic, voc_traffic_build = tj.get('hbm_bytes_per_forward'), tj.get('build_sha256')
    except (OSError, ValueError):
        pass
    return {'config': {'workload': f'BASELINE.json configs[4]: B=1 x T={T_FRAMES} mel generation (FS2 + 100-step DDPM, fp32) + HiFi-GAN vocoder, '
                                   f'{sr} Hz, {T_FRAMES * hop} samples'},
            'dtype': 'f32', 'metric': 'real_time_factor', 'value': dt / audio_s, 'unit': 'seconds of compute per second of audio',
            'higher_is_better': False, 'steps': n, 'warmup': 1, 'ms_per_step': dt * 1e3, 'audio_seconds': audio_s,
            'vocoder_ms': dv * 1e3, 'vocoder_rtf': dv / audio_s, 'finite': bool(torch.isfinite(wav).all()),
            'handoff_timeouts': model.denoise_fn.handoff_timeouts(),
            # HBM-bound by SURVEY §8(d) (channels <= 128, ~10 FLOP/B): `achieved` = HBM-side bytes the forward really moves (PMC) / its wall
            # time; with no PMC summary of this build, the algorithmic bytes (mel in + wav out) — then `frac` says how far the launch chain
            # is from streaming its input and output once.  The matrix work is priced against the pipe it runs on (16-bit: split-fp16 pairs)
            'roofline': (lambda by_: {'kernel': 'HiFi-GAN generator forward (all launches of one vocoder call)', 'bound': 'hbm',
                         'achieved': by_ / dv / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': by_ / dv / 1e9 / PEAK_HBM_GBS,
                         'bytes_priced': 'HBM-side bytes of one forward (PMC)' if voc_traffic else 'algorithmic bytes (no PMC summary of this shape)',
                         'traffic': voc_traffic, 'algorithmic_bytes_per_forward': N_MEL * T_FRAMES * 4 + T_FRAMES * hop * 4,
                         'traffic_over_algorithmic': voc_traffic / (N_MEL * T_FRAMES * 4 + T_FRAMES * hop * 4) if voc_traffic else None,
                         'traffic_build_matches': bool(voc_traffic_build) and voc_traffic_build == build_digest(),
                         'traffic_condition': 'PMC passes of tools/prof_vocoder.py at B=1, T=1000 (profiles/traffic_voc.json): 2 x FETCH_SIZE + WRITE_SIZE '
                                              'summed over the launches of one forward; algorithmic bytes: mel in 0.32 MB + wav out 1.02 MB',
                         'mfma_tflops_algorithmic': voc_tf, 'mfma_frac_of_pipe_in_use': voc_tf / PEAK_BF16_MFMA_TFLOPS,
                         'mfma_frac_executed_of_pipe_in_use': 3.0 * voc_tf / PEAK_BF16_MFMA_TFLOPS,
                         'note': '38.51 MFLOP per mel frame (SURVEY §8d) / vocoder wall time for the matrix figures: the 32/64/16-channel ResBlock pairs run as '
                                 'split-fp16 products (3 fp16 MFMAs per fp32 product) on the 16-bit pipe, so that pipe\'s dense peak is the price; the 8-channel '
                                 'stage and conv_post run on the vector pipe'})(voc_traffic or (N_MEL * T_FRAMES * 4 + T_FRAMES * hop * 4))}


# ------------------------------------------------------------------------------------------------------------------
# process management: `python bench.py --gpus N` from a bare shell starts its own workers
# ------------------------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """Parent of an N-rank run.  Touches no GPU (device_count() does not initialise HIP on this image): it only starts one
    fresh child per GPU with the torchrun environment, forwards rank 0's record and returns the worst exit code."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.selftest_procs and not args.rehearse_one_gpu:
        sys.exit(f'bench.py --gpus {args.gpus}: this node exposes {n_dev} GPUs')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    # rank 0 is back: the others finish with it (they share its barriers).  If it died before the rendezvous they would sit in the
    # RCCL / gloo initialisation for minutes — give them a short grace period, then end exactly the processes started here
    grace = 30 if procs[0].returncode == 0 else 5
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=grace))
        except subprocess.TimeoutExpired:
            p.terminate()
            try:
                rcs.append(p.wait(timeout=10))
            except subprocess.TimeoutExpired:
                p.kill()
                rcs.append(p.wait())
    lines = [l for l in (out0 or '').splitlines() if l.strip()]
    rec = next((l for l in reversed(lines) if l.startswith('{')), None)
    for l in lines:
        if l is not rec:
            print(l, file=sys.stderr)
    if rec:
        print(rec, flush=True)
    bad = [rc for rc in rcs if rc != 0]
    sys.exit(bad[0] if bad else (0 if rec else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--cpu-steps', type=int, default=10,
                    help='0 = no CPU baseline and no parity replay.  Otherwise the baseline is ONE FULL 100-step pass of the oracle at the fastest '
                         'thread count of a short sweep (~40 s), and this is the number of sampler steps of the bounded sample that is run 3 times '
                         'at SURVEY §8(d)\'s thread setting (one socket\'s physical cores) and extrapolated, reported beside it')
    ap.add_argument('--dtype', choices=('f32', 'bf16'), default='f32',
                    help='arithmetic of the fused residual layers of the HEADLINE: f32 = configs[1] (default); bf16 = configs[2] (with --batch 64)')
    ap.add_argument('--batch', type=int, default=None, help='utterances per GPU of the headline (default: 16 at N=1, 64/N at N>1)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs[2] / configs[4] secondaries (N = 1)')
    ap.add_argument('--no-weak', action='store_true', help='skip the weak-scaling figure (N > 1)')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL and run the all-gather even with one rank (self-test)')
    ap.add_argument('--rehearse-one-gpu', action='store_true',
                    help='rehearsal of the N > 1 code path on a box with ONE GPU (tests/test_gpu_dist.py): every rank computes on cuda:0, the '
                         'barrier / all-gather / max-over-ranks go over gloo through host memory.  Exercises the sharded workload, the timing '
                         'protocol and the record; its numbers mean nothing (the ranks share one chip)')
    ap.add_argument('--selftest-procs', action='store_true',
                    help='process-management self-test (tests/test_dist_cpu.py): the N workers only rendezvous over gloo on the CPU, '
                         'all-gather their ranks and rank 0 prints a record; measures nothing')
    args = ap.parse_args()

    env_world = int(os.environ.get('WORLD_SIZE', '0') or 0)
    if args.gpus > 1 and env_world != args.gpus:
        if env_world > 1:
            sys.exit(f'--gpus {args.gpus} but WORLD_SIZE={env_world}')
        return self_launch(args, sys.argv[1:])

    import torch
    from bisinger_amd import dist as bdist
    rank, local_rank, world = bdist.env_world()
    if args.gpus == 1:
        world, rank, local_rank = 1, 0, 0
    if args.selftest_procs:
        import torch.distributed as dist
        bdist.init_distributed('gloo')
        got = bdist.all_gather_rows(torch.tensor([[float(rank)]]), world, world, rank)
        dist.barrier()
        if rank == 0:
            print('noise on stdout before the record')
            print(json.dumps({'selftest': True, 'n_gpus': world, 'ranks': [int(v) for v in got.reshape(-1).tolist()]}), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X: the product path has no CPU fallback (GPUs visible: 0)')
    rehearse = bool(args.rehearse_one_gpu) and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if rehearse:
        bdist.init_distributed('gloo')
    elif world > 1:
        bdist.init_distributed('nccl')
    elif args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=device)

    torch.set_grad_enabled(False)
    model = build_model(device)
    bf16 = args.dtype == 'bf16'
    model.denoise_fn.set_compute('bf16' if bf16 else 'fp32')

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], device='cpu' if rehearse else device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    strong = world > 1 and args.batch is None
    B_total = B_CFG3_TOTAL if strong else (args.batch or B_CFG1) * world
    wl = Workload(model, device, B_total, rank, world)
    wl.gather_on_host = rehearse
    if args.force_dist and world == 1:
        _step = wl.step

        def step_fd(seed):
            mel = _step(seed)
            full = torch.empty_like(mel)
            dist.all_gather_into_tensor(full, mel.contiguous())
            return full
        wl.step = step_fd
    dt, layer_ms, n_layer, mel = timed(wl, args.steps, args.warmup, fence)
    clock = model.denoise_fn.clock_read()
    dt = max_over_ranks(dt)
    assert mel.shape == (B_total, T_FRAMES, N_MEL) and bool(torch.isfinite(mel).all())
    timeouts = model.denoise_fn.handoff_timeouts()

    weak = None
    if world > 1 and strong and not args.no_weak:
        wl2 = Workload(model, device, B_CFG1 * world, rank, world)
        wl2.gather_on_host = rehearse
        k2 = max(2, min(args.steps, 5))
        dt2, _, _, mel2 = timed(wl2, k2, 1, fence, profile=False)
        dt2 = max_over_ranks(dt2)
        weak = {'scaling': 'weak', 'global_batch': B_CFG1 * world, 'utterances_per_gpu': B_CFG1, 'steps': k2, 'warmup': 1,
                'value': B_CFG1 * world * T_FRAMES * k2 / dt2, 'unit': 'mel-frames/s', 'ms_per_step': dt2 / k2 * 1e3,
                'finite': bool(torch.isfinite(mel2).all())}
        del wl2

    # every rank flushes what native libraries buffered on stdout (RCCL prints its load path) before rank 0 prints the record,
    # so that the record is the last line of the job's merged stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if use_dist:
        dist.barrier()
    rc = 0
    if rank == 0:
        value = B_total * T_FRAMES * args.steps / dt
        cfg_name = 'configs[3]' if strong else ('configs[2]' if bf16 else 'configs[1]')
        rec = {
            'metric': 'mel_frames_per_sec', 'value': value, 'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'bf16' if bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE.json {cfg_name}: B={B_total} total = {wl.b_local}/GPU x T={T_FRAMES} x {N_MEL}-mel, FS2-MIDI enc+dec + '
                                   f'{N_DIFF_STEPS}-step DDPM sampler (20-layer DiffNet, 256 ch), '
                                   + ('bf16 MFMA operands / fp32 accumulate in the residual layers (FS2, projections, sampler fp32)'
                                      if bf16 else 'fp32') + ', formula weights',
                       'global_batch': B_total, 'utterances_per_gpu': wl.b_local, 'frames_per_utt': T_FRAMES, 'diffusion_steps': N_DIFF_STEPS,
                       'parallelism': f'utterance-sharded x{world}, one RCCL all-gather of the mels per pass' if world > 1 else 'single GPU'},
            'roofline': roofline(bf16, layer_ms, n_layer, args.steps, wl.b_local, model.denoise_fn.last_path(), clock),
            'handoff_timeouts': timeouts,
        }
        if not bf16 and model.denoise_fn.last_path().startswith('stack_h2'):
            rec['arithmetic'] = ('fp32 tensors and fp32 results; in the residual stack, the step tail and the GEMMs every fp32 product is formed on the '
                                 '16-bit matrix pipe as ah*bh + ah*bl + al*bh from EXACT hi + lo fp16 splits of both operands (fp32 accumulate; '
                                 'product error a few 2^-24, worst case 2^-21).  Error against float64 equals that of the fp32-MFMA forms, also at the edges of the scheme (tests/test_gpu_h2.py); '
                                 'secondary.f32_matrix_pipe is the same workload with every product on the fp32 matrix pipe')
        if use_dist:
            rec['ranks_seen_by_rccl'] = dist.get_world_size()
            rec['collective_backend'] = dist.get_backend()
        if rehearse:
            rec['rehearsal'] = 'every rank on cuda:0, collectives over gloo through host memory: the numbers of this line mean nothing'
        if weak:
            rec['weak_scaling'] = weak
        if world == 1 and not args.no_secondary and not bf16 and args.batch is None:
            rec['secondary'] = {'bf16_b64': secondary_bf16(model, device, fence), 'e2e_rtf_b1': secondary_e2e(model, device, fence),
                                'cfg3_rank': secondary_cfg3_rank(model, device, fence),
                                'captured_sampler': secondary_captured(model, device, fence),
                                'f32_matrix_pipe': secondary_fp32_pipe(args.steps, args.warmup)}
            c3 = rec['secondary'].get('cfg3_rank') or {}
            if c3.get('value_b64_one_gpu'):
                # the base of a 1 -> 8 STRONG-scaling ratio: the N > 1 runs shard B = 64 utterances (configs[3]); this line's own `value` is
                # configs[1] (B = 16) and must not be divided into them
                rec['strong_denominator'] = {'B': B_CFG3_TOTAL, 'value': c3['value_b64_one_gpu'], 'ms_per_step': c3['ms_per_step_b64_one_gpu'],
                                             'unit': 'mel-frames/s', 'note': 'configs[3] on ONE GPU: all 64 utterances, same pass as the N > 1 runs'}
        if world == 1 and not bf16 and not args.no_secondary and args.batch is None:
            # (with the secondaries: the probe runs per-layer launches of the fp32 matrix pipe, which do not belong into a profile of the headline)
            rec['range_headroom'] = range_headroom(model, wl, device)
        if world == 1 and args.cpu_steps > 0:
            base, parity = cpu_baseline_and_parity(model, wl.inp_np, device, args.cpu_steps)
            rec['cpu_baseline'] = base
            rec['parity'] = parity
            rec['gpu_over_cpu'] = value / base['value']          # both sides measured end to end; the CPU at its fastest thread setting
            if base.get('survey_setting'):
                rec['gpu_over_cpu_at_survey_thread_setting'] = value / base['survey_setting']['value']
            if not parity['ok']:
                print(f'bench.py: PARITY FAILED {json.dumps(parity)}', file=sys.stderr)
                rc = 1
        # native libraries (RCCL prints its load path) write through C stdio, which is block-buffered when stdout is a pipe:
        # flush it first so that the JSON record is the LAST line of stdout
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(rec), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == '__main__':
    main()
