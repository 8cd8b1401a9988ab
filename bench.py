#!/usr/bin/env python3
"""bench.py — mel-frames/s of the BiSinger mel-generation hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic utterances already resident in HBM:
GaussianDiffusion.forward(infer=True) = FastSpeech2-MIDI encoder+decoder, 100 DDPM ancestral sampler
steps of the 20-layer DiffNet (on-device Philox noise), de-normalisation — and, for N > 1, the RCCL
all-gather of the generated mels.  Workload at N = 1: BASELINE.json configs[1] (B=16, T=1000, 80 mel,
fp32).  Weak scaling: every rank generates 16 utterances of the 16*N batch.  Inside a rank the sampler runs
the batch as two concurrent launch chains over half the rows each (two HIP streams, BSG_DUAL=0 disables).

The JSON line also carries
  roofline     : the dominant kernel (fused residual layer), timed live with HIP events on the launch
                 stream during the timed steps, against the fp32 MFMA peak (algorithmic FLOPs of the reference's
                 direct convolution; the kernel's Winograd form executes 3/4 of them); `traffic` = the kernel's
                 measured HBM-side bytes per launch (profiles/traffic*.json, PMC);
  --dtype bf16 --batch 64 : BASELINE configs[2] (bf16 MFMA operands / fp32 accumulate), priced against the HBM roof;
  cpu_baseline : the oracle (PyTorch-CPU restatement of the reference, oracle/) timed on this host on a
                 bounded sample of the same workload (rank 0, N = 1 only), extrapolated to 100 steps;
                 the same sample is also replayed on the GPU with the same supplied noise -> `parity`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FRAME_LAYER = 2 * (512 * 768 + 512 * 256)   # dilated conv + output projection
PEAK_F32_MFMA_TFLOPS = 157.3                           # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0                                 # MI355X_MICROARCH.md, HBM3E spec (achievable ~6.3 TB/s)
HBM_BYTES_PER_FRAME_LAYER_BF16 = 4 * 256 * 4          # bf16 config: x in + x out fp32 (2 KB), conditioner term bf16 (1 KB), skip sum bf16 r+w (1 KB)
B_PER_GPU, T_FRAMES, T_TXT, N_MEL, N_DIFF_STEPS = 16, 1000, 100, 80, 100


class PhoneEncoder:
    """Stand-in for TokenTextEncoder: the model only needs len() and pad() (fastspeech/fs2.py:28,92)."""

    def __len__(self):
        return 65

    def pad(self):
        return 0


def build_model(device):
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


def cpu_baseline_and_parity(model, inp_np, device, n_sample_steps):
    """Oracle on the host CPU for FS2 + n_sample_steps sampler steps; the same steps on the GPU with the same noise."""
    from bisinger_amd import synth
    from oracle import fs2 as ofs2, melgen as omg
    B = inp_np['txt_tokens'].shape[0]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    inp = {k: torch.from_numpy(v) for k, v in inp_np.items()}
    noise = torch.from_numpy(synth.synth_noise(n_sample_steps, B, N_MEL, T_FRAMES, seed=1))
    # thread count: PyTorch's default (all hardware threads) is far from the fastest on a many-core host, so time one
    # DiffNet call at a few settings and keep the best -- the baseline should be the CPU path at its best
    from oracle import diffnet as odn
    default_threads = torch.get_num_threads()
    xs_, cs_ = torch.randn(B, 1, N_MEL, T_FRAMES), torch.randn(B, 256, T_FRAMES)
    ts_ = torch.full((B,), 50, dtype=torch.long)
    trials = {}
    for n in sorted({8, 16, 32, 64, default_threads}):
        if n > (os.cpu_count() or 8):
            continue
        torch.set_num_threads(n)
        with torch.no_grad():
            odn.diffnet_forward(sd, xs_, ts_, cs_, 'denoise_fn.')
            t0 = time.perf_counter()
            odn.diffnet_forward(sd, xs_, ts_, cs_, 'denoise_fn.')
            trials[n] = time.perf_counter() - t0
    best = min(trials, key=trials.get)
    torch.set_num_threads(best)
    with torch.no_grad():
        t0 = time.perf_counter()
        fs2_out = ofs2.fs2_forward(sd, inp)
        t1 = time.perf_counter()
        ret = omg.mel_gen(sd, inp, noise, fs2_out=fs2_out, n_steps=n_sample_steps)
        t2 = time.perf_counter()
    est = (t1 - t0) + (t2 - t1) / n_sample_steps * N_DIFF_STEPS
    # replay on the GPU
    d = {k: v.to(device) for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    g = model.fs2(d['txt_tokens'], d['mel2ph'], d['spk_embed'], None, None, None, None, skip_decoder=False, infer=True, **kw)
    x = noise[0][:, None].to(device).contiguous()
    x = model.sample(g['decoder_inp'].transpose(1, 2).contiguous(), x, noise=noise[1:].to(device), n_steps=n_sample_steps)
    torch.cuda.synchronize()
    parity = {
        'what': f'GPU vs fp32 oracle after FS2 + the first {n_sample_steps} sampler steps, same supplied noise, full bench shape',
        'x_max_abs': float((x.cpu() - ret['x']).abs().max()),
        'cond_max_abs': float((g['decoder_inp'].cpu() - fs2_out['decoder_inp']).abs().max()),
        'fs2_mel_max_abs': float((g['mel_out'].cpu() - fs2_out['mel_out']).abs().max()),
    }
    base = {
        'value': B * T_FRAMES / est, 'unit': 'mel-frames/s', 'cores': best, 'kind': 'port',
        'thread_trials_s_per_diffnet_call': {str(k): round(v, 3) for k, v in trials.items()}, 'host_cpus': os.cpu_count(),
        'sample': f'oracle (PyTorch-CPU restatement, fp32): full FS2-MIDI enc+dec ({t1 - t0:.2f} s) + {n_sample_steps} of '
                  f'{N_DIFF_STEPS} sampler steps ({t2 - t1:.2f} s) at B={B}, T={T_FRAMES}; steps extrapolated x{N_DIFF_STEPS}/{n_sample_steps}',
        'est_seconds_per_pass': est,
    }
    return base, parity


def main():
    global B_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--cpu-steps', type=int, default=24, help='sampler steps of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--dtype', choices=('f32', 'bf16'), default='f32',
                    help="arithmetic of the fused residual layers: f32 = BASELINE configs[1] (default, the parity configuration); "
                         "bf16 = configs[2] (bf16 MFMA operands, fp32 accumulate; run with --batch 64)")
    ap.add_argument('--batch', type=int, default=B_PER_GPU, help='utterances per GPU')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL and run the all-gather even with one rank (self-test)')
    args = ap.parse_args()

    from bisinger_amd import dist as bdist, synth
    rank, local_rank, world = bdist.env_world()
    if args.gpus != world:
        if args.gpus > 1:
            sys.exit(f'--gpus {args.gpus} needs one process per GPU: launch with '
                     f'python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py ...')
        world, rank, local_rank = 1, 0, 0
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X: the product path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        bdist.init_distributed('nccl')
    elif args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend='nccl', rank=0, world_size=1)

    torch.set_grad_enabled(False)
    model = build_model(device)
    model.denoise_fn.set_compute('bf16' if args.dtype == 'bf16' else 'fp32')
    B_PER_GPU = args.batch
    B_total = B_PER_GPU * world
    inp_np = synth.synth_inputs(B_total, T_TXT, T_FRAMES, seed=1)
    d = {k: torch.from_numpy(v).to(device) for k, v in inp_np.items()}     # inputs resident in HBM
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    rows = bdist.shard_rows(B_total, rank, world)

    def step(seed):
        out = model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True,
                    seed=seed, rows=rows if world > 1 else None, **kw)
        if args.force_dist and world == 1:
            full = torch.empty_like(out['mel_out'])
            dist.all_gather_into_tensor(full, out['mel_out'].contiguous())
            return full
        return bdist.all_gather_rows(out['mel_out'], B_total, world, rank)

    def fence():
        if world > 1 or args.force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        mel = step(1000 + i)
    fence()
    model.denoise_fn.profile(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        mel = step(i)
    fence()
    dt = time.perf_counter() - t0
    layer_ms, n_layer = model.denoise_fn.profile_read()
    model.denoise_fn.profile(False)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert mel.shape == (B_total, T_FRAMES, N_MEL) and bool(torch.isfinite(mel).all())

    # every rank flushes what native libraries buffered on stdout (RCCL prints its load path) before rank 0 prints the record,
    # so that the record is the last line of the job's merged stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if world > 1 or args.force_dist:
        dist.barrier()
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = B_total * T_FRAMES * args.steps / dt
        layers_per_launch = 20 if os.environ.get('BSG_PERSIST', '0') not in ('', '0') else 1
        avg_ms = layer_ms / max(n_layer, 1)
        # what one launch of the dominant kernel covers, from the counts: the sampler runs the batch as two concurrent launch
        # chains over half the rows each (bsg_ddpm_sample, BSG_DUAL), so a launch is B/2 x T frames and two are in flight
        frame_layers = args.steps * N_DIFF_STEPS * 20 * B_PER_GPU * T_FRAMES
        frames_per_launch = frame_layers / max(n_layer * layers_per_launch, 1)
        concurrent = max(1, round(B_PER_GPU * T_FRAMES / frames_per_launch))
        per_launch = FLOP_PER_FRAME_LAYER * frames_per_launch * layers_per_launch / (avg_ms * 1e-3) / 1e12 if n_layer else None
        achieved = per_launch * concurrent if n_layer else None
        bf16 = args.dtype == 'bf16'
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_bf16.json' if bf16 else 'traffic.json')
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if abs(tj.get('frames_per_launch', 16000) - frames_per_launch) < 1:
                traffic = tj.get('residual_layer_kernel_hbm_bytes_per_launch')
            if traffic is not None:
                traffic *= layers_per_launch
        if bf16:
            # 16x the fp32 MFMA rate moves the layer under the HBM roof: AI = 1,048,576 FLOP / 4 KB = 256 FLOP/B against a
            # ridge of 2,500 TFLOP/s / 8 TB/s = 312 FLOP/B (and the measured fabric traffic is 1.4x the algorithmic bytes)
            ach_gbs = HBM_BYTES_PER_FRAME_LAYER_BF16 * frames_per_launch * concurrent / (avg_ms * 1e-3) / 1e9 if n_layer else None
            roof = {'kernel': 'residual_layer_bf16_kernel (fused DiffNet residual block, bf16 MFMA operands)', 'bound': 'hbm',
                    'achieved': ach_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach_gbs / PEAK_HBM_GBS if ach_gbs else None,
                    'traffic': traffic, 'avg_launch_us': avg_ms * 1e3, 'launches_timed': n_layer,
                    'bytes_per_launch': HBM_BYTES_PER_FRAME_LAYER_BF16 * frames_per_launch,
                    'mfma_tflops': achieved, 'mfma_frac_of_bf16_peak': achieved / 2516.0 if achieved else None,
                    'frames_per_launch': frames_per_launch, 'concurrent_launches': concurrent,
                    'note': 'achieved = algorithmic bytes of one launch / its average duration (HIP events on its own stream) x '
                            'launches in flight'}
        else:
            roof = {'kernel': ('persistent_layers_kernel (20 fused DiffNet residual blocks per launch)' if layers_per_launch > 1
                               else 'residual_layer_kernel<false,true> (fused DiffNet residual block, Winograd GEMM1)'), 'bound': 'mfma',
                    'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': achieved / PEAK_F32_MFMA_TFLOPS if achieved else None, 'traffic': traffic,
                    'avg_launch_us': avg_ms * 1e3, 'launches_timed': n_layer,
                    'flop_per_launch': FLOP_PER_FRAME_LAYER * frames_per_launch * layers_per_launch,
                    'frames_per_launch': frames_per_launch, 'concurrent_launches': concurrent, 'achieved_per_launch': per_launch,
                    'executed_tflops': achieved * 0.75 if achieved and os.environ.get('BSG_WINO', '1') != '0' else achieved,
                    'note': 'achieved = algorithmic FLOPs (direct conv; the Winograd form executes 3/4 of them) of one launch / its '
                            'average duration (HIP events on its own stream) x launches in flight'}
        cfg_name = 'configs[2]' if bf16 else 'configs[1]'
        rec = {
            'metric': 'mel_frames_per_sec', 'value': value, 'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE.json {cfg_name}: B={B_PER_GPU}/GPU x T={T_FRAMES} x {N_MEL}-mel, FS2-MIDI enc+dec + '
                                   f'{N_DIFF_STEPS}-step DDPM sampler (20-layer DiffNet, 256 ch), '
                                   + ('bf16 MFMA operands / fp32 accumulate in the residual layers (FS2, projections, sampler fp32)'
                                      if bf16 else 'fp32') + ', formula weights',
                       'global_batch': B_total, 'frames_per_utt': T_FRAMES, 'diffusion_steps': N_DIFF_STEPS,
                       'parallelism': f'utterance-sharded x{world}, RCCL all-gather of mels' if world > 1 else 'single GPU'},
            'roofline': roof,
        }
        if world == 1 and args.cpu_steps > 0:
            base, parity = cpu_baseline_and_parity(model, inp_np, device, args.cpu_steps)
            rec['cpu_baseline'] = base
            rec['parity'] = parity
            rec['gpu_over_cpu'] = value / base['value']
        # native libraries (RCCL prints its load path) write through C stdio, which is block-buffered when stdout is a pipe:
        # flush it first so that the JSON record is the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(rec), flush=True)
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
