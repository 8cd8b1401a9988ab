#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/run_profiles.sh into the small files kept under profiles/.

    summarize_profiles.py <rocprof output dir> <summary dir>

Per configuration (f32 = BASELINE configs[1], bf16 = configs[2], voc = HiFi-GAN alone at B=16): the kernel-stats CSV as
rocprofv3 wrote it, and a PMC summary: per kernel the mean over all launches of every counter collected, plus
  hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE  (KB -> bytes): MI355X_MICROARCH.md "HBM" - on gfx950 FETCH_SIZE tallies the
      L2's 128-B fabric read requests at 64 B, so it is doubled; WRITE_SIZE is exact;
  mfma_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)   (both in shader cycles);
  avg_us from the kernel-stats pass (PMC passes serialise kernels: a SOLO launch, not two chains in flight).
The `traffic` block restates this for the configuration's dominant kernel, with the algorithmic bytes beside it.
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:      # source digest of the library the passes ran on (bisinger_amd/build.py): bench.py compares it with the build it times
    BUILD = open(os.path.join(ROOT, 'bisinger_amd', 'lib', 'build.sha256')).read().strip()
except OSError:
    BUILD = None


def short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '').replace('bsg::', '')
    return n.split('(')[0].strip()


for cfg in ('f32', 'bf16', 'voc', 'voc1', 'rank', 'b1'):
    st = glob.glob(f'{src}/{cfg}/stats/*/*_kernel_stats.csv')
    avg_us = {}
    if st:
        rows = list(csv.DictReader(open(st[0])))
        with open(f'{dst}/bench_{cfg}_kernel_stats.csv', 'w') as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        for r in rows:
            avg_us[short(r['Name'])] = (float(r['AverageNs']) / 1e3, int(r['Calls']), float(r['Percentage']))
    out = collections.defaultdict(dict)
    for d in sorted(glob.glob(f'{src}/{cfg}/*/')):
        fs = glob.glob(d + '*/*_counter_collection.csv')
        if not fs:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fs[0])):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        for k in agg:
            for c, v in agg[k].items():
                out[k][c] = {'n': len(v), 'mean': sum(v) / len(v)}
    if not out and not avg_us:
        continue
    per_kernel = {}
    for k in sorted(set(out) | set(avg_us), key=lambda k: -avg_us.get(k, (0, 0, 0))[2]):
        if avg_us.get(k, (0, 0, 0))[2] < 0.3 and k not in out:
            continue
        g = lambda c: out.get(k, {}).get(c, {}).get('mean')
        e = {}
        if k in avg_us:
            e['avg_us'], e['calls'], e['pct_of_device_time'] = round(avg_us[k][0], 2), avg_us[k][1], avg_us[k][2]
        if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None:
            e['FETCH_SIZE_KB_raw'], e['WRITE_SIZE_KB'] = g('FETCH_SIZE'), g('WRITE_SIZE')
            e['hbm_bytes_per_launch'] = 2 * 1024 * g('FETCH_SIZE') + 1024 * g('WRITE_SIZE')
            if 'avg_us' in e:
                e['hbm_GBps_at_avg_duration'] = round(e['hbm_bytes_per_launch'] / (e['avg_us'] * 1e-6) / 1e9, 1)
        if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('GRBM_GUI_ACTIVE'):
            e['mfma_pipe_utilisation'] = round(g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (g('GRBM_GUI_ACTIVE') / 8), 4)
        if e:
            per_kernel[k] = e
    summ = {'configuration': cfg, 'per_kernel': per_kernel}
    # dominant kernel of the configuration: a stack launch (all 20 layers of a launch group in one kernel) or a per-layer launch.
    # Figures are normalised to ONE LAYER OVER THE WHOLE BATCH for the stack forms (launch bytes x launch groups / 20), which is the unit
    # bench.py's roofline uses for them; for per-layer launches they are per launch.
    batch_frames = 64000 if cfg == 'bf16' else 16000
    cands = {'f32': [('residual_stack_h2_kernel<true>', 'stack_h2'), ('residual_stack_h2_kernel<false>', 'stack_h2'),
                     ('residual_stack_f43_kernel<1>', 'stack_f43'), ('residual_stack_f43_kernel<0>', 'stack_f43'),
                     ('residual_layer_kernel<false, true>', 'layer')],
             'bf16': [('residual_stack_bf16_kernel<true>', 'stack_bf16'), ('residual_stack_bf16_kernel<false>', 'stack_bf16'),
                      ('residual_layer_bf16_kernel<false>', 'bf16')]}.get(cfg, [])
    cands = ([(k, 'stack_h2q') for k in per_kernel if k.startswith('residual_stack_q_kernel')] +          # 16-row matrix tiles (round 5's default)
             [(k, 'stack_h2') for k in per_kernel if k.startswith('residual_stack_h2_kernel')] + cands)   # any instantiation (<FAIR, TAIL>)
    for dom, path in cands:
        if dom not in per_kernel or 'hbm_bytes_per_launch' not in per_kernel[dom]:
            continue
        n = out[dom]['FETCH_SIZE']['n']
        by = per_kernel[dom]['hbm_bytes_per_launch']
        if path.startswith('stack'):
            groups = max(1, n // 100)                     # the PMC passes run one 100-step pass: launches per DiffNet evaluation
            frames, by = batch_frames, by * groups / 20   # one layer over the batch
            # what the on-chip form must move per frame and layer: conditioner term (fp32 2 KB / bf16 1 KB) + running skip sum r+w
            # (fp32 2 KB; bf16 form: in registers) + x in / skip out once per 20 layers + edges through L2
            if path in ('stack_h2', 'stack_h2q'):      # conditioner term fp32 2 KB + x in / skip out once per 20 layers + two fp16 planes of the edges through L2
                alg_form = (2048 + 2048 / 20 + 2 * 16384 / 64) * frames
            else:
                alg_form = (2048 + 2048 + 1024 / 20) * frames if cfg == 'f32' else (1024 + 2048 / 20 + 2 * 8192 / 64) * frames
        else:
            frames = batch_frames * 2000 // max(n, 1)     # one pass = 100 steps x 20 layers; more launches = half-batch chains
            alg_form = (4 if cfg == 'bf16' else 6) * 256 * 4 * frames
        alg = (4 if cfg == 'bf16' else 6) * 256 * 4 * frames   # one fused kernel per layer
        summ['traffic'] = {'kernel': dom, 'path': path, 'build_sha256': BUILD, 'frames_per_launch': frames, 'algorithmic_bytes_per_launch': alg,
                           'algorithmic_bytes_of_this_form': alg_form,
                           'residual_layer_kernel_hbm_bytes_per_launch': by,
                           'traffic_over_algorithmic': round(by / alg, 3),
                           'mfma_pipe_utilisation': per_kernel[dom].get('mfma_pipe_utilisation'),
                           'unit': 'one layer over the whole batch (launch bytes x launch groups / 20)' if path.startswith('stack') else 'one launch',
                           'condition': 'solo launch (PMC passes serialise kernels)'}
        break
    if cfg == 'voc1':
        # HBM-side bytes of ONE vocoder forward at B = 1, T = 1000: sum over the kernels of (mean bytes per launch x launches) / forwards
        # (tools/prof_vocoder.py runs PN = 5 forwards per pass) -> profiles/traffic_voc.json, read by bench.py's configs[4] roofline
        # forwards = launches of a once-per-forward kernel (conv_post: the warm-up forwards of prof_vocoder.py are profiled too, and PN may change)
        once = [k for k in out if 'conv_post_kernel' in k]
        fw = min((out[k]['FETCH_SIZE']['n'] for k in once), default=0) or int(os.environ.get('PN', 5))
        tot = sum(e['hbm_bytes_per_launch'] * out[k]['FETCH_SIZE']['n'] for k, e in per_kernel.items() if 'hbm_bytes_per_launch' in e and k in out)
        # attribution: bytes of one forward per kernel, and per generator stage.  The stage of a ResBlock kernel is its
        # channel count (template argument C: 64 -> stage 0 at 8 000 positions, 32 -> stage 1 at 64 000, 16 -> stage 2 at 128 000, 8 -> stage 3
        # at 256 000); the up-sampling GEMM / transposed convolutions and conv_pre / conv_post are listed by name
        import re
        by_kernel, by_stage = {}, collections.defaultdict(float)
        for k, e in per_kernel.items():
            if 'hbm_bytes_per_launch' not in e or k not in out:
                continue
            b = e['hbm_bytes_per_launch'] * out[k]['FETCH_SIZE']['n'] / fw
            by_kernel[k] = {'launches_per_forward': out[k]['FETCH_SIZE']['n'] / fw, 'hbm_bytes_per_forward': round(b), 'avg_us': e.get('avg_us')}
            m = re.match(r'resblock_\w+_kernel<(\d+), *(\d+)', k)
            if m:
                by_stage[f'ResBlocks C={m.group(2)}'] += b
            elif 'upsample' in k or 'gemm_h2w' in k or 'h2w_split' in k or 'conv1d_kernel' in k:
                by_stage['up-sampling (transposed convolutions as GEMM / polyphase kernels, incl. conv_pre on the GEMM)'] += b
            else:
                by_stage['other (conv_post, padding, NSF ...)'] += b
        json.dump({'B': 1, 'T': 1000, 'forwards': fw, 'hbm_bytes_per_forward': tot / fw, 'build_sha256': BUILD,
                   'algorithmic_bytes_per_forward': 80 * 1000 * 4 + 256000 * 4,
                   'by_stage_bytes_per_forward': {k: round(v) for k, v in sorted(by_stage.items(), key=lambda kv: -kv[1])},
                   'by_kernel': dict(sorted(by_kernel.items(), key=lambda kv: -kv[1]['hbm_bytes_per_forward'])),
                   'source': 'bench_voc1_pmc_summary.json of the same passes (tools/run_profiles.sh): 2 x FETCH_SIZE + WRITE_SIZE over every launch'},
                  open(f'{dst}/traffic_voc.json', 'w'), indent=1)
    json.dump(summ, open(f'{dst}/bench_{cfg}_pmc_summary.json', 'w'), indent=1)
    if 'traffic' in summ and cfg in ('f32', 'bf16'):   # what bench.py reads (profiles/traffic.json, profiles/traffic_bf16.json)
        json.dump(dict(summ['traffic'], source=f'bench_{cfg}_pmc_summary.json of the same passes (tools/run_profiles.sh)'),
                  open(f'{dst}/traffic' + ('_bf16' if cfg == 'bf16' else '') + '.json', 'w'), indent=1)
    print(cfg, json.dumps(summ.get('traffic', {k: v for k, v in list(per_kernel.items())[:3]}), indent=1)[:1500])
