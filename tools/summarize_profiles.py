#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/run_profiles.sh into the small files kept under profiles/.

    summarize_profiles.py <rocprof output dir> <summary dir>

Per configuration (f32 = BASELINE configs[1], bf16 = configs[2]): the kernel-stats CSV as rocprofv3 wrote it, and a
PMC summary (per-kernel means over all launches) with the HBM-side traffic of the dominant kernel per launch:
  traffic = 2 * FETCH_SIZE + WRITE_SIZE  (KB -> bytes): MI355X_MICROARCH.md "HBM" - on gfx950 FETCH_SIZE tallies the L2's 128-B
  fabric read requests at 64 B, so it is doubled; WRITE_SIZE is exact.  Cross-check kept beside it from the request counters:
  128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B  +  64*WRREQ_64B + 32*(WRREQ - WRREQ_64B)  (agrees within 0.5 % here).
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
KEYS = {'residual_layer_bf16': 'residual_layer_bf16_kernel', 'residual_layer_kernel': 'residual_layer_kernel',
        'step_tail': 'step_tail_kernel', 'gemm_f32': 'gemm_f32_kernel'}
for cfg in ('f32', 'bf16'):
    st = glob.glob(f'{src}/{cfg}/stats/*/*_kernel_stats.csv')
    if st:
        rows = list(csv.DictReader(open(st[0])))
        with open(f'{dst}/bench_{cfg}_kernel_stats.csv', 'w') as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    out = {}
    for d in sorted(glob.glob(f'{src}/{cfg}/*/')):
        fs = glob.glob(d + '*/*_counter_collection.csv')
        if not fs:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fs[0])):
            k = next((v for p, v in KEYS.items() if p in r['Kernel_Name']), None)
            if k:
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k in agg:
            for c, v in agg[k].items():
                out.setdefault(k, {})[c] = {'n': len(v), 'mean': sum(v) / len(v)}
    if not out:
        continue
    dom = 'residual_layer_bf16_kernel' if cfg == 'bf16' else 'residual_layer_kernel'
    r = out.get(dom, {})
    g = lambda c: r.get(c, {}).get('mean')
    summ = {'per_kernel_counter_means': out}
    if g('TCC_EA0_RDREQ_sum') is not None and g('TCC_EA0_WRREQ_sum') is not None:
        rd = 128 * g('TCC_EA0_RDREQ_128B_sum') + 64 * g('TCC_EA0_RDREQ_64B_sum') + 32 * g('TCC_EA0_RDREQ_32B_sum')
        wr = 64 * g('TCC_EA0_WRREQ_64B_sum') + 32 * (g('TCC_EA0_WRREQ_sum') - g('TCC_EA0_WRREQ_64B_sum'))
        frames = 64000 if cfg == 'bf16' else 16000
        n_launch = r.get('FETCH_SIZE', r.get('TCC_EA0_RDREQ_sum', {})).get('n')
        if n_launch:   # one pass = 100 steps x 20 layers; more launches than that = the batch runs as concurrent half-batch chains
            frames = frames * 2000 // n_launch
        t = {'kernel': dom, 'frames_per_launch': frames, 'fabric_read_bytes': rd, 'fabric_write_bytes': wr,
             'algorithmic_bytes_per_launch': 6 * 256 * 4 * frames,
             'FETCH_SIZE_KB_raw': g('FETCH_SIZE'), 'WRITE_SIZE_KB': g('WRITE_SIZE'),
             'l2_hit_requests': g('TCC_HIT_sum'), 'l2_miss_requests': g('TCC_MISS_sum')}
        if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None:
            t['fetch_bytes_corrected'] = 2 * 1024 * g('FETCH_SIZE')
            t['write_bytes'] = 1024 * g('WRITE_SIZE')
            t['residual_layer_kernel_hbm_bytes_per_launch'] = t['fetch_bytes_corrected'] + t['write_bytes']
        else:
            t['residual_layer_kernel_hbm_bytes_per_launch'] = rd + wr
        if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('GRBM_GUI_ACTIVE'):
            t['mfma_busy_cycles_per_simd'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024
            t['kernel_cycles_grbm_gui_active_div8'] = g('GRBM_GUI_ACTIVE') / 8
            t['mfma_pipe_utilisation'] = t['mfma_busy_cycles_per_simd'] / t['kernel_cycles_grbm_gui_active_div8']
        summ['traffic'] = t
    json.dump(summ, open(f'{dst}/bench_{cfg}_pmc_summary.json', 'w'), indent=1)
    print(cfg, json.dumps(summ.get('traffic', {}), indent=1))
