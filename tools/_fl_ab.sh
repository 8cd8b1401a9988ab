# kernel stats of one pass with the pre-split attention: tools/_fl_ab.sh  (PB / PW as tools/prof_rank.py)
set -e
mkdir -p gpurun_out/fl
if [ -n "$FL_TESTS" ]; then
timeout -k 10 500 python -m pytest tests/test_gpu_fs2.py tests/test_gpu_melgen.py tests/test_gpu_f4.py -x -q > gpurun_out/fl/tests.log 2>&1 || { tail -30 gpurun_out/fl/tests.log; exit 1; }
tail -3 gpurun_out/fl/tests.log
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PN=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fl/p1 -o p1 -- python3 tools/prof_rank.py > gpurun_out/fl/p1.log 2>&1
python - <<'PY'
import csv,glob,collections,os
rows=list(csv.DictReader(open('gpurun_out/fl/p1/p1_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
starts=[i for i,n in enumerate(names) if 'embed_tokens' in n]
last=rows[starts[-1]:]
agg=collections.OrderedDict()
for r in last:
    k=(r['Kernel_Name'].replace('bsg::(anonymous namespace)::','').replace('_ZN3bsg12_GLOBAL__N_1','')[:60], r['Grid_Size_X'], r['Grid_Size_Y'])
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    a=agg.setdefault(k,[0,0]); a[0]+=1; a[1]+=d
tot=0
flt=os.environ.get('FL_ONLY')
for k,(c,d) in agg.items():
    if 'residual' in k[0] or 'step_tail' in k[0]: continue
    tot+=d
    if flt and not any(f in k[0] for f in flt.split(',')): continue
    print(f'{k[0]:60s} g=({k[1]},{k[2]}) n={c:3d} total={d/1e3:8.1f} us avg={d/c/1e3:7.1f}')
print('non-sampler kernels of the pass: %.1f us; wall %.1f us' % (tot/1e3, (int(last[-1]['End_Timestamp'])-int(last[0]['Start_Timestamp']))/1e3))
PY
