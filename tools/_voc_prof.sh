# kernel breakdown of one vocoder forward: PB=<batch> bash tools/_voc_prof.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/vocp; rm -rf $O; mkdir -p $O
PN=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o v -- python3 tools/prof_vocoder.py > $O/log 2>&1
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/vocp/v_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last forward: from the last split_transposed / conv1d<7,16> (conv_pre) on
idx=[i for i,r in enumerate(rows) if 'h2w_split_transposed' in r['Kernel_Name'] and int(r['Grid_Size_Y'])<=4*256]
# conv_pre's split is the first of 3 per forward
last=rows[idx[-3]:]
agg=collections.OrderedDict(); tot=0
for r in last:
    k=r['Kernel_Name'].replace('bsg::(anonymous namespace)::','').replace('_ZN3bsg12_GLOBAL__N_1','')[:64]
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp']); tot+=d
    a=agg.setdefault(k,[0,0]); a[0]+=1; a[1]+=d
for k,(c,d) in sorted(agg.items(), key=lambda x:-x[1][1]): print(f'{k:64s} n={c:3d} total={d/1e3:8.1f} us avg={d/c/1e3:7.1f}')
print('kernels of one forward: %.1f us, wall %.1f us, launches %d' % (tot/1e3, (int(last[-1]['End_Timestamp'])-int(last[0]['Start_Timestamp']))/1e3, len(last)))
PY
