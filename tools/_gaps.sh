# gaps between consecutive kernels of the rank pass (configs[3] as one of its 8 ranks): rocprofv3 kernel trace -> end(k) .. start(k+1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gaps; rm -rf $O
PB=${PB:-64} PW=${PW:-8} PN=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/prof_rank.py > $O.log 2>&1
python3 - <<'PY'
import csv,glob,os,collections
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/gaps/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
gaps=collections.defaultdict(list); dur=collections.defaultdict(list)
for a,b in zip(rows,rows[1:]):
    ka,kb=a['Kernel_Name'],b['Kernel_Name']
    key=(ka.split('(')[0][-40:], kb.split('(')[0][-40:])
    if 'residual_part' in ka or 'step_tail' in ka:
        gaps[key].append(int(b['Start_Timestamp'])-int(a['End_Timestamp']))
        dur[ka.split('(')[0][-40:]].append(int(a['End_Timestamp'])-int(a['Start_Timestamp']))
for k,v in gaps.items():
    if len(v)>20: print(k, 'n',len(v),'gap mean us', round(sum(v)/len(v)/1e3,2), 'median', sorted(v)[len(v)//2]/1e3)
for k,v in dur.items(): print(k,'dur mean us', round(sum(v)/len(v)/1e3,2))
PY
find $O -name "*.csv" -delete
