# kernel stats of one pass with the pre-split attention: tools/_fl_ab.sh  (PB / PW as tools/prof_rank.py)
set -e
mkdir -p gpurun_out/fl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PN=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fl/p1 -o p1 -- python3 tools/prof_rank.py > gpurun_out/fl/p1.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fl/p1/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
for r in rows:
    if 'flash' in r['Name'] or 'qkv_split' in r['Name'] or 'layernorm' in r['Name']: print('>>',r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
