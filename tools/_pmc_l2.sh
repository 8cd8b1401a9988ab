# L2 (TCC) counters of the dominant launches: tools/_pmc_l2.sh   (one counter set per run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/l2pmc; rm -rf $O; mkdir -p $O
for cfg in f32 bf16; do
  case $cfg in
    f32) CMD="python3 bench.py --dtype f32 --no-secondary --cpu-steps 0 --steps 1 --warmup 0";;
    bf16) CMD="python3 bench.py --dtype bf16 --batch 64 --cpu-steps 0 --steps 1 --warmup 0";;
  esac
  i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/$cfg/s$i -o s$i -- $CMD > $O/$cfg.s$i.log 2>&1 || echo "pmc $cfg $set failed"
  done <<SETS
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum
SETS
done
python3 - <<'PY'
import csv,glob,collections
for cfg in ('f32','bf16'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f'gpurun_out/l2pmc/{cfg}/s*/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'residual_stack' not in k: continue
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items():
        print(cfg, k, {a:round(sum(b)/len(b)) for a,b in c.items()}, 'launches', {a:len(b) for a,b in c.items()})
PY
