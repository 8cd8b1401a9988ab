# PMC passes over one pass, reported for kernels matching $FL_ONLY: tools/_fl_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/flpmc; rm -rf $O; mkdir -p $O
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  PN=1 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/s$i -o s$i -- python3 ${PROG:-tools/prof_rank.py} > $O/s$i.log 2>&1 || echo "pmc $set failed"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU
SETS
python3 - <<'PY'
import csv,glob,os,collections
flt=os.environ.get('FL_ONLY','flash').split(',')
for f in sorted(glob.glob('gpurun_out/flpmc/s*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if not any(x in k for x in flt): continue
        key=(k[:50], r['Grid_Size'])
        agg[key][r['Counter_Name']]+=float(r['Counter_Value'])
    for key,c in agg.items():
        print(key, {a:round(b) for a,b in c.items()})
PY
