# Round profiles (run on the GPU box through gpurun): rocprofv3 kernel stats and PMC passes over bench.py for the fp32
# (BASELINE configs[1]) and bf16 (configs[2]) configurations.  One counter set per run (gpurun refuses --pmc combined
# with traces; FETCH_SIZE and WRITE_SIZE do not fit one pass).  Summaries: tools/summarize_profiles.py -> profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROF_OUT:-prof}; mkdir -p $O
for cfg in f32 bf16; do
  if [ $cfg = f32 ]; then ARGS="--dtype f32"; else ARGS="--dtype bf16 --batch 64"; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$cfg/stats -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --cpu-steps 0 > $O/$cfg.stats.log 2>&1 || echo "stats $cfg failed"
  while read -r set; do
    [ -z "$set" ] && continue
    n=$(echo $set | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/$cfg/$n -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --cpu-steps 0 > $O/$cfg.$n.log 2>&1 || echo "pmc $cfg $n failed"
  done <<SETS
FETCH_SIZE
WRITE_SIZE
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
SETS
done
# keep only the small csv files (the merge-back limit is 64 MiB)
find $O -name '*_agent_info.csv' -delete
python3 $R/tools/summarize_profiles.py $O $O/summary
ls -la $O/summary
