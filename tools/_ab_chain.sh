# same-box A/B of the ResBlock chain launches on the vocoder alone after its parity tests; kernel stats of the builds of the conv loop
tools/bin/split_asm_check
timeout -k 10 900 python -m pytest tests/test_gpu_hifigan.py -x -q -m gpu > gpurun_out/chain_tests.log 2>&1; tail -4 gpurun_out/chain_tests.log
R=$PWD
for rep in 1 2; do
  for spec in "pf2:" "pf0:BSG_LIB=$R/bisinger_amd/lib/alt/libbisinger_pf0.so" "pf2_pairs:BSG_HG_CHAIN=0"; do
    label=${spec%%:*}; envs=${spec#*:}
    echo "$label rep$rep: $(env $envs timeout -k 10 200 python tools/bench_vocoder.py 2>/dev/null | tail -1)"
  done
done
cd /tmp && export TMPDIR=/tmp
for spec in "pf2:" "pf0:BSG_LIB=$R/bisinger_amd/lib/alt/libbisinger_pf0.so"; do
  label=${spec%%:*}; envs=${spec#*:}
  for kv in $envs; do export $kv; done
  export PB=16
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chainprof_$label -- python3 $R/tools/prof_vocoder.py > $R/gpurun_out/chainprof_$label.log 2>&1
  unset BSG_LIB BSG_HG_CHAIN
  f=$(find $R/gpurun_out/chainprof_$label -name "*kernel_stats.csv" | head -1)
  echo "---- $label"
  python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in rows[:24]:
    if 'h16' in r['Name'] or 'pair_kernel' in r['Name']: print(r['Name'][33:80], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
  find $R/gpurun_out/chainprof_$label -name "*kernel_trace.csv" -delete; find $R/gpurun_out/chainprof_$label -name "*agent_info.csv" -delete
done
