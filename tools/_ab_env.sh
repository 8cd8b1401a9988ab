# same-box A/B of ENVIRONMENT switches on the tree's build: tools/_ab_env.sh "label1:VAR=val VAR2=val" "label2:..."   (extra bench.py
# arguments via AB_ARGS, timed passes via AB_STEPS); two interleaved repetitions
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    out=$(env $envs timeout -k 10 200 python bench.py --no-secondary --cpu-steps 0 --steps ${AB_STEPS:-5} $AB_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],2), round(r['avg_launch_us'],2), r.get('sustained_mhz'), r['kernel'][:28], j['handoff_timeouts'])")
    echo "$label rep$rep: $out"
  done
done
