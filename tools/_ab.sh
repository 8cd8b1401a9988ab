# same-box A/B of library builds: tools/_ab.sh <label=path-or-empty> ...   (empty path = the tree's build); extra env via AB_ENV_<label>,
# extra bench.py arguments via AB_ARGS (e.g. "--dtype bf16 --batch 64"), timed passes via AB_STEPS
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; path=${spec#*=}
    envs=$(eval echo \$AB_ENV_$label)
    out=$(env BSG_LIB=$path $envs timeout -k 10 200 python bench.py --no-secondary --cpu-steps 0 --steps ${AB_STEPS:-5} $AB_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],2), round(j['roofline']['avg_launch_us'],2), j['roofline']['kernel'][:28])")
    echo "$label rep$rep: $out"
  done
done
