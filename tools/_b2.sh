run() { env "$@" python bench.py --steps 4 --warmup 2 --no-secondary --cpu-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac_executed'],3), d['handoff_timeouts'])"; }
run A=1
run BSG_DUAL=0
