run() { env "$@" python bench.py --steps 6 --warmup 2 --no-secondary --cpu-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac_executed'],3))"; }
runb() { env "$@" python bench.py --steps 3 --warmup 1 --dtype bf16 --batch 64 --cpu-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 $*', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],2))"; }
for i in 1 2; do
run BSG_LIB=$GRAFT_REPO_ROOT/bisinger_amd/lib_abl/libold.so
run NEW=1
done
runb BSG_LIB=$GRAFT_REPO_ROOT/bisinger_amd/lib_abl/libold.so
runb NEW=1
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_melgen.py -m gpu -q 2>&1 | tail -2
