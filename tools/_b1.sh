python bench.py --steps 3 --warmup 1 --dtype bf16 --batch 64 --cpu-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],3))"
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -m gpu -q -k "bf16" 2>&1 | tail -3
