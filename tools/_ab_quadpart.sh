set -e
timeout -k 10 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_h2.py -x -q -m gpu > gpurun_out/quadpart_tests.log 2>&1; tail -2 gpurun_out/quadpart_tests.log
for rep in 1 2 3; do
  for q in 1 0; do
    echo "quad=$q rep$rep: $(BSG_COND_QUAD=$q timeout -k 10 200 python tools/bench_small.py 1 4 8 2>/dev/null | tail -1)"
  done
done
