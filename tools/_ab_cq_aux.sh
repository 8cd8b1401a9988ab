# same-box A/B of the cache-policy bits on the conditioner term's loads in the pair form (B = 8): tree = nt + sc0 (3)
for rep in 1 2; do
  echo "nt+sc0 (tree): $(timeout -k 10 200 python tools/bench_small.py 8 5 6 2>/dev/null | tail -1)"
  for aux in 2 1 18 19; do
    echo "aux $aux: $(BSG_LIB=$PWD/bisinger_amd/lib/alt/lib_pq$aux.so timeout -k 10 200 python tools/bench_small.py 8 5 6 2>/dev/null | tail -1)"
  done
  echo "aux 0 (prev): $(BSG_LIB=$PWD/tools/bin/lib_prev.so timeout -k 10 200 python tools/bench_small.py 8 5 6 2>/dev/null | tail -1)"
done
