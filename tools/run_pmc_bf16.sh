# PMC passes over the bf16 residual-layer kernel at B=64 (one counter set per rocprofv3 run, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PMC_OUT:-bf16pmc}; mkdir -p $O
export PB=${PB:-64} PD=${PD:-bf16} PN=10
while read -r set; do
  [ -z "$set" ] && continue
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/$n -- python3 $R/tools/prof_layer.py > $O/$n.log 2>&1 || echo "pass $n failed"
done <<SETS
${PMC_SETS:-TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE WRITE_SIZE}
SETS
python3 $R/tools/pmc_sum.py $O ${PMC_KERNEL:-residual_layer_bf16}
