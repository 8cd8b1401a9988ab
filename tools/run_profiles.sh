# Round profiles (run on the GPU box through gpurun): rocprofv3 kernel stats and PMC passes for
#   f32  : bench.py headline (BASELINE configs[1], B=16)            -> residual_layer_kernel, step_tail_kernel, FS2 kernels
#   bf16 : bench.py --dtype bf16 --batch 64 (configs[2])             -> residual_layer_bf16_kernel
#   voc  : tools/prof_vocoder.py (HiFi-GAN alone, B=16, T=1000)      -> resblock_pair_mfma_kernel, resblock_pair_kernel, upsample_kernel
#   voc1 : the same at B=1 (configs[4]'s vocoder)                     -> profiles/traffic_voc.json (HBM bytes of one forward, bench.py e2e roofline)
#   rank : tools/prof_rank.py (configs[3] as one of its 8 ranks), b1 : the single-utterance pass — kernel stats only
# One counter set per run (gpurun refuses --pmc combined with traces; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Summaries: tools/summarize_profiles.py -> profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROF_OUT:-prof}; mkdir -p $O
for cfg in ${CFGS:-f32 bf16 voc voc1 rank b1}; do
  export PB=16 PW=1 PN=5
  case $cfg in
    f32) CMD="python3 $R/bench.py --dtype f32 --no-secondary --cpu-steps 0";;
    bf16) CMD="python3 $R/bench.py --dtype bf16 --batch 64 --cpu-steps 0";;
    voc) CMD="python3 $R/tools/prof_vocoder.py";;
    voc1) CMD="python3 $R/tools/prof_vocoder.py"; export PB=1;;
    rank) CMD="python3 $R/tools/prof_rank.py"; export PB=64 PW=8 PN=3;;
    b1) CMD="python3 $R/tools/prof_rank.py"; export PB=1 PW=1 PN=3;;
  esac
  case $cfg in f32|bf16) S1="--steps 2 --warmup 1"; S2="--steps 1 --warmup 0";; *) S1=""; S2="";; esac
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$cfg/stats -- $CMD $S1 > $O/$cfg.stats.log 2>&1 || echo "stats $cfg failed"
  case $cfg in rank|b1) continue;; esac      # kernel stats only
  while read -r set; do
    [ -z "$set" ] && continue
    n=$(echo $set | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/$cfg/$n -- $CMD $S2 > $O/$cfg.$n.log 2>&1 || echo "pmc $cfg $n failed"
  done <<SETS
FETCH_SIZE
WRITE_SIZE
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES
SETS
done
# keep only the small csv files (the merge-back limit is 64 MiB)
find $O -name '*_agent_info.csv' -delete
find $O -name '*kernel_trace.csv' -delete
python3 $R/tools/summarize_profiles.py $O $O/summary
ls -la $O/summary
