# rocprofv3 kernel stats of the configs[3] rank shape (tools/prof_rank.py) and of the single-utterance pass -> gpurun_out/<PROF_OUT>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROF_OUT:-prof_rank}; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rank -- python3 $R/tools/prof_rank.py > $O/rank.log 2>&1 || echo "rank stats failed"
PB=1 PW=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -- python3 $R/tools/prof_rank.py > $O/b1.log 2>&1 || echo "b1 stats failed"
find $O -name '*_agent_info.csv' -delete
find $O -name '*kernel_trace.csv' -delete
for c in rank b1; do cp $O/$c/*/*_kernel_stats.csv $O/${c}_kernel_stats.csv; done
tail -n 2 $O/rank.log; tail -n 2 $O/b1.log
