cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pcw -- python tools/bench_flavours.py --pcw > gpurun_out/prof_pcw.json 2>/dev/null
f=$(find gpurun_out/prof_pcw -name "*kernel_stats.csv" | head -1)
head -40 $f | cut -c1-220
