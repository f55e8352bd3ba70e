# Round profile: rocprofv3 kernel stats + HBM traffic counters of the default bench workload.
#   bash tools/profile_round.sh <tag> [git-head]      (on the GPU box; outputs under gpurun_out/prof_<tag>/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r01}
O=gpurun_out/prof_$T
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timer > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timer > /dev/null 2>&1
python tools/make_pmc_traffic.py $O $T ${2:-}
rm -rf $O/stats $O/pmc_fetch $O/pmc_write   # raw traces: tens of MB, the summaries are what is kept
