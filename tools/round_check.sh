# Round-end check on the GPU box: the whole -m gpu suite, the default bench line, the round profile.
#   bash tools/round_check.sh <tag>     (outputs under gpurun_out/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r02}
python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider > gpurun_out/${T}_gputests.log 2>&1
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/${T}_gputests.log | cut -c1-300 | tail -20
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
bash tools/profile_round.sh $T > gpurun_out/${T}_prof.log 2>&1
tail -2 gpurun_out/${T}_prof.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$T -- python tools/trace_loop.py > /dev/null 2>&1
python tools/trace_loop.py --parse gpurun_out/trace_$T > gpurun_out/${T}_loop_census.txt 2>&1
rm -rf gpurun_out/trace_$T
head -3 gpurun_out/${T}_loop_census.txt
