# Everything the round's record under profiles/ is made of, in ONE call on the GPU box (about 15 minutes):
#   bash tools/round_profiles.sh r06      -> gpurun_out/<tag>_*  (copy the summaries to profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r06}
O=gpurun_out
mkdir -p $O
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err;                                   tail -c 200 $O/${T}_bench.err
python bench.py --workload kitti12 --no-extras > $O/${T}_bench_kitti12.json 2>/dev/null
python bench.py --workload kitti15 --steps 2 --warmup 1 --no-extras > $O/${T}_bench_kitti15.json 2>/dev/null
bash tools/profile_round.sh $T > $O/${T}_prof.log 2>&1;                                       tail -2 $O/${T}_prof.log
bash tools/sq_counters_round.sh $T > $O/${T}_sq.log 2>&1;                                     tail -1 $O/${T}_sq.log
DV_SQ_ARGS="tools/bench_flavours.py --pcw" bash tools/sq_counters_round.sh ${T}_config4 > $O/${T}_sq4.log 2>&1
DV_SQ_ARGS="tools/bench_flavours.py --igev-model --ddim-steps 2" bash tools/sq_counters_round.sh ${T}_config5 > $O/${T}_sq5.log 2>&1
bash tools/profile_flavour.sh ${T}_config4 --pcw > $O/${T}_prof4.log 2>&1
bash tools/profile_flavour.sh ${T}_config5 --igev-model --ddim-steps 2 > $O/${T}_prof5.log 2>&1
python tools/e2e_stages.py > $O/${T}_e2e.log 2>&1 && cp $O/e2e_stages.json $O/${T}_e2e_stages.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_prof -- python tools/e2e_stages.py --once > /dev/null 2>&1
cp "$(find $O/e2e_prof -name '*kernel_stats.csv' | head -1)" $O/${T}_e2e_kernel_stats.csv; rm -rf $O/e2e_prof
ls -la $O | grep ${T}_ | head -40
