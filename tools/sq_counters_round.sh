# SQ / GRBM counters of the default bench workload at HEAD, two --pmc passes (no trace domains besides kernel-trace):
#   bash tools/sq_counters_round.sh <tag>     (on the GPU box; outputs under gpurun_out/sq_<tag>/, summary json beside them)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r03}
O=gpurun_out/sq_$T
mkdir -p $O
# DV_SQ_ARGS: another python command line to profile (e.g. "tools/bench_flavours.py --igev" for the 2-D kernels of config 5)
ARGS=${DV_SQ_ARGS:-"bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timer"}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/p1 -- python $ARGS > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/p2 -- python $ARGS > $O/p2.log 2>&1
python tools/sq_summary.py $O $T ${2:-}
rm -rf $O/p1 $O/p2                       # raw counter dumps
