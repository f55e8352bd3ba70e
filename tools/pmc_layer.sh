# usage: bash tools/pmc_layer.sh <layer> <tag>   (runs two PMC passes of tools/bench_conv.py <layer>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=$1; T=$2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_${T}_1 -- python tools/bench_conv.py $L > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${T}_2 -- python tools/bench_conv.py $L > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_${T}_3 -- python tools/bench_conv.py $L > /dev/null 2>&1
find gpurun_out/pmc_${T}_1 gpurun_out/pmc_${T}_2 gpurun_out/pmc_${T}_3 -name "*counter_collection.csv"
