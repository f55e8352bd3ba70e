cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/bench_conv.py dc64 dc64r dc128 dc128r
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_dc1 -- python tools/bench_conv.py dc64r > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_dc2 -- python tools/bench_conv.py dc64r > /dev/null 2>&1
find gpurun_out/pmc_dc1 gpurun_out/pmc_dc2 -name "*counter_collection.csv" | head
