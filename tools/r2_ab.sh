cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-extras --steps 3 > gpurun_out/r2_ab_pk.json 2>/dev/null
DV_WINO_XFORM=scalar python bench.py --no-cpu-baseline --no-extras --steps 3 > gpurun_out/r2_ab_scalar.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --steps 3 > gpurun_out/r2_ab_pk2.json 2>/dev/null
DV_WINO_XFORM=scalar python bench.py --no-cpu-baseline --no-extras --steps 3 > gpurun_out/r2_ab_scalar2.json 2>/dev/null
for f in pk scalar pk2 scalar2; do python - <<PY
import json
d=json.loads(open("gpurun_out/r2_ab_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],2), d["kernels_ms_per_step"]["conv3d_k3s1_co32"], d["kernels_ms_per_step"]["conv3d_k3s1_co32_filter"], d["kernels_ms_per_step"]["conv3d_k3s1_co64"])
PY
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_loop -- python tools/trace_loop.py > /dev/null 2>&1
python tools/trace_loop.py --parse gpurun_out/trace_loop > gpurun_out/r2_loop_census.txt; cat gpurun_out/r2_loop_census.txt
rm -rf gpurun_out/trace_loop
python -m pytest tests/test_igev_model.py tests/test_gpu_dropin.py "tests/test_gpu_parity.py::test_ddim_sample_golden" "tests/test_gpu_parity.py::test_ddim_sample_vs_oracle_batch2" "tests/test_gpu_parity.py::test_ddim_loop_with_split_fp16_convs" "tests/test_gpu_parity.py::test_ddim_sample_other_step_counts" tests/test_gpu_fullsize.py::test_fullsize_oracle_5step -m gpu -q --timeout 1500 -p no:cacheprovider > gpurun_out/r2_t4.log 2>&1; grep -n "^E  \|^FAILED\|passed\|failed\|rerun\|shard" gpurun_out/r2_t4.log | cut -c1-300 | tail -30
