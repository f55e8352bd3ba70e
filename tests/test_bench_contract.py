"""The committed bench line of the round (profiles/r*_bench.json, copied from the GPU box) keeps the driver's
contract: one JSON object with the metric of BASELINE.json, whole-job throughput, a hardware roofline fraction that
is a fraction (achieved / peak < 1, priced with the flops the kernel issues), the CPU baseline timed on the same box
without extrapolation, and the parity leg's verdict."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def latest():
    files = sorted((ROOT / "profiles").glob("r*_bench.json"))
    assert files, "no committed bench line"
    return json.loads(files[-1].read_text())


def test_contract_keys_and_roofline():
    d = latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0        # a hardware fraction
    assert abs(r["algorithmic_tflops"] / r["multiply_reduction"] - r["achieved"]) < 1e-6
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    for k in d.get("roofline_kernels", []):
        assert 0.0 < k["frac"] < 1.0 and k["avg_ms"] > 0
    assert len(d.get("roofline_kernels", [])) >= 3


def test_cpu_baseline_and_parity_leg():
    d = latest()
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["extrapolated"] is False and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    p = d["parity_vs_oracle"]
    if "within_raw_bars" in p:      # round 3 on: the raw contract figure and the builder's spread-scaled one are separate flags
        assert isinstance(p["within_raw_bars"], bool) and p["within_spread_scaled_bars"] is True
        assert p["within_raw_bars_where_unc_lt_3"] is True and p["final_within_raw_bars"] is True
        assert all("share_unc_lt_3" in s and "frac_gt_1e-3" in s for s in p["teacher_forced"])
        assert 0.0 < d["mfma_f32_issued_frac_whole_path"] < 1.0
        assert "mfma_f32_roofline_frac_whole_path" not in d                    # (was algorithmic flops / hardware peak)
        assert d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["logical_cpus"]
    else:
        assert p["teacher_forced_within_bars"] is True
    assert len(p["teacher_forced"]) == d["config"]["ddim_steps"]
    assert all(s["epe_delta"] < p["bars"]["epe"] for s in p["teacher_forced"] + p["free_run"])
    assert p["free_run_final"]["epe_delta"] < p["bars"]["epe"]


def test_calibrated_network_meets_the_raw_bars_on_all_pixels():
    """Round 4 on: the bench line carries a second parity leg on the calibrated network (BatchNorm buffers = statistics
    of the data), where the contract holds as written -- every step, all pixels (a tight ceiling, not a
    multiple of the contract)."""
    d = latest()
    p = d.get("parity_vs_oracle_calibrated")
    if p is None:                       # a bench line of an earlier round
        return
    assert "error" not in p, p
    assert p["within_raw_bars"] is True and p["final_within_raw_bars"] is True
    assert all(s["frac_gt_1e-3"] <= p["bars"]["frac"] and s["epe_delta"] < p["bars"]["epe"] for s in p["teacher_forced"])
    if sum(s["flips_mask_zero"] for s in p["free_run"]) == 0:
        assert all(s["frac_gt_1e-3"] <= p["bars"]["frac"] for s in p["free_run"])


def test_design_kernel_table_is_the_newest_bench_record():
    """DESIGN.md's kernel table is generated from the newest profiles/r*_bench.json (tools/design_kernel_table.py): the
    numbers in the document cannot drift from the committed record."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "design_kernel_table.py"), "--check"])
    assert r.returncode == 0, "run python tools/design_kernel_table.py"
