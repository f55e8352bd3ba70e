"""Condense the two --pmc passes of tools/sq_counters_round.sh into <dir>/<tag>_sq_counters.json: per kernel name the
per-launch averages of every counter plus the derived figures (matrix pipe busy share, instructions per wave).
    python tools/sq_summary.py gpurun_out/sq_r03 r03 [git-head]"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
head = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else None
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diffuvolume_amd._build import csrc_sha16  # noqa: E402
N_SIMD, N_XCD = 1024, 8            # 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(str(src / "p*" / "**" / "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k, cs in agg.items():
    if not any(s in k for s in ("conv", "gwc_rows", "concat_rows", "window_attn", "upsample_softmax", "ddim_step",
                                "geo_", "corr", "refine_inputs", "rank1", "pw_expand", "patch_volume")):
        continue
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    rec = {"kernel": k, "launches": max(len(v) for v in cs.values()), "counters_per_launch": {n: round(v, 1) for n, v in sorted(c.items())}}
    d = {}
    if c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / N_XCD
        d["kernel_cycles"] = round(cyc)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / cyc, 4)
    if c.get("SQ_WAVES"):
        w = c["SQ_WAVES"]
        for n, key in (("SQ_INSTS_VALU", "valu_per_wave"), ("SQ_INSTS_SALU", "salu_per_wave"),
                       ("SQ_INSTS_VALU_MFMA_MOPS_F32", "mfma_mops_f32_per_wave")):
            if n in c:
                d[key] = round(c[n] / w, 1)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        for n, key in (("SQ_WAIT_INST_ANY", "wait_inst_any_frac_of_wave_cycles"), ("SQ_WAIT_ANY", "wait_any_frac_of_wave_cycles"),
                       ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac_of_wave_cycles")):
            if n in c:
                d[key] = round(c[n] / wc, 4)
    rec["derived"] = d
    out.append(rec)
out.sort(key=lambda r: -r["counters_per_launch"].get("GRBM_GUI_ACTIVE", 0) * r["launches"])
if head is None:
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        head = None
prog = os.environ.get("DV_SQ_ARGS") or "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timer"
doc = {"command": "rocprofv3 --kernel-trace --pmc <8 counters> (two passes) -- python " + prog, "git_head": head, "csrc_sha16": csrc_sha16(),
       "notes": "per-launch averages; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); "
                "SQ_INSTS_VALU counts MFMAs too; MOPS_F32 counts 512 flops-units per v_mfma_f32_16x16x4 (see r01_wino_pmc.txt)",
       "kernels": out}
(src / f"{tag}_sq_counters.json").write_text(json.dumps(doc, indent=1))
for r in out[:14]:
    print(r["kernel"][:70], r["launches"], r["derived"])
