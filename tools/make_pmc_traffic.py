"""Condense a tools/profile_round.sh run into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_traffic.json.
FETCH_SIZE on gfx950 counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
Both are reported in KB by rocprofv3.   python tools/make_pmc_traffic.py gpurun_out/prof_r01 r01"""
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
head = sys.argv[3] if len(sys.argv) > 3 else None         # the GPU box has no .git: the caller passes the commit id
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diffuvolume_amd._build import csrc_sha16  # noqa: E402
dst = src / "summary"      # gpurun_out/ travels back from the GPU box; copy the files into profiles/ afterwards
dst.mkdir(parents=True, exist_ok=True)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def pmc(dirname, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(str(src / dirname / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
kernels = []
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    if not any(s in k for s in ("conv3d", "deconv3d", "gwc_rows", "concat_rows", "window_attn", "upsample_softmax",
                                "ddim_step", "noise_prepare", "encode_two_hot", "masked_metrics", "rank1", "pw_expand", "patch_volume",
                                "softmax_d", "mul2")):
        continue
    f_kb = sum(fetch[k]) / len(fetch[k])
    w_kb = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
    kernels.append({"kernel": k, "launches": len(fetch[k]), "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb,
                    "hbm_read_bytes_corrected": 2 * f_kb * 1024, "hbm_write_bytes": w_kb * 1024,
                    "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024})
out = {"git_head": head, "csrc_sha16": csrc_sha16(),
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py "
                  "--steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timer",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM section); "
                     "check: gwc_rows_kernel 2*FETCH ~= 629 MB (its algorithmic input); WRITE_SIZE exact",
       "kernels": kernels}
(dst / f"{tag}_pmc_traffic.json").write_text(json.dumps(out, indent=1))
stats = glob.glob(str(src / "stats" / "**" / "*kernel_stats.csv"), recursive=True)
if stats:
    text = open(stats[0]).read()
    (dst / f"{tag}_kernel_stats.csv").write_text(
        '"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras   (MI355X)"\n' + text)
b = src / "bench_under_rocprof.json"
if b.exists():
    lines = [l for l in b.read_text().splitlines() if l.startswith("{")]
    if lines:
        (dst / f"{tag}_bench_under_rocprof.json").write_text(lines[-1] + "\n")
print("wrote", sorted(p.name for p in dst.glob(f"{tag}_*")))
