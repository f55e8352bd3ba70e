"""Print the headline figures of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(f"{d['value']:.2f} pairs/s  {d['ms_per_step']:.2f} ms/step  issued MFMA frac whole path {d.get('mfma_f32_issued_frac_whole_path')}")
if "roofline" in d:
    r = d["roofline"]
    print(f"roofline: {r['kernel'][:40]} frac {r['frac']:.3f} avg {r['avg_ms']:.3f} ms traffic {r['traffic']}")
for k in d.get("roofline_kernels", []):
    print(f"  {k['kernel'][:60]:60s} {k['bound']:5s} frac {k['frac']:.3f}  {k['ms_per_step']:.2f} ms/step")
print(d.get("kernels_ms_per_step"))
for key in ("cpu_baseline", "extras"):
    if key in d:
        print(key, json.dumps(d[key])[:1500])
