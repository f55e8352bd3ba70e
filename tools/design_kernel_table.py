"""Regenerate the kernel table of DESIGN.md (between the KERNEL_TABLE markers) from the newest committed bench record
(profiles/r*_bench.json): the table IS that file's `kernels_ms_per_step` / `roofline` / `roofline_kernels` content.
    python tools/design_kernel_table.py            rewrite DESIGN.md in place
    python tools/design_kernel_table.py --check    exit 1 if DESIGN.md's table differs from the newest record"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
BEGIN, END = "<!-- KERNEL_TABLE_BEGIN -->", "<!-- KERNEL_TABLE_END -->"
WHAT = {
    "conv3d_k3s1_co32": ("Winograd 3×3×3 stride 1, 32→32 (20 launches per step)", "4b"),
    "conv3d_k3s1_co64": ("Winograd F(2×2×2), 64→64 (4×4 tiles; issued = algorithmic / 3.375)", "4b′"),
    "conv3d_k3s1_co128": ("Winograd F(2×2×2), 128→128 (2×8 tiles on the 60-wide plane)", "4b′"),
    "deconv3d_k3s2_redir": ("transposed convolution + fused redir, persistent with loader waves", "4e"),
    "conv3d_k3s2_co64": ("polyphase stride 2, 32→64", "4d"),
    "conv3d_k3s2_co128": ("polyphase stride 2, 64→128", "4d"),
    "conv3d_k3s1_co32_filter_rank1": ("filter + `dres0[0]` as a rank-1 stencil (VALU)", "4a"),
    "window_attn3d": ("window attention", "small"),
    "conv3d_k3s1_co1": ("32→1 head, z-march (VALU)", "small"),
    "upsample_softmax_regress": ("trilinear ×4 + softmax + regression (+uncertainty)", "small"),
    "pointwise_expand_co864": ("rank-1 tables (once per pair)", "4a"),
    "gwc_volume": ("group-wise correlation volume (once per pair)", "small"),
}


def newest():
    files = sorted((ROOT / "profiles").glob("r*_bench.json"))
    return files[-1], json.loads(files[-1].read_text())


def table():
    f, d = newest()
    ks = d["kernels_ms_per_step"]
    frac = {}
    r = d["roofline"]
    for t in ("conv3d_k3s1_co32",):
        frac[t] = (r["frac"], r.get("traffic"), r.get("algorithmic_bytes_per_launch"))
    for k in d.get("roofline_kernels", []):
        for t in k.get("tags", []):
            frac[t] = (k["frac"], k.get("traffic"), k.get("algorithmic_bytes_per_launch"))
    lines = [f"Source: `profiles/{f.name}` — {d['value']:.2f} pairs/s, {d['ms_per_step']:.1f} ms per batch of {d['config']['global_batch']}; "
             f"whole path issued / peak {d.get('mfma_f32_issued_frac_whole_path', float('nan')):.3f}.", "",
             "| KernelTimer tag | what (section) | ms per step | issued fraction of the fp32 matrix pipe | HBM bytes per launch: PMC / algorithmic |",
             "|---|---|---|---|---|"]
    for tag, ms in sorted(ks.items(), key=lambda kv: -kv[1]):
        what, sec = WHAT.get(tag, (tag, ""))
        fr = frac.get(tag)
        fs = f"{fr[0]:.3f}" if fr else "—"
        tr = "—"
        if fr and fr[1] and fr[2]:
            tr = f"{fr[1] / 1e6:.0f} MB / {fr[2] / 1e6:.0f} MB = {fr[1] / fr[2]:.2f}×"
        lines.append(f"| `{tag}` | {what} ({sec}) | {ms:.2f} | {fs} | {tr} |")
    lines.append(f"| sum of the tags | | {sum(ks.values()):.1f} | | |")
    return "\n".join(lines)


def main():
    p = ROOT / "DESIGN.md"
    s = p.read_text()
    a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
    new = s[:a] + "\n" + table() + "\n" + s[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == s else 1)
    p.write_text(new)
    print(table())


if __name__ == "__main__":
    main()
