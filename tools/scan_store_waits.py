"""Scan compiled gfx950 ISA for `s_waitcnt vmcnt(N)` that drain STORES (on gfx9 stores count in vmcnt): a wait that follows
stores with no load in between and allows fewer outstanding operations than stores are pending.  In a persistent kernel such a
wait serialises the epilogue behind a write round trip (DESIGN.md lesson 22).
    python tools/scan_store_waits.py [kernel.s ...]        (default: compile every csrc/*.hip to ISA under /tmp first)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def scan(path):
    name, pend_store, cnt, stores = None, 0, {}, {}
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, pend_store = m.group(1), 0
            cnt[name], stores[name] = 0, 0
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith('s_endpgm'):
            name = None
            continue
        op = t.split()[0] if t else ''
        if op.startswith(('global_store', 'buffer_store')):
            pend_store += 1
            stores[name] += 1
        elif op.startswith(('global_load', 'buffer_load', 'global_atomic')):
            pend_store = 0                      # a wait after a load is legitimate (conservative)
        elif op == 's_waitcnt' and 'vmcnt' in t:
            n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
            if pend_store > n:                  # waits for at least one store
                cnt[name] += 1
            pend_store = min(pend_store, n)
    return [(k, v, stores[k]) for k, v in cnt.items() if v > 0]


def main(files):
    if not files:
        sys.path.insert(0, str(ROOT))
        from diffuvolume_amd import _build
        out = Path(tempfile.mkdtemp(prefix="dv_isa_"))
        flags = [f for f in _build.FLAGS if f != "-fPIC"]
        for src in sorted((ROOT / "diffuvolume_amd" / "csrc").glob("*.hip")):
            dst = out / (src.stem + ".s")
            subprocess.run([_build._hipcc(), *flags, "-S", "--cuda-device-only", str(src), "-o", str(dst)], check=True,
                           capture_output=True)
            files.append(str(dst))
    for f in files:
        for kernel, waits, nstores in scan(f):
            print(Path(f).stem, kernel[:70], 'store-waits', waits, 'of stores', nstores)


if __name__ == "__main__":
    main(sys.argv[1:])
