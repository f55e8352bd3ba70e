"""In-tree build of libdiffuvolume_hip.so (hipcc, gfx950 only).

``python -m diffuvolume_amd._build`` or ``__graft_entry__.build()``.  Objects are
rebuilt only when their source or a header is newer.  The .so stays in-tree so it
travels with the snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
LIB = PKG / "libdiffuvolume_hip.so"
ARCH = "gfx950"
# -fno-slp-vectorize: under plain -O3 the compiler packs adjacent scalar fp32 adds / multiplies into v_pk_* instructions
# (plus the v_pk_mov that build the operand pairs); on gfx950 that is slower than the scalar form -- measured on the
# bench: single-channel head 2.90 -> 2.48 ms per step, Winograd epilogues -0.5 %, nothing slower (48.2 -> 48.5 pairs/s).
# The packed instructions this library wants (the Winograd input transform) are written out by hand.
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=on", "-fno-slp-vectorize"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libdiffuvolume_hip.so cannot be built")
    return exe


def csrc_sha16() -> str:
    """Fingerprint of the kernel sources (csrc/*.hip, csrc/*.h, include/*.h): stamped into the profiles collected by
    tools/ and compared by bench.py, so that counter figures of an older kernel are never quoted for the current one
    (the GPU box has no .git, so the commit id is not available there)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + list((PKG.parent / "include").glob("*.h"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(verbose: bool = False, force: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list((PKG.parent / "include").glob("*.h"))
    sources = sorted(CSRC.glob("*.hip"))
    jobs = []
    for src in sources:
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return src.name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for name in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[build] compiled {name}")
    objs = [OBJ / (s.stem + ".o") for s in sources]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(verbose=True, force="--force" in sys.argv)
