"""Build libmovenet_hip.so for gfx950 in-tree (no JIT cache: the .so travels
with the repo snapshot to the GPU box).

    python -m movenet_amd.csrc.build [--force] [--keep-temps]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmovenet_hip.so")
STAMP = os.path.join(LIB_DIR, "libmovenet_hip.stamp")

SOURCES = ["common.hip", "generate.hip", "generate_pipe.hip", "generate_pipe_h16.hip", "generate_fold.hip", "sequence.hip", "video.hip", "trainer.hip"]
HEADERS = ["common.h", "gen_common.h", "pipe_common.h", "gemm_family.h", "fused_layer.h", "fused_bwd.h", "fused_bwd_l.h", "fused_fwd.h", "fused_fwd_bf3.h", "bf3.h", os.path.join(ROOT, "include", "movenet_hip.h")]
FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
    "-ffp-contract=off",  # keep every mul/add as written: the kernels spell out fmaf themselves
]
# Per-source extra flags (none at present; e.g. {"sequence.hip": ["-fno-slp-vectorize"]}).
EXTRA_FLAGS: dict = {}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _source_paths() -> List[str]:
    return [os.path.join(HERE, s) for s in SOURCES]


def _digest() -> str:
    h = hashlib.sha256()
    for p in _source_paths() + [p if os.path.isabs(p) else os.path.join(HERE, p) for p in HEADERS]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()


def _compile_and_link(out: str, defs: List[str], keep_temps: bool, verbose: bool) -> None:
    """One hipcc -c per source (in parallel, each with its own flags), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(LIB_DIR, "obj_" + os.path.basename(out).replace(".so", ""))
    os.makedirs(objdir, exist_ok=True)
    base = [_hipcc()] + FLAGS + defs + ["-I", os.path.join(ROOT, "include"), "-I", HERE]
    if keep_temps:
        base += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]

    def one(src: str):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = base + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print("[movenet_amd] " + " ".join(cmd), flush=True)
        return obj, subprocess.run(cmd, cwd=LIB_DIR, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        results = list(pool.map(one, SOURCES))
    for obj, proc in results:
        if proc.returncode != 0:
            sys.stderr.write(proc.stdout + proc.stderr)
            raise RuntimeError(f"hipcc failed compiling {obj}")
        if verbose and (proc.stderr.strip() or proc.stdout.strip()):
            print(proc.stdout + proc.stderr)
    link = [_hipcc(), "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC"]
    link += [obj for obj, _ in results] + ["-o", out]
    if verbose:
        print("[movenet_amd] " + " ".join(link), flush=True)
    proc = subprocess.run(link, cwd=LIB_DIR, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError(f"hipcc failed linking {out}")
    if keep_temps:  # the .s files land next to the objects: keep them where they used to be
        for f in os.listdir(objdir):
            if f.endswith(".s"):
                shutil.copy(os.path.join(objdir, f), os.path.join(LIB_DIR, f))


def build_stamps(verbose: bool = True, exp: int = 0) -> str:
    """Diagnostic twin of the library with in-kernel wall-clock stamps in the PIPE
    generator (never loaded by the product; see scripts/pipe_stamps.py)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    out = os.path.join(LIB_DIR, "libmovenet_hip_stamps.so" if exp == 0 else f"libmovenet_hip_exp{exp}.so")
    _compile_and_link(out, ["-DMVN_PIPE_STAMPS", f"-DMVN_EXP={exp}"], False, verbose)
    return out


def build(force: bool = False, keep_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    digest = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == digest:
                return LIB_PATH
    _compile_and_link(LIB_PATH, [], keep_temps, verbose)
    with open(STAMP, "w") as f:
        f.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        exps = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--exp=")] or [0]
        for e in exps:
            print(build_stamps(verbose=False, exp=e))
        sys.exit(0)
    path = build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
    print(path)
