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

SOURCES = ["common.hip", "generate.hip", "generate_pipe.hip", "sequence.hip", "video.hip"]
HEADERS = ["common.h", "gen_common.h", "gemm_family.h", os.path.join(ROOT, "include", "movenet_hip.h")]
FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
    "-ffp-contract=off",  # keep every mul/add as written: the kernels spell out fmaf themselves
]


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
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_stamps(verbose: bool = True, exp: int = 0) -> str:
    """Diagnostic twin of the library with in-kernel wall-clock stamps in the PIPE
    generator (never loaded by the product; see scripts/pipe_stamps.py)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    out = os.path.join(LIB_DIR, "libmovenet_hip_stamps.so" if exp == 0 else f"libmovenet_hip_exp{exp}.so")
    cmd = [_hipcc()] + FLAGS + ["-DMVN_PIPE_STAMPS", f"-DMVN_EXP={exp}",
                                "-I", os.path.join(ROOT, "include"), "-I", HERE]
    cmd += _source_paths() + ["-o", out]
    if verbose:
        print("[movenet_amd] " + " ".join(cmd), flush=True)
    proc = subprocess.run(cmd, cwd=LIB_DIR, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("hipcc failed building libmovenet_hip_stamps.so")
    return out


def build(force: bool = False, keep_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    digest = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == digest:
                return LIB_PATH
    cmd = [_hipcc()] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", HERE]
    if keep_temps:
        tmp = os.path.join(LIB_DIR, "temps")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    cmd += _source_paths() + ["-o", LIB_PATH]
    if verbose:
        print("[movenet_amd] " + " ".join(cmd), flush=True)
    cwd = os.path.join(LIB_DIR, "temps") if keep_temps else LIB_DIR
    proc = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("hipcc failed building libmovenet_hip.so")
    if verbose and (proc.stderr.strip() or proc.stdout.strip()):
        print(proc.stdout + proc.stderr)
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
