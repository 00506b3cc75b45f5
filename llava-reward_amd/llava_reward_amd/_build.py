"""Build libllava_reward_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(PKG_DIR), "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libllava_reward_hip.so")
SOURCES = ["gemm.hip", "gemm8.hip", "gemm8_narrow.hip", "attention.hip", "gemm_f32.hip", "rowops.hip", "rowops_qwen.hip", "engine.hip", "qwen.hip", "preprocess.hip"]
HEADERS = ["common.h", "kernels.h", "engine.h", os.path.join("..", "..", "include", "llava_reward_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _newer(src_paths, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in src_paths)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        deps = [src] + hdrs + ([os.path.join(CSRC, "gemm8.hip")] if s == "gemm8_narrow.hip" else [])     # (it includes gemm8.hip)
        if force or _newer(deps, obj):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(objs, LIB_PATH):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in os.sys.argv, verbose=True))
