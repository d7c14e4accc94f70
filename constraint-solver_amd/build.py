"""Builds the HIP shared library for gfx950 in-tree:
constraint-solver_amd/lib/libls_amd.so (git-ignored, travels with gpurun).

    python constraint-solver_amd/build.py [--force] [--keep-temps]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libls_amd.so")
SOURCES = ["ls_api.hip", "nq_kernels.hip", "nq_server.hip", "sched_wave.hip", "sched_delta.hip", "sched_server.hip", "sched_grid.hip", "ls_solver.hip"]
ARCH = "gfx950"


def _deps():
    out = []
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files]
    out.append(os.path.join(os.path.dirname(HERE), "include", "ls_amd.h"))
    out.append(os.path.abspath(__file__))
    return out


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, keep_temps: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return SO
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-c", os.path.join(CSRC, src), "-o", obj]
        if keep_temps:
            cmd += ["-save-temps=obj"]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd, cwd=LIBDIR), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    tmp = SO + ".tmp"
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link, cwd=LIBDIR)
    os.replace(tmp, SO)  # atomic: concurrent ranks never see a half-written library
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv))
