"""Build libfluidnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m fluidnet_amd.build [--force]

The .so lands in fluidnet_amd/lib/ (git-ignored, but it travels to the GPU box with the tree).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfluidnet_hip.so")
SOURCES = ["fnet_elementwise.hip", "fnet_loss.hip", "fnet_conv_ref.hip", "fnet_conv_mfma.hip", "fnet_optim.hip", "fnet_plan.hip", "fnet_rollout.hip"]
HEADERS = [os.path.join(CSRC, "fnet_common.h"), os.path.join(os.path.dirname(HERE), "include", "fluidnet_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        # -z defs: an unresolved kernel stub must fail the build here, not the dlopen on the GPU box
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
