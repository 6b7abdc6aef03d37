#!/usr/bin/env python3
"""VGPR count, occupancy (waves per SIMD) and scratch bytes of every gfx950 kernel in one csrc file, compiled with the build's flags.
usage: python tools/kernel_resources.py fnet_elementwise.hip [substring filter] [-DMACRO=..]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidnet_amd.build import FLAGS  # noqa: E402

src = os.path.join(ROOT, "fluidnet_amd", "csrc", sys.argv[1])
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
defs = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    flags = [f for f in FLAGS if f != "-fPIC"]
    subprocess.run(["hipcc", *flags, *defs, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, src], check=True,
                   stderr=subprocess.DEVNULL)
    s = open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?); Occupancy: (\d+)", s, re.S):
    name, body = m.group(1), m.group(2)
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d*", "", name)
    if filt and not all(f in name for f in filt):
        continue
    v = re.search(r"; NumVgprs: (\d+)", body).group(1)
    sc = re.search(r"; ScratchSize: (\d+)", body).group(1)
    print(f"vgpr {v:>4}  occupancy {m.group(3)}  scratch {sc:>4}  {name[:80]}")
