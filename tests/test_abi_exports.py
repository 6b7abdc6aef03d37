"""The C ABI: libfluidnet_hip.so loads on a GPU-less host and exports every function include/fluidnet_hip.h declares; the
header's record layout and opcode numbering match the Python binding.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from fluidnet_amd.engine import runtime as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "fluidnet_hip.h")


def _header_text():
    with open(HDR) as f:
        return re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)


def test_every_declared_function_is_exported():
    decls = re.findall(r"^\s*(?:const\s+char\s*\*|int)\s+(fnet_\w+)\s*\(", _header_text(), flags=re.M)
    assert len(decls) >= 15
    L = rt.lib()
    missing = [d for d in decls if not hasattr(L, d)]
    assert not missing, missing
    assert sorted(decls) == sorted(rt.EXPORTS)
    assert L.fnet_abi_version() == rt.ABI_VERSION == 2
    assert L.fnet_strerror(-1).decode() and L.fnet_strerror(-2).decode()


def test_opcodes_match_header():
    body = re.search(r"enum FnetOpCode \{(.*?)\}", _header_text(), flags=re.S).group(1)
    pairs = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"FNET_OP_(\w+)\s*=\s*(\d+)", body))
    assert pairs == rt.OPC


def test_record_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fluidnet_ops.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(FnetOp), offsetof(FnetOp,dtype), offsetof(FnetOp,i),'
                   ' offsetof(FnetOp,f), offsetof(FnetOp,p)); return 0;}\n')
    exe = tmp_path / "sz.bin"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    d = rt.OP_DTYPE
    assert got == [d.itemsize, d.fields["dtype"][1], d.fields["i"][1], d.fields["f"][1], d.fields["p"][1]]


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(rt, "_lib", None)
    monkeypatch.setattr(rt, "LIB_PATH", "/nonexistent/libfluidnet_hip.so")
    with pytest.raises(rt.FluidNetHipError, match="no CPU fallback"):
        rt.lib()


def test_plan_run_rejects_bad_record_without_a_gpu():
    # host-side validation comes before any launch: an unknown opcode is FNET_E_BADARG and reports its index
    import numpy as np
    ops = np.zeros(2, dtype=rt.OP_DTYPE)
    ops[1]["op"] = 999
    failed = ctypes.c_int32(-1)
    rc = rt.lib().fnet_plan_run(ops.ctypes.data, 2, None, ctypes.byref(failed))
    assert rc in (-1, -3) and failed.value == 1


def test_native_param_count_matches_the_state_dict(tmp_path):
    """fnet_unet_param_count (host-only, no GPU) from a plain C program linked against the library: the native rollout's
    parameter table must describe exactly the reference state dict (6,031,491 floats for UNet_medium)."""
    import fluidnet_oracle as orc
    want = sum(int(np.prod(s)) for s in orc.param_shapes(orc.medium_config()).values())
    src = tmp_path / "pc.c"
    src.write_text('#include <stdio.h>\n#include "fluidnet_hip.h"\n'
                   'int main(void){FnetUNetConfig c={6,3,32,3,FNET_ACT_GELU,FNET_PAD_REPLICATE,1,1,1,FNET_OUT_LINEAR_CLAMP};'
                   'int64_t n=0; int rc=fnet_unet_param_count(&c,&n); printf("%d %lld\\n", rc, (long long)n);'
                   'c.depth=0; rc=fnet_unet_param_count(&c,&n); printf("%d\\n", rc); return 0;}\n')
    exe = tmp_path / "pc.bin"
    libdir = os.path.dirname(rt.LIB_PATH)
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lfluidnet_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)]).split()
    assert int(out[0]) == 0 and int(out[1]) == want
    assert int(out[2]) == -1                       # FNET_E_BADARG for an unsupported configuration


def test_knob_setter_is_host_only_and_rejects_unknown_names():
    """fnet_set_knob (no getenv on the launch path; tests override the cached values through an exported setter)."""
    from fluidnet_amd.engine import runtime as rt
    rt.set_knob("wg_px", 128)
    rt.set_knob("wg_px", -1)                 # back to the environment's value / default
    with pytest.raises(rt.FluidNetHipError):
        rt.set_knob("no_such_knob", 1)
