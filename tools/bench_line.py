"""Print the interesting fields of a bench.py JSON line: python tools/bench_line.py FILE [LABEL]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b = d.get("breakdown_ms_per_step", {})
keys = ("CONV3X3_MFMA", "WGRAD_MFMA", "PACK_S2D", "UNPACK_S2D", "SUBSAMPLE_BIAS", "ZERO_INSERT", "PAD_RING")
print(sys.argv[2] if len(sys.argv) > 2 else "", d["value"], d["unit"], d["ms_per_step"], "ms; conv", d.get("roofline", {}).get("achieved"),
      {k: b[k] for k in keys if k in b}, "loss", d["config"].get("final_loss"))
