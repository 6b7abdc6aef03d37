"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter block per pass, as gpurun requires) into the per-launch
HBM traffic summary bench.py reports as roofline.traffic.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-rollout
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-rollout
    python tools/pmc_traffic.py out/pmc_f/f_counter_collection.csv out/pmc_w/w_counter_collection.csv profiles/r1d_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B
requests of a wide coalesced stream (16 B per lane, global_load and buffer_load...lds alike) at 64 B, so the read side is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Infinity-Cache hits are counted, so this is fabric traffic.
"""
import collections
import csv
import json
import re
import sys


def _load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "").strip()
            name = re.sub(r"<.*", "", name)          # fold template instances: the plan mixes them per layer shape
            e = per[name]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return per


def main(fetch_csv, write_csv, out_json, command=""):
    F = _load(fetch_csv, "FETCH_SIZE")
    W = _load(write_csv, "WRITE_SIZE")
    out = {"command": command or "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-rollout",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 FETCH_SIZE counts 128-B requests as 64 B)",
           "kernels": {}}
    for name, (n, kb) in sorted(F.items(), key=lambda kv: -kv[1][1]):
        wn, wkb = W.get(name, (0, 0.0))
        if n == 0 or wn != n:
            continue                                  # the two passes must have seen the same launches
        out["kernels"][name] = {"launches": n, "fetch_bytes_per_launch": round(2 * kb * 1024 / n), "write_bytes_per_launch": round(wkb * 1024 / n),
                                "hbm_bytes_per_launch": round((2 * kb + wkb) * 1024 / n)}
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in list(out["kernels"].items())[:12]:
        print(f"{k:28s} n={v['launches']:5d} fetch={v['fetch_bytes_per_launch'] / 1e6:8.1f} MB write={v['write_bytes_per_launch'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:])
