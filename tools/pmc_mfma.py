"""Matrix-core utilisation per kernel from ONE rocprofv3 --pmc pass (counters in their own run, program directly after `--`):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d out/pmc_m -o m \\
        -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-rollout --no-config4
    python tools/pmc_mfma.py out/pmc_m/m_counter_collection.csv profiles/r3_pmc_mfma.json

Reading (MI355X_MICROARCH.md, 'rocprofv3 PMC slots' and the latency table): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's
matrix pipe is busy, summed over all SIMDs (32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles, so a
dispatch lasted GRBM_GUI_ACTIVE / 8 cycles and
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (N_SIMD * GRBM_GUI_ACTIVE / 8),   N_SIMD = 256 CUs * 4.
SQ_INSTS_VALU_MFMA_MOPS_BF16 counts 512-FLOP units ("MOPS"), so bf16 FLOPs = 512 * MOPS (checked below against 32 busy cycles per 32768-FLOP
instruction: busy_cycles_per_kflop).  With kernel durations of a --kernel-trace pass (optional third argument: its kernel_stats.csv) the
effective clock = GRBM_GUI_ACTIVE / 8 / duration and the clock-adjusted peak follow.
"""
import collections
import csv
import json
import re
import sys

N_SIMD = 256 * 4


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "").strip()
    return re.sub(r"<.*", "", name)


def main(pmc_csv, out_json, stats_csv=None, command=""):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    own_ns = collections.defaultdict(float)               # dispatch durations inside the counter pass itself (kernels run serialised there)
    with open(pmc_csv) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in launches[k]:
                own_ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            launches[k].add(r["Dispatch_Id"])
    dur = {}
    if stats_csv:
        with open(stats_csv) as f:
            for r in csv.DictReader(f):
                k = short(r["Name"])
                d = dur.setdefault(k, [0, 0.0])
                d[0] += int(r["Calls"])
                d[1] += float(r["TotalDurationNs"])
    out = {"command": command, "n_simd": N_SIMD,
           "reading": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); bf16 FLOP = 512 * SQ_INSTS_VALU_MFMA_MOPS_BF16",
           "kernels": {}}
    for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        e = {"launches": len(launches[k]), "mfma_busy_cycles": busy, "grbm_gui_active": gui, "mfma_util": round(busy / (N_SIMD * gui / 8), 4)}
        if "SQ_BUSY_CYCLES" in c:
            e["sq_busy_cycles"] = c["SQ_BUSY_CYCLES"]
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16")
        if mops:
            e["bf16_tflop_per_launch"] = round(512.0 * mops / len(launches[k]) / 1e12, 4)
            e["busy_cycles_per_kflop"] = round(busy / (512.0 * mops / 1e3), 5)          # 32 / 32.768 = 0.977 for v_mfma_f32_32x32x16_bf16
        ns = own_ns[k] / len(launches[k])
        cyc = gui / 8 / len(launches[k])
        e["avg_duration_us_in_this_pass"] = round(ns / 1e3, 2)
        # (GRBM_GUI_ACTIVE / 8 / duration reads high on dispatches shorter than ~0.3 ms: MI355X_MICROARCH.md, DVFS give-back)
        e["gui_cycles_per_ns"] = round(cyc / ns, 3)
        if mops:
            e["executed_tflops"] = round(512.0 * mops / len(launches[k]) / ns / 1e3, 1)
            e["executed_frac_of_2500"] = round(512.0 * mops / len(launches[k]) / ns / 1e3 / 2500.0, 4)
        if k in dur and dur[k][0]:
            e["avg_duration_us_kernel_trace"] = round(dur[k][1] / dur[k][0] / 1e3, 2)
        out["kernels"][k] = e
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    for k, e in out["kernels"].items():
        print(f"{k:28s} n={e['launches']:5d} mfma_util={e['mfma_util']:.3f} " + " ".join(f"{a}={b}" for a, b in e.items() if a not in ("launches", "mfma_util", "mfma_busy_cycles", "grbm_gui_active", "sq_busy_cycles")))


if __name__ == "__main__":
    main(*sys.argv[1:])
