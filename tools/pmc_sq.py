"""Where the waves of each kernel spend their cycles, and how busy the LDS array is, from TWO rocprofv3 --pmc passes (each in a run of its
own, the program directly after `--`; 8 SQ counters + GRBM_GUI_ACTIVE per pass):

    A: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
    B: SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
    python tools/pmc_sq.py A_counter_collection.csv B_counter_collection.csv profiles/<tag>_pmc_sq.json "<command>"

Reading (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall
(dependency / pipe), SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to SQ_WAVE_CYCLES (all in quad-cycles, summed over
waves).  SQ_LDS_IDX_ACTIVE = LDS-array cycles (shader cycles, summed over the chip's 256 LDS arrays), SQ_LDS_BANK_CONFLICT the extra cycles
among them; a dispatch lasted GRBM_GUI_ACTIVE / 8 cycles, so  lds_busy = SQ_LDS_IDX_ACTIVE / (256 * GRBM_GUI_ACTIVE / 8).
"""
import collections
import csv
import json
import re
import sys

N_CU = 256


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "").strip()
    return re.sub(r"<.*", "", name)


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return per, n


def main(a_csv, b_csv, out_json, command=""):
    A, nA = load(a_csv)
    Bc, _ = load(b_csv)
    out = {"command": command, "kernels": {}}
    for k, c in sorted(A.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0 or len(nA[k]) < 4:
            continue
        e = {"launches": len(nA[k]),
             "wave_parked": round(c.get("SQ_WAIT_ANY", 0) / wc, 3), "issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
             "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "issue_stalled_on_lds": round(c.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
             "issuing_vmem": round(c.get("SQ_ACTIVE_INST_VMEM", 0) / wc, 3), "issuing_lds": round(c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3),
             "issuing_valu": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)}
        b = Bc.get(k)
        if b and b.get("GRBM_GUI_ACTIVE", 0) > 0:
            cyc = b["GRBM_GUI_ACTIVE"] / 8
            e["lds_busy"] = round(b.get("SQ_LDS_IDX_ACTIVE", 0) / (N_CU * cyc), 3)
            if b.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
                e["lds_conflict_share"] = round(b.get("SQ_LDS_BANK_CONFLICT", 0) / b["SQ_LDS_IDX_ACTIVE"], 3)
            e["lds_data_fifo_full_per_cu_cycle"] = round(b.get("SQ_LDS_DATA_FIFO_FULL", 0) / (N_CU * cyc), 3)
            e["lds_cmd_fifo_full_per_cu_cycle"] = round(b.get("SQ_LDS_CMD_FIFO_FULL", 0) / (N_CU * cyc), 3)
            if b.get("SQ_INSTS_VMEM_RD", 0) > 0:
                e["vmem_rd_issue_quadcycles_per_inst"] = round(b.get("SQ_INST_CYCLES_VMEM_RD", 0) / b["SQ_INSTS_VMEM_RD"], 1)
            e["lds_insts_per_launch"] = round(b.get("SQ_INSTS_LDS", 0) / max(1, len(nA[k])))
            e["vmem_rd_insts_per_launch"] = round(b.get("SQ_INSTS_VMEM_RD", 0) / max(1, len(nA[k])))
        out["kernels"][k] = e
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    for k, e in list(out["kernels"].items())[:12]:
        print(f"{k:26s} " + " ".join(f"{a}={v}" for a, v in e.items()))


if __name__ == "__main__":
    main(*sys.argv[1:])
