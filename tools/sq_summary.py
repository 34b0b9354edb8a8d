#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc SQ passes of bench.py into per-kernel instruction mix and issue/wait fractions.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
        --output-format csv -d gpurun_out/sq1 -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-render
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD \
        --output-format csv -d gpurun_out/sq2 -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-render
    python tools/sq_summary.py gpurun_out/sq1/p_counter_collection.csv gpurun_out/sq2/p_counter_collection.csv --grid 256 --iters 40 \
        > profiles/rNN_sq_counters.json

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); WAIT_ANY (parked in
s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (issuing) ~= WAVE_CYCLES.  `limiter` names the largest of the
three.  The per-wave
instruction counts are SQ_INSTS_* / SQ_WAVES.  Counter passes are separate runs (8 SQ slots per pass).
"""
import argparse
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidx12_amd.build import kernel_source_hash   # noqa: E402  (stamps each kernel's summary with the code it was measured on)


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0].split("<")[0]
    return n.split("::")[-1]


def load(fn):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        if "fx::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pass1_csv")
    ap.add_argument("pass2_csv")
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--storage", default="fp32")
    ap.add_argument("--mode", default="fixed", help="bench.py --mode of the profiled run (fixed / faithful)")
    a = ap.parse_args()
    p1, p2 = load(a.pass1_csv), load(a.pass2_csv)
    out = {"grid": a.grid, "iters": a.iters, "storage": a.storage, "mode": a.mode,
           "method": "rocprofv3 --pmc, two SQ passes (8 slots each), averages per dispatch; fractions are of SQ_WAVE_CYCLES", "kernels": {}}
    for k in sorted(set(p1) | set(p2)):
        m1 = {c: sum(v) / len(v) for c, v in p1.get(k, {}).items()}
        m2 = {c: sum(v) / len(v) for c, v in p2.get(k, {}).items()}
        wc = m1.get("SQ_WAVE_CYCLES", 0.0)
        waves = m2.get("SQ_WAVES", 0.0)
        e = {"dispatches": len(next(iter(p1.get(k, p2.get(k)).values()))), "waves": waves, "source_hash": kernel_source_hash(k)}
        if wc:
            e["frac_issuing"] = m1.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            e["frac_issue_stalled"] = m1.get("SQ_WAIT_INST_ANY", 0.0) / wc
            e["frac_parked_waitcnt"] = m1.get("SQ_WAIT_ANY", 0.0) / wc
            e["frac_valu"] = m1.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
            e["frac_lds"] = m1.get("SQ_ACTIVE_INST_LDS", 0.0) / wc
            e["frac_vmem"] = m1.get("SQ_ACTIVE_INST_VMEM", 0.0) / wc
            e["wave_quad_cycles"] = wc
            top = max(("issue", e["frac_issuing"]), ("pipe-stall (a full VMEM / LDS queue holds the next instruction back)", e["frac_issue_stalled"]),
                      ("latency (waves parked in s_waitcnt)", e["frac_parked_waitcnt"]), key=lambda t: t[1])[0]
            e["limiter"] = "%s: %.0f %% of wave cycles issuing, %.0f %% stalled at issue, %.0f %% parked in s_waitcnt" % (
                top, 100 * e["frac_issuing"], 100 * e["frac_issue_stalled"], 100 * e["frac_parked_waitcnt"])
        if waves:
            for c, key in (("SQ_INSTS_VALU", "valu_per_wave"), ("SQ_INSTS_SALU", "salu_per_wave"), ("SQ_INSTS_LDS", "lds_per_wave"),
                           ("SQ_INSTS_VMEM_RD", "vmem_rd_per_wave"), ("SQ_INSTS_VMEM_WR", "vmem_wr_per_wave")):
                if c in m2:
                    e[key] = m2[c] / waves
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
