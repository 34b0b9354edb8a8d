#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into per-kernel HBM-side traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv \
        --grid 256 --iters 40 --storage fp32 > profiles/rNN_pmc_traffic.json

Corrections, exactly as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: the two counters
are collected in SEPARATE passes (TCC slots), both are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of a
wide coalesced read (128-B requests tallied at 64 B), so it is doubled (`wide_loads`); WRITE_SIZE is
taken as reported (it matches the algorithmic store bytes of every kernel here to the byte).  Infinity-Cache hits are
counted by these counters: at 256^3 the figure is L2<->fabric traffic, an upper bound on HBM traffic.
"""
import argparse
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidx12_amd.build import kernel_source_hash   # noqa: E402  (stamps each kernel's summary with the code it was measured on)

# Every kernel here streams whole 128-B lines (coalesced 4-B or 16-B per lane), i.e. 128-B fabric requests that the
# counter tallies at 64 B: the doubling applies to all of them.  Evidence: k_divergence must read >= 201 MB (three
# velocity planes) and reports 142 MB raw.
WIDE = {"k_jacobi_v4": True, "k_advect": True, "k_divergence": True, "k_project": True,
        "k_jacobi_generic": True}


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0].split("<")[0]
    return n.split("::")[-1]


def load(fn, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == counter and "fx::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--storage", default="fp32")
    ap.add_argument("--mode", default="fixed", help="bench.py --mode of the profiled run (fixed / faithful)")
    ap.add_argument("--steps-profiled", type=int, default=5, help="--steps + --warmup of the profiled bench.py run (dispatches / this = launches per step)")
    a = ap.parse_args()
    f, w = load(a.fetch_csv, "FETCH_SIZE"), load(a.write_csv, "WRITE_SIZE")
    out = {"grid": a.grid, "iters": a.iters, "storage": a.storage, "mode": a.mode, "steps_profiled": a.steps_profiled, "unit": "bytes per launch",
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB*1024; FETCH x2 for 16-B-load kernels "
                     "(gfx950 correction, MI355X_MICROARCH.md); includes Infinity-Cache hits", "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fr = sum(f[k]) / max(len(f[k]), 1) * 1024.0
        wr = sum(w[k]) / max(len(w[k]), 1) * 1024.0
        fc = fr * (2.0 if WIDE.get(k, True) else 1.0)
        out["kernels"][k] = {"dispatches": len(f[k]), "fetch_raw": fr, "fetch_corrected": fc, "write": wr, "traffic": fc + wr,
                             "wide_loads": bool(WIDE.get(k, True)), "source_hash": kernel_source_hash(k)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
