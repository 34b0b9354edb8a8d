#!/usr/bin/env python3
"""Summarise the render kernels of a profiled bench.py run: average duration (rocprofv3 --kernel-trace --stats) and cache hit rates
(two --pmc passes) at the frame bench.py pins its render leg to (132).

    B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline [--config 5]"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- $B
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o p -- $B
    rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/l1 -o p -- $B
    python tools/render_pmc_summary.py $O/kt/k_kernel_stats.csv $O/l2/p_counter_collection.csv $O/l1/p_counter_collection.csv \
        --grid 256 --storage fp32 [--has-sh] > profiles/rNN_render_pmc.json

l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS) (MI355X_MICROARCH.md, L2); l1_hit_rate = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES
(read requests the vector L1 passed on to the L2 per L1 access).  Counter passes are separate runs; every dispatch of a kernel is
averaged (the render leg repeats each pass a few times on the same state)."""
import argparse
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidx12_amd.build import kernel_source_hash   # noqa: E402

RENDER = ("k_build_fill", "k_occupancy_blocks", "k_occupancy_blocks_a4", "k_occupancy_dilate", "k_mask_coarsen", "k_light_cells", "k_light_classify", "k_light_rays", "k_light_gi_dirs",
          "k_light_march", "k_view_slots", "k_view_march", "k_direct_march", "k_raymarch_light", "k_raymarch_view", "k_raycast_direct", "k_resolve_cube")


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0].split("<")[0]
    return n.split("::")[-1]


def counters(fn):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        k = short(r["Kernel_Name"])
        if k in RENDER:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats_csv")
    ap.add_argument("l2_csv")
    ap.add_argument("l1_csv")
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--storage", default="fp32")
    ap.add_argument("--has-sh", action="store_true")
    ap.add_argument("--frame", type=int, default=132)
    a = ap.parse_args()
    out = {"grid": a.grid, "storage": a.storage, "has_sh": a.has_sh, "frame": a.frame,
           "method": "rocprofv3 --kernel-trace --stats; --pmc TCC_HIT_sum TCC_MISS_sum; --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum (separate passes)",
           "kernels": {}}
    stats = {}
    for r in csv.DictReader(open(a.stats_csv)):
        k = short(r["Name"])
        if k in RENDER:
            e = stats.setdefault(k, [0, 0.0, float("inf")])
            e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"]); e[2] = min(e[2], float(r["MinNs"]))
    l2, l1 = counters(a.l2_csv), counters(a.l1_csv)
    for k in RENDER:
        if k not in stats and k not in l2:
            continue
        e = {"source_hash": kernel_source_hash(k)}
        if k in stats:
            e["calls"], e["avg_us"] = stats[k][0], stats[k][1] / stats[k][0] / 1e3
            e["min_us"] = stats[k][2] / 1e3                          # (a kernel's first call of a process costs several hundred us more: the average of a dozen calls carries it)
            e["l1_accesses_per_cycle_and_cu_at_min"] = None
        h, m = l2.get(k, {}).get("TCC_HIT_sum"), l2.get(k, {}).get("TCC_MISS_sum")
        if h is not None and m is not None and h + m > 0:
            e["l2_requests"], e["l2_hit_rate"] = h + m, h / (h + m)
            if e.get("avg_us"):
                e["l2_request_GBps_at_128B"] = (h + m) * 128 / (e["avg_us"] * 1e-6) / 1e9      # what the L2 was asked for, against its ~34 TB/s
        acc, req = l1.get(k, {}).get("TCP_TOTAL_CACHE_ACCESSES_sum"), l1.get(k, {}).get("TCP_TCC_READ_REQ_sum")
        if acc and req is not None:
            e["l1_accesses"], e["l1_hit_rate"] = acc, 1.0 - req / acc
            if e.get("min_us"):
                e["l1_accesses_per_cycle_and_cu_at_min"] = acc / 256.0 / (e["min_us"] * 1e-6 * 2.4e9)   # the gathers' bound: ~0.6 tag look-ups per cycle and CU
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
