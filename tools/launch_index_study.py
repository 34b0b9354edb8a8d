#!/usr/bin/env python3
"""Launch time of the metric's kernel by its position inside a step and by step.

usage:  tools/launch_index_study.py <kernel_trace.csv> [--kernel k_jacobi_strip4o] [--per-step 10] [--out summary.json]

Input: the raw `rocprofv3 --kernel-trace --output-format csv` trace of a bench.py run (one row per dispatch with its start / end
timestamps).  The launches of the kernel are taken in dispatch order and cut into steps of `--per-step` launches (the default schedule
of 256^3 / 40 sweeps: ten launches of four sweeps per step); reported: mean / min / max duration per index within the step, the gap
to the previous dispatch (what ran in front of it and for how long), and the per-step mean of the first and of the other launches.
"""
import argparse, csv, json, re, statistics as st, sys


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("fx::", "")
    return re.sub(r"[(<].*", "", n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--kernel", default="k_jacobi_strip4o")
    ap.add_argument("--per-step", type=int, default=10)
    ap.add_argument("--skip-steps", type=int, default=0, help="steps at the head of the trace to leave out (warm-up, scratch contexts)")
    ap.add_argument("--out")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    idx = [i for i, e in enumerate(ev) if e[0] == a.kernel]
    if not idx:
        print("no launches of", a.kernel, "in", a.trace); sys.exit(1)
    steps = [idx[i:i + a.per_step] for i in range(0, len(idx) - a.per_step + 1, a.per_step)][a.skip_steps:]
    by_pos = [[] for _ in range(a.per_step)]
    gap_pos = [[] for _ in range(a.per_step)]
    prev_pos = [dict() for _ in range(a.per_step)]
    first_of_step, rest_of_step = [], []
    for s in steps:
        d = [(ev[i][2] - ev[i][1]) / 1e3 for i in s]
        for k, i in enumerate(s):
            by_pos[k].append(d[k])
            if i > 0:
                gap_pos[k].append((ev[i][1] - ev[i - 1][2]) / 1e3)
                prev_pos[k][ev[i - 1][0]] = prev_pos[k].get(ev[i - 1][0], 0) + 1
        first_of_step.append(d[0]); rest_of_step.append(st.mean(d[1:]) if len(d) > 1 else d[0])
    alld = [x for p in by_pos for x in p]
    out = {
        "kernel": a.kernel, "launches": len(alld), "steps": len(steps), "per_step": a.per_step,
        "all_us": {"mean": st.mean(alld), "median": st.median(alld), "stdev": st.pstdev(alld), "min": min(alld), "max": max(alld)},
        "by_index_in_step_us": [{"index": k + 1, "mean": st.mean(p), "median": st.median(p), "min": min(p), "max": max(p), "stdev": st.pstdev(p),
                                 "gap_before_us_mean": st.mean(g) if g else None, "launch_in_front": max(pp, key=pp.get) if pp else None}
                                for k, (p, g, pp) in enumerate(zip(by_pos, gap_pos, prev_pos))],
        "first_launch_of_step_us": {"mean": st.mean(first_of_step), "min": min(first_of_step), "max": max(first_of_step)},
        "other_launches_of_step_us": {"mean": st.mean(rest_of_step), "min": min(rest_of_step), "max": max(rest_of_step)},
        "per_step_mean_us_deciles": [sorted(st.mean([(ev[i][2] - ev[i][1]) / 1e3 for i in s]) for s in steps)[int(q * (len(steps) - 1) / 10)] for q in range(11)],
    }
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print("%s: %d launches in %d steps; mean %.2f us, median %.2f, sigma %.2f, min %.2f, max %.2f" % (
        a.kernel, len(alld), len(steps), out["all_us"]["mean"], out["all_us"]["median"], out["all_us"]["stdev"], out["all_us"]["min"], out["all_us"]["max"]))
    print("index  mean    median  min     max     sigma   gap-before  in front")
    for r in out["by_index_in_step_us"]:
        print("%5d  %6.2f  %6.2f  %6.2f  %6.2f  %6.2f  %8.2f    %s" % (r["index"], r["mean"], r["median"], r["min"], r["max"], r["stdev"], r["gap_before_us_mean"] or 0, r["launch_in_front"]))


if __name__ == "__main__":
    main()
