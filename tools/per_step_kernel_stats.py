#!/usr/bin/env python3
"""Per-STEP kernel statistics from two `rocprofv3 --kernel-trace --stats` runs of bench.py that differ only in --steps: the difference of the two
tables divided by the difference in steps removes everything that runs once (model build: one cast / copy kernel per parameter tensor, caption
preparation, warm-up, capture).      python tools/per_step_kernel_stats.py stats_a.csv steps_a stats_b.csv steps_b [out.csv]"""
import csv
import sys


def load(path):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(path))}


def main():
    a, na, b, nb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    d = nb - na
    rows = []
    for name in set(a) | set(b):
        ca, ta = a.get(name, (0, 0.0))
        cb, tb = b.get(name, (0, 0.0))
        if cb - ca or tb - ta:
            rows.append((name, (cb - ca) / d, (tb - ta) / d / 1e3))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    fam = {"hipBLASLt": 0.0, "at::native": 0.0, "rocclr": 0.0, "hand-written": 0.0}
    calls = dict.fromkeys(fam, 0.0)
    for name, c, us in rows:
        k = "hipBLASLt" if name.startswith("Cijk") else "rocclr" if "rocclr" in name else "at::native" if "at::" in name or "anonymous namespace" in name else "hand-written"
        fam[k] += us
        calls[k] += c
    print(f"per step: {tot / 1e3:.3f} ms of kernels in {sum(calls.values()):.0f} launches")
    for k in fam:
        print(f"  {k:13s} {fam[k] / 1e3:7.3f} ms  {100 * fam[k] / tot:5.1f} %  {calls[k]:6.1f} launches")
    if len(sys.argv) > 5:
        with open(sys.argv[5], "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "CallsPerStep", "UsPerStep", "Percentage"])
            for name, c, us in rows:
                w.writerow([name, round(c, 2), round(us, 2), round(100 * us / tot, 2)])
    for name, c, us in rows[:12]:
        print(f"  {us:8.1f} us {c:6.1f} x  {name[:100]}")


if __name__ == "__main__":
    main()
