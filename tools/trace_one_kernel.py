"""Durations of one kernel in a rocprofv3 kernel-trace CSV, grouped by grid: python tools/trace_one_kernel.py <dir> <name substring>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
bins = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        bins[(r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append(us)
for g, v in sorted(bins.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"grid {g}: n={len(v)} total {sum(v) / 1e3:.2f} ms  median {v[len(v) // 2]:.1f} us  max {v[-1]:.1f} us")
