"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel name (mean per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", 0))):
    n = max(cnt[k].values())
    print(f"{k}  dispatches={n}")
    for c in sorted(acc[k]):
        print(f"    {c:32s} total {acc[k][c]:.4g}  per-dispatch {acc[k][c] / cnt[k][c]:.4g}")
