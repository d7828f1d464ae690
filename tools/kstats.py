"""Print a rocprofv3 kernel_stats.csv as a compact table: python tools/kstats.py <dir-or-csv> [rows]"""
import csv, glob, sys
src = sys.argv[1]
f = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':68s} {'calls':>6s} {'total ms':>9s} {'avg us':>8s} {'%':>5s}")
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:68]
    print(f"{name:68s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}")
print(f"total kernel time {tot/1e6:.1f} ms")
