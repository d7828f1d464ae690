# GPU box: kernel-trace statistics of a short generation-only bench run (top kernels, per-launch averages)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/qprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qprof -- python3 bench.py --no-cpu-baseline --no-eager-reference --no-events --no-edm --steps ${3:-5} --warmup ${4:-2} --train-steps ${1:-0} > gpurun_out/qprof.json 2> gpurun_out/qprof.err
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/qprof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:int("${2:-34}")]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
find gpurun_out/qprof -name "*kernel_trace.csv" -size +20M -delete
