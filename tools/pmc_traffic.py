"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes: TCC has 4
slots, FETCH_SIZE costs 3 and WRITE_SIZE 2) into profiles/<tag>_pmc_traffic.json: per kernel, mean KB per dispatch and
the corrected HBM-side bytes per launch = 2 * FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE.
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_traffic.json"""
import csv, glob, json, sys
from collections import defaultdict


def per_kernel(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-conv-events --train-steps 0; unit KB per dispatch (mean); "
                  "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    if k in write:
        out["kernels"][k] = {"dispatches": fetch[k][1], "fetch_size_kb": round(fetch[k][0], 1), "write_size_kb": round(write[k][0], 1),
                             "hbm_bytes_per_launch": int((2 * fetch[k][0] + write[k][0]) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k[:60]:60s} n={v['dispatches']:5d} fetch {v['fetch_size_kb']/1e3:8.1f} MB  write {v['write_size_kb']/1e3:8.1f} MB  hbm/launch {v['hbm_bytes_per_launch']/1e6:8.1f} MB")
