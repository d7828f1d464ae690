"""Rebuild profiles/ from the gpurun_out/ directories of one measurement pass:
    python tools/refresh_profiles.py <suffix>     (prof_bench<s>, pmc<s>_FETCH_SIZE/_WRITE_SIZE, prof_train<s>, prof_edm<s>, pmc_sq<a>,<b>)"""
import glob, json, shutil, subprocess, sys
s, sqa, sqb = sys.argv[1], sys.argv[2], sys.argv[3]
run = lambda *a: subprocess.run(["python", *a], capture_output=True, text=True).stdout
print(run("tools/pmc_traffic.py", f"gpurun_out/pmc{s}_FETCH_SIZE", f"gpurun_out/pmc{s}_WRITE_SIZE", "profiles/r01_pmc_traffic.json")[:400])
open("profiles/r01_bench_kernel_stats_summary.txt", "w").write(
    "# rocprofv3 --kernel-trace --stats summary, round 1 (final kernels)\n"
    "# command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0   (1x MI355X)\n"
    "# (7 generation steps of 256 images x T=10 in the trace: 2 warm-up + 5 timed; bench line of the same run in r01_bench_under_rocprof.json)\n\n"
    + run("tools/kstats.py", f"gpurun_out/prof_bench{s}", "40"))
shutil.copy(glob.glob(f"gpurun_out/prof_bench{s}/**/*kernel_stats.csv", recursive=True)[0], "profiles/r01_bench_kernel_stats.csv")
open("profiles/r01_bench_under_rocprof.json", "w").write(open(f"gpurun_out/prof_bench{s}.json").read().strip().split("\n")[-1] + "\n")
shutil.copy("gpurun_out/bench_final.json", "profiles/r01_bench_line.json")
edm = [l for l in open(f"gpurun_out/prof_edm{s}.log") if l.startswith("imagenet64")][0].strip()
open("profiles/r01_edm_imagenet64_kernel_stats_summary.txt", "w").write(
    f"# rocprofv3 --kernel-trace --stats: python3 tools/edm_bench.py imagenet64_T10 100 (ImageNet-64 EDM backbone, 1x MI355X; 4 sample() calls)\n# {edm}\n"
    + run("tools/kstats.py", f"gpurun_out/prof_edm{s}", "16"))
open("profiles/r01_train_kernel_stats_summary.txt", "w").write(
    "# rocprofv3 --kernel-trace --stats: python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-conv-events --train-steps 3 "
    "(train leg: 1 warm-up + 3 timed DxMI steps, B=256, T=10; also 2 generation steps)\n" + run("tools/kstats.py", f"gpurun_out/prof_train{s}", "30"))
hdr = open("profiles/r01_pmc_sq_counters.txt").read().split("\n\n")[0] + "\n\n"
out = hdr
for d in (sqa, sqb):
    keep = False
    for l in run("tools/pmc_summary.py", f"gpurun_out/{d}").split("\n"):
        if not l.startswith("    "):
            keep = any(k in l for k in ("conv_pipe_kernel<4, ", "conv1x1_stream_kernel", "conv_pipe_kernel<2, 6, 3", "gn_silu_kernel<4, 32>", "attention_kernel"))
        if keep:
            out += l + "\n"
open("profiles/r01_pmc_sq_counters.txt", "w").write(out)
d = json.load(open("profiles/r01_bench_line.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], d["train_steps_per_sec"], r["achieved"], r["frac"], r["traffic"], r["algorithmic_bytes_per_launch"], r["avg_launch_us"], r["kernel"], d["cpu_baseline"]["value"])
