"""Rebuild profiles/<tag>_* from the gpurun_out/ directories written by tools/r0N_profile.sh:
    python tools/refresh_profiles.py r04"""
import glob, json, os, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
run = lambda *a: subprocess.run(["python", *a], capture_output=True, text=True).stdout
print(run("tools/pmc_traffic.py", f"gpurun_out/{tag}_pmc_FETCH_SIZE", f"gpurun_out/{tag}_pmc_WRITE_SIZE", f"profiles/{tag}_pmc_traffic.json")[:1200])
B = "python3 bench.py --no-cpu-baseline --no-eager-reference --no-events" + (" --no-edm" if tag >= "r03" else "") + (" --no-small-batch" if tag >= "r06" else "")
open(f"profiles/{tag}_bench_kernel_stats_summary.txt", "w").write(
    f"# rocprofv3 --kernel-trace --stats summary, round {tag[1:]}\n"
    f"# command: rocprofv3 --kernel-trace --stats --output-format csv -- {B} --steps 5 --warmup 2 --train-steps 0   (1x MI355X)\n"
    + (f"# (7 generation steps of 256 images x T=10 in the trace: 2 warm-up + 5 timed; bench line of the same run in {tag}_bench_under_rocprof.json)\n\n" if tag < "r06" else
     f"# (9 generation steps of 256 images x T=10 in the trace: 2 warm-up + the eager first call and the capture call of the hipGraph, python-issued, + 5 timed REPLAYS;\n"
     f"#  bench line of the same run in {tag}_bench_under_rocprof.json)\n\n")
    + run("tools/kstats.py", f"gpurun_out/{tag}_prof_bench", "40"))
shutil.copy(glob.glob(f"gpurun_out/{tag}_prof_bench/**/*kernel_stats.csv", recursive=True)[0], f"profiles/{tag}_bench_kernel_stats.csv")
open(f"profiles/{tag}_bench_under_rocprof.json", "w").write(open(f"gpurun_out/{tag}_prof_bench.json").read().strip().split("\n")[-1] + "\n")
shutil.copy(f"gpurun_out/{tag}_bench_final.json", f"profiles/{tag}_bench_line.json")
open(f"profiles/{tag}_train_kernel_stats_summary.txt", "w").write(
    f"# rocprofv3 --kernel-trace --stats: {B} --steps 1 --warmup 1 --train-steps 3 "
    "(train leg: 1 warm-up + 3 timed DxMI steps, B=256, T=10; also 2 generation steps)\n" + run("tools/kstats.py", f"gpurun_out/{tag}_prof_train", "34"))
hdr = (f"# rocprofv3 --pmc (two separate passes, no trace domains) -- {B} --steps 2 --warmup 1 --train-steps 0\n"
       "# pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT\n"
       "# pass 2: SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE\n"
       "# units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles\n"
       "# (= 32 x SQ_INSTS_MFMA for 32x32x16 bf16) summed over SIMDs; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs.\n\n")
out = hdr
for d in (f"{tag}_pmc_sq1", f"{tag}_pmc_sq2"):
    keep = False
    for l in run("tools/pmc_summary.py", f"gpurun_out/{d}").split("\n"):
        if not l.startswith("    "):
            keep = any(k in l for k in ("conv_ws_kernel", "conv_ws8_kernel", "conv_sm_kernel", "conv_head_kernel", "conv_pipe_kernel<4, ", "conv1x1_rw_kernel", "conv1x1_stream_kernel", "conv_pipe_kernel<2, 6, 3", "gn_silu_kernel", "gn_apply_kernel", "attention_kernel", "attention256_kernel", "attn_block256_kernel"))
        if keep:
            out += l + "\n"
open(f"profiles/{tag}_pmc_sq_counters.txt", "w").write(out)
d = json.load(open(f"profiles/{tag}_bench_line.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], d["train_steps_per_sec"], r["achieved"], r["frac"], r["traffic"], r["algorithmic_bytes_per_launch"], r["avg_launch_us"], r["kernel"], d["cpu_baseline"])
print(d["reference_eager_gpu"])

# ---- round 3 extras: train-leg PMC passes, EDM kernel statistics
if os.path.isdir(f"gpurun_out/{tag}_pmc_train_FETCH_SIZE"):
    print(run("tools/pmc_traffic.py", f"gpurun_out/{tag}_pmc_train_FETCH_SIZE", f"gpurun_out/{tag}_pmc_train_WRITE_SIZE", f"profiles/{tag}_pmc_train_traffic.json")[:1500])
if os.path.isdir(f"gpurun_out/{tag}_pmc_train_sq1"):
    out = ("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT\n"
           f"# -- {B} --steps 1 --warmup 0 --train-steps 1 (train leg: backward kernels)\n\n")
    keep = False
    for l in run("tools/pmc_summary.py", f"gpurun_out/{tag}_pmc_train_sq1").split("\n"):
        if not l.startswith("    "):
            keep = any(k in l for k in ("conv_wgrad", "wgrad_reduce", "gn_silu_bwd", "gn_gen_bwd", "bgemm", "softmax_bwd"))
        if keep:
            out += l + "\n"
    open(f"profiles/{tag}_pmc_train_sq_counters.txt", "w").write(out)
for name, cmd in (("edm_in64", "tools/edm_bench.py imagenet64_T10 100"), ("edm_lsun", "tools/edm_bench.py lsun_bedroom_T4 16"),
                  ("edm_train", "tools/edm_train_bench.py imagenet64_T10 16 2")):
    d = f"gpurun_out/{tag}_prof_{name}"
    if os.path.isdir(d):
        head = open(f"gpurun_out/{tag}_prof_{name}.out").read().strip().split("\n")[-1]
        open(f"profiles/{tag}_{name}_kernel_stats_summary.txt", "w").write(
            f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 {cmd}   (1x MI355X; the run's own line, under the profiler:)\n# {head}\n\n"
            + run("tools/kstats.py", d, "36"))
if "edm" in d_line if (d_line := json.load(open(f"profiles/{tag}_bench_line.json"))) else False:
    print({k: (v.get("images_per_sec"), v.get("train_steps_per_sec")) for k, v in d_line["edm"].items()})
