# Round-3: kernel trace of the generation bench with the streaming GroupNorm path (per-kernel in-situ times)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-eager-reference --no-events"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof_gn -- $B --steps 5 --warmup 2 --train-steps 0 > gpurun_out/r03_prof_gn.json 2> gpurun_out/r03_prof_gn.err
python3 tools/kstats.py gpurun_out/r03_prof_gn 30
