# Round-6 profile passes (the --pmc passes issue every launch from python, DXMI_GRAPH=0: counters are collected per dispatch) (run on the GPU box through gpurun; writes under gpurun_out/).  rocprofv3 is given the program
# itself (python3 ...), counters in their own passes without trace domains, as MI355X_MICROARCH.md prescribes.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/r06_prof_* gpurun_out/r06_pmc_*
B="python3 bench.py --no-cpu-baseline --no-eager-reference --no-events --no-edm --no-small-batch"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_prof_bench -- $B --steps 5 --warmup 2 --train-steps 0 > gpurun_out/r06_prof_bench.json 2> gpurun_out/r06_prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_prof_train -- $B --steps 1 --warmup 1 --train-steps 3 > gpurun_out/r06_prof_train.json 2> gpurun_out/r06_prof_train.err
DXMI_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r06_pmc_FETCH_SIZE -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r06_pmc_f.err
DXMI_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r06_pmc_WRITE_SIZE -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r06_pmc_w.err
# train leg: HBM traffic of the backward kernels (wgrad, GroupNorm backward ...)
DXMI_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r06_pmc_train_FETCH_SIZE -- $B --steps 1 --warmup 0 --train-steps 1 > /dev/null 2> gpurun_out/r06_pmc_tf.err
DXMI_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r06_pmc_train_WRITE_SIZE -- $B --steps 1 --warmup 0 --train-steps 1 > /dev/null 2> gpurun_out/r06_pmc_tw.err
DXMI_GRAPH=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r06_pmc_sq1 -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r06_pmc_s1.err
DXMI_GRAPH=0 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r06_pmc_sq2 -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r06_pmc_s2.err
# train leg SQ pass (MFMA busy of the wgrad kernel)
DXMI_GRAPH=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r06_pmc_train_sq1 -- $B --steps 1 --warmup 0 --train-steps 1 > /dev/null 2> gpurun_out/r06_pmc_ts1.err
# EDM: ImageNet-64 generation, LSUN generation, ImageNet-64 train step
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_prof_edm_in64 -- python3 tools/edm_bench.py imagenet64_T10 100 > gpurun_out/r06_prof_edm_in64.out 2> gpurun_out/r06_prof_edm_in64.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_prof_edm_lsun -- python3 tools/edm_bench.py lsun_bedroom_T4 16 > gpurun_out/r06_prof_edm_lsun.out 2> gpurun_out/r06_prof_edm_lsun.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_prof_edm_train -- python3 tools/edm_train_bench.py imagenet64_T10 16 2 > gpurun_out/r06_prof_edm_train.out 2> gpurun_out/r06_prof_edm_train.err
# the hipGraph-replayed steps at the reference's per-rank batch of an 8-GPU run (kernel time vs wall time: DESIGN 5.6)
for m in gen train; do
  MODE=$m B=32 N=10 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_graph_${m}_b32 -- python3 tools/graph_profile.py > gpurun_out/r06_graph_${m}_b32.out 2> gpurun_out/r06_graph_${m}_b32.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_prof_edm_train_graph -- python3 tools/edm_train_bench.py imagenet64_T10 16 2 graph > gpurun_out/r06_prof_edm_train_graph.out 2> gpurun_out/r06_prof_edm_train_graph.err
python3 tools/graph_step_bench.py > gpurun_out/r06_graph_step_bench.txt 2>&1
python3 tools/gn_apply_ceiling.py > gpurun_out/r06_gn_apply_ceiling.txt 2>&1
python3 tools/graph_node_cost.py > gpurun_out/r06_graph_node_cost.txt 2>&1
python3 tools/td_fused_ab.py > gpurun_out/r06_td_fused_ab.txt 2>&1
python3 tools/small_batch_routing.py > gpurun_out/r06_small_batch_routing.txt 2>&1
python3 bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
# summaries are made HERE (the raw traces / counter dumps exceed what gpurun merges back): profiles/r06_* -> gpurun_out/r06_profiles/
python3 tools/refresh_profiles.py r06 > gpurun_out/r06_refresh.log 2>&1
mkdir -p gpurun_out/r06_profiles && cp profiles/r06_* gpurun_out/r06_profiles/
for f in gpurun_out/r06_graph_*_b32 gpurun_out/r06_prof_edm_train_graph; do python3 tools/kstats.py $f 45 > gpurun_out/r06_profiles/$(basename $f)_kernel_stats_summary.txt 2>&1; done
cp gpurun_out/r06_graph_*.out gpurun_out/r06_prof_edm_train_graph.out gpurun_out/r06_*.txt gpurun_out/r06_profiles/ 2>/dev/null
tail -3 gpurun_out/*.err | tail -40
rm -rf gpurun_out/r06_prof_* gpurun_out/r06_pmc_* gpurun_out/r06_graph_gen_b32 gpurun_out/r06_graph_train_b32
ls gpurun_out | grep r06 | head -40; du -sh gpurun_out
