# Round-2 profile passes (run on the GPU box through gpurun; writes under gpurun_out/).  rocprofv3 is given the program
# itself (python3 ...), counters in their own passes without trace domains, as MI355X_MICROARCH.md prescribes.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-eager-reference --no-events"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof_bench -- $B --steps 5 --warmup 2 --train-steps 0 > gpurun_out/r02_prof_bench.json 2> gpurun_out/r02_prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof_train -- $B --steps 1 --warmup 1 --train-steps 3 > gpurun_out/r02_prof_train.json 2> gpurun_out/r02_prof_train.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_pmc_FETCH_SIZE -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r02_pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_pmc_WRITE_SIZE -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r02_pmc_w.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r02_pmc_sq1 -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r02_pmc_s1.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r02_pmc_sq2 -- $B --steps 2 --warmup 1 --train-steps 0 > /dev/null 2> gpurun_out/r02_pmc_s2.err
python3 bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
ls gpurun_out | head -30; du -sh gpurun_out
