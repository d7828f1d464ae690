# Round-3: kernel trace of the EDM (ImageNet-64) DxMI train step
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/r03_prof_edm_train
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof_edm_train -- python3 tools/edm_train_bench.py imagenet64_T10 16 2 > gpurun_out/r03_prof_edm_train.out 2> gpurun_out/r03_prof_edm_train.err
cat gpurun_out/r03_prof_edm_train.out
python3 tools/kstats.py gpurun_out/r03_prof_edm_train 60
