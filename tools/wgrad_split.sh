cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/wgsplit
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wgsplit -- python3 tools/wgrad_split.py > /dev/null 2> gpurun_out/wgsplit.err
python3 tools/wgrad_split_parse.py gpurun_out/wgsplit
rm -rf gpurun_out/wgsplit
