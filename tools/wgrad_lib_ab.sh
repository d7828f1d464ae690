# GPU box: tools/wgrad_split.py under several library builds (DXMI_LIB=libdxmi_<name>.so), 3x3 shapes only
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
D=$PWD/diffusion-by-maxentirl_amd/dxmi_hip
for l in "$@"; do
  echo "== $l"
  rm -rf gpurun_out/wgsplit
  DXMI_LIB=$D/libdxmi_$l.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wgsplit -- python3 tools/wgrad_split.py > /dev/null 2> gpurun_out/wgsplit.err
  python3 tools/wgrad_split_parse.py gpurun_out/wgsplit | grep k3
done
rm -rf gpurun_out/wgsplit
