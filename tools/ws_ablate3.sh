# GPU box: timing-only ablations of the round-4 conv_ws_kernel through the stamp variant (tools/build_variant.sh g3stamps conv_ws.hip -DDXMI_CONV_STAMPS):
# DXMI_CONV_WS_DBG bits 2 no halo stream, 4 no drain, 8 no residual / table, 16 no group barriers (wrong results)
cd "$GRAFT_REPO_ROOT"
export DXMI_LIB=$PWD/diffusion-by-maxentirl_amd/dxmi_hip/libdxmi_g3stamps.so
for d in 0 2 4 8 14 30; do echo "DBG=$d"; DXMI_CONV_WS_DBG=$d python tools/conv_ws_ab.py 2>&1 | grep "^N256" | awk '{print $2,$3,$4,$5,$(NF-3),$(NF-2)}'; done
