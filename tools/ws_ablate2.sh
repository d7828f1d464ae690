# GPU box: timing-only ablations of conv_ws_kernel through the stamp VARIANT library (tools/build_variant.sh stamps conv_ws.hip -DDXMI_CONV_STAMPS):
# DXMI_CONV_WS_DBG bits 1 no weight stream, 2 no halo stream, 4 no drain, 8 no residual / table, 16 no step barriers (wrong results)
cd "$GRAFT_REPO_ROOT"
export DXMI_LIB=$PWD/diffusion-by-maxentirl_amd/dxmi_hip/libdxmi_stamps.so
for d in 0 1 2 4 8 15 31; do echo "DBG=$d"; DXMI_CONV_WS_DBG=$d python tools/conv_ws_ab.py 2>&1 | grep "^N256" | awk '{print $2,$3,$4,$5,$(NF-3),$(NF-2)}'; done
for s in "128 128 32 1" "128 128 32 0" "512 256 16 0" "256 256 16 1"; do python tools/ws_stamps.py $s; done
