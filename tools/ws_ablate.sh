# GPU box: timing-only ablations of conv_ws_kernel (stamp build: DXMI_CONV_WS_DBG bits 1 no weight stream, 2 no halo stream, 4 no drain, 8 no residual / table)
cd "$GRAFT_REPO_ROOT/diffusion-by-maxentirl_amd/csrc" && rm -f conv_ws.o && make STAMPS=1 -j8 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for d in 0 15 7; do echo "DBG=$d"; DXMI_CONV_WS_DBG=$d python tools/conv_ws_ab.py 2>&1 | grep "^N256" | cut -c1-110; done
