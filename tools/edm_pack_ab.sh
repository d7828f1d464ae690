cd "$GRAFT_REPO_ROOT"
for v in 0 1 0 1; do DXMI_PACK_BATCH=$v python tools/edm_train_bench.py imagenet64_T10 16 3 2>&1 | tail -1 | cut -c1-150; done
