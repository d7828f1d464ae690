cd "$GRAFT_REPO_ROOT"
for nt in 0 1 2 3 0; do echo "NT=$nt"; DXMI_GN_APPLY_NT=$nt python bench.py --no-cpu-baseline --no-eager-reference --no-edm --steps 10 --train-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['roofline_classes']['groupnorm']
print('   img/s %.0f  step %.2f ms  GN class %.3f ms frac %.3f  conv3x3 %.3f' % (d['value'], d['ms_per_step'], g['ms_per_step'], g['frac'], d['roofline_classes']['conv3x3']['ms_per_step']))
"; done
