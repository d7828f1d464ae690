cd "$GRAFT_REPO_ROOT"
for cfg in "4 2" "4 1" "2 2" "2 1" "4 4" "8 1" "4 2"; do set -- $cfg; echo "U=$1 trips=$2"; DXMI_GN_APPLY_U=$1 DXMI_GN_APPLY_TRIPS=$2 python bench.py --no-cpu-baseline --no-eager-reference --no-edm --steps 10 --train-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['roofline_classes']['groupnorm']
print('   img/s %.0f  step %.2f ms  GN class %.3f ms frac %.3f' % (d['value'], d['ms_per_step'], g['ms_per_step'], g['frac']))
"; done
