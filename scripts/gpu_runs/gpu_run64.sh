one() { python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; }
echo "fold up to 256 rows (default):"; one
echo "fold up to 2048 rows:"; HOTRACK_LN_LINEAR_MAX_ROWS=2048 one
echo "default:"; one
for b in 16 32; do for r in 256 2048; do echo "B=$b rows<=$r: $(HOTRACK_LN_LINEAR_MAX_ROWS=$r python bench.py --no-cpu-baseline --no-legs --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])")"; done; done
