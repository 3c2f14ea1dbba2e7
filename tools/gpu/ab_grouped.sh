for i in 1 2; do
for v in 0 1; do
CSM_GROUPED_VALU=$v python bench.py --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('valu=$v', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
done
