time python bench.py > gpurun_out/b.log 2>&1; grep '^{' gpurun_out/b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['dtype'])
for k in ('parity_mode','exact_mode','other_fast_mode'):
    m=d.get(k); print(k, m and (m['dtype'], m['value'], m['ms_per_step'], (m.get('roofline') or {}).get('frac'), (m.get('roofline') or {}).get('achieved')))
print(d.get('cpu_baseline'))
"
tail -3 gpurun_out/b.log | cut -c1-200
