python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head
python bench.py > gpurun_out/r3_head_bench.log 2>&1; grep '^{' gpurun_out/r3_head_bench.log > gpurun_out/r3_final_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_final_bench.json'))
print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])
for k in ('parity_mode','exact_mode','other_fast_mode'):
    m=d.get(k); print(k, m and (m['dtype'], m['value'], m['ms_per_step'], (m.get('roofline') or {}).get('frac'), (m.get('roofline') or {}).get('achieved')))
PY
