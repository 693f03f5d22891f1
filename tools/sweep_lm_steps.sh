mkdir -p gpurun_out
for s in 8 9 10 12 16; do
  PLBA_LM_STEPS=$s timeout -k 10 200 python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b5_s$s.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/b5_s$s.log').read().strip().splitlines()[-1])
ph=d['phase_ms_per_iteration']
print($s, round(d['value'],1), round(d['ms_per_step'],4), 'schur', round(ph['linearize_launch'],4), 'trial', round(ph['backsub_update'],4))
PY
done
