mkdir -p gpurun_out
for s in 3 4 5 6 7 8; do
  PLBA_CHAIN_SEG=$s PLBA_PREP_TIMING=1 timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config5-leg > gpurun_out/b3_seg$s.log 2>&1 || exit 1
  python - <<PY
import json
L=open('gpurun_out/b3_seg$s.log').read().splitlines()
d=json.loads(L[-1]); ph=d['phase_ms_per_iteration']
plan=[l for l in L if 'dense system' in l][0:1]+[l for l in L if 'dependent launches' in l][0:1]
print($s, round(d['value'],1), round(d['ms_per_step'],4), 'dense', d['config']['dense_dim'], 'lin', round(ph['linearize_launch']*1e3,1), 'fact', round(ph['factorisation_launches']*1e3,1), 'solve', round(ph['dense_solve']*1e3,1), 'trial', round(ph['trial_errors']*1e3,1), plan[-1][10:] if plan else '')
PY
done
