# BASELINE configs[2..4] at their FULL batch on one GPU, then each of them in the tolerance mode next to the default path on the same
# 16-bit checkpoint (profiles/r6_configs_bench.jsonl)
for c in 2 3 4; do
  python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-dvfs --no-strict-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('metric','value','ms_per_step','n_gpus','steps','config','roofline','kernel_ms_per_step')}))"
done
for c in 3 2 4; do
for mode in "--tolerance-mode" ""; do
  python bench.py --config $c $mode --f16-weights --steps 2 --warmup 1 --no-cpu-baseline --no-dvfs --no-strict-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('metric','value','ms_per_step','n_gpus','steps','config','roofline','kernel_ms_per_step')}))"
done
done
