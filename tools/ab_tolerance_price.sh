# Round 6: error (eight draws, configs[1] and [4]) and price (bench config, interleaved with the default model) of tolerance-mode
# settings 'blocks:attention blocks' with e4m3 (1) / 16-bit (0) lo products -> profiles/r6_tolerance_sweep_fp8.txt, r6_tolerance_price.txt
#   gpurun -- 'bash tools/ab_tolerance_price.sh'
python tools/sweep_tolerance.py --seeds 8 --configs 1,4 12:10 14:8 14:10 16:8 > gpurun_out/r6_seeds_fp8_b.txt 2>&1; tail -10 gpurun_out/r6_seeds_fp8_b.txt
for setting in "12:8 1" "12:8 0" "14:8 1" "12:10 1" "8:5 1" "8:5 0"; do set -- $setting
EVENTCLIP_TOLERANCE_MODE=$1 EVENTCLIP_LO_FP8=$2 python bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --no-from-host --no-strict-line --no-dvfs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['tolerance_mode']; print('$1 fp8=$2', 'default', round(d['ms_per_step'],1), [(l['weights'], round(l['ms_per_step'],1), round(l['ratio_to_default'],3)) for l in t['lines']])"
done
