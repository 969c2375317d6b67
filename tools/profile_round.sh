# One measurement round on the GPU box: bench line, kernel trace, the two PMC passes.
#   gpurun -- 'bash tools/profile_round.sh r2_a <commit>'
# Every profiled run passes --no-dvfs: bench.py's clock / power sampling must not start anything
# while rocprofv3's preload is in the environment.
set -x
TAG=${1:-r2_a}
COMMIT=${2:-unrecorded}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 1500 gpurun_out/bench_$TAG.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dvfs --no-tolerance-mode --no-other-configs --no-from-host --no-strict-line > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dvfs --no-tolerance-mode --no-other-configs --no-from-host --no-strict-line > gpurun_out/pmc_fetch_$TAG.json 2> gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dvfs --no-tolerance-mode --no-other-configs --no-from-host --no-strict-line > gpurun_out/pmc_write_$TAG.log 2>&1
F=$(find gpurun_out/pmc_fetch_$TAG -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write_$TAG -name "*counter_collection.csv" | head -1)
S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
# the same two passes over BASELINE configs[3] (N-ImageNet @336: the row-band events kernel, the 480 x 640 -> 336
# preprocess, attention at S = 577), one quarter of its global batch on this one GPU
C3="--config 3 --batch 512 --steps 1 --warmup 1 --no-cpu-baseline --no-dvfs --no-strict-line"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_c3_$TAG -- python3 bench.py $C3 > gpurun_out/pmc_fetch_c3_$TAG.json 2> gpurun_out/pmc_fetch_c3_$TAG.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_c3_$TAG -- python3 bench.py $C3 > gpurun_out/pmc_write_c3_$TAG.log 2>&1
F3=$(find gpurun_out/pmc_fetch_c3_$TAG -name "*counter_collection.csv" | head -1)
W3=$(find gpurun_out/pmc_write_c3_$TAG -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W gpurun_out/pmc_fetch_$TAG.json $COMMIT --config 3 $F3 $W3 gpurun_out/pmc_fetch_c3_$TAG.json > gpurun_out/traffic_$TAG.json
cp $S gpurun_out/${TAG}_kernel_stats.csv
# the tolerance mode's own kernel table (eventclip_amd.clip.TOLERANCE_MODE on 16-bit-representable weights)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tol_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dvfs --no-from-host --no-strict-line --tolerance-mode --f16-weights > gpurun_out/prof_tol_$TAG.log 2>&1
cp $(find gpurun_out/prof_tol_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_tolerance_kernel_stats.csv
ls -la gpurun_out/*$TAG*
