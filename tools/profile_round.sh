set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r1_f.json 2> gpurun_out/bench_r1_f.err
tail -c 600 gpurun_out/bench_r1_f.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1_f.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch_f.json 2> gpurun_out/pmc_fetch_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write_f.log 2>&1
find gpurun_out/prof_r1_f gpurun_out/pmc_fetch_f gpurun_out/pmc_write_f -name "*.csv" | head -20
