set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r1_e.json 2> gpurun_out/bench_r1_e.err
tail -c 600 gpurun_out/bench_r1_e.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1_e -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1_e.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_d -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch_d.json 2> gpurun_out/pmc_fetch_d.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_d -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write_d.log 2>&1
find gpurun_out/prof_r1_e gpurun_out/pmc_fetch_d gpurun_out/pmc_write_d -name "*.csv" | head -20
