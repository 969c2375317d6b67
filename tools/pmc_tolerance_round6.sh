# Round 6: HBM traffic of the tolerance mode's kernels (profiles/r6_tolerance_traffic.json)
#   gpurun -- 'bash tools/pmc_tolerance_round6.sh'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="bench.py --tolerance-mode --f16-weights --steps 1 --warmup 1 --no-cpu-baseline --no-dvfs --no-from-host --no-strict-line"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_tol -- python3 $B > gpurun_out/pmc_fetch_tol.json 2> gpurun_out/pmc_fetch_tol.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_tol -- python3 $B > gpurun_out/pmc_write_tol.log 2>&1
F=$(find gpurun_out/pmc_fetch_tol -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write_tol -name "*counter_collection.csv" | head -1)
python tools/tolerance_traffic.py $F $W > gpurun_out/r6_tolerance_traffic.json
