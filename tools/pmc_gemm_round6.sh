# MFMA-pipe busy share of the GEMM kernel with a lo product as 16-bit and as e4m3 operands (round 6), rocprofv3 counters, on the
# QKV and c_proj shapes at the bench size.   gpurun -- 'bash tools/pmc_gemm_round6.sh > gpurun_out/r6_gemm_pmc.txt'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_r6
for spec in "qkv 3072 1024 store16" "c_proj 1024 4096 resid_hl"; do
  set -- $spec
  for lo in none f16 e4m3; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_r6/$1_$lo -- python3 tools/pmc_gemm.py --variant 0 --n $2 --k $3 --frames 2560 --epi $4 --iters 3 --lo $lo > gpurun_out/pmc_r6/$1_$lo.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
print('# bash tools/pmc_gemm_round6.sh   (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES over tools/pmc_gemm.py')
print('#   --variant 0 --frames 2560 --lo none | f16 | e4m3: the product kernel, plain / with the lo product A_lo W^T as 16-bit / as e4m3 operands; M = 657 920, 3 launches each)')
for name in ('qkv', 'c_proj'):
    for lo in ('none', 'f16', 'e4m3'):
        f = glob.glob(f'gpurun_out/pmc_r6/{name}_{lo}/**/*counter_collection.csv', recursive=True)
        if not f:
            print(name, lo, 'no counters'); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if 'gemm2pp' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        m = {k: sum(v) / len(v) for k, v in acc.items()}
        gui = m.get('GRBM_GUI_ACTIVE', 0) / 8
        busy = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024
        print(f'{name:7s} lo = {lo:5s} GRBM_GUI_ACTIVE / 8 = {gui:10.0f} cycles per launch; SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs = {busy:10.0f}; MFMA pipe busy {100 * busy / max(gui, 1):5.1f} %; launches {len(acc.get("GRBM_GUI_ACTIVE", []))}')
PY
