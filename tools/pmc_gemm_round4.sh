# MFMA-pipe busy share of the product GEMM kernel on the tower's four launches (full bench size), rocprofv3 counters.
#   gpurun -- 'bash tools/pmc_gemm_round4.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_r4
for spec in "qkv 3072 1024 store16_ln" "out_proj 1024 1024 resid_hl" "c_fc 4096 1024 gelu16_ln" "c_proj 1024 4096 resid_hl"; do
  set -- $spec
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_r4/$1 -- python3 tools/pmc_gemm.py --variant 0 --n $2 --k $3 --frames 2560 --epi $4 --iters 3 > gpurun_out/pmc_r4/$1.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for name in ('qkv', 'out_proj', 'c_fc', 'c_proj'):
    f = glob.glob(f'gpurun_out/pmc_r4/{name}/**/*counter_collection.csv', recursive=True)
    if not f:
        print(name, 'no counters'); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'gemm2pp' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    gui = m.get('GRBM_GUI_ACTIVE', 0) / 8
    busy = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024
    print(f'{name:9s} GRBM_GUI_ACTIVE / 8 = {gui:10.0f} cycles per launch; SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs = {busy:10.0f}; MFMA pipe busy {100 * busy / max(gui, 1):5.1f} %; launches {len(acc.get("GRBM_GUI_ACTIVE", []))}')
PY
