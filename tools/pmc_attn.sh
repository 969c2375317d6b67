#!/bin/bash
# SQ counters of the attention kernel, one rocprofv3 pass per counter group and variant (diagnostic build).
#   bash tools/pmc_attn.sh [S]      -> gpurun_out/pmc_attn_S<S>.txt
S=${1:-257}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_attn_S$S.txt
: > $OUT
for V in 0 1; do
  for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
    D=/tmp/pmc_${V}_$RANDOM
    rocprofv3 --kernel-trace --pmc $G -d $D --output-format csv -- python3 $R/tools/pmc_attn.py $S $V > /dev/null 2>&1
    F=$(find $D -name "*counter_collection.csv" | head -1)
    echo "variant $V: $G" >> $OUT
    python3 - "$F" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'attention_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'  {k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
  done
done
cat $OUT
