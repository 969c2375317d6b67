#!/bin/bash
# SQ / TA counters of the preprocess kernel, one rocprofv3 pass per counter group.
#   bash tools/pmc_preprocess.sh [caltech|imagenet]      -> gpurun_out/pmc_preprocess_<geometry>.txt
GEO=${1:-caltech}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_preprocess_$GEO.txt
: > $OUT
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  D=/tmp/pmcp_$RANDOM
  rocprofv3 --kernel-trace --pmc $G -d $D --output-format csv -- python3 $R/tools/pmc_preprocess.py $GEO > /dev/null 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  echo "$G" >> $OUT
  python3 - "$F" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
if sys.argv[1]:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'preprocess_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'  {k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
done
cat $OUT
