#!/bin/bash
# PMC passes over the block Gram kernel (p = 512): HBM traffic, L2 hit rate, MFMA busy
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/pmcblk
mkdir -p $o
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES SQ_WAIT_INST_ANY"; do
  d=$o/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o blk -- python3 tools/gram_diag.py 1000000 512 > /dev/null 2> $d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gram_blk" in r["Kernel_Name"] or "gram_sb" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "mean per dispatch", sum(v) / len(v), "dispatches", len(v))
PY
done
