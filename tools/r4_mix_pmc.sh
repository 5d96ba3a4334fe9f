#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/mix; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
for kind in gn cp2; do
  i=0
  for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $pmc --output-format csv -d $O/${kind}_$i -o c -- python $R/tools/r4_mix_one.py $kind > $O/${kind}_$i.log 2>&1; echo "$kind $i rc=$?"
  done
done
cd $R
python - <<'PY'
import csv, glob, statistics
for kind in ("gn", "cp2"):
    agg = {}
    for f in glob.glob(f'gpurun_out/mix/{kind}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'eval_kernel' in r['Kernel_Name']: agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print(kind, {k: statistics.median(v) for k, v in sorted(agg.items())})
PY
