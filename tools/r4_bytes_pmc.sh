#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/bpmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in "CFMM_COMPACT=-1" "CFMM_COMPACT=0"; do for cfg in C4 C4x4; do
  env $c timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${cfg}_${c#*=} -o c -- python $R/tools/profile_eval.py --config $cfg > $O/log_${cfg}_${c#*=} 2>&1
done; done
cd $R
python - <<'PY'
import csv, glob, statistics
for d in sorted(glob.glob('gpurun_out/bpmc/C4*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'eval_kernel' in r['Kernel_Name']]
        ns = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'eval_kernel' in r['Kernel_Name']]
        print(d.split('/')[-1], 'FETCH_SIZE KB median', statistics.median(v), '-> MB moved (x2 gfx950)', 2 * statistics.median(v) / 1024, 'dispatch us', statistics.median(ns) / 1e3)
PY
