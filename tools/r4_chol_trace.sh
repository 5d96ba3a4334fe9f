#!/bin/bash
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/ctr; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $R/tools/profile_newton.py --solves 2 > $O/log 2>&1; echo rc=$?
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ctr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))) for r in csv.DictReader(open(f)))
ch = [(b, e, g) for b, e, k, g in rows if 'chol_step' in k]
last = ch[-17:] if any('step2' in k for _, _, k, _ in rows) else ch[-32:]
print('launch durations (us) / workgroups:', ' '.join('%.1f/%d' % ((e - b) / 1e3, g) for b, e, g in last))
print('sum %.1f' % sum((e - b) / 1e3 for b, e, g in last))
PY
