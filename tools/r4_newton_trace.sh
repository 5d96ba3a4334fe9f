#!/bin/bash
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/ntr; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $R/tools/profile_newton.py --solves 3 > $O/log 2>&1; echo rc=$?
tail -1 $O/log | cut -c1-400
