#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3 4; do python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29720+i)) tools/r6_gkdbg.py > /tmp/o.log 2>&1; grep "^leg" /tmp/o.log | awk '{print $0}' | tail -3; echo --; done
python -m pytest tests -m gpu -x -q -k "kink or sum or tie or sweep or shipped or two_asset or k_asset" 2>&1 | tail -4
