#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02_probe3
mkdir -p $O
for dbg in 0 1 6; do
  VDL2GPU_K1_DBG=$dbg python bench.py --no-cpu --no-ring --no-parity --steps 6 --warmup 2 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('dbg $dbg', round(d['value']), round(d['ms_per_step'],4), 'k1 live', round(d['roofline']['avg_launch_ms'],4), 'alone', round(d['roofline']['alone']['avg_launch_ms'],4))"
done > $O/dbg.txt 2>&1
cat $O/dbg.txt
