#!/bin/bash
# dev aid (round 5): library variants x environment settings, alternating; steady state and repairs shown
cd "$(dirname "$0")/../.."
tag=${1:-r05}; shift
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in variants/*.so; do
for k in "$@"; do
  env $k VDL2GPU_LIB=$PWD/$v python bench.py --no-cpu --no-extra --no-ring --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %-60s' % ('$v', '$k'), round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4), d['stats']['repairs'], d['stats']['serial_redos'])
"
done; done; done
} > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
