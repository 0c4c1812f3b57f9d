#!/bin/bash
# dev aid (round 5): the driver's command (--steps 20 --warmup 5, no extras) under environment settings, alternating
cd "$(dirname "$0")/../.."
tag=${1:-r05}; shift
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for k in "$@"; do
  env $k python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s' % '$k', round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4), d['host_ms_per_step']['enqueue'], d['host_ms_per_step']['wait_for_ring'])
"
done; done
} > gpurun_out/${tag}_drv.txt 2>&1
cat gpurun_out/${tag}_drv.txt
