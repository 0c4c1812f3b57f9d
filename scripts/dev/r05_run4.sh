#!/bin/bash
# dev aid (round 5): reserved CUs (VDL2GPU_RESERVE_CUS) -- alternating bench runs, then a Gantt and a parity run at one setting
cd "$(dirname "$0")/../.."
tag=${1:-r05}; shift
mkdir -p gpurun_out
{
for rep in 1 2; do
for k in "" "$@"; do
  env $k python bench.py --no-cpu --no-extra --no-ring --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s' % '$k', round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4))
"
done; done
echo "== gantt $1"
env $1 VDL2GPU_STAGE_DUMP=1 python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 32 --warmup 5 2>/tmp/gantt_err.txt >/dev/null
python scripts/dev/stage_gantt.py /tmp/gantt_err.txt
echo "== parity $1"
env $1 python bench.py --no-cpu --no-extra --no-ring 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity'))"
} > gpurun_out/${tag}_exp.txt 2>&1
cat gpurun_out/${tag}_exp.txt
