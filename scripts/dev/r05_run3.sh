#!/bin/bash
# dev aid (round 5): what binds the period now -- environment knobs alternating, light loads, a detailed stage Gantt
cd "$(dirname "$0")/../.."
tag=${1:-r05}
mkdir -p gpurun_out
{
echo "== knobs"
for rep in 1 2; do
for k in "" "VDL2GPU_K2B_FRONT=1" "VDL2GPU_FRONT2=1" "VDL2GPU_K2B_FRONT=1 VDL2GPU_FRONT2=1"; do
  env $k python bench.py --no-cpu --no-extra --no-ring --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s' % '$k', round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4))
"
done; done
echo "== loads"
for bps in 0.05 1 2 4 8; do
  python bench.py --no-cpu --no-extra --no-ring --no-parity --bursts-per-s $bps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bps %-6s' % '$bps', round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4), d.get('bursts'), d.get('stats'))
"
done
echo "== gantt"
VDL2GPU_STAGE_DUMP=1 python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 32 --warmup 5 2>/tmp/gantt_err.txt >/dev/null
python scripts/dev/stage_gantt.py /tmp/gantt_err.txt
} > gpurun_out/${tag}_exp.txt 2>&1
cat gpurun_out/${tag}_exp.txt
