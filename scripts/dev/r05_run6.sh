#!/bin/bash
# dev aid (round 5): -m gpu tests, environment settings alternating (steady state and repairs shown), a Gantt and a parity run under the last setting
cd "$(dirname "$0")/../.."
tag=${1:-r05}; shift
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.txt
tail -4 gpurun_out/${tag}_tests.txt
fi
{
for rep in 1 2; do
for k in "$@"; do
  env $k python bench.py --no-cpu --no-extra --no-ring --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s' % '$k', round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4), d['stats']['repairs'], d['stats']['serial_redos'])
"
done; done
last="${@: -1}"
echo "== gantt $last"
env $last VDL2GPU_STAGE_DUMP=1 python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 32 --warmup 5 2>/tmp/gantt_err.txt >/dev/null
python scripts/dev/stage_gantt.py /tmp/gantt_err.txt | sed -n "1p;6,12p"
echo "== parity $last"
env $last python bench.py --no-cpu --no-extra --no-ring 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity']['equal'], d['parity']['bursts_checked'], d['parity']['mismatches'][:3])"
} > gpurun_out/${tag}_exp.txt 2>&1
cat gpurun_out/${tag}_exp.txt
