#!/bin/bash
# dev aid (round 6): GPU tests, then the driver's bench command, then a stage Gantt -- one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_${1:-a}_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06_${1:-a}_tests.txt
tail -3 gpurun_out/r06_${1:-a}_tests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_${1:-a}_bench.json 2> gpurun_out/r06_${1:-a}_bench.err; echo "bench rc $?"
python - <<P
import json
d=json.loads(open('gpurun_out/r06_${1:-a}_bench.json').read().strip().splitlines()[-1])
print('value',round(d['value']),'ms',round(d['ms_per_step'],4),'steady',d.get('steady_state',{}).get('ms_per_step'))
print({k:(round(v['value']) if isinstance(v,dict) and 'value' in v else None) for k,v in d.get('configs',{}).items()})
print(d.get('stats'))
P
VDL2GPU_STAGE_DUMP=1 timeout 300 python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 32 --warmup 5 2>/tmp/gantt_err.txt >/dev/null
python scripts/dev/stage_gantt.py /tmp/gantt_err.txt > gpurun_out/r06_${1:-a}_gantt.txt 2>&1; tail -12 gpurun_out/r06_${1:-a}_gantt.txt
