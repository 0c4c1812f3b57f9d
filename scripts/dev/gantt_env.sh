#!/bin/bash
# dev aid: stage Gantt (VDL2GPU_STAGE_DUMP) of the headline bench under environment settings: scripts/dev/gantt_env.sh "A=1 B=2" "C=3" ...
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for k in "" "$@"; do
  env $k VDL2GPU_STAGE_DUMP=1 python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 32 --warmup 5 2>/tmp/gantt_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('== %-60s' % '$k', round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4))"
  python scripts/dev/stage_gantt.py /tmp/gantt_err.txt | tail -4
done; done
