#!/bin/bash
# dev aid: the bench with and without environment settings, same box, alternating: scripts/dev/ab_env.sh "VDL2GPU_FOO=1" ["A=1 B=2" ...]
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
for k in "" "$@"; do
  env $k python bench.py --no-cpu --no-extra --no-ring --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s' % '$k', round(d['value']), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernels_ms'].items() if k!='note'})
"
done; done
