#!/bin/bash
# dev aid: bench the library variants variants/*.so (built by scripts/dev/mkvariant.sh) against each other on ONE box, alternating;
# usage: scripts/dev/abv.sh [reps] [bench args...]
cd "$(dirname "$0")/../.."
reps=${1:-2}; shift
for rep in $(seq $reps); do
for v in variants/*.so; do
  VDL2GPU_LIB=$PWD/$v python bench.py --no-cpu --no-extra --no-ring --no-parity "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s' % '$v', round(d['value']), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernels_ms'].items() if k!='note'}, round(d['roofline']['avg_launch_ms'],4))
"
done; done
