#!/bin/bash
# dev aid: same-box A/B of (library variant, environment) pairs, alternating:  scripts/dev/abx.sh <reps> "name[:VAR=1 VAR2=2]" ...
#   name = a file variants/<name>.so (scripts/dev/mkvariant.sh), or "-" for the tree's own library
cd "$(dirname "$0")/../.."
reps=$1; shift
for rep in $(seq $reps); do
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=${spec#*:}
  lib=""; [ "$name" != "-" ] && lib="VDL2GPU_LIB=$PWD/variants/$name.so"
  env $lib $envs python bench.py --no-cpu --no-extra --no-ring --no-parity $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s' % '$spec', round(d['value']), round(d['ms_per_step'],4), 'steady', round(d.get('steady_state',{}).get('ms_per_step',0),4), {k:round(v,3) for k,v in d['kernels_ms'].items() if k!='note'}, 'k1', round(d['roofline']['avg_launch_ms'],4), d['stats']['repairs'], d['stats']['serial_redos'])
"
done; done
