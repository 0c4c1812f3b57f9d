#!/bin/bash
# dev aid (round 6): GPU tests, then the default bench command under each of the given environment settings, alternating (A B A B)
#   scripts/dev/r06_try.sh <tag> "VAR=1" "VAR=2" ...      (an empty string = the default)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
tag=$1; shift
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_${tag}_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06_${tag}_tests.txt
tail -3 gpurun_out/r06_${tag}_tests.txt
fi
for rep in 1 2; do
for e in "$@"; do
  env $e timeout 300 python bench.py --no-cpu --no-extra --no-ring 2>/dev/null | tail -1 > /tmp/line.json
  python - "$e" <<'P'
import json,sys
d=json.load(open('/tmp/line.json'))
print('%-40s value %8.0f ms %.4f steady %s parity %s repairs %s redos %s' % (sys.argv[1] or 'default', d['value'] or 0, d['ms_per_step'], round(d.get('steady_state',{}).get('ms_per_step',0),4), d['parity']['equal'], d['stats']['repairs'], d['stats']['serial_redos']))
P
done
done 2>&1 | tee -a gpurun_out/r06_${tag}_ab.txt
