#!/bin/bash
# dev aid: the same bench arguments on the round-2 tree and on this tree, same box, alternating.  Needs the round-2 tree beside
# this one:  git worktree add -f .ab_r02 6c1981f && (cd .ab_r02 && python -c 'import __graft_entry__ as g; g.build()')   [git-ignored]
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for t in .ab_r02 .; do
  ( cd $t && python bench.py "$@" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', round(d['value']), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernels_ms'].items() if k!='note'}, d['roofline']['avg_launch_ms'])
" )
done; done
