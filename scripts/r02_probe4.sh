#!/bin/bash
cd "$(dirname "$0")/.."
VDL2GPU_K1_PROF=1 python bench.py --no-cpu --no-ring --no-parity --steps 6 --warmup 2 2>/dev/null | tail -1 > /tmp/b.json
python -c "
import json; d=json.load(open('/tmp/b.json')); g=d['dbg']; n=g[54]
print('alone', d['roofline']['alone']['avg_launch_ms'], 'waves', n)
names=['barrier1','loader(vmcnt+cvt+write+issue)','barrier2','mix','finalize','iter']
tot=sum(g[48:54])
for i,nm in enumerate(names): print('%-32s %10.0f ticks/wave  %5.1f%%' % (nm, g[48+i]/max(1,n), 100*g[48+i]/max(1,tot)))
"
