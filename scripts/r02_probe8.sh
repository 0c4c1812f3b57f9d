#!/bin/bash
# whole-library optimisation level vs kernel times (instruction-cache start-up costs): rebuild on the box, rocprof stats
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
F="--offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fPIC -shared -I include"
cp vdlm2dec_amd/libvdl2gpu.so /tmp/keep.so
here=$(pwd)
while read -r v; do
  /opt/rocm/bin/hipcc $F $v vdlm2dec_amd/csrc/vdl2gpu.hip -o vdlm2dec_amd/libvdl2gpu.so 2>/dev/null || { echo "build failed [$v]"; continue; }
  rm -rf /tmp/pr8
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr8 -- python $here/bench.py --no-cpu --no-ring --steps 10 --warmup 3 > /tmp/pr8.log 2>&1 )
  echo "variant [$v]"
  tail -1 /tmp/pr8.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('   step', round(d['ms_per_step'],4), 'parity', d['parity']['equal'], d['parity']['bursts_checked'])
except Exception as e: print('   bench failed', e)"
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pr8/**/*kernel_stats.csv',recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print("   %-40s avg %8.1f us" % (r["Name"][:40], float(r["AverageNs"])/1e3))
PY
done
cp /tmp/keep.so vdlm2dec_amd/libvdl2gpu.so
