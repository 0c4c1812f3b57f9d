#!/bin/bash
# dev aid: rebuild libvdl2gpu.so with extra -D flags on the GPU box and bench each variant (two rounds, interleaved)
#   scripts/variants.sh "-DK2A_THREADS=512" "-DK2A_TS=512" ...
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fPIC -shared -I include"
cp vdlm2dec_amd/libvdl2gpu.so /tmp/libvdl2gpu.keep
i=0
for v in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc $F $v vdlm2dec_amd/csrc/vdl2gpu.hip -o /tmp/libvariant_$i.so 2>/dev/null || echo "variant $v: build failed"
done
for round in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    [ -f /tmp/libvariant_$i.so ] || continue
    cp /tmp/libvariant_$i.so vdlm2dec_amd/libvdl2gpu.so
    timeout 300 python bench.py --steps ${STEPS:-12} --warmup 3 --no-cpu 2>&1 | grep "^{" > /tmp/kv.json
    python -c "import json; d=json.load(open('/tmp/kv.json')); print('variant $v', round(d['value']), round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['kernels_ms'].items() if k != 'note'}, d['parity']['equal'])"
  done
done
cp /tmp/libvdl2gpu.keep vdlm2dec_amd/libvdl2gpu.so
