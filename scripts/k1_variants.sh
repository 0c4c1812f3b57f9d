#!/bin/bash
# dev aid: k1_fast ablation (1 = no stores, 2 = no refills, 4 = no mixing)
for v in 0 1 2 3 4 7; do
  VDL2GPU_K1_VARIANT=$v timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-parity 2>&1 | grep "^{" > /tmp/kv.json
  python -c "import json; d=json.load(open('/tmp/kv.json')); print('variant $v', d['roofline']['avg_launch_ms'])"
done
