#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02_probe2
mkdir -p $O
for ns in 1 2 4 7 12 28; do
  VDL2GPU_K1_NSUB=$ns python bench.py --no-cpu --no-ring --no-parity --steps 8 --warmup 2 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('nsub $ns', round(d['value']), round(d['ms_per_step'],4), 'k1 live', round(d['roofline']['avg_launch_ms'],4), 'alone', round(d['roofline']['alone']['avg_launch_ms'],4))"
done > $O/nsub.txt 2>&1
cat $O/nsub.txt
timeout 900 scripts/pmc_kernel.sh \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
  2>&1 | grep "k1_" > $O/pmc.txt
cat $O/pmc.txt
