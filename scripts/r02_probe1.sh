#!/bin/bash
# round-2 probe: instruction-rate microbenchmark + SQ counters of the current kernels
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02_probe1
mkdir -p $O
timeout 300 scripts/micro/valu_rate.bin > $O/valu_rate.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > $O/counters.txt
timeout 900 scripts/pmc_kernel.sh \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
  > $O/pmc.txt 2>&1
tail -5 /tmp/pmc_1.log /tmp/pmc_2.log /tmp/pmc_3.log >> $O/pmc.txt 2>&1
cat $O/valu_rate.txt
cat $O/pmc.txt | tail -40
