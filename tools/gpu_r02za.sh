#!/bin/bash
# SQ counters of the matrix-core motion compensation (16x16 uni, 8 bit): where do the cycles go
TAG=${1:-r02zj}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python tools/bench_kernels.py --resident --planes 8 --only mc --mc-variant 4 --mc-config 16,16,0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python tools/rocpd_summary.py pmc $OUT/pmc$i/p_results.db mc4_kernel 2>&1 | cut -c1-220 | tee -a $OUT/counters.txt
done
find $OUT -name '*.db' -delete
