#!/bin/bash
TAG=${1:-r02s2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant 0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python tools/rocpd_summary.py pmc $OUT/pmc$i/p_results.db sao_kernel 2>&1 | cut -c1-220 | tee -a $OUT/counters.txt
done
find $OUT -name '*.db' -delete
