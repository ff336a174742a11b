#!/bin/bash
# SQ / TCC counter passes on the headline bench (each pass its own run; --kernel-trace only, per the gpurun rules)
OUT=gpurun_out/${1:-ctr}; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -c . $OUT/counters_list.txt
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python tools/rocpd_summary.py pmc $OUT/pmc$i/p_results.db tu_idct 2>&1 | cut -c40-200 | tee -a $OUT/counters.txt
done
find $OUT -name '*.db' -delete
