#!/bin/bash
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/ab_tu_variants.py 144,6288@2048,6288@768,6288@1536,6288@512 32 2>&1 | tail -4 | tee $OUT/ab_tu_variants.txt
export OHEVC_TU_VARIANT=6288
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python tools/rocpd_summary.py pmc $OUT/pmc$i/p_results.db tu_idct 2>&1 | cut -c40-200 | tee -a $OUT/counters.txt
done
find $OUT -name '*.db' -delete
