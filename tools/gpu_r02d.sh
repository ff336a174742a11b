#!/bin/bash
# SQ counter passes on the headline kernel, dot2 form (variant 1168) vs matrix-core form (variant 1424 = 16+128+256+1024... no nt there: 400)
TAG=${1:-r02d}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in 144 400; do
  export OHEVC_TU_VARIANT=$v
  CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/v${v}_pmc$i -o p -- $CMD > $OUT/v${v}_pmc$i.log 2>&1
    echo "variant $v set $i" | tee -a $OUT/counters.txt
    python tools/rocpd_summary.py pmc $OUT/v${v}_pmc$i/p_results.db tu_idct 2>&1 | cut -c40-200 | tee -a $OUT/counters.txt
  done
done
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/sq_counter_names.txt
find $OUT -name '*.db' -delete
