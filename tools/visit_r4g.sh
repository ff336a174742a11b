TAG=r4g; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for s in natural flat; do
  python tools/diag_chain_clocks.py $s 2>/dev/null | grep '^{' | tee -a $OUT/chain_clocks.jsonl
  OHEVC_CHAIN_AGENT_ACQUIRE=1 python tools/diag_chain_clocks.py $s 2>/dev/null | grep '^{' | sed 's/^{/{"agent_acquire": 1, /' | tee -a $OUT/chain_clocks.jsonl
  OHEVC_INTRA_CHAIN_WAVES=8 python tools/diag_chain_clocks.py $s 2>/dev/null | grep '^{' | sed 's/^{/{"chain_waves": 8, /' | tee -a $OUT/chain_clocks.jsonl
done
