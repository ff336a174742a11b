#!/bin/bash
# what dividing the slice-data parse over processes buys on ONE GPU (planes host-staged through gloo): bench.py --mode frames with 1, 2, 4 ranks
TAG=${1:-r02fp}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python bench.py --mode frames --steps 2 --warmup 1 --frames-pictures 33 2>/dev/null | tail -1 > $OUT/frames_1rank.json
for n in 2 4; do
  ( cd /tmp && timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2964$n $GRAFT_REPO_ROOT/bench.py --gpus $n --mode frames --frames-one-gpu --steps 2 --warmup 1 --frames-pictures 33 2>/dev/null | tail -1 ) > $OUT/frames_${n}ranks_one_gpu.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02fp/frames_*.json")):
    try: d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["unit"], d["fps"], "fps", d["config"]["exchange"])
    except Exception as e: print(f, "unreadable", e, open(f).read()[:300])
PY
