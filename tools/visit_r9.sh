# round 4: two more rows of the two-layer bench - qp22-like density at 1080p, and an 8K enhancement layer over a 4K base layer
TAG=${1:-r9}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 150 python tools/bench_shvc.py --size 1920x1088 --frames 17 --passes 2 --dense 2> /dev/null | tail -1 > $OUT/bench_shvc_1080p_dense.json; cut -c1-800 $OUT/bench_shvc_1080p_dense.json
timeout 200 python tools/bench_shvc.py --size 7680x4352 --frames 5 --passes 1 --kinds sse,hip 2> /dev/null | tail -1 > $OUT/bench_shvc_8k.json; cut -c1-800 $OUT/bench_shvc_8k.json
