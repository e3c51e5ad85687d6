#!/bin/bash
# e2e rate of the chunk-sharded multi-process mode, rehearsed on one GPU (gloo, all ranks on GPU 0)
# usage: tools/e2e_dist.sh <ranks> <pairs per chunk> <chunks>
set -e
R=${1:-2}; N=${2:-1000000}; C=${3:-8}
W=$(mktemp -d /tmp/quade_e2e_dist.XXXX)
python tools/e2e_bench.py $N 1 $C --prepare-only $W > /dev/null
cd $W/out
S=$(date +%s.%N)
QUADE_DIST_BACKEND=gloo QUADE_DEVICE=0 PYTHONPATH=$GRAFT_REPO_ROOT python -m torch.distributed.run --nnodes=1 --nproc-per-node $R \
  --master-addr 127.0.0.1 --master-port 29544 -m quade_amd.quade -c $W/conf.txt > $W/log.txt 2>&1 || { tail -20 $W/log.txt; exit 1; }
E=$(date +%s.%N)
python - <<PY
n = $N * $C
dt = $E - $S
print('{"mode": "e2e multi-process", "ranks": $R, "chunks": $C, "pairs": %d, "seconds": %.2f, "pairs_per_s": %.0f}' % (n, dt, n / dt))
PY
grep "Total pair" $W/out/Quade_report.csv | head -1
rm -rf $W
