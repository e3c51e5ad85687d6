#!/bin/bash
# Diagnostic: the multi-process CLI with the nccl (RCCL) backend and two ranks on ONE GPU -- RCCL is
# expected to refuse duplicate devices; the real thing needs one GPU per rank.  Prints the log tail.
W=$(mktemp -d /tmp/q2.XXXX); cd $W
python $GRAFT_REPO_ROOT/tools/e2e_bench.py 20000 1 4 --prepare-only $W > /dev/null
cd $W/out
QUADE_DEVICE=0 PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29599 -m quade_amd.quade -c $W/conf.txt > $W/log.txt 2>&1
echo "exit code $?"
grep -v "^$" $W/log.txt | grep -iE "error|duplicate|Traceback|raise|nccl|rccl" | head -12 | cut -c1-250
