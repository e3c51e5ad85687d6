set -e
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -20 gpurun_out/bench1.err; exit 1; }
cat gpurun_out/bench1.json
tail -3 gpurun_out/bench1.err
