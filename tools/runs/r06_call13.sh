#!/bin/bash
# round 6, GPU call 13 (final build): full GPU suite, smoke, GEMM traffic PMC (stamped on this build), serial + default kernel tables, two default bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c13; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log | tail -2; grep "^FAILED" $O/gpu_suite.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 600 python tools/measure_gemm_traffic.py $O/gemm_traffic.json > $O/gemm_traffic.log 2>&1; tail -2 $O/gemm_traffic.log | cut -c1-400
PROF_STEPS=3 bash tools/prof_bench.sh r06_serial --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --no-opt-overlap --no-wgrad-stream --steps 3 2>&1 | tail -1 | cut -c1-200
PROF_STEPS=3 bash tools/prof_bench.sh r06_default --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 3 2>&1 | tail -1 | cut -c1-200
cp $O/gemm_traffic.json profiles/r06_gemm_traffic.json 2>/dev/null
for i in 1 2; do
  timeout 900 python bench.py > $O/bench_default_$i.out 2> $O/bench_default_$i.err; tail -1 $O/bench_default_$i.out | wc -c; tail -1 $O/bench_default_$i.out | cut -c1-700
  cp gpurun_out/bench_detail.json $O/bench_detail_$i.json
done
