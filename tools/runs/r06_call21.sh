#!/bin/bash
# round 6, GPU call 21 (final build): full GPU suite, smoke, GEMM traffic stamp, default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c21; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log | tail -2; grep "^FAILED" $O/gpu_suite.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 600 python tools/measure_gemm_traffic.py $O/gemm_traffic.json > $O/gemm_traffic.log 2>&1; tail -1 $O/gemm_traffic.log | cut -c1-200
cp $O/gemm_traffic.json profiles/r06_gemm_traffic.json 2>/dev/null
timeout 900 python bench.py > $O/bench_default.out 2> $O/bench_default.err; tail -1 $O/bench_default.out | wc -c; tail -1 $O/bench_default.out | cut -c1-500
cp gpurun_out/bench_detail.json $O/bench_detail.json
