#!/bin/bash
# round 6, GPU call 53 (final build of the last session): full GPU suite, smoke, GEMM traffic stamp, default bench line x 2, serial + default-schedule kernel tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c53; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log | tail -2; grep "^FAILED" $O/gpu_suite.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 600 python tools/measure_gemm_traffic.py $O/gemm_traffic.json > $O/gemm_traffic.log 2>&1; tail -1 $O/gemm_traffic.log | cut -c1-200
cp $O/gemm_traffic.json profiles/r06_gemm_traffic.json 2>/dev/null
for i in 1 2; do
timeout 900 python bench.py > $O/bench_default_$i.out 2> $O/bench_default_$i.err; tail -1 $O/bench_default_$i.out | wc -c; tail -1 $O/bench_default_$i.out | cut -c1-300
cp gpurun_out/bench_detail.json $O/bench_detail_$i.json
done
bash tools/prof_bench.sh c53serial --steps 3 --warmup 1 --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --no-graph --no-wgrad-stream --no-opt-overlap
cp gpurun_out/prof_c53serial/stats.md $O/kernel_stats_serial.md
bash tools/prof_bench.sh c53default --steps 3 --warmup 1 --no-eager-baseline --no-long-audio --no-extra-legs --no-parity
cp gpurun_out/prof_c53default/stats.md $O/kernel_stats_default.md; cp gpurun_out/prof_c53default/timeline.md $O/timeline_default.md
bash tools/prof_decode.sh 8; cp gpurun_out/prof_decode/stats.md $O/decode_b8_stats.md
bash tools/prof_decode.sh 1; cp gpurun_out/prof_decode/stats.md $O/decode_b1_stats.md
for b in 1 2 4 8 16 32; do echo "decode B=$b: $(timeout 200 python tools/bench_decode.py $b 2>&1 | tail -1 | cut -c100-230)"; done > $O/decode_steps.txt; cat $O/decode_steps.txt
python tools/bench_decode_chain_batched.py 8 | tail -1 > $O/decode_b8_launches.json
