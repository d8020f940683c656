#!/bin/bash
# round 6, GPU call 3: failing dp test with its output, peaked-weights calibration (v2), smoke, DynamicCache test, GEMM per-shape table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c3; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -k "one_gpu_gloo and sharp" > $O/dp_test.log 2>&1; tail -40 $O/dp_test.log | cut -c1-600
python tools/parity_fulldepth.py --only-peaked --out $O/peaked.json > $O/peaked_summary.json 2> $O/peaked.err; tail -4 $O/peaked.err; cut -c1-3000 $O/peaked_summary.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "dynamic_cache or past_key_values or smoke_entry" 2>&1 | tail -15
AFK_PROF_DUMP=$O/gemm_launches.csv python bench.py --no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 6 --warmup 2 --no-graph 2>$O/shapes.err | tail -1 | cut -c1-300
python tools/gemm_shapes.py $O/gemm_launches.csv > $O/gemm_shapes.md; head -40 $O/gemm_shapes.md
