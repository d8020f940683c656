#!/bin/bash
# round 6, GPU call 59: last decoder layer behind its attention on the labelled rows only - tests + step A/B
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_custom_ops_gpu.py tests/test_dp_gpu.py -q -x 2>&1 | tail -4
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2 3; do
for v in 1 0; do
AFK_LAST_LAYER_ROWS=$v python bench.py $F 2>gpurun_out/c59.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd last_layer_rows=$v', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d.get('loss'), d.get('executed_tflops_per_gpu'))" || tail -5 gpurun_out/c59.err
done; done
