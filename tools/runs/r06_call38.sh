#!/bin/bash
# round 6, GPU call 38: bias sums where their operand is produced - tests + step A/B
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "colsum or layernorm or gelu or norm_bwd" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "fused_bias or golden or checkpointing or wgrad_stream or graphed or accumulation" 2>&1 | tail -4
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2 3; do
for v in 1 0; do
AFK_FUSE_BIAS_SUMS=$v python bench.py $F 2>gpurun_out/call38_err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd fuse_bias_sums=$v', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d.get('loss'))" || tail -5 gpurun_out/call38_err.log
done; done
