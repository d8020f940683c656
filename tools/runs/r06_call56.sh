#!/bin/bash
# round 6, GPU call 56: checkpoint re-runs without their dead tail GEMM (encoder fc2 / decoder down_proj) - tests + the long-audio legs, alternating
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "checkpointing or long_audio or golden" 2>&1 | tail -3
for rnd in 1 2; do
for v in 1 0; do
AFK_RECOMPUTE_SKIP_TAIL=$v python bench.py --workload long5min --steps 3 --warmup 2 --no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity 2>gpurun_out/c56.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd long5min skip_tail=$v', d['ms_per_step'], d.get('loss'))" || tail -3 gpurun_out/c56.err
done; done
AFK_RECOMPUTE_SKIP_TAIL=1 python bench.py --workload long10min --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity 2>gpurun_out/c56.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('long10min skip_tail=1', d['ms_per_step'], d.get('loss'))" || tail -3 gpurun_out/c56.err
