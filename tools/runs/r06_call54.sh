#!/bin/bash
# round 6, GPU call 54: AdamW with two vectors per thread in flight (AFK_ADAMW_UNROLL=2): test + step A/B, alternating, three rounds
cd $GRAFT_REPO_ROOT
AFK_ADAMW_UNROLL=2 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "adamw or optimizer" 2>&1 | tail -2
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2 3; do
  for u in 1 2; do
    AFK_ADAMW_UNROLL=$u python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd unroll=$u', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d.get('roofline_hbm',{}).get('achieved'), d.get('roofline_hbm',{}).get('ms'))"
  done
done
