#!/bin/bash
# round 6, GPU call 24: AdamW thin-launch block count under the low-priority side stream (AFK_THIN_BLOCKS); alternating, three rounds
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2 3; do
  for n in 256 384 512 768 1024 2048; do
    AFK_THIN_BLOCKS=$n python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd thin=$n', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  done
done
