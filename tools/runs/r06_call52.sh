#!/bin/bash
# round 6, GPU call 52: AdamW thin-launch block count on the final build (default 768 since this call), alternating, three rounds
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2 3; do
  for n in 256 768; do
    AFK_THIN_BLOCKS=$n python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd thin=$n', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  done
done
