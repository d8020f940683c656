#!/bin/bash
# round 6, GPU call 22: tail peel of the TN weight-gradient GEMMs (AFK_PEEL_TAIL) under the prioritised eager schedule - re-test of the round-2 / 3 A/B; alternating, three rounds
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2 3; do
  for v in 0 1; do
    AFK_PEEL_TAIL=$v python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd peel=$v', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  done
done
