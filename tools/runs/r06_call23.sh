#!/bin/bash
# round 6, GPU call 23: side-stream knobs under the prioritised schedule - AdamW thin-launch block count, wgrad stream on / off, optimizer overlap on / off; alternating, two rounds
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2; do
  for v in "default:" "thin128:AFK_THIN_BLOCKS=128" "thin192:AFK_THIN_BLOCKS=192" "thin384:AFK_THIN_BLOCKS=384" "thintr64:AFK_THIN_TRANSPOSE=64"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd $name', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  done
  python bench.py $F --no-wgrad-stream 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd no-wgrad-stream', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  python bench.py $F --no-opt-overlap 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd no-opt-overlap', d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done
