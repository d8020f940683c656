#!/bin/bash
# round 6, GPU call 25: what the optimizer costs inside the overlapped step (AFK_PROBE_SKIP_ADAMW: invalid timing probe) and the CU-masked side stream
# (AFK_SIDE_CUS: AdamW on n CUs, n / 8 per XCD); alternating, three rounds
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
run() {  # label, env...
  local label=$1; shift
  env "$@" python bench.py $F 2>gpurun_out/call25_err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d.get('loss'))" || tail -5 gpurun_out/call25_err.log
}
for rnd in 1 2 3; do
  run "$rnd base" AFK_X=0
  run "$rnd skip_adamw" AFK_PROBE_SKIP_ADAMW=1
  run "$rnd side16" AFK_SIDE_CUS=16
  run "$rnd side32" AFK_SIDE_CUS=32
  run "$rnd side64" AFK_SIDE_CUS=64
  run "$rnd side32_thin1024" AFK_SIDE_CUS=32 AFK_THIN_BLOCKS=1024
  run "$rnd side64_thin2048" AFK_SIDE_CUS=64 AFK_THIN_BLOCKS=2048
  run "$rnd side64_main192" AFK_SIDE_CUS=64 AFK_MAIN_CUS=192
done
