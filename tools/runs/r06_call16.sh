#!/bin/bash
# round 6, GPU call 16: which dgrads read W itself on the NN kernel (no W^T shadow) - re-test of the round-3 A/B on this round's build; alternating, two rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c16; mkdir -p $O
F="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 8 --warmup 2"
for rnd in 1 2; do
  for v in "default:mlp.gate_up.weight" "down:mlp.gate_up.weight,mlp.down_proj.weight" "attn:mlp.gate_up.weight,self_attn.o_proj.weight,self_attn.qkv.weight" "all4:mlp.gate_up.weight,mlp.down_proj.weight,self_attn.o_proj.weight,self_attn.qkv.weight"; do
    name=${v%%:*}; val=${v#*:}
    AFK_NN_DGRAD=$val python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd $name', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['peak_mem_gib'])"
  done
  AFK_BWD_FORM=direct python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$rnd direct', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['peak_mem_gib'])"
done
