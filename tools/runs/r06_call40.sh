#!/bin/bash
# round 6, GPU call 40: RMSNorm in the prologue of the batched qkv / gate|up / lm_head launches
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "decode_chain" 2>&1 | tail -5
python tools/bench_decode_chain_batched.py 8 | tail -1
for r in 1 2; do
for m in prologue producer launch; do
echo "step B=8 norm=$m: $(AFK_DECODE_NORM=$m python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c100-200)"
done; done
echo "step B=4 prologue: $(python tools/bench_decode.py 4 2>&1 | tail -1 | cut -c100-200)"
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_hf_plugin_gpu.py -q -x -k "generate or decode or cache" 2>&1 | tail -3
