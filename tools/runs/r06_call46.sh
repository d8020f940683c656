#!/bin/bash
# round 6, GPU call 46: 9 .. 32 sequences per decode step as groups of eight through the norm-in-prologue launches
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "norm_in_prologue" 2>&1 | tail -5
for b in 8 9 12 16 17 24 32; do echo "B=$b: $(timeout 300 python tools/bench_decode.py $b 2>&1 | tail -1 | cut -c100-260)"; done
echo "B=16 old path: $(AFK_DECODE_CHAIN_BATCH_MAX=8 timeout 300 python tools/bench_decode.py 16 2>&1 | tail -1 | cut -c100-260)"
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py -q -x -k "generate or decode" 2>&1 | tail -3
