#!/bin/bash
# round 6, GPU call 31: fused Linear + residual + RMSNorm launches of the batched decode step
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "decode_chain" 2>&1 | tail -8
B="python tools/bench_decode_chain_batched.py 8"
for rnd in 1 2; do
echo "default: $($B | tail -1 | cut -c120-)"
done
for rnd in 1 2; do
echo "fused norm:   $(python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c1-260)"
echo "unfused norm: $(AFK_DECODE_FUSE_NORM=0 python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c1-260)"
done
timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "generate" 2>&1 | tail -3
