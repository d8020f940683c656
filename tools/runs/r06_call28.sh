#!/bin/bash
# round 6, GPU call 28: batched decode, narrow Linears with the input fragments one stage ahead (AFK_CHAIN_AHEAD), group shapes per Linear
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -q -k "decode_chain_batched" 2>&1 | tail -2
B="python tools/bench_decode_chain_batched.py 8"
export ONLY=qkv,o_proj,down
for rnd in 1 2; do
echo "ahead default shapes (32,8/32,8/16,16): $($B | tail -1 | cut -c120-)"
echo "ahead0 default shapes:  $(AFK_CHAIN_AHEAD=0 $B | tail -1 | cut -c120-)"
echo "ahead 16,8:     $(AFK_CHAIN_MFMA_NARROW=16,8 $B | tail -1| cut -c120-)"
echo "ahead0 16,8:    $(AFK_CHAIN_AHEAD=0 AFK_CHAIN_MFMA_NARROW=16,8 $B | tail -1| cut -c120-)"
echo "ahead 32,8:     $(AFK_CHAIN_MFMA_NARROW=32,8 $B | tail -1| cut -c120-)"
done
unset ONLY
AFK_CHAIN_MFMA_NARROW=32,8/16,8/16,8 python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c1-300
AFK_CHAIN_MFMA_NARROW=32,8/16,16/16,8 python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c1-300
AFK_CHAIN_AHEAD=0 python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c1-300
