#!/bin/bash
# round 6, GPU call 30: batched decode, stages of the K loop dealt round-robin to the waves of a block (AFK_CHAIN_KIL)
cd $GRAFT_REPO_ROOT
export AFK_CHAIN_AHEAD=0
AFK_CHAIN_KIL=1 python -m pytest tests/test_ops_gpu.py -q -k "decode_chain_batched" 2>&1 | tail -2
B="python tools/bench_decode_chain_batched.py 8"
for rnd in 1 2; do
echo "kil0 default:   $($B | tail -1 | cut -c120-)"
echo "kil1 default:   $(AFK_CHAIN_KIL=1 $B | tail -1 | cut -c120-)"
echo "kil1 16,8:      $(ONLY=qkv,o_proj,down AFK_CHAIN_KIL=1 AFK_CHAIN_MFMA_NARROW=16,8 $B | tail -1| cut -c120-)"
echo "kil1 32,8:      $(ONLY=qkv,o_proj,down AFK_CHAIN_KIL=1 AFK_CHAIN_MFMA_NARROW=32,8 $B | tail -1| cut -c120-)"
echo "kil1 16,16:     $(ONLY=qkv,o_proj,down AFK_CHAIN_KIL=1 AFK_CHAIN_MFMA_NARROW=16,16 $B | tail -1| cut -c120-)"
done
