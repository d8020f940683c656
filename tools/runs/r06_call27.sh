#!/bin/bash
# round 6, GPU call 27: batched decode launches per kernel (M = 8): lane-masked / hoisted input fragments, group shapes, dot-product form
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -q -k "decode_chain" 2>&1 | tail -3
B="python tools/bench_decode_chain_batched.py 8"
for rnd in 1 2; do
echo "default:        $($B | tail -1)"
echo "xmask0:         $(AFK_CHAIN_XMASK=0 $B | tail -1)"
echo "narrow16,8:     $(AFK_CHAIN_MFMA_NARROW=16,8 ONLY=qkv,o_proj,down $B | tail -1)"
echo "narrow32,8:     $(AFK_CHAIN_MFMA_NARROW=32,8 ONLY=qkv,o_proj,down $B | tail -1)"
echo "narrow16,16:    $(AFK_CHAIN_MFMA_NARROW=16,16 ONLY=qkv,o_proj,down $B | tail -1)"
echo "dot:            $(AFK_CHAIN_MFMA=0 $B | tail -1)"
echo "dot S=8 narrow: $(AFK_CHAIN_MFMA=0 AFK_CHAIN_S=8,8,8,1,1 ONLY=qkv,o_proj,down $B | tail -1)"
done
python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c1-400
