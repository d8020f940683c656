#!/bin/bash
# round 6, GPU call 35: decode attention, half-depth register batches with 7 resident blocks per CU
cd $GRAFT_REPO_ROOT
export ONLY=attn
for w in 4 7 8; do
echo "B=8 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 8 | tail -1 | cut -c120-)"
echo "B=8 wpe=$w keys=1500: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 8 1500 | tail -1 | cut -c120-)"
echo "B=4 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 4 | tail -1 | cut -c120-)"
echo "B=1 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 1 | tail -1 | cut -c120-)"
done
unset ONLY
for w in 4 7; do
echo "step B=8 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c100-260)"
done
