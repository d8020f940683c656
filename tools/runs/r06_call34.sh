#!/bin/bash
# round 6, GPU call 34: decode attention register budget (AFK_ATTN_DECODE_WPE) at B = 8 / 1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "attn_decode or hand_over or handover" 2>&1 | tail -3
export ONLY=attn
for w in 3 4 5 6; do
echo "B=8 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 8 | tail -1 | cut -c120-)"
done
for w in 3 4 5; do
echo "B=1 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 1 | tail -1 | cut -c120-)"
echo "B=4 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 4 | tail -1 | cut -c120-)"
done
unset ONLY
for w in 3 4 5; do
echo "step B=8 wpe=$w: $(AFK_ATTN_DECODE_WPE=$w python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c100-260)"
done
echo "step B=1 default: $(python tools/bench_decode.py 1 2>&1 | tail -1 | cut -c100-260)"
