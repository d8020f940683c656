#!/bin/bash
# round 6, GPU call 36: decode attention at B = 8: splits x register budget
cd $GRAFT_REPO_ROOT
export ONLY=attn
for ns in 3 4 5 6; do for w in 4 7; do
echo "B=8 ns=$ns wpe=$w: $(NS=$ns AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 8 | tail -1 | cut -c120-)"
done; done
for ns in 4 8; do for w in 4 7; do
echo "B=8 keys=1500 ns=$ns wpe=$w: $(NS=$ns AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 8 1500 | tail -1 | cut -c120-)"
echo "B=5 ns=$ns wpe=$w: $(NS=$ns AFK_ATTN_DECODE_WPE=$w python tools/bench_decode_chain_batched.py 5 | tail -1 | cut -c120-)"
done; done
