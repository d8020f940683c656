#!/bin/bash
# round 6, GPU call 43: decode attention, group form on the matrix pipe
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "attn_decode" 2>&1 | tail -6
export ONLY=attn
for b in 8 4 2; do for ns in 8 16; do
echo "B=$b ns=$ns per-head: $(NS=$ns python tools/bench_decode_chain_batched.py $b | tail -1 | cut -c120-)"
echo "B=$b ns=$ns gmma:     $(NS=$ns AFK_ATTN_DECODE_GROUP=3 python tools/bench_decode_chain_batched.py $b | tail -1 | cut -c120-)"
done; done
echo "B=8 keys=1500 gmma: $(AFK_ATTN_DECODE_GROUP=3 python tools/bench_decode_chain_batched.py 8 1500 | tail -1 | cut -c120-)"
echo "B=8 keys=1500 per-head: $(python tools/bench_decode_chain_batched.py 8 1500 | tail -1 | cut -c120-)"
unset ONLY
for r in 1 2; do
echo "step B=8 per-head: $(python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c150-200)"
echo "step B=8 gmma:     $(AFK_ATTN_DECODE_GROUP=3 python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c150-200)"
done
