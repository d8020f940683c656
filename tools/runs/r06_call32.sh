#!/bin/bash
# round 6, GPU call 32: decode attention at B = 8: splits and group form
cd $GRAFT_REPO_ROOT
export ONLY=attn
for ns in 2 4 8 16; do
echo "per-head ns=$ns: $(NS=$ns python tools/bench_decode_chain_batched.py 8 | tail -1 | cut -c120-)"
echo "group    ns=$ns: $(AFK_ATTN_DECODE_GROUP=1 NS=$ns python tools/bench_decode_chain_batched.py 8 | tail -1 | cut -c120-)"
done
echo "per-head ns=8 B=1: $(NS=8 python tools/bench_decode_chain_batched.py 1 | tail -1 | cut -c120-)"
echo "per-head ns=8 B=2: $(NS=8 python tools/bench_decode_chain_batched.py 2 | tail -1 | cut -c120-)"
echo "per-head ns=8 B=4: $(NS=8 python tools/bench_decode_chain_batched.py 4 | tail -1 | cut -c120-)"
echo "per-head ns=8 B=8 keys=200: $(NS=8 python tools/bench_decode_chain_batched.py 8 200 | tail -1 | cut -c120-)"
