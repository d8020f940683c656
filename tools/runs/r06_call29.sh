#!/bin/bash
# round 6, GPU call 29: is it the input rows or the kernel structure?  narrow Linears of the batched decode step at M = 1 / 2 / 4 / 8 in both forms
cd $GRAFT_REPO_ROOT
export ONLY=qkv,o_proj,down AFK_CHAIN_AHEAD=0
for M in 1 2 4 8; do
echo "M=$M mfma: $(AFK_CHAIN_MFMA=1 python tools/bench_decode_chain_batched.py $M | tail -1 | cut -c120-)"
echo "M=$M dot:  $(AFK_CHAIN_MFMA=0 python tools/bench_decode_chain_batched.py $M | tail -1 | cut -c120-)"
done
echo "M=8 mfma 16,8 all: $(AFK_CHAIN_MFMA=1 AFK_CHAIN_MFMA_NARROW=16,8 python tools/bench_decode_chain_batched.py 8 | tail -1 | cut -c120-)"
echo "M=1 mfma 16,8 all: $(AFK_CHAIN_MFMA=1 AFK_CHAIN_MFMA_NARROW=16,8 python tools/bench_decode_chain_batched.py 1 | tail -1 | cut -c120-)"
