#!/bin/bash
# round 6, GPU call 26: decode kernel tables (B = 8 and B = 1) on the current build
cd $GRAFT_REPO_ROOT
bash tools/prof_decode.sh 8; cp gpurun_out/prof_decode/stats.md gpurun_out/decode_b8_stats.md
bash tools/prof_decode.sh 1; cp gpurun_out/prof_decode/stats.md gpurun_out/decode_b1_stats.md
python tools/bench_decode_chain.py 2>&1 | tail -1
