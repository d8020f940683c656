#!/bin/bash
# round 6, GPU call 44: decode tests on the new defaults (matrix-pipe group attention for large launches) + timings
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "decode or hand_over" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_hf_plugin_gpu.py tests/test_fullwidth_gpu.py -q -x -k "generate or decode or cache" 2>&1 | tail -3
for r in 1 2; do echo "step B=8: $(python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c150-200)"; done
echo "step B=6: $(python tools/bench_decode.py 6 2>&1 | tail -1 | cut -c150-200)"
echo "step B=5: $(python tools/bench_decode.py 5 2>&1 | tail -1 | cut -c150-200)"
echo "step B=5 per-head: $(AFK_ATTN_DECODE_GROUP=0 python tools/bench_decode.py 5 2>&1 | tail -1 | cut -c150-200)"
echo "step B=1: $(python tools/bench_decode.py 1 2>&1 | tail -1 | cut -c150-200)"
