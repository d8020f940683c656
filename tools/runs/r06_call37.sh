#!/bin/bash
# round 6, GPU call 37: decode tests on the new defaults + decode step timings
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "decode or attn_decode or hand" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_hf_plugin_gpu.py -q -x -k "generate or decode or cache" 2>&1 | tail -3
for r in 1 2; do
echo "step B=8: $(python tools/bench_decode.py 8 2>&1 | tail -1 | cut -c100-260)"
echo "step B=1: $(python tools/bench_decode.py 1 2>&1 | tail -1 | cut -c100-260)"
done
echo "step B=5: $(python tools/bench_decode.py 5 2>&1 | tail -1 | cut -c100-260)"
echo "step B=4: $(python tools/bench_decode.py 4 2>&1 | tail -1 | cut -c100-260)"
echo "step B=2: $(python tools/bench_decode.py 2 2>&1 | tail -1 | cut -c100-260)"
python tools/bench_decode_chain_batched.py 8 | tail -1
