#!/bin/bash
# round 6, GPU call 6: XCD-aware attention block map A/B, dK/dV parts A/B, SwiGLU backward forms, attention tests, full-depth parity test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c6; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "attention" 2>&1 | tail -6 | cut -c1-300
timeout 900 python tools/bench_attn_sched.py decoder encoder 5-min > $O/attn_xcd_ab.jsonl 2> $O/attn_xcd_ab.err; cat $O/attn_xcd_ab.jsonl | cut -c1-700; tail -3 $O/attn_xcd_ab.err
timeout 600 python tools/bench_attn_parts.py > $O/attn_parts.jsonl 2> $O/attn_parts.err; cat $O/attn_parts.jsonl | cut -c1-600; tail -3 $O/attn_parts.err
for f in 1 2; do AFK_SILU_BWD_FORM=$f python tools/bench_silu_bwd.py 2>/dev/null | tail -1; done
timeout 1200 python -m pytest tests/test_fullwidth_gpu.py -q -x 2>&1 | tail -5 | cut -c1-400
