#!/bin/bash
# round 6, GPU call 5: attention schedule A/B (bit-equality test, event timing, per-kernel trace), peaked sweep 6 / 8, the rest of the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c5; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "attention" 2>&1 | tail -8 | cut -c1-300
timeout 900 python tools/bench_attn_sched.py > $O/attn_sched_ab.jsonl 2> $O/attn_sched_ab.err; cat $O/attn_sched_ab.jsonl | cut -c1-600; tail -3 $O/attn_sched_ab.err
bash tools/runs/attn_sched_prof.sh > $O/attn_sched_prof.txt 2>&1; cat $O/attn_sched_prof.txt | cut -c1-220
for q in 6 8; do
  AFK_PEAK_ENC_QK=$q python tools/parity_fulldepth.py --only-peaked --out $O/peaked_$q.json 2> $O/peaked_$q.err | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('enc_qk $q', r['attention_peak'], r['qk'], 'noise', r['noise_dominated'], 'over', len(r['over_bar']), r['ours_over_floor'])"
done
timeout 3000 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log | tail -3; grep "^FAILED" $O/gpu_suite.log | cut -c1-200
