#!/bin/bash
# round 6, GPU call 4: DynamicCache test, peaked-weights sweep of the encoder q/k scale, full GPU suite, default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c4; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "dynamic_cache or past_key_values" 2>&1 | tail -8 | cut -c1-400
for q in 1.5 2.0 2.8 3.5; do
  AFK_PEAK_ENC_QK=$q python tools/parity_fulldepth.py --only-peaked --out $O/peaked_$q.json 2> $O/peaked_$q.err | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('enc_qk $q', r['attention_peak'], r['qk'], 'noise', r['noise_dominated'], 'over', len(r['over_bar']), r['ours_over_floor'])"
done
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log | cut -c1-400
timeout 900 python bench.py > $O/bench_default.out 2> $O/bench_default.err; tail -1 $O/bench_default.out | wc -c; tail -1 $O/bench_default.out | cut -c1-3900
