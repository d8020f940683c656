#!/bin/bash
# round 6, GPU call 2: priority-mode variants, parity calibration (peaked weights, long train leg), DP pre-flight/probe on one GPU, dp tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c2; mkdir -p $O
Q="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 10 --warmup 3"
for pr in h0 0l hl 0; do
  AFK_STREAM_PRIORITIES=$pr python bench.py $Q --no-graph --detail-name r06c2/ab_eager_$pr.json 2>$O/ab_eager_$pr.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager prio=$pr', d['ms_per_step'])"
done
for pr in h0 0l; do
  AFK_STREAM_PRIORITIES=$pr python bench.py $Q --detail-name r06c2/ab_graph_$pr.json 2>$O/ab_graph_$pr.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph prio=$pr', d['ms_per_step'])"
done
python tools/parity_fulldepth.py --out $O/parity.json > $O/parity_summary.json 2> $O/parity.err; echo "parity rc=$?"
tail -12 $O/parity.err
python - <<'PY'
import json
s=json.load(open('gpurun_out/r06c2/parity_summary.json'))
print(json.dumps({k:s.get(k) for k in ('green','checks','peaked','long5min_train','attention_peak_fp32_reference')}))
PY
python bench.py --force-dp $Q --detail-name r06c2/force_dp.json 2>$O/force_dp.err | tail -1
grep "\[bench\]" $O/force_dp.err | tail -4
timeout 1200 python -m pytest tests/test_dp_gpu.py -x -q 2>&1 | tail -5
