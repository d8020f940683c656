#!/bin/bash
# round 6, GPU call 1: MFMA power ceiling, stream-priority A/B (graph + eager), timeline with priorities, default bench line, GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c1; mkdir -p $O
python tools/measure_mfma_ceiling.py $O/mfma_power_ceiling.json > $O/ceiling.log 2>&1
tail -6 $O/ceiling.log
Q="--no-cpu-baseline --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --steps 10 --warmup 3"
for rep in 1 2; do
  for pr in 0 1; do
    AFK_STREAM_PRIORITIES=$pr python bench.py $Q --detail-name r06c1/ab_graph_p${pr}_$rep.json 2>$O/ab_graph_p${pr}_$rep.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph prio=$pr rep=$rep', d['ms_per_step'], d.get('stream_priorities'))"
  done
done
for pr in 0 1; do
  AFK_STREAM_PRIORITIES=$pr python bench.py $Q --no-graph --detail-name r06c1/ab_eager_p${pr}.json 2>$O/ab_eager_p${pr}.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager prio=$pr', d['ms_per_step'])"
done
AFK_STREAM_PRIORITIES=1 bash tools/prof_bench.sh r06_prio1 --no-eager-baseline --no-long-audio --no-extra-legs --no-parity
AFK_STREAM_PRIORITIES=1 bash tools/prof_bench.sh r06_prio1_eager --no-eager-baseline --no-long-audio --no-extra-legs --no-parity --no-graph
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err
tail -1 $O/bench_default.out | wc -c
tail -1 $O/bench_default.out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
