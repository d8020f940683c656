#!/bin/bash
# round 6, GPU call 9: PMC table of the attention kernels (new defaults), default bench line, serial + overlapped kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c9; mkdir -p $O
timeout 900 python tools/measure_attn_pmc.py $O/attn_pmc.md dec enc long5 > $O/attn_pmc.log 2>&1; cat $O/attn_pmc.md | cut -c1-400
timeout 900 python bench.py > $O/bench_default.out 2> $O/bench_default.err; tail -1 $O/bench_default.out | wc -c; tail -1 $O/bench_default.out | cut -c1-1200
cp gpurun_out/bench_detail.json $O/bench_detail.json
