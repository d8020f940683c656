#!/bin/bash
# round 6, GPU call 7: LDS-DMA wait probe of the forward (probe build), per-kernel trace with the new defaults, attention tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c7; mkdir -p $O
timeout 600 python tools/probes/probe_attn_dma.py 2>&1 | tee $O/probe_attn_dma.jsonl | cut -c1-400
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "attention" 2>&1 | tail -4 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in dec enc long5; do
    OUT=$R/gpurun_out/prof_attn_new_$shape; rm -rf $OUT; mkdir -p $OUT
    reps=10; [ $shape = long5 ] && reps=4
    rocprofv3 --kernel-trace -d $OUT -o attn -- python $R/tools/one_attn.py $shape $reps > $OUT/run.log 2>&1
    DB=$(find $OUT -name "*.db" | head -1)
    python $R/tools/rocpd_stats.py $DB $OUT/stats.md > /dev/null 2>&1
    echo "== new defaults $shape"; grep -i "lds_kernel\|delta\|gqa" $OUT/stats.md | cut -c1-200
done
