#!/bin/bash
# per-kernel durations of the attention microbenchmark (rocprofv3 kernel trace -> markdown table)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_attn
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o attn -- python $GRAFT_REPO_ROOT/tools/bench_attn.py > $OUT/run.log 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $OUT/stats.md > /dev/null 2>&1
grep -i "lds_kernel\|delta\|gqa" $OUT/stats.md
