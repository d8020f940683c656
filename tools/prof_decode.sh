#!/bin/bash
# per-kernel durations of the decode benchmark (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_decode
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o dec -- python $GRAFT_REPO_ROOT/tools/bench_decode.py ${1:-1} > $OUT/run.log 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $OUT/stats.md > /dev/null 2>&1
head -22 $OUT/stats.md | cut -c1-150
tail -1 $OUT/run.log | cut -c1-300
rm -f $DB
