#!/bin/bash
# rocprofv3 kernel trace of the default bench (full AF3-7B) -> gpurun_out/prof_$1/{stats.md,timeline.md}
NAME=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/run.log 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $OUT/stats.md ${PROF_STEPS:+--steps $PROF_STEPS} > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $DB $OUT/timeline.md
tail -1 $OUT/run.log | cut -c1-200
rm -f $DB   # 64 MiB cap on gpurun_out
