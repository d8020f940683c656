#!/bin/bash
# per-kernel durations of the attention kernels under both schedules (rocprofv3 kernel trace of tools/one_attn.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sch in 0 1; do
  for shape in dec enc long5; do
    OUT=$R/gpurun_out/prof_attn_s${sch}_$shape; rm -rf $OUT; mkdir -p $OUT
    reps=10; [ $shape = long5 ] && reps=4
    AFK_ATTN_SCHED=$sch rocprofv3 --kernel-trace -d $OUT -o attn -- python $R/tools/one_attn.py $shape $reps > $OUT/run.log 2>&1
    DB=$(find $OUT -name "*.db" | head -1)
    python $R/tools/rocpd_stats.py $DB $OUT/stats.md > /dev/null 2>&1
    echo "== sched $sch $shape"; grep -i "lds_kernel\|delta\|gqa" $OUT/stats.md | cut -c1-200
  done
done
