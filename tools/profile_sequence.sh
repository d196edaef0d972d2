#!/bin/bash
# rocprofv3 kernel trace of the whole synthetic drive (tools/sequence.py) and of the single-view loops; summaries -> gpurun_out/<tag>/
# usage: tools/profile_sequence.sh <tag>
set -u
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/sequence.py --frames 60 > $OUT/sequence_kitti07.json 2> $OUT/sequence_kitti07.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_seq -o seq -- python3 $ROOT/tools/sequence.py --frames 60 > $OUT/sequence_kitti07_under_rocprof.json 2> $OUT/prof_seq.err
f=$(find $OUT/prof_seq -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats_sequence_kitti07.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_loops -o loops -- python3 $ROOT/tools/single_view_loops.py both > $OUT/single_view_loops_under_rocprof.json 2> $OUT/prof_loops.err
f=$(find $OUT/prof_loops -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats_single_view_loops.csv
python3 $ROOT/tools/single_view_loops.py both > $OUT/single_view_loops.json 2> $OUT/single_view_loops.err
rm -rf $OUT/prof_seq $OUT/prof_loops
ls -la $OUT
# the opaque-surface tracking iteration, one-level and two-level grouping (kernel traces) and the surface mapping window A/B
for M in 0 auto; do
  LVDGS_SUPER_TILES=$M rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_surface_$M -o s -- python3 $ROOT/bench.py --workload surface_100k_1920x1080 --steps 100 --warmup 60 --no-cpu-baseline --no-side > $OUT/bench_surface_super_$M.json 2> $OUT/prof_surface_$M.err
  f=$(find $OUT/prof_surface_$M -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_surface_100k_1920x1080_super_$M.csv
  rm -rf $OUT/prof_surface_$M
  LVDGS_SUPER_TILES=$M python3 $ROOT/tools/window_bench.py surface_100k_1920x1080 2>/dev/null | sed "s/^/super=$M /" >> $OUT/window_surface_ab.txt
  LVDGS_SUPER_TILES=$M python3 $ROOT/tools/track_ab.py surface_100k_1920x1080 kitti07_geom cfg3_500k_1920x1080 2>/dev/null | sed "s/^/super=$M /" >> $OUT/track_ab.txt
done
cat $OUT/window_surface_ab.txt $OUT/track_ab.txt
