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
