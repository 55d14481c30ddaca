#!/bin/bash
# Dev (GPU box): kernel trace of a few bench steps -> timeline of the last step (gpurun_out/<tag>_timeline.txt).
#   bash tools/trace_step.sh <tag> [bench args...]
set -e
TAG=${1:-step}; shift || true
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT.log 2>&1
python3 tools/prof_db.py $OUT --timeline > gpurun_out/${TAG}_timeline.txt
python3 tools/prof_queues.py $OUT > gpurun_out/${TAG}_queues.txt 2>&1 || true
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete; find $OUT -type d -empty -delete
tail -12 gpurun_out/${TAG}_timeline.txt
