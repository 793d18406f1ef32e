#!/bin/bash
# rocprofv3 kernel-trace summary of the captured configs[4] run (tag = $1).
tag=${1:-r3_long}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/$tag; mkdir -p $o
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o long -- python3 bench.py --workload long --steps 1 --warmup 1 > $o/long_line_under_rocprof.json 2> $o/prof.err || exit 1
find $o/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/long_kernel_stats.csv
rm -rf $o/prof
head -12 $o/long_kernel_stats.csv | cut -c1-180
