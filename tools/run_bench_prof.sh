#!/bin/bash
# bench.py plain, then under rocprofv3 --kernel-trace --stats; summaries land in gpurun_out/ (tag = $1)
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -u bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_${tag}_prof.json 2> gpurun_out/prof_$tag.err
python3 - "$tag" <<'PY'
import json, sys
t = sys.argv[1]
d = json.load(open(f'gpurun_out/bench_{t}.json'))
print('plain  :', d['value'], 'seq/s', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'TF/s', d['roofline']['frac'], 'launch', d['roofline']['avg_launch_ms'], 'cpu', d['cpu_baseline']['value'])
d = json.load(open(f'gpurun_out/bench_{t}_prof.json'))
print('rocprof:', d['value'], 'seq/s', 'launch', d['roofline']['avg_launch_ms'])
PY
head -2 gpurun_out/prof_$tag/${tag}_kernel_stats.csv | cut -c1-150
