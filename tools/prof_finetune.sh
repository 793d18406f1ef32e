#!/bin/bash
# rocprofv3 kernel stats of bench.py --workload finetune (tag = $1, batch = $2); summaries land in gpurun_out/
set -e
tag=${1:-x}; batch=${2:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --workload finetune --batch $batch --steps 3 --warmup 1 > gpurun_out/prof_${tag}_line.json 2> gpurun_out/prof_$tag.err
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
python3 - "$tag" <<'PY'
import csv, sys
t = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/{t}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.1f} ms over 4 passes -> {tot/4e6:.1f} ms per pass')
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  per pass {float(r['TotalDurationNs'])/4e6:7.2f} ms")
PY
