#!/bin/bash
# rocprofv3 kernel stats of tools/bench_finetune.py --loop (tag = $1, batch = $2)
set -e
tag=${1:-x}; batch=${2:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 tools/bench_finetune.py --dtype bf16 --batch $batch --reps 3 --loop > gpurun_out/prof_${tag}_out.txt 2> gpurun_out/prof_$tag.err
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
tail -1 gpurun_out/prof_${tag}_out.txt | cut -c1-300
python3 - "$tag" <<'PY'
import csv, sys
t = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/{t}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.1f} ms over 4 steps -> {tot/4e6:.1f} ms per step')
for r in rows[:30]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  per step {float(r['TotalDurationNs'])/4e6:7.2f} ms")
PY
