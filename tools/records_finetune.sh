#!/bin/bash
# The fine-tune records: bench lines, rocprofv3 kernel-trace summaries, pass and loop timings (tag = $1).
tag=${1:-r3_ft}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/$tag; mkdir -p $o
stats() {
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$name -o $name -- python3 "$@" > $o/${name}_line_under_rocprof.json 2> $o/prof_$name.err || return 1
  find $o/prof_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${name}_kernel_stats.csv
  rm -rf $o/prof_$name
}
timeout -k 10 200 python -u bench.py --workload finetune --steps 5 --warmup 2 > $o/bench_finetune_line.json 2> $o/ft.err || exit 1
timeout -k 10 200 python -u bench.py --workload finetune --batch 32 --steps 5 --warmup 2 > $o/bench_finetune_b32_line.json 2>> $o/ft.err || exit 1
stats bench_finetune bench.py --workload finetune --steps 3 --warmup 1 || exit 1
stats bench_finetune_b32 bench.py --workload finetune --batch 32 --steps 3 --warmup 1 || exit 1
timeout -k 10 200 python tools/bench_finetune.py --dtype bf16 --batch 128 --reps 3 --json $o/finetune_c3_b128.json > /dev/null 2>&1
timeout -k 10 200 python tools/bench_finetune.py --dtype bf16 --batch 32 --reps 3 --json $o/finetune_c3_b32.json > /dev/null 2>&1
timeout -k 10 200 python tools/bench_finetune.py --dtype bf16 --batch 128 --reps 4 --loop --json $o/finetune_loop_b128.json > /dev/null 2>&1
timeout -k 10 200 python tools/bench_finetune.py --dtype bf16 --batch 32 --reps 4 --loop --json $o/finetune_loop_b32.json > /dev/null 2>&1
grep -h "seconds_per_step\|total_s" $o/finetune_*.json
cut -c200-330 $o/bench_finetune_line.json; cut -c200-330 $o/bench_finetune_b32_line.json
