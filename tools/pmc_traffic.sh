#!/bin/bash
# HBM-side traffic of ONE full k_sample_bf16 launch (B=1024, 128 steps + noise removal): separate --pmc passes,
# one counter group per run (MI355X_MICROARCH.md, "rocprofv3 PMC slots"); CSVs land in gpurun_out/pmc_<tag>_*/.
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo "$c" | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$name -o p -- python3 tools/pmc_run.py 1024 128 \
    > gpurun_out/pmc_${tag}_$name.log 2>&1
done
python3 tools/pmc_summarize.py $tag
