#!/bin/bash
# Cache-side counters of drk_layer_bf16 launches (tools/bench_layer.py, 2 timed + 3 warm-up launches per dilation):
# one --pmc pass per counter group.  usage: tools/pmc_layer.sh TAG [bench_layer args]
set -e
tag=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcl_${tag}_$i -o p -- python3 tools/bench_layer.py --reps 2 "$@" \
    > gpurun_out/pmcl_${tag}_$i.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(f'gpurun_out/pmcl_{tag}_*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_layer_bf16' in r['Kernel_Name']:
            per[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    for c, v in per.items():
        v.sort()
        n = len(v) // 4                              # launches per dilation
        for k, d in enumerate((1, 4, 16, 64)):
            last = v[k * n + n - 1]
            rows[d][c] = last[1]
            rows[d]['us_under_' + c] = last[2]
for d, r in rows.items():
    if 'FETCH_SIZE' in r:
        r['fetch_MB_corrected'] = 2 * r['FETCH_SIZE'] * 1024 / 1e6
    if 'WRITE_SIZE' in r:
        r['write_MB'] = r['WRITE_SIZE'] * 1024 / 1e6
    if 'TCP_TCC_READ_REQ_sum' in r:
        r['L2_read_requests_M'] = r['TCP_TCC_READ_REQ_sum'] / 1e6
json.dump(rows, open(f'gpurun_out/pmcl_{tag}_summary.json', 'w'), indent=1)
print(json.dumps(rows, indent=1))
PY
