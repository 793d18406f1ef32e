#!/bin/bash
# HBM-side and cache-side counters of the grouped long-sequence forward's launches (tools/bench_group.py --pmc: three
# forwards at B=256, L=1024), one --pmc pass per counter group.  usage: tools/pmc_group.sh TAG
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcg_${tag}_$i -o p -- python3 tools/bench_group.py --pmc \
    > gpurun_out/pmcg_${tag}_$i.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(f'gpurun_out/pmcg_{tag}_*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_group_bf16' in r['Kernel_Name']:
            per[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(
                (int(r['Dispatch_Id']), float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    for (k, c), v in per.items():
        v.sort()
        rows[k][c] = v[-1][1]                       # the last forward's last launch of this kernel
        rows[k]['us_under_' + c] = v[-1][2]
for k, r in rows.items():
    if 'FETCH_SIZE' in r:
        r['fetch_MB_corrected'] = 2 * r['FETCH_SIZE'] * 1024 / 1e6      # MI355X_MICROARCH.md: gfx950 reports half of a wide streaming read
    if 'WRITE_SIZE' in r:
        r['write_MB'] = r['WRITE_SIZE'] * 1024 / 1e6
    if 'TCP_TCC_READ_REQ_sum' in r:
        r['L2_read_MB_at_128B'] = r['TCP_TCC_READ_REQ_sum'] * 128 / 1e6
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and r.get('GRBM_GUI_ACTIVE'):
        # GRBM_GUI_ACTIVE sums the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES the 1024 SIMDs (as tools/pmc_mfma.sh reads them)
        cyc = r['GRBM_GUI_ACTIVE'] / 8
        r['clock_GHz_under_pmc'] = cyc / (r['us_under_GRBM_GUI_ACTIVE'] * 1e-6) / 1e9
        r['mfma_pipe_utilisation'] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc
        r['lds_array_busy_frac'] = r.get('SQ_LDS_IDX_ACTIVE', 0.0) / 256.0 / cyc
json.dump(rows, open(f'gpurun_out/pmcg_{tag}_summary.json', 'w'), indent=1)
print(json.dumps(rows, indent=1))
PY
