#!/bin/bash
# MFMA-pipe occupancy of ONE k_sample_bf16 launch (B=1024, 16 steps + noise removal): its own --pmc pass.
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_mfma -o p -- python3 tools/pmc_run.py 1024 16 \
  > gpurun_out/pmc_${tag}_mfma.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
out = {}
for f in glob.glob(f'gpurun_out/pmc_{tag}_mfma/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_sample_bf16' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            out['duration_ms'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
            out['scratch_bytes_per_lane'] = int(r['Scratch_Size']); out['vgpr'] = int(r['VGPR_Count'])
if 'GRBM_GUI_ACTIVE' in out:
    clk = out['GRBM_GUI_ACTIVE'] / 8 / (out['duration_ms'] * 1e-3) / 1e9
    per_simd = out['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0
    out['clock_GHz_under_pmc'] = clk
    out['mfma_busy_cycles_per_SIMD'] = per_simd
    out['mfma_pipe_utilisation'] = per_simd / (clk * 1e9 * out['duration_ms'] * 1e-3)
json.dump(out, open(f'gpurun_out/pmc_{tag}_mfma_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
