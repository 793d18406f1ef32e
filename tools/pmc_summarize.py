"""Per-launch HBM-side traffic of k_sample_bf16 from the --pmc passes of tools/pmc_traffic.sh.

FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies
the 128-B requests of 16-B-per-lane loads at 64 B, so the fetched bytes are 2 x the reported value for this kernel,
whose global reads are all 16-B-per-lane fragment loads; WRITE_SIZE is exact."""
import csv
import glob
import json
import sys

tag = sys.argv[1]
out = {'what': 'rocprofv3 --pmc, one counter group per run, over ONE k_sample_bf16 launch: B=1024, L=200, 128 steps + '
               'noise removal = 129 denoiser forwards (python3 tools/pmc_run.py 1024 128)'}
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'k_sample_bf16' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            out['duration_ms_' + r['Counter_Name']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
            out['scratch_bytes_per_lane'] = int(r['Scratch_Size'])
if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
    fetch = 2.0 * out['FETCH_SIZE'] * 1024.0
    write = out['WRITE_SIZE'] * 1024.0
    out['fetch_bytes_corrected'] = fetch
    out['write_bytes'] = write
    out['traffic_bytes_per_launch'] = fetch + write
    out['traffic_TBps'] = (fetch + write) / (out['duration_ms_FETCH_SIZE'] * 1e-3) / 1e12
json.dump(out, open(f'gpurun_out/pmc_{tag}_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
