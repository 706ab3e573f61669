#!/usr/bin/env python3
"""Turn a `rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv` run of bench.py into per-launch HBM
traffic for the GEMV kernels, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes: FETCH_SIZE is reported in KiB and, on gfx950, counts exactly half of the bytes of a
wide (16 B/lane) coalesced streaming read -> bytes = FETCH_SIZE * 1024 * 2.
usage: python tools/pmc_traffic.py <counter_collection.csv> [out.json]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in rows:
    if ('stripe_gemv' in r['Kernel_Name'] or 'gemv_rowwave' in r['Kernel_Name']) and r['Counter_Name'] == 'FETCH_SIZE':
        per[(r['Kernel_Name'].split('(')[0][:72], int(r['Grid_Size']))].append(float(r['Counter_Value']))
out = {'counter': 'FETCH_SIZE', 'unit_correction': 'KiB * 1024 * 2 (gfx950 wide-load correction, MI355X_MICROARCH.md HBM)',
       'kernels': [], }
tot_b, tot_n = 0.0, 0
for (name, grid), v in sorted(per.items()):
    b = sum(v) / len(v) * 1024 * 2
    out['kernels'].append({'kernel': name, 'grid_threads': grid, 'launches': len(v), 'hbm_bytes_per_launch': round(b)})
    tot_b += b * len(v)
    tot_n += len(v)
out['hbm_bytes_per_launch_avg'] = round(tot_b / max(tot_n, 1))
try:        # the kernel sources this pass was taken on: bench.py refuses a pass of another build (roofline.traffic_source)
    from bench import csrc_sha16
    out['csrc_sha16'] = csrc_sha16()
except Exception as e:
    out['csrc_sha16_error'] = repr(e)[:100]
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
