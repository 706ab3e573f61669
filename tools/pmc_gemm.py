#!/usr/bin/env python3
"""Per-launch PMC averages of the prefill tile GEMM (csrc/gemm8.hip) from one or more `rocprofv3 --pmc ... --kernel-trace -f csv` passes
(tools/run_prefill_once.py as the command), plus the derived matrix-pipe utilisation:
    SQ_VALU_MFMA_BUSY_CYCLES [cycles, summed over SIMDs] / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs).
usage: python tools/pmc_gemm.py out.json pass1_counter_collection.csv [pass2.csv ...]"""
import collections
import csv
import json
import sys

import os
KERNEL = os.environ.get('PMC_KERNEL', 'gemm8_kernel')      # substring of the kernel name (PMC_KERNEL=stripe_gemm_kernel for csrc/stripe_mm.inc)
acc = collections.defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if KERNEL in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
out = {'kernel': ('gptq::gemm8_kernel<false, false> (csrc/gemm8.hip), M = 16384, K = N = 4096 (tools/run_prefill_once.py)' if KERNEL == 'gemm8_kernel' else
                  KERNEL + ' (' + os.environ.get('PMC_NOTE', 'tools/run_prefill_once.py') + ')'),
       'note': 'rocprofv3 --pmc, separate passes; SQ_*_CYCLES in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles); GRBM_GUI_ACTIVE summed over 8 XCDs'}
for k, v in sorted(acc.items()):
    out[k] = sum(v) / len(v)
if 'SQ_VALU_MFMA_BUSY_CYCLES' in out and 'GRBM_GUI_ACTIVE' in out:
    out['mfma_pipe_utilisation'] = round(out['SQ_VALU_MFMA_BUSY_CYCLES'] / (out['GRBM_GUI_ACTIVE'] / 8 * 256 * 4), 4)
if 'SQ_WAVE_CYCLES' in out:
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
        if k in out:
            out[k + '_frac_of_wave_cycles'] = round(out[k] / out['SQ_WAVE_CYCLES'], 4)
if 'SQ_LDS_BANK_CONFLICT' in out and 'SQ_LDS_IDX_ACTIVE' in out:
    out['lds_bank_conflict_frac'] = round(out['SQ_LDS_BANK_CONFLICT'] / max(out['SQ_LDS_IDX_ACTIVE'], 1), 4)
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1], 'w'), indent=1)
