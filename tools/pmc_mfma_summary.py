#!/usr/bin/env python3
"""Per-kernel and per-update-step MFMA utilisation from the `rocprofv3 --pmc` passes of tools/pmc_mfma.sh.

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)   (the gfx94x derived-metric formula; ROCm 7.2
ships no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots").  GRBM_GUI_ACTIVE of a dispatch = its busy cycles at the
clock it ran at, so the ratio is clock-independent.  SQ_INSTS_VALU_MFMA_MOPS_F32 counts fp32 matrix operations in units of
512 FLOP-pairs... recorded raw; the FLOP figure used for TFLOP/s is the algorithmic one."""
import json
import sys
import os
import re

out = sys.argv[1]


def load(name):
    rows = {}
    path = os.path.join(out, f'step_{name}.txt')
    if not os.path.exists(path):
        return rows
    for line in open(path):
        p = line.rstrip('\n').split('\t')
        if len(p) == 4 and p[0] != 'TOTAL':
            rows.setdefault(p[1], {})[p[0]] = (int(p[2]), float(p[3]))
    return rows


a = load('SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE')
b = load('SQ_INSTS_VALU_MFMA_MOPS_F32+SQ_INSTS_VALU_MFMA_F32')
kern = []
tot_busy = tot_active = tot_mops = 0.0
for k, v in a.items():
    busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0.0))[1]
    act = v.get('GRBM_GUI_ACTIVE', (0, 0.0))[1]
    calls = v.get('GRBM_GUI_ACTIVE', (0, 0.0))[0]
    mops = b.get(k, {}).get('SQ_INSTS_VALU_MFMA_MOPS_F32', (0, 0.0))[1]
    insts = b.get(k, {}).get('SQ_INSTS_VALU_MFMA_F32', (0, 0.0))[1]
    tot_busy += busy
    tot_active += act
    tot_mops += mops
    if busy > 0:
        kern.append(dict(kernel=k, calls=calls, mfma_busy_cycles=busy, gui_active_cycles=act,
                         mfma_util=round(busy / (act * 1024.0), 4) if act else None, mfma_mops_f32=mops, mfma_insts_f32=insts))
kern.sort(key=lambda r: -r['mfma_busy_cycles'])
print(json.dumps(dict(method='rocprofv3 --kernel-trace --pmc <counters> in separate passes over bench.py --steps 2 --warmup 1 '
                             '(3 update-steps); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs), summed over '
                             'the libcdrl kernels; kernels of the two streams overlap in time, so the step figure is a lower bound '
                             'of the utilisation while an MFMA kernel is resident',
                      bench_args=sys.argv[2:], mfma_util_all_kernels=round(tot_busy / (tot_active * 1024.0), 4) if tot_active else None,
                      mfma_busy_cycles=tot_busy, gui_active_cycles=tot_active, mfma_mops_f32=tot_mops, kernels=kern[:25]), indent=1))
