#!/usr/bin/env python3
"""Per-kernel and per-update-step MFMA utilisation from the `rocprofv3 --pmc` passes of tools/pmc_mfma.sh.

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)   (the gfx94x derived-metric formula;
ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots").  As collected here SQ_VALU_MFMA_BUSY_CYCLES is
summed over all 1024 SIMDs (it equals 64 cycles x SQ_INSTS_VALU_MFMA_F32 for v_mfma_f32_32x32x2_f32, checked on every kernel)
and GRBM_GUI_ACTIVE is summed over the 8 XCDs (8 x the dispatch's busy cycles), so the ratio is clock-independent.
SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = FLOPs issued to the matrix pipe, padding included (1 fp32 32x32x2 MFMA = 8 MOPS)."""
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
c = load('SQ_INSTS_VALU_MFMA_MOPS_BF16')        # x3 split-precision kernels and the bf16-operand mode (1 MOPS = 512 FLOPs as well)
kern = []
tot_busy = tot_active = tot_mops = tot_mops16 = 0.0
for k, v in a.items():
    busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0.0))[1]
    act = v.get('GRBM_GUI_ACTIVE', (0, 0.0))[1]
    calls = v.get('GRBM_GUI_ACTIVE', (0, 0.0))[0]
    mops = b.get(k, {}).get('SQ_INSTS_VALU_MFMA_MOPS_F32', (0, 0.0))[1]
    insts = b.get(k, {}).get('SQ_INSTS_VALU_MFMA_F32', (0, 0.0))[1]
    mops16 = c.get(k, {}).get('SQ_INSTS_VALU_MFMA_MOPS_BF16', (0, 0.0))[1]
    tot_mops16 += mops16
    tot_busy += busy
    tot_active += act
    tot_mops += mops
    if busy > 0:
        kern.append(dict(kernel=k, calls=calls, mfma_busy_cycles=busy, gui_active_cycles=act,
                         mfma_util=round(busy / (act / 8.0 * 1024.0), 4) if act else None, mfma_flops_issued=mops * 512.0, mfma_insts_f32=insts,
                         mfma_bf16_flops_issued=mops16 * 512.0))
kern.sort(key=lambda r: -r['mfma_busy_cycles'])
print(json.dumps(dict(method='rocprofv3 --kernel-trace --pmc <counters> in separate passes over bench.py --steps 2 --warmup 1 '
                             '(3 update-steps); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), summed over '
                             'the libcdrl kernels; kernels of the two streams overlap in time, so the step figure is a lower bound '
                             'of the utilisation while an MFMA kernel is resident',
                      bench_args=sys.argv[2:], mfma_util_all_kernels=round(tot_busy / (tot_active / 8.0 * 1024.0), 4) if tot_active else None,
                      mfma_flops_issued_per_update_step=tot_mops * 512.0 / 3.0,
                      mfma_bf16_flops_issued_per_update_step=tot_mops16 * 512.0 / 3.0,
                      mfma_busy_cycles=tot_busy, gui_active_cycles=tot_active, mfma_mops_f32=tot_mops, kernels=kern[:25]), indent=1))
