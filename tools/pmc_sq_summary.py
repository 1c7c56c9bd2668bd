#!/usr/bin/env python3
"""Per-kernel SQ issue / stall picture from the `rocprofv3 --pmc` passes of tools/pmc_sq.sh.

Usage: tools/pmc_sq_summary.py gpurun_out/pmc_sq_<tag> [bench args]  > sq.json

Derived figures (gfx94x formulas of ROCm's derived_counters.xml; ROCm 7.2 ships no gfx950 section -- MI355X_MICROARCH.md
"rocprofv3 PMC slots").  As collected here every SQ counter is summed over the whole chip and GRBM_GUI_ACTIVE over the 8 XCDs:
  valu_busy   = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)     share of SIMD cycles issuing VALU (MFMA included)
  lds_busy    = SQ_ACTIVE_INST_LDS x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)      ... issuing LDS instructions
  lds_array   = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256 CUs)              share of cycles the LDS arrays are indexed
  bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                       extra LDS-array cycles per LDS-array cycle
  wait_inst_any, wait_any, active_inst_any: / SQ_WAVE_CYCLES                    (disjoint: issue stall, parked on s_waitcnt / barrier, issuing)
  occupancy   = SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)          resident waves per SIMD, averaged over the kernel
  valu_per_wave, lds_per_wave, vmem_per_wave: instructions per wave
The kernels of the engine's streams overlap in time, GRBM_GUI_ACTIVE of a dispatch counts the whole chip: per-kernel busy shares are
lower bounds wherever another stream's kernel was resident too."""
import glob
import json
import os
import sys

out = sys.argv[1]
ctr = {}
for path in sorted(glob.glob(os.path.join(out, 'step_*.txt'))):
    for line in open(path):
        p = line.rstrip('\n').split('\t')
        if len(p) == 4 and p[0] != 'TOTAL':
            ctr.setdefault(p[1], {})[p[0]] = (int(p[2]), float(p[3]))


def g(v, k):
    return v.get(k, (0, 0.0))[1]


rows = []
for k, v in ctr.items():
    act = g(v, 'GRBM_GUI_ACTIVE')
    if not act:
        continue
    calls = v['GRBM_GUI_ACTIVE'][0]
    simd = act / 8.0 * 1024.0
    wc = g(v, 'SQ_WAVE_CYCLES')
    waves = g(v, 'SQ_WAVES')
    idx = g(v, 'SQ_LDS_IDX_ACTIVE')

    def r(x, d, nd=4):
        return round(x / d, nd) if d else None

    rows.append(dict(
        kernel=k, calls=calls, gui_active_cycles_per_call=round(act / 8.0 / calls),
        waves_per_call=r(waves, calls, 0), occupancy_waves_per_simd=r(wc * 4.0, simd, 3),
        valu_busy=r(g(v, 'SQ_ACTIVE_INST_VALU') * 4.0, simd), lds_busy=r(g(v, 'SQ_ACTIVE_INST_LDS') * 4.0, simd),
        vmem_busy=r(g(v, 'SQ_ACTIVE_INST_VMEM') * 4.0, simd), salu_busy=r(g(v, 'SQ_ACTIVE_INST_SCA') * 4.0, simd),
        lds_array_busy=r(idx, act / 8.0 * 256.0), lds_bank_conflict=r(g(v, 'SQ_LDS_BANK_CONFLICT'), idx),
        wait_inst_any=r(g(v, 'SQ_WAIT_INST_ANY'), wc), wait_inst_lds=r(g(v, 'SQ_WAIT_INST_LDS'), wc), wait_any=r(g(v, 'SQ_WAIT_ANY'), wc),
        active_inst_any=r(g(v, 'SQ_ACTIVE_INST_ANY'), wc),
        valu_per_wave=r(g(v, 'SQ_INSTS_VALU'), waves, 1), salu_per_wave=r(g(v, 'SQ_INSTS_SALU'), waves, 1),
        lds_per_wave=r(g(v, 'SQ_INSTS_LDS'), waves, 1),
        vmem_rd_per_wave=r(g(v, 'SQ_INSTS_VMEM_RD'), waves, 1), vmem_wr_per_wave=r(g(v, 'SQ_INSTS_VMEM_WR'), waves, 1),
        total_gui_active=act))
rows.sort(key=lambda x: -x['total_gui_active'])
print(json.dumps(dict(
    method='rocprofv3 --kernel-trace --pmc <<= 4 counters> in separate passes over bench.py --steps 2 --warmup 1 (3 update-steps); '
           'formulas in tools/pmc_sq_summary.py; kernels sorted by their share of GRBM_GUI_ACTIVE (device time); SQ_WAVE_CYCLES, '
           'SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave',
    bench_args=sys.argv[2:], kernels=rows[:30]), indent=1))
