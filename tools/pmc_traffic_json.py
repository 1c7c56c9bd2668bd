#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh.
Usage: tools/pmc_traffic_json.py gpurun_out/pmc_<tag> "<commit note>" [B T H W [dtype [calibration dir]]] > profiles/rNN_pmc_traffic.json
(dtype f32 | bf16 | bf16s = bench.py --dtype; the calibration passes default to the step directory)"""
import json
import sys


def total(path):
    for line in open(path):
        p = line.split('\t')
        if p[0] == 'TOTAL':
            return float(p[2]) * 1024.0
    raise SystemExit(f'no TOTAL in {path}')


def main():
    d, note = sys.argv[1], sys.argv[2]
    B, T, H, W = (int(x) for x in sys.argv[3:7]) if len(sys.argv) >= 7 else (256, 4, 90, 120)
    dtype = sys.argv[7] if len(sys.argv) >= 8 else 'f32'
    cal = sys.argv[8] if len(sys.argv) >= 9 else d
    true_bytes = 2 * 1024 ** 3 * 10.0                      # tools/pmc_calibrate.py: cdrl_gather_rows 2 GiB x 10 per direction
    cf, cw = total(f'{cal}/cal_FETCH_SIZE.txt'), total(f'{cal}/cal_WRITE_SIZE.txt')
    fc, wc = true_bytes / cf, true_bytes / cw
    steps = 3                                              # bench.py --steps 2 --warmup 1
    rd, wr = total(f'{d}/step_FETCH_SIZE.txt') * fc / steps, total(f'{d}/step_WRITE_SIZE.txt') * wc / steps
    alg = 2 * 3 * (2 if dtype == 'bf16s' else 4) * B * T * {(90, 120): 979500, (90, 360): 2902212, (135, 180): 2275248}[(H, W)]
    print(json.dumps(dict(workload=dict(B=B, T=T, H=H, W=W, dtype=dtype), update_steps_profiled=steps,
                          calibration=dict(workload='cdrl_gather_rows 2 GiB x 10 per direction (16-byte lanes)', true_bytes_per_direction=true_bytes,
                                           FETCH_SIZE_bytes=cf, WRITE_SIZE_bytes=cw, fetch_correction=round(fc, 4), write_correction=round(wc, 4)),
                          read_bytes_per_update_step=rd, write_bytes_per_update_step=wr, bytes_per_update_step=rd + wr,
                          algorithmic_bytes_per_update_step=alg, ratio_to_algorithmic=round((rd + wr) / alg, 3),
                          method='rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py --steps 2 '
                                 '--warmup 1; sum over libcdrl kernels / 3 update-steps; FETCH_SIZE x2 gfx950 correction '
                                 '(MI355X_MICROARCH.md HBM section) confirmed by the calibration run of the same session; exact for '
                                 '16-byte-lane streams, an upper bound for narrower accesses', commit=note), indent=1))


if __name__ == '__main__':
    main()
