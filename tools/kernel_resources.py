#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (stderr file) per kernel:
VGPRs / AGPRs / spills / occupancy.  Usage: tools/kernel_resources.py <remarks.txt> [name filter]"""
import re
import subprocess
import sys


def main():
    t = open(sys.argv[1]).read()
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    names, rows = [], []
    for b in re.split(r'remark: Function Name: ', t)[1:]:
        name = b.split()[0]
        g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
        names.append(name)
        rows.append((g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')))
    if not names:
        print('no kernel-resource remarks in', sys.argv[1])
        return
    dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=60).stdout.splitlines()
    for n, r in zip(dem, rows):
        if filt in n:
            print(f'{n[:90]:90s} vgpr {r[0]:3d} agpr {r[1]:3d} spill {r[2]:3d} scratch {r[3]:4d} occ {r[4]}')


if __name__ == '__main__':
    main()
