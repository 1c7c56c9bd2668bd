#!/usr/bin/env python3
"""Largest idle gaps on the critical stream of the last update-step of a tools/timeline_bench.sh trace.
usage: tools/timeline_gaps.py gpurun_out/tl_<tag>/timeline.tsv <kernels per update-step> [queue id = 2]"""
import sys

rows = [l.rstrip('\n').split('\t') for l in open(sys.argv[1])][1:]
n = int(sys.argv[2])
q = sys.argv[3] if len(sys.argv) > 3 else '2'
R = [(float(r[0]), float(r[1]), r[3], r[6]) for r in rows][-n:]
main = [r for r in R if r[2] == q]
t0 = R[0][0]
gaps = [(b[0] - (a[0] + a[1]), a[0] - t0, a[3][:48], b[3][:48]) for a, b in zip(main, main[1:])]
print('queue', q, 'kernels', len(main), 'idle %.2f ms' % (sum(g for g, _, _, _ in gaps if g > 0) / 1e3), 'of %.2f ms' % ((main[-1][0] + main[-1][1] - main[0][0]) / 1e3),
      '(gaps >= 100 us: %.2f ms)' % (sum(g for g, _, _, _ in gaps if g >= 100) / 1e3))
for g, t, a, b in sorted(gaps, reverse=True)[:8]:
    print('%8.1f us at %7.2f ms  %s -> %s' % (g, t / 1e3, a, b))
