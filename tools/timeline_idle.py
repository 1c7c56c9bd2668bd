#!/usr/bin/env python3
"""Itemises the idle time of the critical stream over exactly one update-step of a tools/timeline_bench.sh trace.
usage: tools/timeline_idle.py gpurun_out/tl_<tag>/timeline.tsv [queue = 2]
Gap classes: boundary (< 2.6 us: a dependent kernel boundary), bubble (2.6-12 us: a cross-stream event record / wait in front of the
next kernel), hole (>= 12 us: the stream waits for another stream or for the host)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1]), delimiter='\t'))
q = sys.argv[2] if len(sys.argv) > 2 else '2'
idx = [i for i, r in enumerate(rows) if 'stem_fwd' in r['kernel'] and 'stem_bwd' not in r['kernel']]
sub = [r for r in rows[idx[-3]:idx[-1]] if r['queue'] == q]
ev = [(float(r['start_us']), float(r['dur_us']), r['kernel']) for r in sub]
gaps = [(b[0] - (a[0] + a[1]), a[2], b[2]) for a, b in zip(ev, ev[1:])]
cls = lambda g: 'boundary' if g < 2.6 else ('bubble' if g < 12 else 'hole')
tot = defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    t = tot[cls(max(g, 0.0))]
    t[0] += 1
    t[1] += max(g, 0.0)
busy = sum(e[1] for e in ev)
span = ev[-1][0] + ev[-1][1] - ev[0][0]
print(f'queue {q}: {len(ev)} kernels, span {span / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms, idle {(span - busy) / 1e3:.2f} ms')
for k in ('boundary', 'bubble', 'hole'):
    n, t = tot[k]
    print(f'  {k:9s} {n:4d} gaps  {t / 1e3:6.3f} ms  mean {t / max(n, 1):5.2f} us')
pair = defaultdict(lambda: [0, 0.0])
short = lambda s: s.split('<')[0][:28]
for g, a, b in gaps:
    if g >= 2.6:
        p = pair[(short(a), short(b))]
        p[0] += 1
        p[1] += g
print('  gaps >= 2.6 us by (previous kernel -> next kernel):')
for (a, b), (n, t) in sorted(pair.items(), key=lambda x: -x[1][1])[:28]:
    print(f'   {t / 1e3:6.3f} ms {n:4d} x {t / n:6.1f} us  {a} -> {b}')
