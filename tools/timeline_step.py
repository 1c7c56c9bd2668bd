#!/usr/bin/env python3
"""Per-queue kernel breakdown of exactly one update-step (two stem_fwd-delimited passes) of a timeline.tsv
written by tools/rocpd_timeline.py.  Usage: tools/timeline_step.py <timeline.tsv> [top_n]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1]), delimiter='\t'))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    idx = [i for i, r in enumerate(rows) if 'stem_fwd' in r['kernel'] and 'stem_bwd' not in r['kernel']]     # (names of __bf16 instantiations come out mangled)
    a, b = idx[-3], idx[-1]
    sub = rows[a:b]
    s = [float(r['start_us']) for r in sub]
    d = [float(r['dur_us']) for r in sub]
    print(f'kernels per update-step {len(sub)}  span {(s[-1] + d[-1] - s[0]) / 1e3:.2f} ms  sum of durations {sum(d) / 1e3:.2f} ms')
    for q in sorted(set(r['queue'] for r in sub)):
        agg = defaultdict(lambda: [0, 0.0])
        for r in sub:
            if r['queue'] == q:
                x = agg[r['kernel']]
                x[0] += 1
                x[1] += float(r['dur_us'])
        tot = sum(v[1] for v in agg.values())
        n = sum(v[0] for v in agg.values())
        print(f'queue {q}: {n} kernels, busy {tot / 1e3:.2f} ms')
        for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
            print(f'  {t / 1e3:7.2f} ms {n:5d} {t / n:7.1f} us  {k}')


if __name__ == '__main__':
    main()
