#!/usr/bin/env python3
"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
Usage: tools/rocpd_summary.py <results.db> [top_n]   -> markdown table on stdout."""
import re
import sqlite3
import sys


def demangle_bf16(name):
    """rocprofv3 leaves names with the __bf16 builtin (Itanium `DF16b`) mangled; spell it as a vendor type and ask c++filt."""
    if not name.startswith('_Z') or 'DF16b' not in name:
        return name
    import subprocess
    try:
        out = subprocess.run(['c++filt', name.replace('DF16b', 'u6__bf16')], capture_output=True, text=True, timeout=5).stdout.strip()
        return out or name
    except Exception:
        return name


def short(name):
    name = demangle_bf16(name)
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('void ', '').replace('cdrl::', '')
    return name[:90]


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute(f'select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) '
                     f'from kernels group by {name_col} order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    ncalls = sum(r[1] for r in rows)
    print(f'total kernel time {total / 1e6:.3f} ms over {ncalls} dispatches\n')
    print('| kernel | calls | total ms | avg us | min us | max us | share |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for n, cnt, tot, avg, mn, mx in rows[:top]:
        print(f'| {short(n)} | {cnt} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f}% |')


if __name__ == '__main__':
    main()
