#!/usr/bin/env python3
"""Dumps the last N kernel dispatches of a rocprofv3 (rocpd sqlite) trace in start order:
start_us (relative), dur_us, gap to previous end on the same queue, queue, grid, workgroup, kernel.
Usage: tools/rocpd_timeline.py <results.db> [last_n]   -> tsv on stdout."""
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
    return name.replace('void ', '').replace('cdrl::', '')[:70]


def main():
    db = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    print('# columns:', cols, file=sys.stderr)
    name_col = 'name' if 'name' in cols else 'kernel_name'
    q_col = next((x for x in ('queue_id', 'queue', 'stream_id', 'stream') if x in cols), None)
    g_col = next((x for x in ('grid_x', 'grid_size_x', 'grid_size') if x in cols), None)
    w_col = next((x for x in ('workgroup_x', 'workgroup_size_x', 'workgroup_size') if x in cols), None)
    sel = ', '.join([name_col, 'start', 'end'] + [x or '0' for x in (q_col, g_col, w_col)])
    rows = c.execute(f'select {sel} from kernels order by start').fetchall()[-last:]
    t0 = rows[0][1]
    last_end = {}
    print('start_us\tdur_us\tgap_us\tqueue\tgrid\twg\tkernel')
    for n, s, e, q, g, w in rows:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f'{(s - t0) / 1e3:.1f}\t{(e - s) / 1e3:.1f}\t{gap:.1f}\t{q}\t{g}\t{w}\t{short(n)}')


if __name__ == '__main__':
    main()
