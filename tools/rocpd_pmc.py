#!/usr/bin/env python3
"""Sums a rocprofv3 --pmc counter per kernel from the rocpd sqlite output.
Usage: tools/rocpd_pmc.py <results.db> [name-filter]   -> 'counter kernel calls sum' lines + total."""
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


def main():
    db = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute('pragma table_info(counters_collection)')]
    # counters_collection view: one row per (dispatch, counter)
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = c.execute(f'select counter_name, {name_col}, count(*), sum(value) from counters_collection '
                     f'group by counter_name, {name_col} order by 4 desc').fetchall()
    tot = {}
    for cn, kn, n, s in rows:
        if flt and flt not in kn:
            continue
        short = re.sub(r'\(.*$', '', demangle_bf16(kn)).replace('void ', '').replace('cdrl::', '')[:70]
        print(f'{cn}\t{short}\t{n}\t{s:.0f}')
        tot[cn] = tot.get(cn, 0.0) + s
    for k, v in tot.items():
        print(f'TOTAL\t{k}\t{v:.0f}')


if __name__ == '__main__':
    main()
