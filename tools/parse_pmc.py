#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel; FETCH_SIZE / WRITE_SIZE are
reported in bytes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads: x2; both are in KiB).  The summary carries `_meta` = {source_sha: hash of dlwp_amd/csrc, members}: bench.py
only quotes a summary taken on the kernel source it is running.  Usage: parse_pmc.py out.json dir1 [dir2 ...] [--members M]"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    argv = list(sys.argv[1:])
    members = 256
    if '--members' in argv:
        i = argv.index('--members')
        members = int(argv[i + 1])
        del argv[i:i + 2]
    out, dirs = argv[0], argv[1:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name']
                if 'dlwp' not in k and 'conv2d' not in k and 'pad2d' not in k and 'kernel' not in k:
                    continue
                if k.startswith('void at::') or 'rocclr' in k:
                    continue
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {}
    for k, c in agg.items():
        e = {n: sum(v) / len(v) for n, v in c.items()}
        if 'FETCH_SIZE' in e:
            e['hbm_read_bytes'] = e['FETCH_SIZE'] * 1024.0 * 2.0
        if 'WRITE_SIZE' in e:
            e['hbm_write_bytes'] = e['WRITE_SIZE'] * 1024.0
        e['launches'] = max(len(v) for v in c.values())
        res[k] = e
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from dlwp_amd import _lib
        res['_meta'] = {'source_sha': _lib.kernel_source_hash(), 'members': members}
    except Exception as e:  # noqa: BLE001
        res['_meta'] = {'source_sha': None, 'members': members, 'error': repr(e)}
    json.dump(res, open(out, 'w'), indent=1)
    for k, e in res.items():
        if k == '_meta':
            continue
        print(k[:90], {n: ('%.4g' % v) for n, v in e.items()})


if __name__ == '__main__':
    main()
