#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel; FETCH_SIZE / WRITE_SIZE are
reported in bytes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads: x2; both are in KiB).  Usage: parse_pmc.py out.json dir1 [dir2 ...]"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
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
    json.dump(res, open(out, 'w'), indent=1)
    for k, e in res.items():
        print(k[:90], {n: ('%.4g' % v) for n, v in e.items()})


if __name__ == '__main__':
    main()
