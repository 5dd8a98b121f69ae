#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel name, grid size): calls, mean / min duration in microseconds.
usage: python tools/trace_summary.py <dir or *_kernel_trace.csv> [name filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    acc = defaultdict(list)
    order = []
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            if flt and flt not in name:
                continue
            key = (name[:100], r.get('Grid_Size', r.get('Grid_Size_X', '')))
            if key not in acc:
                order.append(key)
            acc[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
    for key in order:
        v = acc[key]
        v2 = sorted(v)[:max(1, len(v) * 3 // 4)]     # drop the slowest quarter (first launches)
        print('%-100s grid %-9s calls %3d  mean %8.2f us  min %8.2f us' % (key[0], key[1], len(v), sum(v2) / len(v2), v2[0]))


if __name__ == '__main__':
    main()
