#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace --memory-copy-trace run: memory copies by direction (count, bytes, time, GB/s), and the
blit kernels (__amd_rocclr_copyBuffer ...) of the kernel trace beside them -- are the big transfers copy-engine transfers?
usage: python tools/trace_copies.py <rocprofv3 output directory>"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    f = glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    if rows:
        print('columns:', list(rows[0].keys()))
    by = collections.defaultdict(lambda: [0, 0, 0])
    for r in rows:
        k = r.get('Direction') or r.get('Kind') or '?'
        dt = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        nb = int(r.get('Size', r.get('Bytes', 0)) or 0)
        by[k][0] += 1
        by[k][1] += nb
        by[k][2] += dt
    for k, (n, nb, dt) in by.items():
        print('%-34s %5d copies %10.1f MB %9.2f ms  %6.1f GB/s while active' % (k, n, nb / 1e6, dt / 1e6, nb / max(dt, 1)))
    big = sorted(rows, key=lambda r: -(int(r['End_Timestamp']) - int(r['Start_Timestamp'])))[:5]
    for r in big:
        print('  longest:', {k: r[k] for k in r if k in ('Direction', 'Size', 'Bytes', 'Kind')},
              round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 3), 'ms')
    ks = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
    if ks:
        for r in csv.DictReader(open(ks[0])):
            if 'copyBuffer' in r['Name'] or 'blit' in r['Name'].lower() or 'fillBuffer' in r['Name']:
                print('kernel %-40s calls %6s total %9.3f ms avg %8.1f us max %8.1f us' % (
                    r['Name'][:40], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3))


if __name__ == '__main__':
    main()
