#!/usr/bin/env python3
"""HBM traffic of the standalone halo / pooling / up-sampling kernels from rocprofv3 counters, per (kernel, shape):
joins the rows tools/bench_pad.py wrote (--out, with HIP-event times) with the per-dispatch FETCH_SIZE / WRITE_SIZE of two
separate `rocprofv3 --kernel-trace --pmc` passes over the SAME command (tools/profile_pads.sh).  bench_pad.py launches
every (kernel, shape) 3 + iters times in a row, so the k-th block of 3 + iters dispatches of a kernel symbol belongs to
the k-th row of that kernel.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts the 128-byte requests of wide coalesced
reads as 64 bytes on gfx950 (MI355X_MICROARCH.md, HBM section): x2.
Usage: parse_pad_pmc.py rows.json iters out.json fetch_dir write_dir"""
import collections
import csv
import glob
import json
import os
import sys

SYMBOL = {'pad2d_fwd': 'pad2d_fwd_', 'pad2d_bwd': 'pad2d_bwd', 'maxpool2_fwd': 'maxpool2_fwd_kernel',
          'upsample2_fwd': 'upsample2_fwd', 'series_merge_time': 'copy_runs_kernel'}


def per_dispatch(d, counter):
    """[(dispatch id, kernel name, value)] in dispatch order"""
    out = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                out.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    return sorted(out)


def main():
    rows = json.load(open(sys.argv[1]))
    per = 3 + int(sys.argv[2])
    out, fetch_dir, write_dir = sys.argv[3], sys.argv[4], sys.argv[5]
    fetch, write = per_dispatch(fetch_dir, 'FETCH_SIZE'), per_dispatch(write_dir, 'WRITE_SIZE')
    seen = collections.Counter()
    for r in rows:
        key = SYMBOL[r['kernel']]
        k = seen[key]
        seen[key] += 1

        def block(table):
            vals = [v for _, name, v in table if key in name]     # (all template instances of the kernel, dispatch order)
            blk = vals[k * per:(k + 1) * per]
            return sum(blk[3:]) / max(1, len(blk[3:])) if len(blk) == per else None
        f, w = block(fetch), block(write)
        alg = r['gbs'] * r['ms'] * 1e6
        r['algorithmic_bytes'] = alg
        if f is not None and w is not None:
            r['hbm_read_bytes'] = f * 1024.0 * 2.0
            r['hbm_write_bytes'] = w * 1024.0
            r['traffic_over_algorithmic'] = (r['hbm_read_bytes'] + r['hbm_write_bytes']) / alg
            r['hbm_gbs_from_counters'] = (r['hbm_read_bytes'] + r['hbm_write_bytes']) / r['ms'] / 1e6
        r['frac_of_8TBs'] = r['gbs'] / 8000.0
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dlwp_amd import _lib
    json.dump({'_meta': {'source_sha': _lib.kernel_source_hash(), 'peak_gbs': 8000.0,
                         'note': 'ms / gbs: HIP events (un-profiled run); bytes: rocprofv3 PMC passes of the same command; '
                                 'the 256 MB Infinity Cache absorbs part of the traffic of tensors under ~100 MB'},
               'rows': rows}, open(out, 'w'), indent=1)
    for r in rows:
        print('%-18s %-26s %8.3f ms %7.1f GB/s algorithmic  traffic/algorithmic %s' %
              (r['kernel'], r['shape'], r['ms'], r['gbs'],
               '%.2f' % r['traffic_over_algorithmic'] if 'traffic_over_algorithmic' in r else 'n/a'))


if __name__ == '__main__':
    main()
