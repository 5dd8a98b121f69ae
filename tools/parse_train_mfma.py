#!/usr/bin/env python3
"""Executed matrix-core instructions of ONE training step from a `rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA` pass of
tools/bench_train.py: the counter summed over every dispatch of the run / the number of steps the run made (--steps: warm-up + timed
train_on_batch calls; every matrix kernel must have been launched a whole number of times per step).  Writes {'_meta': {source_sha, batch}, 'mfma_per_step': N, 'per_kernel': {...}}
-- what bench.py's train_cfg3.executed_frac reads (profiles/r4_train_mfma_b<batch>.json; quoted only on the same kernel source).
Usage: parse_train_mfma.py out.json pmc_dir --batch B --steps S"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, d = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 0
    total = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != 'SQ_INSTS_MFMA':
                continue
            total[r['Kernel_Name']] += float(r['Counter_Value'])
            calls[r['Kernel_Name']] += 1
    steps = int(sys.argv[sys.argv.index('--steps') + 1])     # train_on_batch calls of the profiled run: warm-up + timed
    odd = {k: c for k, c in calls.items() if total[k] > 0 and c % steps}
    if odd:
        sys.exit('kernels not launched a whole number of times per step (%d steps?): %r' % (steps, odd))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dlwp_amd import _lib
    res = {'_meta': {'source_sha': _lib.kernel_source_hash(), 'batch': batch, 'steps_in_run': steps,
                     'definition': 'sum of SQ_INSTS_MFMA over all dispatches of tools/bench_train.py / steps (v_mfma_f32_16x16x4_f32: 2048 FLOP each)'},
           'mfma_per_step': sum(total.values()) / steps,
           'per_kernel': {k[:100]: {'mfma_per_step': v / steps, 'launches_per_step': calls[k] / steps}
                          for k, v in sorted(total.items(), key=lambda kv: -kv[1]) if v > 0}}
    json.dump(res, open(out, 'w'), indent=1)
    print('steps', steps, 'mfma_per_step', res['mfma_per_step'], '= %.2f GFLOP' % (res['mfma_per_step'] * 2048 / 1e9))


if __name__ == '__main__':
    main()
