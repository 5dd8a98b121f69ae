#!/usr/bin/env python3
"""Executed matrix-core instructions of ONE training step from a `rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA` pass of
tools/bench_train.py: the counter summed over every dispatch of the run / the number of steps the run made (--steps: warm-up + timed
train_on_batch calls; the run's total must be a whole multiple of it).  Writes {'_meta': {source_sha, batch}, 'mfma_per_step': N, 'per_kernel': {...}}
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
    # every step executes the same matrix instructions (its first steps run launch by launch, the later ones replay the recorded
    # step, where a weight gradient and a data gradient may share a launch: other kernel NAMES, the same bodies)
    if sum(total.values()) % steps:
        sys.exit('SQ_INSTS_MFMA of the run is not a whole multiple of %d steps: %r' % (steps, dict(calls)))
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
