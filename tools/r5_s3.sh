#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
export TMPDIR=/tmp
# FIRST thing on a fresh box: forked graphs launched directly on a real stream
DLWP_BENCH_SIDE_STREAM=1 timeout 300 python tools/bench_cfg4.py --members 8 > $O/cfg4_side.json 2> $O/cfg4.err; echo "cfg4 side rc=$? $(python -c "
import json
d=json.load(open('$O/cfg4_side.json'))
print(round(d['six_hour_steps_per_s']), round(d['ms_per_forward'],4))" 2>&1 | tail -1)"
timeout 1500 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/suite.log)"
grep -E "^FAILED|^ERROR" $O/suite.log | head -20
python tools/bench_pad.py --iters 20 > $O/pad.txt 2>&1; grep pad2d $O/pad.txt
