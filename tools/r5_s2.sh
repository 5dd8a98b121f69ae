#!/bin/bash
# round-5 session 2: the suite on the fixed step validation + flat pad kernels, config-4 launch-stream A/B, pads, host-visible rate
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/suite.log)"
grep -E "^FAILED|^ERROR" $O/suite.log | head -20
python tools/bench_pad.py --iters 20 > $O/pad.txt 2>&1; cat $O/pad.txt
for cfg in "1 0" "0 0" "1 1" "0 1"; do
  set -- $cfg
  DLWP_ROLLOUT_OWN_STREAM=$1 DLWP_BENCH_SIDE_STREAM=$2 timeout 300 python tools/bench_cfg4.py --members 8 > $O/cfg4_own$1_side$2.json 2> $O/cfg4.err
  echo "cfg4 own=$1 side_stream=$2: $(python -c "
import json
d=json.load(open('$O/cfg4_own$1_side$2.json'))
print(round(d['six_hour_steps_per_s']), round(d['ms_per_forward'],4))" 2>&1 | tail -1)"
done
timeout 300 python tools/bench_host_rollout.py --reps 5 > $O/host.json 2> $O/host.err; cut -c1-500 $O/host.json
