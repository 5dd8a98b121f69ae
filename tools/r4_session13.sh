#!/bin/bash
# session 13: the streaming position-split kernel (conv_fwd_wino2s.hip) -- parity, then A/B on the bench's forward
OUT=gpurun_out/s13
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "position_split or phase_channels or few_channel or 16_output" > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)" | tee $OUT/summary.txt
grep -n "^FAILED\|^E  " $OUT/tests.log | head -12 >> $OUT/summary.txt
for m in 0 1; do
  DLWP_FEW_STREAM=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_fs$m.json 2> $OUT/bench_fs$m.err
  python - <<PY >> $OUT/summary.txt
import json
d = json.loads(open("$OUT/bench_fs$m.json").read().strip().splitlines()[-1])
print("few_stream=$m value", d["value"], "ms_per_step", d["ms_per_step"])
for r in d.get("layers", []):
    print("   ", r.get("layer"), r.get("kernel"), "ms", r.get("ms"), "iso", r.get("ms_isolated"))
PY
done
cat $OUT/summary.txt
