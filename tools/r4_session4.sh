#!/bin/bash
# r4 GPU session 3: the library's training step (tape) + the host-only loader; full GPU suite.
OUT=gpurun_out/s4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests_all.log 2>&1
echo "all gpu tests rc=$?" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR|passed|failed" $OUT/tests_all.log | tail -40 >> $OUT/summary.txt
for form in python graph lanes branches; do
  for b in 8 64; do
    if [ $form = python ]; then env="DLWP_TRAIN_GRAPH=0"; else env="DLWP_TRAIN_STEP=$form"; fi
    env $env timeout 300 python tools/bench_fit_generator.py --batch $b --samples $((b * 40)) --epochs 3 > $OUT/fitgen_b${b}_$form.json 2> $OUT/fitgen_b${b}_$form.err
    tail -1 $OUT/fitgen_b${b}_$form.json >> $OUT/summary.txt
    tail -2 $OUT/fitgen_b${b}_$form.err | grep -v amdgpu.ids >> $OUT/summary.txt
  done
done
cat $OUT/summary.txt
