#!/bin/bash
OUT=gpurun_out/s5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/bench_pull.py > $OUT/bench_pull.json 2> $OUT/bench_pull.err; cat $OUT/bench_pull.json | tee -a $OUT/summary.txt
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_gpu_model.py -q -k "member_chains or member_sharding or graph_equals" > $OUT/chains_$i.log 2>&1
  echo "chains run $i rc=$?" | tee -a $OUT/summary.txt
done
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests_all.log 2>&1
echo "all gpu tests rc=$?" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR|passed|failed|Fatal" $OUT/tests_all.log | tail -30 >> $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests_all2.log 2>&1
echo "all gpu tests (2nd) rc=$?" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR|passed|failed|Fatal" $OUT/tests_all2.log | tail -30 >> $OUT/summary.txt
for b in 8 64; do
  DLWP_TRAIN_STEP=graph timeout 300 python tools/bench_fit_generator.py --batch $b --samples $((b * 40)) --epochs 3 > $OUT/fitgen_b${b}_graph.json 2> $OUT/fitgen_b${b}_graph.err
  tail -1 $OUT/fitgen_b${b}_graph.json >> $OUT/summary.txt
  DLWP_LOADER_PULL=0 DLWP_TRAIN_STEP=graph timeout 300 python tools/bench_fit_generator.py --batch $b --samples $((b * 40)) --epochs 3 > $OUT/fitgen_b${b}_graph_nopull.json 2> $OUT/fitgen_b${b}_graph_nopull.err
  tail -1 $OUT/fitgen_b${b}_graph_nopull.json >> $OUT/summary.txt
done
cat $OUT/summary.txt
