#!/bin/bash
# round-3 final measurements on ONE box: tests, the bench line, rocprofv3 kernel stats + PMC passes of the bench command, config 4,
# the training step.  Everything lands in gpurun_out/prof (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/r3_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/r3_pytest.log | head -2
bash tools/profile_bench.sh r3 > $O/r3_profile_bench.log 2>&1; tail -30 $O/r3_profile_bench.log | cut -c1-200
bash tools/profile_cfg4.sh r3 8 > $O/r3_profile_cfg4.log 2>&1; tail -12 $O/r3_profile_cfg4.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for b in 64 8; do
  rocprofv3 --kernel-trace --stats -d $O/tr$b -o s --output-format csv -- python $R/tools/bench_train.py --batch $b --steps 40 --warmup 10 > $O/r3_train_b$b.json 2> $O/tr$b.err
  cp $O/tr$b/s_kernel_stats.csv $O/r3_train_b${b}_kernel_stats.csv
  rm -rf $O/tr$b
done
cd $R
for b in 64 8; do python tools/bench_train.py --batch $b --steps 40 --warmup 20 > $O/r3_train_b${b}_noprof.json 2>/dev/null; tail -1 $O/r3_train_b${b}_noprof.json | cut -c1-300; done
bash tools/profile_wgrad.sh r3 > $O/r3_profile_wgrad.log 2>&1
bash tools/profile_few.sh r3 > $O/r3_profile_few.log 2>&1
# the bench line quotes the rocprofv3 / PMC summaries of THIS kernel source from profiles/: put the fresh ones there first
for f in r3_kernel_stats.csv r3_kernel_stats.meta.json r3_hbm_traffic_b256.json r3_mfma_busy.json; do cp $O/$f $R/profiles/$f; done
timeout 600 python bench.py > $O/r3_bench.json 2> $O/r3_bench.err; echo "bench rc=$?"
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/cfg4_stats $O/cfg4_fetch $O/cfg4_write
