for i in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export DLWP_LIB_PATH=$PWD/tools/microbench/old_lib.so; else unset DLWP_LIB_PATH; fi
  for b in 64 8; do echo -n "$lib b$b: "; python tools/bench_wgrad_pooled.py --batch $b; done
done
done
python -m pytest tests/test_gpu_train_fold.py -q -x 2>&1 | tail -2
