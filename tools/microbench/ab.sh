for i in 1 2 3; do
for lib in old new; do
  if [ $lib = old ]; then export DLWP_LIB_PATH=$PWD/tools/microbench/old_lib.so; else unset DLWP_LIB_PATH; fi
  for form in lanes graph; do
    echo -n "$lib $form b64: "; DLWP_TRAIN_STEP=$form python tools/bench_train.py --batch 64 --steps 40 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
done
for lib in old new; do
  if [ $lib = old ]; then export DLWP_LIB_PATH=$PWD/tools/microbench/old_lib.so; else unset DLWP_LIB_PATH; fi
  echo -n "$lib auto b8: "; python tools/bench_train.py --batch 8 --steps 40 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
