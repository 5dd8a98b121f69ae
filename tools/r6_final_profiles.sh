#!/bin/bash
# round-6 final measurements on ONE box: the GPU suite (first run on the box), rocprofv3 kernel stats + PMC passes of the bench command,
# config 4, config 5 at 4 members, 8 members of config 2, the training step (kernel stats + executed MFMA instructions), the bench line.
# Everything lands in gpurun_out/prof (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof; mkdir -p $O
cd $R
export TMPDIR=/tmp
meta() {  # meta <csv> <command>: the kernel source a kernel-stats summary was taken on
  python - "$1" "$2" <<PY
import json, sys
sys.path.insert(0, "$R")
from dlwp_amd import _lib
open(sys.argv[1][:-4] + '.meta.json', 'w').write(json.dumps({"source_sha": _lib.kernel_source_hash(), "command": sys.argv[2]}))
PY
}
if [ "$1" != "--no-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/r6_pytest.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' $O/r6_pytest.log | tail -1)"
  cp gpurun_out/forward_errors.json $O/r6_forward_errors.json 2>/dev/null
fi
bash tools/profile_bench.sh r6 > $O/r6_profile_bench.log 2>&1; tail -30 $O/r6_profile_bench.log | cut -c1-200
bash tools/profile_cfg4.sh r6 8 > $O/r6_profile_cfg4.log 2>&1; tail -12 $O/r6_profile_cfg4.log | cut -c1-200
meta $O/r6_cfg4_bf16_m8_kernel_stats.csv "tools/profile_cfg4.sh r6 8"
cd /tmp
for b in 64 8; do
  rocprofv3 --kernel-trace --stats -d $O/tr$b -o s --output-format csv -- python $R/tools/bench_train.py --batch $b --steps 40 --warmup 10 > $O/r6_train_b$b.json 2> $O/tr$b.err
  cp $(find $O/tr$b -name '*kernel_stats.csv' | head -1) $O/r6_train_b${b}_kernel_stats.csv
  meta $O/r6_train_b${b}_kernel_stats.csv "tools/bench_train.py --batch $b --steps 40 --warmup 10"
  rm -rf $O/tr$b
  # ... and every kernel of the step ALONE (one stream): the isolated durations DESIGN 5.17 / section 8 quote
  DLWP_TRAIN_STEP=graph rocprofv3 --kernel-trace --stats -d $O/trs$b -o s --output-format csv -- python $R/tools/bench_train.py --batch $b --steps 40 --warmup 10 > /dev/null 2> $O/trs$b.err
  cp $(find $O/trs$b -name '*kernel_stats.csv' | head -1) $O/r6_train_b${b}_single_stream_kernel_stats.csv
  meta $O/r6_train_b${b}_single_stream_kernel_stats.csv "DLWP_TRAIN_STEP=graph tools/bench_train.py --batch $b --steps 40 --warmup 10"
  rm -rf $O/trs$b
  DLWP_TAPE_VALIDATE=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA -d $O/trm$b -o p --output-format csv -- python $R/tools/bench_train.py --batch $b --steps 20 --warmup 10 > /dev/null 2> $O/trm$b.err
  python $R/tools/parse_train_mfma.py $O/r6_train_mfma_b$b.json $O/trm$b --batch $b --steps 30
  rm -rf $O/trm$b
done
# config 5 (1-degree grid, 12 channels) at 4 members = one GPU's share of 32 members on 8; config 2 at 8 members
rocprofv3 --kernel-trace --stats -d $O/c5 -o s --output-format csv -- python $R/bench.py --grid 180x360 --channels 12 --members 4 --forwards 40 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/r6_bench_cfg5_m4.json 2> $O/c5.err
cp $(find $O/c5 -name '*kernel_stats.csv' | head -1) $O/r6_cfg5_m4_kernel_stats.csv; rm -rf $O/c5
meta $O/r6_cfg5_m4_kernel_stats.csv "bench.py --grid 180x360 --channels 12 --members 4 --forwards 40 --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/c2 -o s --output-format csv -- python $R/bench.py --members 8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/r6_bench_cfg2_m8.json 2> $O/c2.err
cp $(find $O/c2 -name '*kernel_stats.csv' | head -1) $O/r6_cfg2_m8_kernel_stats.csv; rm -rf $O/c2
meta $O/r6_cfg2_m8_kernel_stats.csv "bench.py --members 8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
# the rollout examples/validate.py runs (TimeSeriesEstimator with insolation): kernels of the fed rollout, 256 samples x 28 calls
rocprofv3 --kernel-trace --stats -d $O/est -o s --output-format csv -- python $R/tools/experiments/r6_estimator_api_breakdown.py > $O/r6_estimator_profile.txt 2> $O/est.err
cp $(find $O/est -name '*kernel_stats.csv' | head -1) $O/r6_estimator_kernel_stats.csv; rm -rf $O/est
meta $O/r6_estimator_kernel_stats.csv "tools/experiments/r6_estimator_api_breakdown.py (TimeSeriesEstimator.predict, 256 samples x 28 calls, 6 -> 4 channels)"
cd $R
for b in 64 8; do python tools/bench_train.py --batch $b --steps 40 --warmup 20 > $O/r6_train_b${b}_noprof.json 2>/dev/null; tail -1 $O/r6_train_b${b}_noprof.json | cut -c1-300; done
python tools/bench_layer6.py > $O/r6_layer6.json 2>/dev/null; cat $O/r6_layer6.json
# the weight-gradient launches alone: per-phase cycle counts of the channel-block instances on the config-3 layers (built here:
# hipcc is on the box), the first layer's fused launch, the instance sweep at 64 and 8 samples
(cd tools/microbench && hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDLWP_PHASE_TIMING -o /tmp/wgrad_cb_phase_timing.bin wgrad_cb_phase_timing.hip -I../../include 2> $O/wgcb_build.err \
  && /tmp/wgrad_cb_phase_timing.bin > $O/r6_wgrad_cb_phase_timing.txt 2>&1)
for b in 64 8; do python tools/bench_wgrad_pooled.py --batch $b; done > $O/r6_wgrad_pooled.json 2>/dev/null
for b in 64 8; do python tools/tune_wgrad.py --batch $b --layers L4,L2p,L3p,L5r,L6r --iters 20 2>/dev/null | grep -E "wgrad|cfg" | grep -v failed | awk '/wgrad/ {n=0; print; next} {if (n++ < 6) print}'; done > $O/r6_wgrad_sweep.txt
# the bench line quotes the rocprofv3 / PMC summaries of THIS kernel source from profiles/: put the fresh ones there first
for f in r6_kernel_stats.csv r6_kernel_stats.meta.json r6_hbm_traffic_b256.json r6_mfma_busy.json r6_train_mfma_b64.json r6_train_mfma_b8.json; do cp $O/$f $R/profiles/$f; done
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/cfg4_stats $O/cfg4_fetch $O/cfg4_write
# the padding kernels: HIP-event rates + rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (north_star: HBM GB/s of the padding kernels)
cd $R
bash tools/profile_pads.sh r6 > $O/r6_profile_pads.log 2>&1; tail -3 $O/r6_profile_pads.log | cut -c1-200
meta $O/r6_pad_pool_kernel_stats.csv "tools/profile_pads.sh r6"
for f in r6_pad_pool_hbm.json r6_pad_pool_kernel_stats.csv r6_pad_pool_kernel_stats.meta.json; do cp $O/$f $R/profiles/$f 2>/dev/null; done
# the host-visible rollout with a memory-copy trace: the series leaves through the copy engines (SDMA), not through blit kernels
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/hv -o s --output-format csv -- python $R/tools/bench_host_rollout.py --reps 2 > $O/r6_host_visible.json 2> $O/hv.err
python $R/tools/trace_copies.py $O/hv > $O/r6_host_visible_copies.txt 2>&1
cat $O/r6_host_visible_copies.txt; rm -rf $O/hv
cd $R
timeout 900 python bench.py > $O/r6_bench.json 2> $O/r6_bench.err; echo "bench (with pad counters) rc=$?"; tail -1 $O/r6_bench.json | cut -c1-300
