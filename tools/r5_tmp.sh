R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" python $R/tools/bench_host_rollout.py --reps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', round(d['host_visible_steps_per_s']), 'host-visible', round(d['device_steps_per_s']), 'device')"
}
run default X=1
run blit_size_0 GPU_FORCE_BLIT_COPY_SIZE=0
run blit_wg_4 DEBUG_CLR_LIMIT_BLIT_WG=4
run blit_wg_8 DEBUG_CLR_LIMIT_BLIT_WG=8
run blit_wg_64 DEBUG_CLR_LIMIT_BLIT_WG=64
GPU_FORCE_BLIT_COPY_SIZE=0 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/hv -o s --output-format csv -- python $R/tools/bench_host_rollout.py --reps 2 > $O/host.json 2> $O/hv.err
python $R/tools/trace_copies.py $O/hv
rm -rf $O/hv
