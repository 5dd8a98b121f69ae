for k in ${KNOCKS:-base 1 2 3 4 5}; do
  if [ $k = base ]; then unset DLWP_LIB_PATH; else export DLWP_LIB_PATH=$GRAFT_REPO_ROOT/dlwp_amd/knock/libdlwp_hip_k$k.so; fi
  python tools/bench_cfg4.py --members 8 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
L = {l['layer']: l['ms'] for l in d['launches'] if 'lstm' in l.get('layer', '')}
print('k=$k', round(d['six_hour_steps_per_s'], 1), 'steps/s;', '; '.join('%s %.4f ms' % (k_, v) for k_, v in L.items()))
"
done
