#!/bin/bash
# A/B of engine switches on ONE box: bench.py's reduced shape (timed steps only) once per environment setting given as "NAME=VAL[,NAME=VAL]" arguments
# ("-" = defaults; LIB=build/NAME/libitsx_hip.so selects a variant build of scripts/build_variant.sh).  Prints reads/s, ms per step and pass A's / the MSV filter's time of every run.  usage: scripts/ab_env.sh - ITSX_SHARE_B=16 ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  envs=""; if [ "$v" != "-" ]; then envs=$(echo "$v" | tr ',' ' '); fi
  lib=$(echo " $envs" | sed -n 's/.* LIB=\([^ ]*\).*/\1/p'); [ -z "$lib" ] && lib=-
  out=$(env $envs python -c "import sys, runpy, os; sys.path.insert(0, '.'); import itsxpress_amd._lib as l; l.LIB_PATH = os.path.abspath(sys.argv[1]) if sys.argv[1] != '-' else l.LIB_PATH; sys.argv = ['bench.py'] + sys.argv[2:]; runpy.run_path('bench.py', run_name='__main__')" $lib --steps ${STEPS:-2} --warmup 1 --cpu-sample 0 --handover-steps 0 --full-steps 0 --files-leg 0 --alone-steps 0 ${BENCH_ARGS} 2>/dev/null | grep '^{' | tail -1)
  echo "$out" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; r = d['config'].get('rows_shared_frac', {})
print('%-44s %9.0f reads/s  %8.1f ms/step  bound %7.1f  msv %6.1f  fwd %6.1f bwd %6.1f ens %6.1f complete %6.1f build %5.1f  shared %s' % ('$v', d['value'], d['ms_per_step'], s['ms_bound_kernel'], s['ms_msv_kernel'], s['ms_fwd_kernel'], s['ms_bwd_kernel'], s['ms_ensemble'], s['ms_lazy_complete'], s['ms_share_build'], r.get('k_fwd_bound')))
"
done
