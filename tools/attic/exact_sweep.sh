# developer aid: exact-fp32 mode (FCL_PRECISION=0) tile choices of the LDS-DMA kernels: the eager kernel table and the 4-stream bench value
cd $GRAFT_REPO_ROOT
run() {
  env FCL_PRECISION=0 "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  value %.2f M  replay %.2f M' % (p['value']/1e6, p.get('value_replay_only',0)/1e6))
for k,v in p['kernels'].items():
    if 'plstm' in k or 'pgemm' in k: print('    %-44s %7.1f us %5.1f launches avg %5.1f us %6.1f TF' % (k, v['ms_per_step']*1000, v['launches_per_step'], v['ms_per_step']*1000/v['launches_per_step'], v['tflops']))
"
}
run A=0
run FCL_PLSTM_CFG=4
run FCL_PLSTM_CFG=1
run FCL_PLSTM_CFG=4 FCL_PGEMM_CFG=2
run FCL_PLSTM_CFG=4 FCL_PGEMM_CFG=1
