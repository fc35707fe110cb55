cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v FCL_FP_SPLIT=1 FCL_FP_SPLIT_RT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', p['value'], 'replay', p.get('value_replay_only'), 'roof', p['roofline']['avg_launch_us'])
print({k: round(v['ms_per_step']*1000/v['launches_per_step'],2) for k,v in p['kernels'].items()})
"
done
