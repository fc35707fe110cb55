# developer aid: the default bench line (4 streams, fresh feed) for several feat/prenet kernel configurations, same box
cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 1" "1 2" "2 2" "1 0" "2 0" "0 0" "1 1"; do
  set -- $cfg
  FCL_FP_SPLIT=$1 FCL_FP_SPLIT_RT=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SPLIT=$1 RT=$2 value %.2f M  replay %.2f M  predicted %.2f M  calibrated %.2f M' % (p['value']/1e6, p.get('value_replay_only',0)/1e6, p['predicted_durations']['value']/1e6, (p.get('calibrated_caps') or {}).get('value',0)/1e6))
"
done
