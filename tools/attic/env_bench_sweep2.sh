# developer aid: like env_bench_sweep.sh with extra bench.py arguments per arm: "ENV=.. ENV=.. -- --streams 8"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for arm in "$@"; do
  envs="${arm%%--*}"; args=""; case "$arm" in *--*) args="${arm#*-- }";; esac
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $args 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s value %.2f M  replay %.2f M  predicted %.2f M  ms/step %.3f' % ('''$arm''', p['value']/1e6, p.get('value_replay_only',0)/1e6, p['predicted_durations']['value']/1e6, p['ms_per_step']))
"
done
done
