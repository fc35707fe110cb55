# developer aid: the default bench line (4 streams, fresh feed) under several environment settings, same box, interleaved.
# usage: tools/env_bench_sweep.sh "A=1 B=2" "A=0" ...   (each argument = one arm; run twice round-robin)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for arm in "$@"; do
  env $arm python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s value %.2f M  replay %.2f M  predicted %.2f M' % ('$arm', p['value']/1e6, p.get('value_replay_only',0)/1e6, p['predicted_durations']['value']/1e6))
"
done
done
