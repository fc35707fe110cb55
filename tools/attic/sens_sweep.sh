# developer aid: sensitivity of the 4-stream bench line to each kernel family: a family's debug flag makes its kernels return early (results are
# garbage, the launch sequence and everything else are unchanged): "how fast would the pass be if this family were free"
cd $GRAFT_REPO_ROOT
run() {
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %.2f M  replay %.2f M' % ('$*', p['value']/1e6, p.get('value_replay_only',0)/1e6))
"
}
run A=0
run FCL_FP_DBG=1
run FCL_PLSTM_DBG=1
run FCL_PGEMM_DBG=1
run FCL_FP_DBG=1 FCL_PLSTM_DBG=1
run FCL_FP_DBG=1 FCL_PLSTM_DBG=1 FCL_PGEMM_DBG=1
run A=0
