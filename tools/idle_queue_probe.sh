# What do extra HIP streams created earlier in the process cost the KD update?  n raw streams (hipStreamCreateWithFlags through the library, never used) are
# created before bench.py runs in the same process; FCL_PLACE_STREAMS=0 is the round-5 behaviour (streams as they come), 1 the measured placement.
OUT=gpurun_out/${1:-idleq}
mkdir -p $OUT
for rep in 1 2; do
 for place in 0 1; do
  for n in 0 1 2 3 5; do
    v=$(FCL_PLACE_STREAMS=$place python3 -c "
import sys, runpy, ctypes as C, torch
sys.path.insert(0, '.')
import fcl_taco2_amd
from fcl_taco2_amd import _lib
torch.zeros(1, device='cuda')
keep = []
for _ in range($n):
    h = C.c_void_p(); _lib.check(_lib.load().fcl_stream_create_cus(0, C.byref(h))); keep.append(h)
sys.argv = ['bench.py', '--workload', 'kd_step', '--no-cpu-baseline', '--no-dp-schedule', '--regions', '5']
runpy.run_path('bench.py', run_name='__main__')
" 2>>$OUT/err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])")
    echo "rep $rep FCL_PLACE_STREAMS=$place raw_streams_before=$n kd_step_ms $v" >> $OUT/probe.log
  done
 done
done
cat $OUT/probe.log
