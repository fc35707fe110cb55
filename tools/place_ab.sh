# headline synthesis line (with its decode-driver / calibrated-capacity legs) with and without the measured stream placement, same box
OUT=gpurun_out/${1:-place}
mkdir -p $OUT
for rep in 1 2 3; do
 for place in 0 1; do
  FCL_PLACE_STREAMS=$place python3 bench.py --no-cpu-baseline --no-extras 2>>$OUT/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('rep $rep FCL_PLACE_STREAMS=$place', 'value %.3f M' % (d['value']/1e6), 'calibrated %.3f' % (d.get('value_calibrated_caps',0)/1e6), 'decode_driver %.3f' % (d.get('value_decode_driver',0)/1e6), 'first_call %.3f' % (d['decode_driver']['first_call']['value']/1e6))
" >> $OUT/ab.log
 done
done
cat $OUT/ab.log
