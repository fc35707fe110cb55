# headline (four streams) under different pconv / pgemm / plstm tile thresholds (round 6): gpurun_out/$1/tile_sweep.log
OUT=gpurun_out/${1:-r6B}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,2), ' '.join('%s=%.3f' % (k.split('_kernel')[0]+k.split('_kernel')[1][:12], v['ms_per_step']) for k, v in d['kernels'].items() if 'pconv' in k or 'plstm_kernel<' in k))"; }
for e in "X=0" "FCL_PCONV_BIG_MIN=60" "FCL_PCONV_BIG_MIN=30" "FCL_PCONV_BIG_MIN=30 FCL_PGEMM_BIG_MIN=30" "FCL_PLSTM_CFG=1" "X=1"; do
  v=$(env $e python3 bench.py --no-cpu-baseline --no-extras --regions 7 2>>$OUT/err.log | val)
  echo "$e -> $v" >> $OUT/tile_sweep.log
done
cat $OUT/tile_sweep.log
