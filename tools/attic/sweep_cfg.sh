run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.2fM' % (d['value']/1e6), '%.4f' % d['ms_per_step'])"; }
run default
FCL_PLSTM_CFG=1 run plstm128x128
FCL_PLSTM_CFG=2 run plstm64x128
FCL_PLSTM_CFG=3 run plstm64x64
FCL_PGEMM_CFG=1 run pgemm128x128
FCL_PGEMM_CFG=2 run pgemm64x128
FCL_PGEMM_CFG=3 run pgemm64x64
FCL_LSTM_SMALL_M=512 run small512
FCL_LSTM_SMALL_M=256 run small256
FCL_LSTM_SMALL_M=1600 run small1600
