run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.2fM' % (d['value']/1e6), '%.4f' % d['ms_per_step'], ' '.join('%s=%.3f' % (k.split('_kernel')[0][:6]+k[-12:], v['ms_per_step']) for k, v in d['kernels'].items() if 'lstm' in k))"; }
run default
FCL_LSTM_SMALL_M=512 run small512_64x64
FCL_LSTM_SMALL_M=512 FCL_PLSTM_ROW32_M=1100 run small512_row32
FCL_LSTM_SMALL_M=300 FCL_PLSTM_ROW32_M=1100 run small300_row32
FCL_LSTM_SMALL_M=128 FCL_PLSTM_ROW32_M=1100 run small128_row32
FCL_LSTM_SMALL_M=512 FCL_PLSTM_ROW32_M=1700 run small512_row32_1700
