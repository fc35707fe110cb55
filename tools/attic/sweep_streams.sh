run() { python bench.py --no-cpu-baseline --steps 48 --streams $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 streams=$2', '%.2fM' % (d['value']/1e6), '%.4f' % d['ms_per_step'])"; }
for q in 2 4 8 16; do for s in 4 6 8; do GPU_MAX_HW_QUEUES=$q run hwq=$q $s; done; done
