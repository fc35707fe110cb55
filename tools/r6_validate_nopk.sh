mkdir -p gpurun_out/r6O
python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_train_native.py tests/test_gpu_fused_small.py -x -q 2>&1 | tail -3
python3 tools/stress_bilstm_concurrent.py 2>&1 | tail -1
STRESS_SMALL_FIRST=1 python3 tools/stress_bilstm_concurrent.py 2>&1 | tail -1
python3 tools/hazard_trigger_scan.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | head -4
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do echo -n "diag $i: "; python3 tools/diag_native_kd2.py 2>&1 | grep "^rep" | awk '$4+0 > 1e-4 || $6+0 > 1e-4 || $8+0>1e-4 || $10+0>1e-4' | wc -l; done
python3 tools/bilstm_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -3
BILSTM_BENCH_MODEL=teacher python3 tools/bilstm_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -3
bash tools/ab_kd.sh r6O 2>&1 | tail -6
