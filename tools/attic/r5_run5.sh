mkdir -p gpurun_out/r5h
python3 -m pytest tests/test_gpu_training.py -x -q -k "structure or kd_refuses or g17" > gpurun_out/r5h/test_struct.log 2>&1
python3 -m pytest tests -m gpu -x -q > gpurun_out/r5h/test_all.log 2>&1
