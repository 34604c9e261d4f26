cd /root/repo
python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "skinny or persistent" > gpurun_out/t4.log 2>&1; tail -3 gpurun_out/t4.log
FUZZ_DECODE=1 FUZZ_OVR='{"family": 5}' FUZZ_OUT=gpurun_out/fuzz_skinny.json python tools/gpu_fuzz.py 1500 13 > gpurun_out/fuzz_skinny.log 2>&1; tail -2 gpurun_out/fuzz_skinny.log
python tools/skinny_lab.py check time > gpurun_out/skinny_lab.log 2>&1; tail -2 gpurun_out/skinny_lab.log
