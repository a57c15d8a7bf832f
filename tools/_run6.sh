cd /root/repo
python tools/resample_microbench.py split 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "select" 2>&1 | tail -3
