cd /root/repo
python -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "backbone" 2>&1 | tail -3
python tools/interleave_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -3
python tools/backbone_spt_calib.py f32 2>&1 | grep -v amdgpu.ids
python tools/backbone_time.py 2>&1 | grep -v amdgpu.ids | tail -3
python tools/c3_time.py 2>&1 | grep -v amdgpu.ids
