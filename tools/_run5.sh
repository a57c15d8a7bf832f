cd /root/repo
python tools/resample_microbench.py split 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_k2_gather_split.txt; cat gpurun_out/r06_k2_gather_split.txt
bash tools/perturbation_check.sh
