cd /root/repo
python -m pytest tests/test_fused_gpu.py -q -m gpu -k "mean_score_input_grad or dps_step_without_autograd or gru_pair" -s 2>&1 | grep "mean_score_input_grad \|elements off\|dps fused\|passed\|failed\|Error\|assert"
python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "dps" -s 2>&1 | grep "g26\|passed\|failed"
python tools/dps_profile.py 256 32 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dps -o dps -- python3 /root/repo/tools/dps_profile.py 256 16 > /tmp/dps.log 2>&1
f=$(find /tmp/prof_dps -name "*kernel_stats.csv" | head -1)
head -30 $f > /root/repo/gpurun_out/r06_dps_kernel_stats_after.csv
