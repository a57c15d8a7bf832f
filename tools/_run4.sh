cd /root/repo
python -m pytest tests/test_fused_gpu.py -q -m gpu -k "mean_score_input_grad or dps_step_without_autograd" -s 2>&1 | grep "mean_score_input_grad \|elements off\|dps fused\|passed\|failed"
