set -x
cd /root/repo
python tools/exp_variants.py run bb_f32_small > gpurun_out/r06_bb_small_tile_fixed_cost.txt 2>&1
cat gpurun_out/r06_bb_small_tile_fixed_cost.txt
python -m pytest tests/test_harness_gpu.py -x -q -m gpu -k "eight_rank or gru_round" 2>&1 | tail -4
