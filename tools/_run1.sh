set -x
cd /root/repo
python tools/c3_two_part_ab.py > gpurun_out/r06_c3_two_part_ab.txt 2>&1
tail -12 gpurun_out/r06_c3_two_part_ab.txt
python -m pytest tests/test_skip_gpu.py tests/test_harness_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_t1_tests.txt
cat gpurun_out/r06_t1_tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dps -o dps -- python3 /root/repo/tools/dps_profile.py 256 16 > /tmp/dps.log 2>&1
tail -2 /tmp/dps.log
f=$(find /tmp/prof_dps -name "*kernel_stats.csv" | head -1)
head -45 $f > /root/repo/gpurun_out/r06_dps_kernel_stats_before.csv
