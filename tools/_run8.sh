cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_t2_tests.txt; cat gpurun_out/r06_t2_tests.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_t2_bench.out 2> gpurun_out/r06_t2_bench.err ) 2> gpurun_out/r06_t2_bench.time
tail -c 400 gpurun_out/r06_t2_bench.out; cat gpurun_out/r06_t2_bench.time; cp bench_full.json gpurun_out/r06_t2_bench_full.json
bash tools/check_tools.sh r06 > /dev/null 2>&1
