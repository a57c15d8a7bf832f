cd /root/repo
export SVDD_HIP_LIB=build/exp/bb_f32_timing/timing/libsvdd_hip.so
for s in 1 2 3 4; do python tools/lpt_phase_timing.py f32 --L 50 --spt $s 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_bb_small_tile_phases.txt
python tools/lpt_phase_timing.py f32 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_bb_small_tile_phases.txt
cat gpurun_out/r06_bb_small_tile_phases.txt
