bash tools/gpu_run.sh r06f tests bench
mkdir -p gpurun_out
timeout 600 ./databend_amd/host/block_sweep --out gpurun_out/r06_block_size_sweep.json 2> gpurun_out/r06_block_size_sweep.err; grep -c ok gpurun_out/r06_block_size_sweep.err
GPU_MAX_HW_QUEUES=8 timeout 600 ./databend_amd/host/block_sweep --only-q1 --out gpurun_out/r06_block_size_sweep_hwq8.json 2> gpurun_out/r06_block_size_sweep_hwq8.err; grep "threads 8" gpurun_out/r06_block_size_sweep_hwq8.err | grep pipelined | cut -c1-110
