python -m pytest tests/test_sch_gpu.py tests/test_pdsch_proc_gpu.py tests/test_ldpc_chain_gpu.py tests/test_dropin_gpu.py -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/pkprof -o pk --output-format csv -- python3 bench.py --no-cpu --no-latency --steps 3 --warmup 1 > gpurun_out/pkprof.log 2>&1
grep -i "pdsch_cb\|crc_kernel" gpurun_out/pkprof/pk_kernel_stats.csv | cut -c1-60,150-
