#!/bin/bash
# Times the OFDM demodulator variants (threads per transform) through bench.py's stage timings.
set -e
mkdir -p gpurun_out
python -m pytest tests/test_ofdm_gpu.py -x -q -m gpu > gpurun_out/ofdm_tests.log 2>&1 || { tail -20 gpurun_out/ofdm_tests.log; exit 1; }
tail -2 gpurun_out/ofdm_tests.log
run() { echo "== $*"; env "$@" python bench.py --no-cpu --no-latency --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['kernel_ms'])"; }
run MIPHY_X=0                       # default: 4096-point symbols on the compile-time-stride kernel
run MIPHY_FFT_THREADS_DIV=16        # N/16 threads: the plain kernel (two butterflies per thread)
MIPHY_FFT_THREADS_DIV=16 python -m pytest tests/test_ofdm_gpu.py -x -q -m gpu 2>&1 | tail -2
