#!/bin/bash
# SQ counter passes over bench.py (decoder focus). usage (on the GPU box): bash tools/pmc_decoder.sh <tag>
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/pmc_$tag/a -o a --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > gpurun_out/pmc_$tag/a.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU -d gpurun_out/pmc_$tag/b -o b --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > gpurun_out/pmc_$tag/b.log 2>&1
python3 profiles/summarize_pmc.py gpurun_out/pmc_$tag/a gpurun_out/pmc_$tag/b > gpurun_out/pmc_$tag/summary.csv
grep ldpc gpurun_out/pmc_$tag/summary.csv
