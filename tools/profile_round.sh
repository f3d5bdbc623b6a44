#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench, then FETCH_SIZE / WRITE_SIZE in two separate --pmc passes.
# usage: bash tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/{kernel_stats.csv,bench.json,pmc_fetch.csv,pmc_write.csv}
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/prof_$tag
mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o/kt -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-latency > $o/bench.json 2> $o/bench.err
cp $o/kt/kt_kernel_stats.csv $o/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $o/f -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > $o/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $o/w -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > $o/w.log 2>&1
python3 profiles/summarize_pmc.py $o/f > $o/pmc_fetch.csv
python3 profiles/summarize_pmc.py $o/w > $o/pmc_write.csv
head -6 $o/kernel_stats.csv | cut -c1-60,200-
cat $o/pmc_fetch.csv $o/pmc_write.csv | grep -v Memcpy | grep -v elementwise
