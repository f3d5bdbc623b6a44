#!/bin/bash
# Round profile on the GPU box. usage: bash tools/profile_round.sh <tag> <git commit> [mix cycles from valu_probe]
#   kernel-trace stats of the default bench, FETCH_SIZE / WRITE_SIZE / SQ counters in separate --pmc passes (never combined with
#   other trace domains), the stamped JSON files bench.py reads, and the probes.
# -> gpurun_out/prof_<tag>/{kernel_stats.csv,bench.json,pmc_fetch.csv,pmc_write.csv,pmc_sq.csv,r03_traffic.json,r03_pmc_sq.json,valu_probe.txt}
# Copy what is to be judged into profiles/ (r03_*).
set -e
tag=${1:-x}
commit=${2:-unknown}
mix=${3:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/prof_$tag
mkdir -p $o
B="python3 bench.py --no-cpu --no-latency --no-extra"
rocprofv3 --kernel-trace --stats -d $o/kt -o kt --output-format csv -- $B --steps 20 --warmup 3 > $o/bench.json 2> $o/bench.err
cp $o/kt/kt_kernel_stats.csv $o/kernel_stats.csv
echo "kernel-trace done" 
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $o/f -o f --output-format csv -- $B --steps 3 --warmup 1 > $o/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $o/w -o w --output-format csv -- $B --steps 3 --warmup 1 > $o/w.log 2>&1
echo "fetch/write done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $o/a -o a --output-format csv -- $B --steps 3 --warmup 1 > $o/a.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU -d $o/b -o b --output-format csv -- $B --steps 3 --warmup 1 > $o/b.log 2>&1
echo "sq done"
python3 profiles/summarize_pmc.py $o/f > $o/pmc_fetch.csv
python3 profiles/summarize_pmc.py $o/w > $o/pmc_write.csv
python3 profiles/summarize_pmc.py $o/a $o/b > $o/pmc_sq.csv
if [ -x tools/valu_probe ]; then ./tools/valu_probe > $o/valu_probe.txt 2>&1 || true; fi
if [ -z "$mix" ] && [ -f $o/valu_probe.txt ]; then mix=$(grep "decoder mix" $o/valu_probe.txt | sed 's/.*W=3: *\([0-9.]*\).*/\1/'); fi
python3 tools/make_profile_json.py $o r03 $commit 1024 $mix
# the extra legs on their own: kernel statistics and SQ counters of the polar, encoder / modulator and small-lifting-size kernels
L="python3 tools/legs_profile_run.py"
rocprofv3 --kernel-trace --stats -d $o/lkt -o lkt --output-format csv -- $L > $o/legs.json 2> $o/legs.err
cp $o/lkt/lkt_kernel_stats.csv $o/kernel_stats_legs.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $o/la -o la --output-format csv -- $L > $o/la.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d $o/lb -o lb --output-format csv -- $L > $o/lb.log 2>&1
python3 profiles/summarize_pmc.py $o/la $o/lb > $o/pmc_sq_legs.csv
echo "legs done"
head -8 $o/kernel_stats.csv | cut -c1-70,200-
grep -v Memcpy $o/pmc_fetch.csv | grep -v elementwise | head -12
cat $o/valu_probe.txt 2>/dev/null || true
