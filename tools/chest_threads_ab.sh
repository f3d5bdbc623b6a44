set -e
python -m pytest tests/test_chest_gpu.py tests/test_pusch_proc_gpu.py -x -q -m gpu 2>&1 | tail -2
for t in 512 256; do echo "== threads $t"; MIPHY_CHEST_THREADS=$t python bench.py --no-cpu --no-latency --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['kernel_ms']['dmrs_chest'], d['parity_check'])"; done
