python -m pytest tests/test_ldpc_decode_gpu.py tests/test_sch_gpu.py tests/test_pusch_proc_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu > gpurun_out/bench_prio.json 2>gpurun_out/bench_prio.err; python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_prio.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['kernel_ms'])
print(j.get('single_slot_latency_us'), j.get('single_slot_latency_hip_graph_us'), j.get('single_slot_stage_us'))
for k,v in j['legs'].items():
    print(k, v.get('info_bits_per_s'), v.get('ms_per_step'), v.get('kernel_ms'))
PY
python tools/ldpc_rate_sweep.py --n 8192 > gpurun_out/sweep_prio.txt 2>&1; head -12 gpurun_out/sweep_prio.txt | cut -c1-130
