python -m pytest tests/test_chest_gpu.py tests/test_pusch_demod_gpu.py tests/test_pusch_proc_gpu.py tests/test_pdsch_mod_gpu.py tests/test_dropin_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu --no-extra > gpurun_out/bench_chest.json 2>gpurun_out/bench_chest.err; python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_chest.json').read().strip().splitlines()[-1])
for k in ['value','ms_per_step','kernel_ms','single_slot_latency_us','single_slot_latency_hip_graph_us','single_slot_stage_us','parity_check']:
    print(k, j.get(k))
PY
