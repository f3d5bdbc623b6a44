python -m pytest tests/test_pdsch_mod_gpu.py tests/test_pdsch_proc_gpu.py tests/test_dropin_gpu.py tests/test_pdcch_proc_gpu.py tests/test_ssb_proc_gpu.py tests/test_csi_rs_gpu.py -m gpu -x -q 2>&1 | tail -4
python bench.py --no-cpu > gpurun_out/bench_pk.json 2>gpurun_out/bench_pk.err; python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_pk.json').read().strip().splitlines()[-1])
print(j['value'], j['kernel_ms'])
l=j['legs']['pdsch_tx_chain']
print({k:l[k] for k in l if k not in ('cpu_reference_all_cores','cpu_reference_t1','config')})
PY
