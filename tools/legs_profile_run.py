"""The extra legs of bench.py on their own (polar PDCCH encode / SSC / SCL-8, PDSCH transmit chain, mixed slot) for the profiler passes of
tools/profile_round.sh: rocprofv3 kernel statistics and SQ counters of polar_decode_kernel, polar_scl_kernel, pdcch_encode_kernel,
ldpc_encode_kernel, rate_match_kernel, pdsch_mod_kernel, ofdm_mod_4096_kernel, ldpc_decode_pkw_kernel ... without the headline step's launches
mixed into the per-kernel means. usage: python3 tools/legs_profile_run.py [--slots 256]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, miphy, bench, bench_legs as BL
ap = argparse.ArgumentParser(); ap.add_argument("--slots", type=int, default=256)
a = ap.parse_args()
ctx = miphy.Context(0); dev = torch.device("cuda", 0)
w = bench.pusch_workload()
out = {"polar_pdcch": bench.polar_leg(ctx, miphy, torch, dev)}
out["pdsch_tx_chain"], ok1 = BL.pdsch_tx_leg(ctx, miphy, torch, dev, w, a.slots, 6, 1.0, False, bench.HBM_PEAK_GBS)
out["pusch_mixed_slot"], ok2 = BL.mixed_slot_leg(ctx, miphy, torch, dev, a.slots, 6, 33.0, 777, 1.0, False)
out["verified"] = bool(ok1 and ok2)
print(json.dumps(out))
