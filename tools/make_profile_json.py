#!/usr/bin/env python3
"""Turns the summarised PMC passes of tools/profile_round.sh into the two stamped files bench.py reads:
  <round>_traffic.json  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE
                        reports half the bytes of wide coalesced reads; WRITE_SIZE is exact)
  <round>_pmc_sq.json   SQ instruction counts per launch of every kernel of the step
Both carry `kernel_source_sha` (hash of csrc/*.hip, *.h: bench.py refuses a file collected on other kernel sources) and the git
commit given on the command line.
usage: make_profile_json.py <dir with pmc_fetch.csv pmc_write.csv pmc_sq.csv> <round> <git commit> <slots per step> [mix cycles]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402

STAGE = {"ldpc_decode_pk_kernel": "ldpc_decode", "ldpc_decode_kernel": "ldpc_decode", "ldpc_decode_fused_kernel": "ldpc_decode",
         "rate_dematch_kernel": "rate_dematch", "ofdm_demod_4096_kernel": "ofdm_demod", "ofdm_demod_wide_kernel": "ofdm_demod",
         "ofdm_demod_kernel": "ofdm_demod", "chest_kernel": "dmrs_chest", "pusch_demod_kernel": "pusch_demod", "pusch_rx_kernel": "pusch_frontend",
         "pusch_tb_assemble_kernel": "tb_assemble"}


def read(path):
    rows = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            rows[(r["kernel"], r["counter"])] = (int(r["dispatches"]), float(r["mean_per_dispatch"]))
    return rows


def main():
    d, rnd, commit, slots = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    mix = float(sys.argv[5]) if len(sys.argv) > 5 else None
    sha = kernel_source_sha()
    f, w, sq = read(os.path.join(d, "pmc_fetch.csv")), read(os.path.join(d, "pmc_write.csv")), read(os.path.join(d, "pmc_sq.csv"))
    kern = {}
    for (k, c), (n, v) in f.items():
        if k in STAGE and c == "FETCH_SIZE" and (k, "WRITE_SIZE") in w:
            wr = w[(k, "WRITE_SIZE")][1]
            kern[STAGE[k]] = {"kernel": k, "fetch_kib": v, "write_kib": wr, "hbm_bytes_per_launch": (2 * v + wr) * 1024.0}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in two separate passes over `python3 bench.py --steps 3 --warmup 1 "
                         "--no-cpu --no-latency --no-extra` (tools/profile_round.sh); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
               "git_commit": commit, "kernel_source_sha": sha, "slots_per_gpu_per_step": slots, "kernels": kern},
              open(os.path.join(d, "%s_traffic.json" % rnd), "w"), indent=1)
    kk = {}
    for (k, c), (n, v) in sq.items():
        if k in STAGE:
            e = kk.setdefault(STAGE[k], {"kernel": k})
            e[c] = v
    if "ldpc_decode" in kk:
        kk["ldpc_decode"]["codeblocks_per_launch"] = slots * 38
    j = {"source": "rocprofv3 --kernel-trace --pmc SQ_* over the same command (tools/profile_round.sh), per-dispatch means",
         "git_commit": commit, "kernel_source_sha": sha, "slots_per_gpu_per_step": slots, "kernels": kk}
    if mix:
        j["measured_mix_cycles_per_wave_instruction"] = mix
        j["measured_mix_source"] = "tools/valu_probe 'decoder mix' row at 3 waves per SIMD (profiles/%s_valu_probe.txt)" % rnd
    json.dump(j, open(os.path.join(d, "%s_pmc_sq.json" % rnd), "w"), indent=1)
    print(json.dumps({"traffic_kernels": sorted(kern), "sq_kernels": sorted(kk), "sha": sha}))


if __name__ == "__main__":
    main()
