#!/usr/bin/env python3
"""Summary of the profiler passes over the extra legs (tools/profile_round.sh -> kernel_stats_legs.csv, pmc_sq_legs.csv): per kernel the
rocprofv3 average duration, wave64 VALU instructions per launch and per second against the guide's issue peak (1024 SIMDs x 2.4 GHz / 2),
lane utilisation of the VALU instructions (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)), wave-cycle shares (parked / issue stall /
issuing) and LDS bank-conflict cycles per LDS instruction cycle.
usage: make_legs_profile_json.py <dir> <round> <git commit>"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402

PEAK = 1024 * 2.4 / 2.0  # G wave-instr/s
KEEP = ("polar_decode_kernel", "polar_scl_kernel", "pdcch_encode_kernel", "polar_encode_kernel", "pdsch_cb_encode_kernel", "pdsch_cb_encode_pk_kernel", "pdsch_seq_kernel", "ldpc_encode_kernel", "rate_match_kernel",
        "pdsch_mod_kernel", "dmrs_pdsch_kernel", "ofdm_mod_4096_kernel", "ldpc_decode_pkw_kernel", "crc_kernel")


def short(name):
    k = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", k)[0].split("::")[-1].strip()


def main():
    d, rnd, commit = sys.argv[1], sys.argv[2], sys.argv[3]
    dur = {}
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats_legs.csv"))):
        k = short(r["Name"])
        e = dur.setdefault(k, [0, 0.0])
        e[0] += int(r["Calls"])
        e[1] += float(r["TotalDurationNs"])
    sq = {}
    for r in csv.DictReader(open(os.path.join(d, "pmc_sq_legs.csv"))):
        sq.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_dispatch"])
    out = {}
    for k in KEEP:
        if k not in dur or k not in sq:
            continue
        c, avg_ns = sq[k], dur[k][1] / dur[k][0]
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        e = {"launches_timed": dur[k][0], "avg_us": avg_ns / 1e3, "valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"),
             "valu_G_wave_instr_per_s": c.get("SQ_INSTS_VALU", 0.0) / avg_ns, "valu_issue_frac_of_peak": c.get("SQ_INSTS_VALU", 0.0) / avg_ns / PEAK,
             "valu_lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") else None,
             "wave_cycles_parked": c.get("SQ_WAIT_ANY", 0.0) / wc, "wave_cycles_issue_stall": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
             "wave_cycles_issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "waves_per_launch": c.get("SQ_WAVES"),
             "lds_instructions_per_launch": c.get("SQ_INSTS_LDS"),
             "lds_bank_conflict_cycles_per_lds_instruction_cycle": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_ACTIVE_INST_LDS"]) if c.get("SQ_ACTIVE_INST_LDS") else None}
        out[k] = e
    json.dump({"source": "rocprofv3 --kernel-trace --stats and two --pmc SQ passes over `python3 tools/legs_profile_run.py` (tools/profile_round.sh); per-dispatch means of "
                         "the launches of the three legs (polar PDCCH 16 384 codewords per call at aggregation levels 1-16; PDSCH transmit chain and mixed slot at 256 slots)",
               "git_commit": commit, "kernel_source_sha": kernel_source_sha(), "valu_issue_peak_G_wave_instr_per_s": PEAK, "kernels": out},
              open(os.path.join(d, "%s_pmc_sq_legs.json" % rnd), "w"), indent=1)
    for k, e in out.items():
        print("%-26s %9.1f us  VALU %6.1f G/s = %.2f of peak, lanes %.2f, parked %.2f stall %.2f issuing %.2f" %
              (k, e["avg_us"], e["valu_G_wave_instr_per_s"], e["valu_issue_frac_of_peak"], e["valu_lane_utilisation"] or 0, e["wave_cycles_parked"],
               e["wave_cycles_issue_stall"], e["wave_cycles_issuing"]))


if __name__ == "__main__":
    main()
