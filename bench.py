#!/usr/bin/env python3
"""Headline benchmark: PUSCH receive hot path on synthetic 100 MHz n78 slots (273 PRB, 30 kHz SCS).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over a batch of S synthetic uplink slots per GPU.  The workload is
BASELINE.json configs[2]: 273-PRB PUSCH, 256QAM R=948/1024, 1 layer, 38 codeblocks (BG1, Z=384) per slot,
TBS = 319 784 information bits per slot.  Inputs are resident in HBM before the timed region.

`value` = LDPC information bits / s over the whole step (all ranks).  Extra keys: `slots_per_s` (whole pipeline),
per-kernel HIP-event times, `roofline` of the dominant kernel and `cpu_baseline` (reference AVX2 path from oracle/_ref
when it loads, else the scalar oracle port), as the task contract asks.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def pusch_workload():
    """273 PRB x 156 data RE (14 symbols, 1 DM-RS symbol, 2 CDM groups w/o data), 256QAM R=948/1024, 1 layer."""
    return dict(nprb=273, mod=8, nof_layers=1, nsym=273 * 156, tbs=319784, bg=1, rv=0, Nref=0)


def build_slot_llrs(w, n_unique, sigma, seed):
    """CPU (oracle, test infrastructure) generation of a few unique noisy slots; returns int8 [n_unique, G] and TBs."""
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    G = w["nsym"] * w["mod"]
    llrs = np.zeros((n_unique, G), dtype=np.int8)
    tbs = []
    for u in range(n_unique):
        tb = rng.integers(0, 256, w["tbs"] // 8, dtype=np.uint8)
        cw = O.o_pdsch_encode(w["bg"], w["rv"], w["mod"], w["Nref"], w["nof_layers"], w["nsym"], tb)
        y = (1.0 - 2.0 * cw) + sigma * rng.standard_normal(G)
        llrs[u] = np.round(np.clip(4 * y, -20, 20) / 20 * 120).astype(np.int8)
        tbs.append(tb)
    return llrs, tbs


def cpu_baseline(w, llrs, max_iter, early_stop, budget_s):
    """Times the reference's own pusch_decoder (AVX2 rate dematcher + AVX2 LDPC decoder) on the host cores, or the
    scalar oracle port when oracle/_ref is unavailable.  Bounded sample; T threads, one decoder instance each."""
    import oracle_lib as O
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kind = "reference" if O.ref_available() else "port"
    try:
        if kind == "reference":
            O.ref()
    except OSError:
        kind = "port"
    T = max(1, min(ncores, 16))
    done = [0] * T
    stop_at = time.time() + budget_s
    G = llrs.shape[1]

    def worker(t):
        if kind == "reference":
            dec = O.RefPuschDecoder("avx2")
        k = 0
        while time.time() < stop_at:
            slot = llrs[(t + k) % llrs.shape[0]]
            if kind == "reference":
                dec.decode_sequence(w["bg"], w["mod"], w["Nref"], w["nof_layers"], w["nsym"], w["tbs"] // 8, [0],
                                    slot[None, :], max_iter, early_stop)
            else:
                od = O.OraclePuschDecoder(w["bg"], w["mod"], w["Nref"], w["nof_layers"], w["nsym"], w["tbs"] // 8)
                od.decode(slot, 0, True, max_iter, early_stop)
            k += 1
        done[t] = k

    if kind == "port":
        T = 1
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.time() - t0
    slots = sum(done)
    return dict(value=slots * w["tbs"] / dt, unit="info_bits/s", cores=T, kind=kind,
                sample="%d slots (38 CBs each, same LLRs as the GPU run) in %.1f s, %s" %
                       (slots, dt, "srsRAN pusch_decoder avx2" if kind == "reference" else "scalar C oracle"),
                host_cores_available=ncores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--slots", type=int, default=256, help="slots per GPU per step")
    ap.add_argument("--max-iter", type=int, default=6)
    ap.add_argument("--early-stop", type=int, default=0)
    ap.add_argument("--sigma", type=float, default=0.2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--chunks", type=int, default=1, help="slot groups per step, alternated over two HIP streams (1 = one stream)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-slot latency leg (keeps profiles to the timed step only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import miphy
    import oracle_lib as O
    ctx = miphy.Context(local_rank)
    w = pusch_workload()
    S = args.slots
    seg = O.o_segmentation(w["tbs"], w["bg"], w["mod"], w["nof_layers"], w["nsym"])
    C, Z, N, K, F = seg.nof_cbs, seg.Z, seg.N, seg.K, seg.nof_filler_bits
    G = w["nsym"] * w["mod"]
    n_unique = 4
    llrs_u, tbs_u = build_slot_llrs(w, n_unique, args.sigma, seed=1234 + rank)

    # ---- device-resident inputs and descriptors (everything below is HBM resident before timing starts)
    slot_src = np.arange(S) % n_unique
    llr_d = torch.from_numpy(llrs_u[slot_src].reshape(-1)).to(dev)          # S*G int8 codeword LLRs
    softbuf_d = torch.zeros(S * C * N, dtype=torch.int8, device=dev)        # HARQ soft buffers (device-resident pool)
    bits_d = torch.zeros(S * C * (K // 8), dtype=torch.uint8, device=dev)   # decoded codeblock messages
    iters_d = torch.zeros(S * C, dtype=torch.int32, device=dev)
    rdm = np.zeros(S * C, dtype=miphy.LdpcRdmDesc)
    dec = np.zeros(S * C, dtype=miphy.LdpcDecDesc)
    crc_poly = int(seg.crc_poly)
    # The decoder only needs the part of the soft buffer the dematcher can have written (new data, rv 0: E + fillers,
    # rounded up to a node); the rest is zero, which the reference trims away itself (ldpc_decoder_impl.cpp:86-99).
    dec_in_len = [min(N, max((22 + 2) * Z, -(-(seg.E[c] + F) // Z) * Z)) for c in range(C)]
    for s in range(S):
        for c in range(C):
            i = s * C + c
            rdm[i] = (w["bg"], w["rv"], w["mod"], 1, Z, F, w["Nref"], seg.E[c], s * G + seg.cw_offset[c], i * N)
            dec[i] = (w["bg"], crc_poly if args.early_stop else miphy.CRC_NONE, Z, args.max_iter, F, dec_in_len[c], 0, i * N, i * (K // 8))
    rdm_d = torch.from_numpy(rdm.view(np.uint8)).to(dev)
    dec_d = torch.from_numpy(dec.view(np.uint8)).to(dev)
    stream = torch.cuda.current_stream()

    # ---- front end of the slot: OFDM demodulation of the time-domain slot and DM-RS channel estimation (1 rx port,
    # 1 layer, DM-RS in symbol 2 like pusch_processor_benchmark.cpp:104-105). The soft demapper between the estimator
    # and the decoder is outside this path, so the codeword LLRs above are synthetic.
    ocfg = miphy.OfdmConfig(1, w["nprb"], 4096, 144, 1.0, 0.0, 3.5e9)
    slot_samples = ocfg.slot_size(0)
    nsc = w["nprb"] * 12
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    samples_d = torch.view_as_complex(torch.randn(S * slot_samples, 2, device=dev, generator=g) * 0.7071)
    grid_d = torch.zeros(S * 14 * nsc, dtype=torch.complex64, device=dev)
    ce_d = torch.zeros(S * 14 * nsc, dtype=torch.complex64, device=dev)
    sc_d = torch.zeros(S * 5, dtype=torch.float32, device=dev)
    ojobs = np.zeros(S, dtype=miphy.OfdmJob)
    cjobs = np.zeros(S, dtype=miphy.PuschChestJob)
    for s in range(S):
        ojobs[s] = (s * slot_samples, s * 14 * nsc, s % 2, 0)
        j = cjobs[s]
        j["numerology"], j["slot_in_frame"], j["scrambling_id"], j["scaling"] = 1, s % 20, 1, 1.0
        j["nof_tx_layers"], j["nof_rx_ports"], j["first_symbol"], j["nof_symbols"] = 1, 1, 0, 14
        j["rx_ports"] = [0, 1, 2, 3]
        j["symbols_mask"], j["grid_nof_prb"] = 1 << 2, w["nprb"]
        j["rb_mask"] = [0xFFFFFFFFFFFFFFFF] * 4 + [(1 << (w["nprb"] - 256)) - 1]
        j["grid_offset"], j["ce_offset"], j["scalars_offset"] = s * 14 * nsc, s * 14 * nsc, s * 5
    ojobs_d = torch.from_numpy(ojobs.view(np.uint8)).to(dev)
    cjobs_d = torch.from_numpy(cjobs.view(np.uint8)).to(dev)

    stages = ["ofdm_demod", "dmrs_chest", "rate_dematch", "ldpc_decode"]
    ev = {k: [] for k in stages}

    # One step = the whole batch of S slots. With --chunks G > 1 the batch is cut into G groups of slots that alternate between
    # two HIP streams, so the HBM-bound front end (OFDM, estimator, dematcher) of one group runs under the VALU-bound LDPC
    # decode of the other; every step still processes all S slots and the timed region ends with a device-wide sync.
    G_ch = max(1, min(args.chunks, S))
    bounds = [(S * i // G_ch, S * (i + 1) // G_ch) for i in range(G_ch)]
    streams = [stream] if G_ch == 1 else [torch.cuda.Stream(), torch.cuda.Stream()]
    max_E, dec_lim = max(seg.E[:C]), (Z, max(dec_in_len))

    def step(timed):
        for ci, (a, b) in enumerate(bounds):
            st = streams[ci % len(streams)]
            e = [torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)] if timed else None
            if timed:
                e[0].record(st)
            ctx.ofdm_demodulate_slots(ocfg, ojobs_d[a * 24:b * 24], samples_d, grid_d, st)
            if timed:
                e[1].record(st)
            ctx.dmrs_pusch_estimate_batch(cjobs_d[a * 96:b * 96], grid_d, ce_d, sc_d, st)
            if timed:
                e[2].record(st)
            # rate dematch in launches of <= 65535 codeblocks
            for x in range(a * C, b * C, 65535):
                y = min(b * C, x + 65535)
                ctx.ldpc_rate_dematch_batch(rdm_d[x * 32:y * 32], llr_d, softbuf_d, st, max_E=max_E)
            if timed:
                e[3].record(st)
            ctx.ldpc_decode_batch(dec_d[a * C * 32:b * C * 32], softbuf_d, bits_d, iters_d, st, limits=dec_lim)
            if timed:
                e[4].record(st)
                for i, k in enumerate(stages):
                    ev[k].append((e[i], e[i + 1]))

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- single-slot latency of the same pipeline (one slot = 38 codeblocks: the real-time unit of work), not part of `value`
    lat_us = None
    if rank == 0 and not args.no_latency:
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        for r in range(reps + 3):
            if r == 3:
                ev0.record(stream)
            ctx.ofdm_demodulate_slots(ocfg, ojobs_d[:24], samples_d, grid_d, stream)
            ctx.dmrs_pusch_estimate_batch(cjobs_d[:96], grid_d, ce_d, sc_d, stream)
            ctx.ldpc_rate_dematch_batch(rdm_d[:C * 32], llr_d, softbuf_d, stream, max_E=max(seg.E[:C]))
            ctx.ldpc_decode_batch(dec_d[:C * 32], softbuf_d, bits_d, iters_d, stream, limits=(Z, max(dec_in_len)))
        ev1.record(stream)
        torch.cuda.synchronize()
        lat_us = ev0.elapsed_time(ev1) / reps * 1e3

    # ---- correctness guard on what was just computed (not timed): every slot must decode to its TB
    bits = bits_d.cpu().numpy().reshape(S, C, K // 8)
    iters = iters_d.cpu().numpy().reshape(S, C)
    ok_slots = 0
    for s in range(min(S, 8)):
        od = O.OraclePuschDecoder(w["bg"], w["mod"], w["Nref"], w["nof_layers"], w["nsym"], w["tbs"] // 8)
        ok, tb, _ = od.decode(llrs_u[slot_src[s]], 0, True, args.max_iter, bool(args.early_stop))
        same = np.array_equal(od.cb_msgs.reshape(C, -1), bits[s])
        ok_slots += int(ok and same and np.array_equal(tb, tbs_u[slot_src[s]]))
    checked = min(S, 8)

    # per-launch durations (HIP events on the launching stream); a step has G_ch launches of each kernel
    launch_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in ev[k]])) for k in stages}
    kernel_ms = {k: launch_ms[k] * G_ch for k in stages}
    total_slots = S * args.steps * world
    info_bits = total_slots * w["tbs"]
    value = info_bits / dt
    # Roofline of the dominant kernel (LDPC decode): algorithmic bytes per codeblock = N LLR bytes in + K/8 bytes out
    # + 4 bytes iteration count; units per launch = S*C codeblocks.
    # Algorithmic bytes per launch: decode = LLRs in + K/8 out + 4 B iterations per codeblock;
    # dematch = E in + N out per codeblock; OFDM demod = 61440*8 in + 14*3276*8 out per slot-port; estimator = DM-RS REs in
    # (n_dmrs * 13104 B) + 14*3276*8 out per (slot, port, layer).
    alg = {"ldpc_decode": S * sum(dec_in_len[c] + K // 8 + 4 for c in range(C)),
           "rate_dematch": S * (G + C * N),
           "ofdm_demod": S * (slot_samples * 8 + 14 * nsc * 8),
           "dmrs_chest": S * (1 * (nsc // 2) * 8 + 14 * nsc * 8)}
    gbs = {k: alg[k] / (kernel_ms[k] * 1e-3) / 1e9 for k in stages}
    dom = max(kernel_ms, key=kernel_ms.get)
    achieved = gbs[dom]
    # HBM traffic per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE collected offline in separate rocprofv3 passes on
    # this same workload; the counters cannot be read from inside the process). Scales linearly with the slots per step.
    traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        for k, v in tj["kernels"].items():
            traffic[k] = v["hbm_bytes_per_launch"] * S / tj["slots_per_gpu_per_step"]
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "LDPC info-bits/sec + OFDM slots/sec, 100 MHz n78 273-PRB grid",
        "value": value,
        "unit": "info_bits/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int8",
        "data": "synthetic",
        "config": {"workload": "273-PRB 30kHz PUSCH, 256QAM R=948/1024, 1 layer, 38 CB/slot BG1 Z=384, TBS 319784; "
                               "per slot: OFDM demod (4096-pt, 1 port) + DM-RS channel estimate + rate-dematch + LDPC decode "
                               "(%d it, early_stop=%d)" % (args.max_iter, args.early_stop),
                   "slots_per_gpu_per_step": S, "codeblocks_per_step": S * C * world, "sigma": args.sigma,
                   "parallelism": "slots sharded across GPUs, no data-path collective"},
        "slots_per_s": total_slots / dt,
        "kernel_ms": kernel_ms,
        "kernel_algorithmic_GBps": gbs,
        "ofdm_slots_per_s": S * world / (kernel_ms["ofdm_demod"] * 1e-3),
        "roofline_ofdm": {"kernel": "ofdm_demod", "bound": "hbm", "achieved": gbs["ofdm_demod"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": gbs["ofdm_demod"] / HBM_PEAK_GBS, "traffic": traffic.get("ofdm_demod")},
        "single_slot_latency_us": lat_us,
        "mean_ldpc_iterations": float(iters.mean()) if args.early_stop else float(args.max_iter),
        "parity_check": "%d/%d slots identical to oracle" % (ok_slots, checked),
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get(dom), "algorithmic_bytes": alg[dom],
                     "note": "LDPC decode is LDS/VALU-bound; HBM fraction reported as required"},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(w, llrs_u, args.max_iter, bool(args.early_stop), args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if ok_slots != checked:
        print("PARITY FAILURE in bench", file=sys.stderr)
        sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
