#!/usr/bin/env python3
"""Headline benchmark: PUSCH receive hot path on synthetic 100 MHz n78 slots (273 PRB, 30 kHz SCS).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over a batch of S synthetic uplink slots per GPU: time-domain samples in, transport-block
bits out (OFDM demodulation -> DM-RS channel estimation -> equalise / soft-demap / descramble -> rate dematch -> LDPC decode).
The workload is BASELINE.json configs[2]: 273-PRB PUSCH, 256QAM R=948/1024, 1 layer, 38 codeblocks (BG1, Z=384) per slot,
TBS = 319 784 information bits per slot.  The slots are synthesised once, outside the timed region, by the transmit side of the
same library on the device (SCH encoder, PDSCH-style modulator, DM-RS mapper, OFDM modulator) plus AWGN, and are resident in HBM
before timing starts.

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


RNTI, N_ID, DMRS_SCR_ID = 0x4601, 935, 1
DMRS_SCALING = 1.4125375  # DM-RS boosted by 3 dB with two CDM groups without data (sch_dmrs_power.h)


def build_tx_grids(ctx, miphy, torch, dev, w, n_unique, seed):
    """Transmit side on the device (not timed): n_unique random transport blocks -> SCH encoder -> PDSCH-style modulator (scrambling,
    256QAM, mapping around the DM-RS symbol) -> type-1 DM-RS of every slot of a frame on symbol 2 (CP-OFDM uplink has the downlink's
    structure, so the PDSCH transmit blocks produce a valid PUSCH slot). Returns a device tensor complex64 [20 * 14 * nsc] (slot-in-frame
    k carries transport block k % n_unique) and the transport blocks (numpy)."""
    rng = np.random.default_rng(seed)
    G, nsc, nprb = w["nsym"] * w["mod"], w["nprb"] * 12, w["nprb"]
    tb_bytes = w["tbs"] // 8
    tbs = [rng.integers(0, 256, tb_bytes, dtype=np.uint8) for _ in range(n_unique)]
    td = np.zeros(n_unique, dtype=miphy.PdschTbDesc)
    for u in range(n_unique):
        td[u] = (w["bg"], w["rv"], w["mod"], w["nof_layers"], w["Nref"], w["nsym"], tb_bytes, u * tb_bytes, u * G)
    cw_d = torch.zeros(n_unique * G, dtype=torch.uint8, device=dev)
    ctx.pdsch_encode_batch(td, torch.from_numpy(np.concatenate(tbs)).to(dev), cw_d)
    grids = torch.zeros(20 * 14 * nsc, dtype=torch.complex64, device=dev)
    rb_words = [0xFFFFFFFFFFFFFFFF] * 4 + [(1 << (nprb - 256)) - 1]
    mj = np.zeros(20, dtype=miphy.PdschModJob)
    dj = np.zeros(20, dtype=miphy.DmrsPdschJob)
    for k in range(20):
        j = mj[k]
        j["rnti"], j["n_id"], j["scaling"], j["mod"], j["port"], j["start_symbol"], j["nof_symbols"] = RNTI, N_ID, 1.0, w["mod"], 0, 0, 14
        j["dmrs_type"], j["nof_cdm_groups_without_data"], j["dmrs_symbols_mask"] = 1, 2, 1 << 2
        j["grid_nof_prb"], j["bwp_start_rb"], j["bwp_size_rb"], j["nof_bits"] = nprb, 0, nprb, G
        j["rb_mask"] = rb_words
        j["cw_offset"], j["grid_offset"] = (k % n_unique) * G, k * 14 * nsc
        q = dj[k]
        q["slot_in_frame"], q["scrambling_id"], q["amplitude"], q["dmrs_type"], q["nof_ports"] = k, DMRS_SCR_ID, DMRS_SCALING, 1, 1
        q["symbols_mask"], q["grid_nof_prb"], q["rb_mask"], q["grid_offset"] = 1 << 2, nprb, rb_words, k * 14 * nsc
    assert miphy.pdsch_mod_nof_re(mj[0]) * w["mod"] == G
    ctx.pdsch_modulate_batch(mj, cw_d, grids)
    ctx.dmrs_pdsch_map_batch(dj, grids)
    torch.cuda.synchronize()
    return grids, tbs


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
    ap.add_argument("--slots", type=int, default=1024, help="slots per GPU per step (1024 slots = 38 912 codeblocks = exactly 38 rounds of the 1024 codeblocks the chip holds at once)")
    ap.add_argument("--max-iter", type=int, default=6)
    ap.add_argument("--early-stop", type=int, default=0)
    ap.add_argument("--snr-db", type=float, default=33.0, help="per-RE SNR of the synthesised slots")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--chunks", type=int, default=1, help="slot groups per step, alternated over two HIP streams (1 = one stream)")
    ap.add_argument("--chunk-streams", type=int, default=2, help="1: the slot groups run back to back on one stream (cache blocking only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-slot latency leg (keeps profiles to the timed step only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Rehearsal knob for a one-GPU box: MIPHY_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo for the barrier and the
    # MAX reduction of the elapsed time (RCCL refuses two ranks on one device). The measured multi-GPU runs never set it.
    rehearsal = os.environ.get("MIPHY_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import miphy
    import oracle_lib as O
    ctx = miphy.Context(local_rank)
    w = pusch_workload()
    S = args.slots
    # Segmentation of the transport block through the library's own host logic (ldpc.h:128-207 restated in csrc/sch.hip); the
    # rate-matched length of every codeblock is TS 38.212 5.4.2.1 (ldpc_segmenter_impl.cpp:104-141): the first codeblocks get the
    # floor, the last ones the ceiling of the per-codeblock share of the codeword.
    sg = miphy.sch_segmentation(w["tbs"] // 8, w["bg"])
    C, Z, N, K, F = sg.nof_cbs, sg.Z, sg.N, sg.K, sg.nof_filler_bits
    G = w["nsym"] * w["mod"]
    unit = w["nof_layers"] * w["mod"]
    n_short = C - (G // unit) % C
    seg_E = [unit * ((G // unit) // C) if c < n_short else unit * (-(-(G // unit) // C)) for c in range(C)]
    seg_off = [sum(seg_E[:c]) for c in range(C)]
    assert sum(seg_E) == G
    seg_crc_poly = miphy.CRC24B if C > 1 else (miphy.CRC16 if w["tbs"] <= 3824 else miphy.CRC24A)
    n_unique = 4
    nsc = w["nprb"] * 12
    grids_tx, tbs_u = build_tx_grids(ctx, miphy, torch, dev, w, n_unique, seed=1234 + rank)
    slot_src = np.arange(S) % n_unique  # transport block carried by slot s (slot s is slot-in-frame s % 20)

    # ---- transmit side + channel, once, outside the timed region: OFDM modulation of the S grids on the device and AWGN
    stream = torch.cuda.current_stream()
    mcfg = miphy.OfdmConfig(1, w["nprb"], 4096, 0, 1.0 / 64, 0.0, 3.5e9)
    ocfg = miphy.OfdmConfig(1, w["nprb"], 4096, 144, 1.0 / 64, 0.0, 3.5e9)  # unitary pair: demod(mod(grid)) == grid
    slot_samples = ocfg.slot_size(0)
    ojobs = np.zeros(S, dtype=miphy.OfdmJob)
    for s in range(S):
        ojobs[s] = (s * slot_samples, s * 14 * nsc, s % 2, 0)
    ojobs_d = torch.from_numpy(ojobs.view(np.uint8)).to(dev)
    samples_d = torch.zeros(S * slot_samples, dtype=torch.complex64, device=dev)
    grid_d = grids_tx.reshape(20, -1)[torch.arange(S, device=dev) % 20].reshape(-1).contiguous()
    ctx.ofdm_modulate_slots(mcfg, ojobs_d, grid_d, samples_d, stream)
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    noise_sigma = float(10.0 ** (-args.snr_db / 20.0))
    samples_d += torch.view_as_complex(torch.randn(S * slot_samples, 2, device=dev, generator=g) * (noise_sigma * 0.70710678))
    torch.cuda.synchronize()
    grid_d.zero_()  # from here on: the receiver's resource grid

    # ---- device-resident buffers and descriptors of the receive chain (everything below is HBM resident before timing starts)
    llr_d = torch.zeros(S * G, dtype=torch.int8, device=dev)                # codeword LLRs produced by the demodulator
    softbuf_d = torch.zeros(S * C * N, dtype=torch.int8, device=dev)        # HARQ soft buffers (device-resident pool)
    bits_d = torch.zeros(S * C * (K // 8), dtype=torch.uint8, device=dev)   # decoded codeblock messages
    iters_d = torch.zeros(S * C, dtype=torch.int32, device=dev)
    rdm = np.zeros(S * C, dtype=miphy.LdpcRdmDesc)
    dec = np.zeros(S * C, dtype=miphy.LdpcDecDesc)
    crc_poly = int(seg_crc_poly)
    # The decoder only needs the part of the soft buffer the dematcher can have written (new data, rv 0: E + fillers,
    # rounded up to a node); the rest is zero, which the reference trims away itself (ldpc_decoder_impl.cpp:86-99).
    dec_in_len = [min(N, max((22 + 2) * Z, -(-(seg_E[c] + F) // Z) * Z)) for c in range(C)]
    for s in range(S):
        for c in range(C):
            i = s * C + c
            rdm[i] = (w["bg"], w["rv"], w["mod"], 1, Z, F, w["Nref"], seg_E[c], s * G + seg_off[c], i * N)
            dec[i] = (w["bg"], crc_poly if args.early_stop else miphy.CRC_NONE, Z, args.max_iter, F, dec_in_len[c], 0, i * N, i * (K // 8))
    rdm_d = torch.from_numpy(rdm.view(np.uint8)).to(dev)
    dec_d = torch.from_numpy(dec.view(np.uint8)).to(dev)

    # ---- front end of the slot: 1 rx port, 1 layer, DM-RS type 1 in symbol 2 with two CDM groups without data (like
    # pusch_processor_benchmark.cpp:104-105), all 273 PRB allocated.
    ce_d = torch.zeros(S * 14 * nsc, dtype=torch.complex64, device=dev)
    sc_d = torch.zeros(S * 5, dtype=torch.float32, device=dev)
    cjobs = np.zeros(S, dtype=miphy.PuschChestJob)
    djobs = np.zeros(S, dtype=miphy.PuschDemodJob)
    rb_words = [0xFFFFFFFFFFFFFFFF] * 4 + [(1 << (w["nprb"] - 256)) - 1]
    for s in range(S):
        j = cjobs[s]
        j["numerology"], j["slot_in_frame"], j["scrambling_id"], j["scaling"] = 1, s % 20, DMRS_SCR_ID, DMRS_SCALING
        j["nof_tx_layers"], j["nof_rx_ports"], j["first_symbol"], j["nof_symbols"] = 1, 1, 0, 14
        j["rx_ports"] = [0, 1, 2, 3]
        j["symbols_mask"], j["grid_nof_prb"] = 1 << 2, w["nprb"]
        j["rb_mask"] = rb_words
        j["grid_offset"], j["ce_offset"], j["scalars_offset"] = s * 14 * nsc, s * 14 * nsc, s * 5
        q = djobs[s]
        q["rnti"], q["n_id"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = RNTI, N_ID, w["mod"], 1, 0, 14
        q["dmrs_type"], q["nof_cdm_groups_without_data"], q["ce_nof_symbols"] = 1, 2, 14
        q["rx_ports"] = [0, 1, 2, 3]
        q["dmrs_symbols_mask"], q["grid_nof_prb"], q["nof_llr"] = 1 << 2, w["nprb"], G
        q["rb_mask"] = rb_words
        q["grid_offset"], q["ce_offset"], q["scalars_offset"], q["llr_offset"] = s * 14 * nsc, s * 14 * nsc, s * 5, s * G
    assert miphy.pusch_demod_nof_llr(djobs[0]) == G
    cjobs_d = torch.from_numpy(cjobs.view(np.uint8)).to(dev)
    djobs_d = torch.from_numpy(djobs.view(np.uint8)).to(dev)

    stages = ["ofdm_demod", "dmrs_chest", "pusch_demod", "rate_dematch", "ldpc_decode"]
    ev = {k: [] for k in stages}

    # One step = the whole batch of S slots. With --chunks G > 1 the batch is cut into G groups of slots that alternate between
    # two HIP streams, so the HBM-bound front end (OFDM, estimator, dematcher) of one group runs under the VALU-bound LDPC
    # decode of the other; every step still processes all S slots and the timed region ends with a device-wide sync.
    G_ch = max(1, min(args.chunks, S))
    bounds = [(S * i // G_ch, S * (i + 1) // G_ch) for i in range(G_ch)]
    streams = [stream] if (G_ch == 1 or args.chunk_streams == 1) else [torch.cuda.Stream(), torch.cuda.Stream()]
    max_E, dec_lim = max(seg_E), (Z, max(dec_in_len))

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
            ctx.pusch_demodulate_batch(djobs_d[a * 104:b * 104], grid_d, ce_d, sc_d, llr_d, st)
            if timed:
                e[3].record(st)
            # rate dematch in launches of <= 65535 codeblocks
            for x in range(a * C, b * C, 65535):
                y = min(b * C, x + 65535)
                ctx.ldpc_rate_dematch_batch(rdm_d[x * 32:y * 32], llr_d, softbuf_d, st, max_E=max_E)
            if timed:
                e[4].record(st)
            ctx.ldpc_decode_batch(dec_d[a * C * 32:b * C * 32], softbuf_d, bits_d, iters_d, st, limits=dec_lim)
            if timed:
                e[5].record(st)
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
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
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
            ctx.pusch_demodulate_batch(djobs_d[:104], grid_d, ce_d, sc_d, llr_d, stream)
            ctx.ldpc_rate_dematch_batch(rdm_d[:C * 32], llr_d, softbuf_d, stream, max_E=max(seg_E))
            ctx.ldpc_decode_batch(dec_d[:C * 32], softbuf_d, bits_d, iters_d, stream, limits=(Z, max(dec_in_len)))
        ev1.record(stream)
        torch.cuda.synchronize()
        lat_us = ev0.elapsed_time(ev1) / reps * 1e3

    # ---- correctness guard on what was just computed (not timed). (1) Demodulator: the oracle, fed with the GPU's own resource
    # grid, channel estimate and noise variance of slot 0, must give the same LLRs bit for bit. (2) Decoder: the oracle decoder on the
    # GPU's LLRs must give the same codeblock messages. (3) End to end: every checked slot yields the transport block that was sent.
    checked = min(S, 8)
    llr_h = llr_d[:checked * G].cpu().numpy().reshape(checked, G)
    bits = bits_d.cpu().numpy().reshape(S, C, K // 8)
    iters = iters_d.cpu().numpy().reshape(S, C)
    g0 = grid_d[:14 * nsc].cpu().numpy().reshape(1, 14, nsc)
    h0 = ce_d[:14 * nsc].cpu().numpy().reshape(1, 14, nsc)
    dm = np.zeros(14, np.uint8)
    dm[2] = 1
    o_llr, _, _ = O.o_pusch_demodulate(RNTI, N_ID, w["mod"], 0, 14, dm, 0, 2, np.ones(w["nprb"], np.uint8), g0, h0, float(sc_d[2].item()))
    demod_ok = bool(np.array_equal(o_llr, llr_h[0]))
    ok_slots = 0
    for s in range(checked):
        od = O.OraclePuschDecoder(w["bg"], w["mod"], w["Nref"], w["nof_layers"], w["nsym"], w["tbs"] // 8)
        ok, tb, _ = od.decode(llr_h[s], 0, True, args.max_iter, bool(args.early_stop))
        same = np.array_equal(od.cb_msgs.reshape(C, -1), bits[s])
        ok_slots += int(ok and same and np.array_equal(tb, tbs_u[slot_src[s]]))
    if not demod_ok:
        ok_slots = -1

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
    # (n_dmrs * 13104 B) + 14*3276*8 out per (slot, port, layer); demodulator = 16 B in + mod B out per data RE and port.
    alg = {"ldpc_decode": S * sum(dec_in_len[c] + K // 8 + 4 for c in range(C)),
           "rate_dematch": S * (G + C * N),
           "ofdm_demod": S * (slot_samples * 8 + 14 * nsc * 8),
           "dmrs_chest": S * (1 * (nsc // 2) * 8 + 14 * nsc * 8),
           "pusch_demod": S * (w["nsym"] * (8 + 8) + G)}  # received RE + channel coefficient in, mod LLR bytes out
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
    # VALU-issue roofline of the decoder (it is not an HBM kernel): wave64 VALU instructions per launch from the SQ counters of the
    # same workload (profiles/r01_pmc_sq_v8_all_kernels.csv, SQ_INSTS_VALU of a 38 912-codeblock launch; scales with the codeblocks) over the launch time, against
    # 1024 SIMDs x (2.4 GHz / 4 cycles per wave64 instruction).
    valu = None
    try:
        import csv
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_pmc_sq_v8_all_kernels.csv"))):
            if r["kernel"] == "ldpc_decode_pk_kernel" and r["counter"] == "SQ_INSTS_VALU":
                per_cb = float(r["mean_per_dispatch"]) / 38912.0
                ach = per_cb * S * C / (kernel_ms["ldpc_decode"] * 1e-3) / 1e9
                valu = {"kernel": "ldpc_decode", "bound": "valu_issue", "achieved": ach, "peak": 1024 * 2.4 / 4, "unit": "G wave-instr/s",
                        "frac": ach / (1024 * 2.4 / 4), "valu_wave_instructions_per_codeblock": per_cb}
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
                               "per slot, time-domain samples to transport block: OFDM demod (4096-pt, 1 port) + DM-RS channel estimate + "
                               "equalise/soft-demap/descramble + rate-dematch + LDPC decode (%d it, early_stop=%d); slots synthesised by "
                               "the transmit chain + AWGN" % (args.max_iter, args.early_stop),
                   "slots_per_gpu_per_step": S, "codeblocks_per_step": S * C * world, "snr_db": args.snr_db,
                   "parallelism": "slots sharded across GPUs, no data-path collective"},
        "slots_per_s": total_slots / dt,
        "kernel_ms": kernel_ms,
        "kernel_algorithmic_GBps": gbs,
        "ofdm_slots_per_s": S * world / (kernel_ms["ofdm_demod"] * 1e-3),
        "roofline_ofdm": {"kernel": "ofdm_demod", "bound": "hbm", "achieved": gbs["ofdm_demod"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": gbs["ofdm_demod"] / HBM_PEAK_GBS, "traffic": traffic.get("ofdm_demod")},
        "roofline_valu": valu,
        "single_slot_latency_us": lat_us,
        "mean_ldpc_iterations": float(iters.mean()) if args.early_stop else float(args.max_iter),
        "parity_check": "%d/%d slots: LLRs and codeblocks identical to the oracle, transport block recovered" % (ok_slots, checked),
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get(dom), "algorithmic_bytes": alg[dom],
                     "note": "LDPC decode is LDS/VALU-bound; HBM fraction reported as required"},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(w, llr_h[:4], args.max_iter, bool(args.early_stop), args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if ok_slots != checked:
        print("PARITY FAILURE in bench", file=sys.stderr)
        sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
