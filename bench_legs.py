"""Extra legs of bench.py (round 3): a slot that mixes allocation sizes and MCS, the downlink transmit chain, the polar CPU reference and the
compressed-IQ ingest. Every leg verifies what it computed (transport blocks recovered / oracle parity) and carries the reference's own CPU
chain beside the GPU figure where oracle/_ref is present. Nothing here is part of `value`."""
import time

import numpy as np

RNTI0, N_ID0, DMRS_SCR_ID = 0x4601, 935, 1
DMRS_SCALING = 1.4125375  # DM-RS boosted by 3 dB with two CDM groups without data (sch_dmrs_power.h)
# The 23.5 pdsch_processor takes the rate-matching buffer size from the PDU (encoder Nref = tbs_lbrm_bytes * 8, pdsch_processor_impl.cpp:192,238,
# at most ldpc::MAX_CODEBLOCK_SIZE / 8 = 66 * 384 / 8 bytes); its benchmark passes exactly that maximum: the full circular buffer.
LBRM_BYTES = 66 * 384 // 8

# One 273-PRB slot shared by eight UEs: (first PRB, PRBs, bits per symbol, TBS bits, base graph, R x 1024). 14 symbols, DM-RS in symbol 2
# (12 RE per PRB with two CDM groups without data), one layer. TBS and base graph are what the reference's tbs_calculator_calculate /
# get_ldpc_base_graph return for (PRBs, modulation, R) -- tests/test_oracle_vs_ref.py::test_mixed_slot_table_is_the_reference_calculators pins it.
MIXED_PDUS = [
    (0, 4, 2, 144, 2, 120),        # BG2 Z=28, 1 codeblock, ~41 layers (QPSK R=120/1024: the lowest MCS) -> wave kernel
    (4, 8, 2, 768, 2, 308),        # BG2 Z=80, 1 codeblock, ~24 layers -> packed kernel, one wavefront per codeblock
    (12, 16, 4, 3752, 2, 378),     # BG2 Z=384, 1 codeblock, ~19 layers
    (28, 25, 4, 9992, 1, 658),     # BG1 Z=240, 2 codeblocks, ~14 layers -> two wavefronts per codeblock
    (53, 30, 6, 15624, 1, 567),    # BG1 Z=384, 2 codeblocks, ~19 layers
    (83, 40, 6, 31752, 1, 873),    # BG1 Z=384, 4 codeblocks, 6 layers
    (123, 60, 8, 58384, 1, 797),   # BG1 Z=384, 7 codeblocks, ~9 layers
    (183, 90, 8, 104496, 1, 948),  # BG1 Z=384, 13 codeblocks, 4 layers (256QAM R=948/1024: the highest MCS)
]


def ev_ms(torch, f, reps, stream=None):
    st = stream or torch.cuda.current_stream()
    for _ in range(2):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps):
        f()
    b.record(st)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def rb_mask_words(start, count):
    m = ((1 << count) - 1) << start
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(5)]


def host_threads(O):
    cpus, quota = O.host_cpus()
    t_all = len(cpus) if quota is None else max(1, min(len(cpus), int(round(quota))))
    return cpus, t_all


# ------------------------------------------------------------------------------------------------ mixed slot
def mixed_slot_leg(ctx, miphy, torch, dev, S, max_iter, snr_db, seed, cpu_seconds, with_cpu):
    """S slots of 273 PRB, each carrying the eight PDUs of MIXED_PDUS (4 ... 90 PRB, QPSK R=120 ... 256QAM R=948, BG2 Z=7 next to BG1 Z=384).
    Transmit side on the device (SCH encoder, modulator, DM-RS per PDU, OFDM modulator) + AWGN, once; the timed step is the receive chain
    samples -> transport blocks: OFDM demodulation per slot, then estimator / demodulator / decode plan over the 8 S PDUs. The decode plan
    sorts the codeblocks into launch classes (lifting size, base graph, layers); `ldpc_launches` is their number."""
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    nprb_grid, nsc, NP, n_unique = 273, 273 * 12, len(MIXED_PDUS), 4
    segs = [miphy.sch_segmentation(t // 8, bg) for (_, _, _, t, bg, _) in MIXED_PDUS]
    G = [n * 156 * m for (_, n, m, _, _, _) in MIXED_PDUS]
    tbb = [t // 8 for (_, _, _, t, _, _) in MIXED_PDUS]
    C = [sg.nof_cbs for sg in segs]
    info_bits = sum(t for (_, _, _, t, _, _) in MIXED_PDUS)
    # ---- transmit: n_unique sets of eight transport blocks
    tbs = [[rng.integers(0, 256, tbb[u], dtype=np.uint8) for u in range(NP)] for _ in range(n_unique)]
    td = np.zeros(n_unique * NP, dtype=miphy.PdschTbDesc)
    tb_off, cw_off, tb_offs, cw_offs = 0, 0, [], []
    for k in range(n_unique):
        for u, (_, n, m, t, bg, _) in enumerate(MIXED_PDUS):
            td[k * NP + u] = (bg, 0, m, 1, 0, n * 156, tbb[u], tb_off, cw_off)
            tb_offs.append(tb_off), cw_offs.append(cw_off)
            tb_off += (tbb[u] + 15) // 16 * 16
            cw_off += G[u]
    tb_all = np.zeros(tb_off, dtype=np.uint8)
    for k in range(n_unique):
        for u in range(NP):
            tb_all[tb_offs[k * NP + u]:tb_offs[k * NP + u] + tbb[u]] = tbs[k][u]
    cw_d = torch.zeros(cw_off, dtype=torch.uint8, device=dev)
    ctx.pdsch_encode_batch(td, torch.from_numpy(tb_all).to(dev), cw_d)
    grids = torch.zeros(20 * 14 * nsc, dtype=torch.complex64, device=dev)
    mj = np.zeros(20 * NP, dtype=miphy.PdschModJob)
    dj = np.zeros(20 * NP, dtype=miphy.DmrsPdschJob)
    for k in range(20):
        for u, (rb0, n, m, _, _, _) in enumerate(MIXED_PDUS):
            j = mj[k * NP + u]
            j["rnti"], j["n_id"], j["scaling"], j["mod"], j["port"], j["start_symbol"], j["nof_symbols"] = RNTI0 + u, N_ID0 + u, 1.0, m, 0, 0, 14
            j["dmrs_type"], j["nof_cdm_groups_without_data"], j["dmrs_symbols_mask"] = 1, 2, 1 << 2
            j["grid_nof_prb"], j["bwp_start_rb"], j["bwp_size_rb"], j["nof_bits"] = nprb_grid, 0, nprb_grid, G[u]
            j["rb_mask"] = rb_mask_words(rb0, n)
            j["cw_offset"], j["grid_offset"] = cw_offs[(k % n_unique) * NP + u], k * 14 * nsc
            q = dj[k * NP + u]
            q["slot_in_frame"], q["scrambling_id"], q["amplitude"], q["dmrs_type"], q["nof_ports"] = k, DMRS_SCR_ID, DMRS_SCALING, 1, 1
            q["symbols_mask"], q["grid_nof_prb"], q["rb_mask"], q["grid_offset"] = 1 << 2, nprb_grid, rb_mask_words(rb0, n), k * 14 * nsc
            assert miphy.pdsch_mod_nof_re(mj[k * NP + u]) * m == G[u]
    ctx.pdsch_modulate_batch(mj, cw_d, grids)
    ctx.dmrs_pdsch_map_batch(dj, grids)
    st = torch.cuda.current_stream()
    mcfg = miphy.OfdmConfig(1, nprb_grid, 4096, 0, 1.0 / 64, 0.0, 3.5e9)
    ocfg = miphy.OfdmConfig(1, nprb_grid, 4096, 144, 1.0 / 64, 0.0, 3.5e9)
    ss = ocfg.slot_size(0)
    oj = np.zeros(S, dtype=miphy.OfdmJob)
    for s in range(S):
        oj[s] = (s * ss, s * 14 * nsc, s % 2, 0)
    oj_d = torch.from_numpy(oj.view(np.uint8)).to(dev)
    samples = torch.zeros(S * ss, dtype=torch.complex64, device=dev)
    grid = grids.reshape(20, -1)[torch.arange(S, device=dev) % 20].reshape(-1).contiguous()
    ctx.ofdm_modulate_slots(mcfg, oj_d, grid, samples, st)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    samples += torch.view_as_complex(torch.randn(S * ss, 2, device=dev, generator=gen) * (float(10.0 ** (-snr_db / 20.0)) * 0.70710678))
    torch.cuda.synchronize()
    grid.zero_()
    # ---- receive descriptors: per (slot, PDU) an estimator job, a demodulator job and a transport-block record
    cj = np.zeros(S * NP, dtype=miphy.PuschChestJob)
    qj = np.zeros(S * NP, dtype=miphy.PuschDemodJob)
    tdr = np.zeros(S * NP, dtype=miphy.PuschTbDesc)
    Gsum, Csum, tb_slot = sum(G), sum(C), sum((b + 15) // 16 * 16 for b in tbb)
    g_off, c_off, t_off = np.concatenate([[0], np.cumsum(G)[:-1]]), np.concatenate([[0], np.cumsum(C)[:-1]]), np.concatenate(
        [[0], np.cumsum([(b + 15) // 16 * 16 for b in tbb])[:-1]])
    for s in range(S):
        for u, (rb0, n, m, t, bg, _) in enumerate(MIXED_PDUS):
            i = s * NP + u
            j = cj[i]
            j["numerology"], j["slot_in_frame"], j["scrambling_id"], j["scaling"] = 1, s % 20, DMRS_SCR_ID, DMRS_SCALING
            j["nof_tx_layers"], j["nof_rx_ports"], j["first_symbol"], j["nof_symbols"], j["rx_ports"] = 1, 1, 0, 14, [0, 1, 2, 3]
            j["symbols_mask"], j["grid_nof_prb"], j["ce_compact"], j["rb_mask"] = 1 << 2, nprb_grid, 1, rb_mask_words(rb0, n)
            # the allocations of a slot are disjoint, so its PDUs share ONE estimate row (each job writes its own PRBs)
            j["grid_offset"], j["ce_offset"], j["scalars_offset"] = s * 14 * nsc, s * nsc, i * 5
            q = qj[i]
            q["rnti"], q["n_id"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = RNTI0 + u, N_ID0 + u, m, 1, 0, 14
            q["dmrs_type"], q["nof_cdm_groups_without_data"], q["ce_nof_symbols"], q["ce_compact"], q["rx_ports"] = 1, 2, 14, 1, [0, 1, 2, 3]
            q["dmrs_symbols_mask"], q["grid_nof_prb"], q["nof_llr"], q["rb_mask"] = 1 << 2, nprb_grid, G[u], rb_mask_words(rb0, n)
            q["grid_offset"], q["ce_offset"], q["scalars_offset"], q["llr_offset"] = s * 14 * nsc, s * nsc, i * 5, s * Gsum + int(g_off[u])
            tdr[i] = (bg, 0, m, 1, 1, 0, max_iter, 0, n * 156, tbb[u], s * Csum + int(c_off[u]), s * Gsum + int(g_off[u]), s * tb_slot + int(t_off[u]))
    cj_d, qj_d = (torch.from_numpy(a.view(np.uint8)).to(dev) for a in (cj, qj))
    ce = torch.zeros(S * nsc, dtype=torch.complex64, device=dev)
    sc = torch.zeros(S * NP * 5, dtype=torch.float32, device=dev)
    llr = torch.zeros(S * Gsum, dtype=torch.int8, device=dev)
    soft = torch.zeros(S * Csum * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev)
    msgs = torch.zeros(S * Csum * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)
    crc = torch.zeros(S * Csum, dtype=torch.uint8, device=dev)
    tb = torch.zeros(S * tb_slot, dtype=torch.uint8, device=dev)
    res = torch.zeros(S * NP * miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)
    plan = ctx.pusch_decode_plan(tdr)
    plan.enable_timing(16)
    names = ["ofdm_demod", "dmrs_chest", "pusch_demod"]
    acc = {k: [] for k in names}

    def step(timed):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed:
            e[0].record(st)
        ctx.ofdm_demodulate_slots(ocfg, oj_d, samples, grid, st)
        if timed:
            e[1].record(st)
        ctx.dmrs_pusch_estimate_batch(cj_d, grid, ce, sc, st, max_ports=1, max_layers=1)
        if timed:
            e[2].record(st)
        ctx.pusch_demodulate_batch(qj_d, grid, ce, sc, llr, st)
        if timed:
            e[3].record(st)
            for i, k in enumerate(names):
                acc[k].append((e[i], e[i + 1]))
        plan.run(llr, soft, msgs, crc, tb, res, st)

    for _ in range(2):
        step(False)
    torch.cuda.synchronize()
    plan.read_timing()
    miphy.lib().miphy_debug_ldpc_kernels_used(1)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        step(True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    used = int(miphy.lib().miphy_debug_ldpc_kernels_used(1))
    kms = {k: float(np.mean([a.elapsed_time(b) for a, b in acc[k]])) for k in names}
    kms.update(plan.read_timing())
    # ---- verification: every transport block of the step CRC-ok and equal to the transmitted one; the oracle decoder on the GPU's LLRs of the
    # first slots gives the same verdicts, iteration statistics and bytes
    r = res.cpu().numpy().view(miphy.PuschResult).reshape(S, NP)
    tb_h = tb.cpu().numpy().reshape(S, tb_slot)
    good = 0
    for s in range(S):
        for u in range(NP):
            good += int(r[s, u]["tb_crc_ok"] != 0 and np.array_equal(tb_h[s, int(t_off[u]):int(t_off[u]) + tbb[u]], tbs[(s % 20) % n_unique][u]))
    llr_h = llr[:2 * Gsum].cpu().numpy().reshape(2, Gsum)
    parity = 0
    for s in range(2):
        for u, (_, n, m, t, bg, _) in enumerate(MIXED_PDUS):
            od = O.OraclePuschDecoder(bg, m, 0, 1, n * 156, tbb[u])
            ok, tbo, mm = od.decode(llr_h[s, int(g_off[u]):int(g_off[u]) + G[u]], 0, True, max_iter, False)
            parity += int(bool(ok) == bool(r[s, u]["tb_crc_ok"]) and mm == (int(r[s, u]["iters_min"]), int(r[s, u]["iters_max"])) and
                          (not ok or np.array_equal(tbo, tb_h[s, int(t_off[u]):int(t_off[u]) + tbb[u]])))
    out = {"config": "273-PRB slot shared by 8 PUSCH PDUs: PRBs %s, (Qm, R x 1024) %s, TBS %s, codeblocks %s, lifting sizes %s; samples -> transport blocks, "
                     "%d LDPC iterations, no early stop" % ([p[1] for p in MIXED_PDUS], [(p[2], p[5]) for p in MIXED_PDUS], [p[3] for p in MIXED_PDUS], C,
                                                            [sg.Z for sg in segs], max_iter),
           "slots": S, "pdus_per_slot": NP, "codeblocks_per_slot": Csum, "info_bits_per_slot": info_bits, "ms_per_step": dt * 1e3, "kernel_ms": kms,
           "slots_per_s": S / dt, "info_bits_per_s": S * info_bits / dt, "ldpc_launches": plan.nof_launches(),
           "ldpc_kernels_used": [n_ for b_, n_ in ((1, "one_row_per_lane"), (2, "packed"), (4, "fused_dematch"), (8, "messages_in_global_memory"), (16, "wave_multi_codeblock")) if used & b_],
           "transport_blocks_recovered": good, "transport_blocks": S * NP, "oracle_parity_pdus": "%d/%d" % (parity, 2 * NP)}
    plan.close()
    if with_cpu and O.ref_available():
        cpus, t_all = host_threads(O)
        samples4 = samples[:4 * ss].cpu().numpy().reshape(4, ss)
        pd = [(rb0, n, m, t, bg, RNTI0 + u, N_ID0 + u, R) for u, (rb0, n, m, t, bg, R) in enumerate(MIXED_PDUS)]
        for key, T in (("cpu_reference_all_cores", t_all), ("cpu_reference_t1", 1)):
            dtc, done, ok = O.r_pusch_chain_bench_multi(T, cpus, cpu_seconds, 1, samples4, nprb_grid, pd, DMRS_SCR_ID, 4096, 144, 1.0 / 64, 3.5e9, max_iter, 0)
            out[key] = {"value": done * info_bits / dtc, "unit": "info_bits/s", "cores": T, "kind": "reference", "slots_per_s": done / dtc,
                        "sample": "%d slots (%d of %d transport blocks CRC ok) in %.1f s: srsRAN ofdm_slot_demodulator (generic DFT) once per slot + pusch_processor "
                                  "(AVX2 LDPC) per PDU, one instance per pinned thread, the GPU run's first 4 slots of samples" % (done, ok, done * NP, dtc)}
    return out, (good == S * NP and parity == 2 * NP)


# ------------------------------------------------------------------------------------------------ downlink transmit chain
def pdsch_tx_leg(ctx, miphy, torch, dev, w, S, max_iter, cpu_seconds, with_cpu, hbm_peak):
    """north_star's transmit half: S transport blocks of the headline allocation (273 PRB, 256QAM R=948/1024, TBS 319 784) -> pdsch_processor
    (TB CRC + segmentation + LDPC encode + rate match, scrambling + 256QAM + RE mapping, DM-RS) -> ofdm_slot_modulator. Per-stage HIP-event times
    of the separate entry points, the composed miphy_pdsch_process_batch + miphy_ofdm_modulate_slots as the chain figure; the produced samples
    are received again by the uplink-style chain of this library (same waveform structure) and the transport blocks compared."""
    import oracle_lib as O
    rng = np.random.default_rng(5)
    nprb, nsc, mod, tb_bytes = w["nprb"], w["nprb"] * 12, w["mod"], w["tbs"] // 8
    G = w["nsym"] * mod
    sg = miphy.sch_segmentation(tb_bytes, w["bg"])
    C = sg.nof_cbs
    st = torch.cuda.current_stream()
    n_u = 8
    tb_u = rng.integers(0, 256, (n_u, tb_bytes), dtype=np.uint8)
    tb_stride = (tb_bytes + 15) // 16 * 16
    tb_all = np.zeros((S, tb_stride), dtype=np.uint8)
    tb_all[:, :tb_bytes] = tb_u[np.arange(S) % n_u]
    tb_d = torch.from_numpy(tb_all.reshape(-1)).to(dev)
    rbw = rb_mask_words(0, nprb)
    # separate entry points (per-stage times)
    td = np.zeros(S, dtype=miphy.PdschTbDesc)
    mj = np.zeros(S, dtype=miphy.PdschModJob)
    dj = np.zeros(S, dtype=miphy.DmrsPdschJob)
    pdus = np.zeros(S, dtype=miphy.PdschPdu)
    for s in range(S):
        td[s] = (w["bg"], 0, mod, 1, 8 * LBRM_BYTES, w["nsym"], tb_bytes, s * tb_stride, s * G)
        j = mj[s]
        j["rnti"], j["n_id"], j["scaling"], j["mod"], j["port"], j["start_symbol"], j["nof_symbols"] = RNTI0, N_ID0, 1.0, mod, 0, 0, 14
        j["dmrs_type"], j["nof_cdm_groups_without_data"], j["dmrs_symbols_mask"] = 1, 2, 1 << 2
        j["grid_nof_prb"], j["bwp_start_rb"], j["bwp_size_rb"], j["nof_bits"], j["rb_mask"] = nprb, 0, nprb, G, rbw
        j["cw_offset"], j["grid_offset"] = s * G, s * 14 * nsc
        q = dj[s]
        q["slot_in_frame"], q["scrambling_id"], q["amplitude"], q["dmrs_type"], q["nof_ports"] = s % 20, DMRS_SCR_ID, DMRS_SCALING, 1, 1
        q["symbols_mask"], q["grid_nof_prb"], q["rb_mask"], q["grid_offset"] = 1 << 2, nprb, rbw, s * 14 * nsc
        p = pdus[s]
        p["slot_in_frame"], p["rnti"], p["n_id"], p["dmrs_scrambling_id"], p["tbs_lbrm_bytes"], p["tb_bytes"] = s % 20, RNTI0, N_ID0, DMRS_SCR_ID, LBRM_BYTES, tb_bytes
        p["ratio_pdsch_dmrs_to_sss_dB"], p["ratio_pdsch_data_to_sss_dB"] = -3.0, 0.0  # the processor scales by 10^(-ratio / 20): DM-RS boosted by 3 dB
        p["bg"], p["rv"], p["mod"], p["port"], p["start_symbol"], p["nof_symbols"], p["nof_cdm_groups_without_data"] = w["bg"], 0, mod, 0, 0, 14, 2
        p["dmrs_symbols_mask"], p["grid_nof_prb"], p["bwp_start_rb"], p["bwp_size_rb"], p["rb_mask"] = 1 << 2, nprb, 0, nprb, rbw
        p["tb_offset"], p["grid_offset"] = s * tb_stride, s * 14 * nsc
    mj_d, dj_d = (torch.from_numpy(a.view(np.uint8)).to(dev) for a in (mj, dj))
    cw = torch.zeros(S * G, dtype=torch.uint8, device=dev)
    grid = torch.zeros(S * 14 * nsc, dtype=torch.complex64, device=dev)
    mcfg = miphy.OfdmConfig(1, nprb, 4096, 0, 1.0 / 64, 0.0, 3.5e9)
    ocfg = miphy.OfdmConfig(1, nprb, 4096, 144, 1.0 / 64, 0.0, 3.5e9)
    ss = ocfg.slot_size(0)
    oj = np.zeros(S, dtype=miphy.OfdmJob)
    for s in range(S):
        oj[s] = (s * ss, s * 14 * nsc, s % 2, 0)
    oj_d = torch.from_numpy(oj.view(np.uint8)).to(dev)
    samples = torch.zeros(S * ss, dtype=torch.complex64, device=dev)
    kms = {"pdsch_encode": ev_ms(torch, lambda: ctx.pdsch_encode_batch(td, tb_d, cw, st), 5),
           "pdsch_modulate": ev_ms(torch, lambda: ctx.pdsch_modulate_batch(mj_d, cw, grid, st), 5),
           "dmrs_pdsch": ev_ms(torch, lambda: ctx.dmrs_pdsch_map_batch(dj_d, grid, st), 5),
           "ofdm_mod": ev_ms(torch, lambda: ctx.ofdm_modulate_slots(mcfg, oj_d, grid, samples, st), 5)}

    plan = miphy.PdschProcessPlan(ctx, pdus)

    def chain():
        plan.run(tb_d, grid, st)
        ctx.ofdm_modulate_slots(mcfg, oj_d, grid, samples, st)

    def chain_per_call():  # host descriptors every call: validation, segmentation and MBs of descriptors through the staging ring per call
        ctx.pdsch_process_batch(pdus, tb_d, grid, st)
        ctx.ofdm_modulate_slots(mcfg, oj_d, grid, samples, st)

    t0 = time.perf_counter()
    for _ in range(3):
        chain_per_call()
    torch.cuda.synchronize()
    ms_per_call = (time.perf_counter() - t0) / 3 * 1e3
    grid_pc = grid.clone()
    grid.zero_()
    ms_chain = ev_ms(torch, chain, 5)
    # the encoder stage as the PLAN runs it (TB CRC + codeblock kernels, no host work): the plan alone minus its modulator and DM-RS stages; the
    # entry `pdsch_encode` of kernel_ms above is the per-call entry point with its host-side segmentation and descriptor upload
    ms_plan = ev_ms(torch, lambda: plan.run(tb_d, grid, st), 5)
    kms["pdsch_encode_in_plan"] = max(1e-6, ms_plan - kms["pdsch_modulate"] - kms["dmrs_pdsch"])
    plan_equals_per_call = bool(torch.equal(torch.view_as_real(grid), torch.view_as_real(grid_pc)))
    # algorithmic bytes: TB in + rate-matched codeword out (one byte per bit) | codeword in + data REs out | DM-RS REs out | grid in + samples out
    alg = {"pdsch_encode": S * (tb_bytes + G), "pdsch_encode_in_plan": S * (tb_bytes + G), "pdsch_modulate": S * (G + w["nsym"] * 8), "dmrs_pdsch": S * (nsc // 2) * 8, "ofdm_mod": S * (14 * nsc * 8 + ss * 8)}
    gbs = {k: alg[k] / (kms[k] * 1e-3) / 1e9 for k in kms}
    # ---- verification (not timed): (1) the plan's grid equals the per-call grid and the grid of the separate entry points; (2) the codeword of slot 0
    # equals the oracle's (pinned against the reference encoder); (3) the samples demodulate back to the grid; (4) the receive chain of this library
    # (OFDM demodulator, estimator, demodulator, decoder) recovers the transport blocks of 8 slots from the samples, and the hard decisions of its
    # LLRs are the codeword bits.
    grid_c = grid.clone()
    grid.zero_()
    ctx.pdsch_encode_batch(td, tb_d, cw, st)
    ctx.pdsch_modulate_batch(mj_d, cw, grid, st)
    ctx.dmrs_pdsch_map_batch(dj_d, grid, st)
    torch.cuda.synchronize()
    # (the composed call derives the DM-RS amplitude as 10^(3/20) like the reference, the separate job carries the rounded constant: one unit in the last place)
    grid_diff = float((grid - grid_c).abs().max())
    same_grid = grid_diff <= 1e-6
    cw_oracle = bool(np.array_equal(cw[:G].cpu().numpy(), O.o_pdsch_encode(w["bg"], 0, mod, 8 * LBRM_BYTES, 1, w["nsym"], tb_u[0])))
    nchk = min(S, 8)
    grid_rx = torch.zeros(nchk * 14 * nsc, dtype=torch.complex64, device=dev)
    cj = np.zeros(nchk, dtype=miphy.PuschChestJob)
    qj = np.zeros(nchk, dtype=miphy.PuschDemodJob)
    for s in range(nchk):
        j = cj[s]
        j["numerology"], j["slot_in_frame"], j["scrambling_id"], j["scaling"] = 1, s % 20, DMRS_SCR_ID, DMRS_SCALING
        j["nof_tx_layers"], j["nof_rx_ports"], j["first_symbol"], j["nof_symbols"], j["rx_ports"] = 1, 1, 0, 14, [0, 1, 2, 3]
        j["symbols_mask"], j["grid_nof_prb"], j["ce_compact"], j["rb_mask"] = 1 << 2, nprb, 1, rbw
        j["grid_offset"], j["ce_offset"], j["scalars_offset"] = s * 14 * nsc, s * nsc, s * 5
        q = qj[s]
        q["rnti"], q["n_id"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = RNTI0, N_ID0, mod, 1, 0, 14
        q["dmrs_type"], q["nof_cdm_groups_without_data"], q["ce_nof_symbols"], q["ce_compact"], q["rx_ports"] = 1, 2, 14, 1, [0, 1, 2, 3]
        q["dmrs_symbols_mask"], q["grid_nof_prb"], q["nof_llr"], q["rb_mask"] = 1 << 2, nprb, G, rbw
        q["grid_offset"], q["ce_offset"], q["scalars_offset"], q["llr_offset"] = s * 14 * nsc, s * nsc, s * 5, s * G
    ce = torch.zeros(nchk * nsc, dtype=torch.complex64, device=dev)
    sc = torch.zeros(nchk * 5, dtype=torch.float32, device=dev)
    llr = torch.zeros(nchk * G, dtype=torch.int8, device=dev)
    tdr = np.zeros(nchk, dtype=miphy.PuschTbDesc)
    for s in range(nchk):
        tdr[s] = (w["bg"], 0, mod, 1, 1, 0, max_iter, 8 * LBRM_BYTES, w["nsym"], tb_bytes, s * C, s * G, s * tb_bytes)
    soft = torch.zeros(nchk * C * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev)
    msgs = torch.zeros(nchk * C * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)
    crc = torch.zeros(nchk * C, dtype=torch.uint8, device=dev)
    tbo = torch.zeros(nchk * tb_bytes, dtype=torch.uint8, device=dev)
    res = torch.zeros(nchk * miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)
    ctx.ofdm_demodulate_slots(ocfg, oj[:nchk], samples, grid_rx, st)
    ctx.dmrs_pusch_estimate_batch(cj, grid_rx, ce, sc, st)
    ctx.pusch_demodulate_batch(qj, grid_rx, ce, sc, llr, st)
    ctx.pusch_decode_batch(tdr, llr, soft, msgs, crc, tbo, res, st)
    torch.cuda.synchronize()
    rr = res.cpu().numpy().view(miphy.PuschResult)
    tb_back = int(sum(int(rr[s]["tb_crc_ok"] != 0 and np.array_equal(tbo[s * tb_bytes:(s + 1) * tb_bytes].cpu().numpy(), tb_u[s % n_u])) for s in range(nchk)))
    ref_g = grid[:nchk * 14 * nsc]
    ofdm_err = float((grid_rx - ref_g).abs().max() / ref_g.abs().pow(2).mean().sqrt())
    bit_errors = int(((llr < 0).to(torch.uint8) != cw[:nchk * G]).sum().item())
    verified = same_grid and cw_oracle and ofdm_err < 2e-4 and bit_errors == 0 and plan_equals_per_call and tb_back == nchk
    plan.close()
    out = {"config": "%d transport blocks of the headline allocation (273 PRB, 256QAM R=948/1024, TBS %d, %d codeblocks BG1 Z=%d) -> pdsch_processor (CRC, segmentation, LDPC "
                     "encode, rate match, scrambling, modulation, RE mapping, DM-RS) -> ofdm_slot_modulator (4096-point)" % (S, w["tbs"], C, sg.Z),
           "slots": S, "ms_per_step": ms_chain, "entry_points": "miphy_pdsch_process_plan_run (descriptors prepared once) + miphy_ofdm_modulate_slots",
           "ms_per_step_host_descriptors_every_call": ms_per_call, "info_bits_per_s_host_descriptors_every_call": S * w["tbs"] / (ms_per_call * 1e-3),
           "kernel_ms": kms, "kernel_algorithmic_GBps": gbs,
           "kernel_hbm_frac": {k: gbs[k] / hbm_peak for k in gbs},
           "info_bits_per_s": S * w["tbs"] / (ms_chain * 1e-3), "slots_per_s": S / (ms_chain * 1e-3), "ofdm_mod_slots_per_s": S / (kms["ofdm_mod"] * 1e-3),
           "ldpc_encode_info_bits_per_s": S * w["tbs"] / (kms["pdsch_encode_in_plan"] * 1e-3),
           "ldpc_encode_info_bits_per_s_per_call_entry_point": S * w["tbs"] / (kms["pdsch_encode"] * 1e-3),
           "verification": {"plan_grid_equals_per_call_grid": plan_equals_per_call, "composed_grid_vs_separate_entry_points_max_abs_diff": grid_diff, "codeword_equals_oracle": cw_oracle,
                            "ofdm_demod_of_the_samples_vs_grid_rel_err": ofdm_err, "hard_decisions_of_received_llrs_vs_codeword_bit_errors": bit_errors,
                            "slots_received_again": nchk, "transport_blocks_recovered_by_the_receive_chain": "%d/%d" % (tb_back, nchk)}}
    if with_cpu and O.ref_available():
        cpus, t_all = host_threads(O)
        for key, T in (("cpu_reference_all_cores", t_all), ("cpu_reference_t1", 1)):
            dtc, done = O.r_pdsch_chain_bench(T, cpus, cpu_seconds, 1, tb_u, nprb, mod, w["tbs"], RNTI0, N_ID0, DMRS_SCR_ID, 4096, 1.0 / 64, 3.5e9)
            out[key] = {"value": done * w["tbs"] / dtc, "unit": "info_bits/s", "cores": T, "kind": "reference", "slots_per_s": done / dtc,
                        "sample": "%d slots in %.1f s: srsRAN pdsch_processor (AVX2 LDPC encoder) + ofdm_slot_modulator (generic DFT), one instance per pinned thread" % (done, dtc)}
    return out, verified


# ------------------------------------------------------------------------------------------------ polar CPU reference
def harq_retx_leg(ctx, miphy, torch, dev, n_tb, max_iter):
    """HARQ with soft combining through the transport-block level path: 273 PRB 16QAM R=658/1024 (13 codeblocks BG1 Z=384) sent at a noise
    level where no transport block survives the first transmission (rv 0), then retransmitted at rv 2, 3, 1 (the reference's order) into
    the same soft buffers. Retransmissions are not dematched by the decoder: the rate dematcher combines into the full-length soft buffer
    (rv 2 and 3 wrap around the circular buffer), and the decoder runs over all 46 layers. A retransmission changes the buffers, so every
    timed run starts from a saved copy of the state the transmission before it left (4 runs each, HIP events around plan.run)."""
    bg, mod, nsym, tb_bytes, sigma = 1, 4, 273 * 156, 108552 // 8, 0.75
    sg = miphy.sch_segmentation(tb_bytes, bg)
    C, G = sg.nof_cbs, nsym * mod
    rng = np.random.default_rng(5)
    n_u = min(n_tb, 8)
    tb_u = rng.integers(0, 256, (n_u, tb_bytes), dtype=np.uint8)
    soft = torch.zeros(n_tb * C * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev)
    msgs = torch.zeros(n_tb * C * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)
    crc = torch.zeros(n_tb * C, dtype=torch.uint8, device=dev)
    out = torch.zeros(n_tb * tb_bytes, dtype=torch.uint8, device=dev)
    res = torch.zeros(n_tb * miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)
    idx = torch.arange(n_tb, device=dev) % n_u
    rows = []
    for k, rv in enumerate((0, 2, 3, 1)):
        td = np.zeros(n_u, dtype=miphy.PdschTbDesc)
        for u in range(n_u):
            td[u] = (bg, rv, mod, 1, 0, nsym, tb_bytes, u * tb_bytes, u * G)
        cw = torch.zeros(n_u * G, dtype=torch.uint8, device=dev)
        ctx.pdsch_encode_batch(td, torch.from_numpy(tb_u.reshape(-1)).to(dev), cw)
        g = torch.Generator(device=dev)
        g.manual_seed(100 + k)
        y = (1.0 - 2.0 * cw.reshape(n_u, G)[idx].to(torch.float32)) + sigma * torch.randn(n_tb, G, device=dev, generator=g)
        llr = torch.clamp(torch.round(torch.clamp(4.0 * y, -20, 20) * 6.0), -120, 120).to(torch.int8).reshape(-1)
        del y
        tbd = np.zeros(n_tb, dtype=miphy.PuschTbDesc)
        for t in range(n_tb):
            tbd[t] = (bg, rv, mod, 1, 1 if k == 0 else 0, 0, max_iter, 0, nsym, tb_bytes, t * C, t * G, t * tb_bytes)
        plan = ctx.pusch_decode_plan(tbd)
        plan.enable_timing(16)
        s0, c0, m0 = soft.clone(), crc.clone(), msgs.clone()
        tms = []
        for rep in range(5):
            soft.copy_(s0), crc.copy_(c0), msgs.copy_(m0)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            plan.run(llr, soft, msgs, crc, out, res)
            b.record()
            torch.cuda.synchronize()
            if rep:
                tms.append(a.elapsed_time(b))
            else:
                plan.read_timing()  # (the first run is the warm-up: its stage times are dropped)
        tm = plan.read_timing()
        r = res.cpu().numpy().view(miphy.PuschResult)
        ok = int((r["tb_crc_ok"] != 0).sum())
        same = bool(torch.equal(out.reshape(n_tb, tb_bytes), torch.from_numpy(tb_u).to(dev)[idx])) if ok == n_tb else False
        ms = float(np.mean(tms))
        newly = ok - (rows[-1]["tb_crc_ok"] if rows else 0)  # transport blocks this transmission recovered (the later ones only combine and skip)
        rows.append({"transmission": k, "rv": rv, "tb_crc_ok": ok, "tb_newly_recovered": newly, "transport_blocks_recovered": same, "ms_per_step": ms,
                     "kernel_ms": tm, "info_bits_per_s": newly * tb_bytes * 8 / (ms * 1e-3), "dematch_in_decoder": bool(plan.info()[1])})
        plan.close()
        del s0, c0, m0
    ok_leg = rows[0]["tb_crc_ok"] == 0 and all(r["tb_crc_ok"] == n_tb and r["transport_blocks_recovered"] for r in rows[1:])
    return {"config": "273 PRB 16QAM R=658/1024 (TBS 108552, 13 codeblocks BG1 Z=384), BPSK-AWGN sigma 0.75: rv 0 (nothing decodes), then rv 2, 3, 1 combined "
                      "into the same soft buffers; codeblocks whose CRC is already good are dematched and skipped by the decoder as in pusch_decoder_impl.cpp:183-199",
            "transport_blocks": n_tb, "codeblocks": n_tb * C, "ldpc_iterations": max_iter, "transmissions": rows}, ok_leg


def polar_cpu_leg(ctx, miphy, torch, dev, seconds):
    """The reference's PDCCH polar chains on the host cores beside the GPU's polar figures (polar_chain_test.cpp:156-210 flow): pdcch_encoder::encode
    and rate dematcher + SSC decoder + deallocator, per aggregation level, one thread and all threads."""
    import oracle_lib as O
    if not O.ref_available():
        return None
    cpus, t_all = host_threads(O)
    rng = np.random.default_rng(0)
    A, ncw = 40, 256
    rows = []
    for AL in (1, 2, 4, 8, 16):
        E = 108 * AL
        pay = rng.integers(0, 2, (ncw, A), dtype=np.uint8)
        cwd = np.stack([O.r_pdcch_encode(pay[c], 0x1234 + c, E) for c in range(ncw)])
        sigma = {1: 0.75, 2: 1.0, 4: 1.4, 8: 2.0, 16: 2.8}[AL]
        y = (1.0 - 2.0 * cwd) + sigma * rng.standard_normal(cwd.shape)
        llr = np.clip(np.round(y * (2.0 / sigma ** 2) * 4), -120, 120).astype(np.int8)
        row = {"aggregation_level": AL, "E": E}
        for stage, name in ((0, "encode"), (1, "ssc_decode")):
            for T, tag in ((1, "t1"), (t_all, "all_cores")):
                dt, done = O.r_polar_chain_bench(T, cpus, seconds, stage, A, E, pay, llr)
                row["%s_Mcw_per_s_%s" % (name, tag)] = done / dt / 1e6
        rows.append(row)
    return {"kind": "reference", "cores_all": t_all, "rows": rows,
            "sample": "%.1f s per point; srsRAN pdcch_encoder / polar rate dematcher + SSC decoder + deallocator, objects created once per pinned thread" % seconds}


# ------------------------------------------------------------------------------------------------ compressed-IQ ingest
def ofh_ingest_leg(ctx, miphy, torch, dev, step_from_grid, grid_d, S, nprb, tbs_bits, tb_d, tb_bytes, exp_tb):
    """the fronthaul hands over BFP-compressed frequency-domain IQ, not time samples. The received grids of the headline step are
    compressed to 9-bit BFP U-plane payloads (the library's own compressor, outside the timed region) and kept in pinned host memory; timed:
    H2D of the payloads -> miphy_ofh_iq_decompress_batch into the resource grid -> estimator -> demodulator -> decode plan -> transport blocks
    back to pinned host memory, with the upload of step i + 1 under the compute of step i (two payload buffers, a copy stream)."""
    w_bits, nsc = 9, nprb * 12
    st = torch.cuda.current_stream()
    rec = nprb * (1 + 3 * w_bits)  # bytes of one section: [udCompParam][24 * 9 bits] per PRB
    n = S * 14
    jobs = np.zeros(n, dtype=miphy.OfhIqJob)
    jobs["payload_offset"] = np.arange(n, dtype=np.uint64) * np.uint64(rec)
    jobs["grid_offset"] = np.arange(n, dtype=np.uint64) * np.uint64(nsc)
    jobs["nof_prb"], jobs["data_width"], jobs["compression"] = nprb, w_bits, miphy.OFH_COMPRESSION_BFP
    jobs_d = torch.from_numpy(jobs.view(np.uint8)).to(dev)
    pay = [torch.zeros(n * rec, dtype=torch.uint8, device=dev) for _ in range(2)]
    scale = 0.2  # the grid holds unit-power symbols and 3 dB boosted DM-RS: inside (-1, 1) after this factor (the estimator absorbs it)
    ctx.ofh_iq_compress_batch(jobs_d, grid_d, pay[0], scale, st)
    torch.cuda.synchronize()
    h_in = torch.empty(n * rec, dtype=torch.uint8).pin_memory()
    h_in.copy_(pay[0])
    h_out = torch.empty(S * tb_bytes, dtype=torch.uint8).pin_memory()
    copy_s = torch.cuda.Stream()
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    done = [torch.cuda.Event(), torch.cuda.Event()]
    for e_ in done:
        e_.record(st)

    def pipelined(k_steps):
        for i in range(k_steps):
            b_ = i % 2
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(done[b_])
                pay[b_].copy_(h_in, non_blocking=True)
                ready[b_].record(copy_s)
            st.wait_event(ready[b_])
            ctx.ofh_iq_decompress_batch(jobs_d, pay[b_], grid_d, True, st)
            step_from_grid()
            h_out.copy_(tb_d, non_blocking=True)
            done[b_].record(st)

    tb_d.zero_()
    pipelined(2)
    torch.cuda.synchronize()
    kp = 8
    t0 = time.perf_counter()
    pipelined(kp)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / kp * 1e3
    ok = bool(torch.equal(tb_d.reshape(S, tb_bytes), exp_tb))
    ms_h2d = ev_ms(torch, lambda: pay[0].copy_(h_in, non_blocking=True), 3)
    ms_dec = ev_ms(torch, lambda: ctx.ofh_iq_decompress_batch(jobs_d, pay[0], grid_d, True, st), 5)
    ms_dev = ev_ms(torch, lambda: (ctx.ofh_iq_decompress_batch(jobs_d, pay[0], grid_d, True, st), step_from_grid()), 3)
    return {"config": "BFP-9 U-plane payloads (%d B per slot-port instead of %d B of fp32 time samples) in pinned host memory -> H2D -> miphy_ofh_iq_decompress_batch -> "
                      "estimator -> demodulator -> decode plan -> TB D2H, upload of step i+1 under the compute of step i" % (14 * rec, 61440 * 8),
            "bits_per_s": S * tbs_bits / (ms * 1e-3), "ms_per_step": ms, "h2d_ms": ms_h2d, "h2d_GBps": n * rec / ms_h2d / 1e6, "decompress_ms": ms_dec,
            "decompress_algorithmic_GBps": (n * rec + n * nsc * 8) / ms_dec / 1e6, "device_only_ms": ms_dev, "device_only_bits_per_s": S * tbs_bits / (ms_dev * 1e-3),
            "payload_bytes_per_step": n * rec, "transport_blocks_ok": ok}, ok
