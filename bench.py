#!/usr/bin/env python3
"""Headline benchmark: PUSCH receive hot path on synthetic 100 MHz n78 slots (273 PRB, 30 kHz SCS).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this script starts `python -m torch.distributed.run --nproc-per-node N` on itself as a CHILD process
before anything touches the GPU, one rank per GPU over RCCL (when the machine has fewer than N GPUs every rank shares cuda:0 over
gloo and the line says `"rehearsal_shared_gpu": true`). Under a launcher (WORLD_SIZE set) it is one rank of that job.

One "step" = one pass of the hot path over a batch of S synthetic uplink slots per GPU: time-domain samples in, transport blocks
out (OFDM demodulation -> DM-RS channel estimation -> equalise / soft-demap / descramble -> rate dematch -> LDPC decode ->
transport-block assembly + TB CRC). The workload is BASELINE.json configs[2]: 273-PRB PUSCH, 256QAM R=948/1024, 1 layer, 38
codeblocks (BG1, Z=384) per slot, TBS = 319 784 information bits per slot. The slots are synthesised once, outside the timed
region, by the transmit side of the same library on the device (SCH encoder, modulator, DM-RS mapper, OFDM modulator) plus AWGN,
and are resident in HBM before timing starts.

`value` = LDPC information bits / s over the whole step (all ranks). Extra keys: per-kernel HIP-event times and
algorithmic GB/s, `roofline` of the dominant kernel, `roofline_valu` (the decoder is a VALU kernel), `cpu_baseline` (the
reference's own receive chain on all host cores; `cpu_baseline_t1`, `cpu_baseline_decoder_only`), `pcie_inclusive`, the other
codeblock configurations (`legs`: 16QAM R=658, BG1 Z=384 R=1/3, polar AL 1-16) and, with --ingest scatter, `ingest_scatter` (one ingest
GPU scatters LLR slabs over RCCL, every rank decodes, results gathered).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

import bench_legs as BL

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "srsran_project_23.5_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NOF_SIMDS, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md constants table)
PROFILE_ROUND = "r03"

RNTI, N_ID, DMRS_SCR_ID = 0x4601, 935, 1
DMRS_SCALING = 1.4125375  # DM-RS boosted by 3 dB with two CDM groups without data (sch_dmrs_power.h)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--slots", type=int, default=1024, help="slots per GPU per step (1024 slots = 38 912 codeblocks)")
    ap.add_argument("--max-iter", type=int, default=6)
    ap.add_argument("--early-stop", type=int, default=0)
    ap.add_argument("--snr-db", type=float, default=33.0, help="per-RE SNR of the synthesised slots")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="duration of each CPU baseline leg")
    ap.add_argument("--chunks", type=int, default=1, help="slot groups per step, alternated over two HIP streams (1 = one stream)")
    ap.add_argument("--chunk-streams", type=int, default=2, help="1: the slot groups run back to back on one stream")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-slot latency leg (keeps profiles to the timed step only)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs (other MCS / rates, polar, PCIe-inclusive)")
    ap.add_argument("--ingest", choices=["local", "scatter"], default="local",
                    help="scatter: add the separately timed single-ingest leg (rank 0 scatters the LLR slabs over RCCL, results all-gathered); "
                         "local (default): every rank owns its slots, no collective in the data path")
    ap.add_argument("--ingest-slots", type=int, default=128, help="slots per rank in the scatter leg")
    ap.add_argument("--ldpc-force", type=int, default=0, help="A-B knob of tools/ab_bench.py: miphy_debug_force_ldpc_kernel mode (0 = automatic)")
    ap.add_argument("--extra-multi", action="store_true", help="run the extra GPU legs on rank 0 of a multi-GPU job as well (default: single GPU only)")
    return ap.parse_args()


def pusch_workload():
    """273 PRB x 156 data RE (14 symbols, 1 DM-RS symbol, 2 CDM groups w/o data), 256QAM R=948/1024, 1 layer."""
    return dict(nprb=273, mod=8, nof_layers=1, nsym=273 * 156, tbs=319784, bg=1, rv=0, Nref=0)


def kernel_source_sha():
    """Hash of the kernel sources and their build recipe: profiles collected on another build must not be quoted."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "srsran_project_23.5_amd", "csrc")
    h.update(open(os.path.join(ROOT, "srsran_project_23.5_amd", "Makefile"), "rb").read())
    for dirpath, _, files in sorted(os.walk(d)):
        for f in sorted(files):
            if f.endswith((".hip", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(dirpath, f), "rb").read())
    return h.hexdigest()[:16]


def launch_children(args):
    """--gpus N > 1 and no launcher: start the N ranks as a child job (never an exec of this process; nothing here has
    touched the GPU: torch.cuda.device_count() does not initialise it on this image)."""
    import torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus:
        if args.gpus > 4:
            print("bench.py: --gpus %d needs %d GPUs (found %d); the shared-GPU rehearsal is limited to 4 ranks" % (args.gpus, args.gpus, ndev),
                  file=sys.stderr)
            return 2
        env["MIPHY_BENCH_REHEARSAL"] = "1"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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


def ev_ms(torch, f, reps, stream=None):
    """Mean HIP-event time of f() over `reps` back-to-back calls (two untimed calls first)."""
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


def ports_leg(ctx, miphy, torch, dev, w, grids_tx, tbs_u, nports, S, max_iter, snr_db, seed):
    """The headline slot received on `nports` antenna ports (maximum-ratio combining in the demodulator): every port sees the transmitted slot
    through its own flat complex gain plus independent AWGN. OFDM demodulation of S x nports slot-ports, one channel estimate row per
    port, nports-port equalise / demap / descramble, then the same decode plan. Per-stage HIP-event times; every transport block must
    come back CRC-ok and equal to the transmitted one."""
    sg = miphy.sch_segmentation(w["tbs"] // 8, w["bg"])
    C, G, nsc, tb_bytes, n_unique = sg.nof_cbs, w["nsym"] * w["mod"], w["nprb"] * 12, w["tbs"] // 8, len(tbs_u)
    mcfg = miphy.OfdmConfig(1, w["nprb"], 4096, 0, 1.0 / 64, 0.0, 3.5e9)
    ocfg = miphy.OfdmConfig(1, w["nprb"], 4096, 144, 1.0 / 64, 0.0, 3.5e9)
    ss = ocfg.slot_size(0)
    st = torch.cuda.current_stream()
    # transmit once per slot, then one copy per port with its own gain and noise
    tj = np.zeros(S, dtype=miphy.OfdmJob)
    for s_ in range(S):
        tj[s_] = (s_ * ss, s_ * 14 * nsc, s_ % 2, 0)
    x = torch.zeros(S * ss, dtype=torch.complex64, device=dev)
    ctx.ofdm_modulate_slots(mcfg, tj, grids_tx.reshape(20, -1)[torch.arange(S, device=dev) % 20].reshape(-1).contiguous(), x, st)
    gains = torch.tensor([0.9 * np.exp(1j * (0.7 * p_ + 0.3)) for p_ in range(nports)], dtype=torch.complex64, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    sigma = float(10.0 ** (-snr_db / 20.0)) * 0.70710678
    samples = (x.reshape(S, 1, ss) * gains.reshape(1, nports, 1)).contiguous()
    samples += torch.view_as_complex(torch.randn(S, nports, ss, 2, device=dev, generator=gen) * sigma)
    samples = samples.reshape(-1)
    del x
    oj = np.zeros(S * nports, dtype=miphy.OfdmJob)
    cj = np.zeros(S, dtype=miphy.PuschChestJob)
    dj = np.zeros(S, dtype=miphy.PuschDemodJob)
    rb_words = [0xFFFFFFFFFFFFFFFF] * 4 + [(1 << (w["nprb"] - 256)) - 1]
    for s_ in range(S):
        for p_ in range(nports):
            oj[s_ * nports + p_] = ((s_ * nports + p_) * ss, (s_ * nports + p_) * 14 * nsc, s_ % 2, 0)
        j = cj[s_]
        j["numerology"], j["slot_in_frame"], j["scrambling_id"], j["scaling"] = 1, s_ % 20, DMRS_SCR_ID, DMRS_SCALING
        j["nof_tx_layers"], j["nof_rx_ports"], j["first_symbol"], j["nof_symbols"], j["rx_ports"] = 1, nports, 0, 14, [0, 1, 2, 3]
        j["symbols_mask"], j["grid_nof_prb"], j["ce_compact"], j["rb_mask"] = 1 << 2, w["nprb"], 1, rb_words
        j["grid_offset"], j["ce_offset"], j["scalars_offset"] = s_ * nports * 14 * nsc, s_ * nports * nsc, s_ * nports * 5
        q = dj[s_]
        q["rnti"], q["n_id"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = RNTI, N_ID, w["mod"], nports, 0, 14
        q["dmrs_type"], q["nof_cdm_groups_without_data"], q["ce_nof_symbols"], q["ce_compact"], q["rx_ports"] = 1, 2, 14, 1, [0, 1, 2, 3]
        q["dmrs_symbols_mask"], q["grid_nof_prb"], q["nof_llr"], q["rb_mask"] = 1 << 2, w["nprb"], G, rb_words
        q["grid_offset"], q["ce_offset"], q["scalars_offset"], q["llr_offset"] = s_ * nports * 14 * nsc, s_ * nports * nsc, s_ * nports * 5, s_ * G
    oj_d, cj_d, dj_d = (torch.from_numpy(a.view(np.uint8)).to(dev) for a in (oj, cj, dj))
    grid = torch.zeros(S * nports * 14 * nsc, dtype=torch.complex64, device=dev)
    ce = torch.zeros(S * nports * nsc, dtype=torch.complex64, device=dev)
    sc = torch.zeros(S * nports * 5, dtype=torch.float32, device=dev)
    llr = torch.zeros(S * G, dtype=torch.int8, device=dev)
    soft = torch.zeros(S * C * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev)
    msgs = torch.zeros(S * C * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)
    crc = torch.zeros(S * C, dtype=torch.uint8, device=dev)
    tb = torch.zeros(S * tb_bytes, dtype=torch.uint8, device=dev)
    res = torch.zeros(S * miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)
    td = np.zeros(S, dtype=miphy.PuschTbDesc)
    for s_ in range(S):
        td[s_] = (w["bg"], w["rv"], w["mod"], w["nof_layers"], 1, 0, max_iter, w["Nref"], w["nsym"], tb_bytes, s_ * C, s_ * G, s_ * tb_bytes)
    plan = ctx.pusch_decode_plan(td)
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
        ctx.dmrs_pusch_estimate_batch(cj_d, grid, ce, sc, st, max_ports=nports, max_layers=1)
        if timed:
            e[2].record(st)
        ctx.pusch_demodulate_batch(dj_d, grid, ce, sc, llr, st)
        if timed:
            e[3].record(st)
            for i, k in enumerate(names):
                acc[k].append((e[i], e[i + 1]))
        plan.run(llr, soft, msgs, crc, tb, res, st)

    for _ in range(2):
        step(False)
    torch.cuda.synchronize()
    plan.read_timing()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        step(True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms = {k: float(np.mean([a.elapsed_time(b) for a, b in acc[k]])) for k in names}
    kms.update(plan.read_timing())
    r = res.cpu().numpy().view(miphy.PuschResult)
    ok = int((r["tb_crc_ok"] != 0).sum())
    idx = torch.from_numpy((np.arange(S) % 20) % n_unique).to(dev)
    same = bool(torch.equal(tb.reshape(S, tb_bytes), torch.from_numpy(np.stack(tbs_u)).to(dev)[idx])) if ok == S else False
    plan.close()
    return {"config": "headline slot on %d rx ports (flat gain per port, %g dB per port), MRC in the demodulator" % (nports, snr_db), "slots": S, "rx_ports": nports,
            "ms_per_step": dt * 1e3, "kernel_ms": kms, "slots_per_s": S / dt, "info_bits_per_s": S * w["tbs"] / dt,
            "ofdm_slot_ports_per_s": S * nports / (kms["ofdm_demod"] * 1e-3), "tb_crc_ok": ok, "transport_blocks_recovered": same}


# ------------------------------------------------------------------------------------------------ CPU baseline legs
def cpu_legs(w, samples4, llr4, ocfg_args, max_iter, early_stop, seconds):
    """The reference's receive chain on the host cores (oracle/_ref, the reference compiled in place), driven like its
    pusch_processor_benchmark (pinned threads, one processor instance per thread): full chain on all cores = `cpu_baseline`, on one
    core = `cpu_baseline_t1`; pusch_decoder alone (rate dematcher + LDPC decoder, what round 1 reported) on 1 / all cores.
    Falls back to the scalar oracle port (one thread, decoder only) when oracle/_ref is absent."""
    import oracle_lib as O
    cpus, quota = O.host_cpus()
    t_all = len(cpus) if quota is None else max(1, min(len(cpus), int(round(quota))))
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {}
    host = dict(cpu_model=model, cpus_in_affinity_mask=len(cpus), cgroup_cpu_quota=quota)
    kind = "reference"
    try:
        if not O.ref_available():
            raise OSError("no oracle/_ref")
        O.ref()
    except OSError:
        kind = "port"
    if kind == "port":
        od = O.OraclePuschDecoder(w["bg"], w["mod"], w["Nref"], w["nof_layers"], w["nsym"], w["tbs"] // 8)
        t0, k = time.time(), 0
        while time.time() - t0 < seconds:
            od.decode(llr4[k % llr4.shape[0]], 0, True, max_iter, early_stop)
            k += 1
        dt = time.time() - t0
        out["cpu_baseline"] = dict(value=k * w["tbs"] / dt, unit="info_bits/s", cores=1, kind="port",
                                   sample="%d slots in %.1f s, scalar C oracle, rate dematch + LDPC decode only" % (k, dt), **host)
        return out

    def chain(T, stage):
        dt, done, ok = O.r_pusch_chain_bench(T, cpus, seconds, stage, samples4, w["nprb"], w["mod"], w["tbs"], RNTI, N_ID, DMRS_SCR_ID, *ocfg_args,
                                             max_iter, early_stop)
        return dict(value=done * w["tbs"] / dt, unit="info_bits/s", cores=T, kind="reference",
                    sample="%d slots (%d TB CRC ok) in %.1f s; srsRAN ofdm_slot_demodulator (generic DFT) + pusch_processor (AVX2 LDPC), one instance per "
                           "pinned thread, same 4 slots of time-domain samples as the GPU run" % (done, ok, dt), slots_per_s=done / dt, **host)

    def dec(T):
        dt, done, ok = O.r_pusch_decoder_bench(T, cpus, seconds, llr4, w["mod"], w["nsym"], w["tbs"], max_iter, early_stop)
        return dict(value=done * w["tbs"] / dt, unit="info_bits/s", cores=T, kind="reference",
                    sample="%d slots (%d TB CRC ok) in %.1f s; srsRAN pusch_decoder avx2 only (rate dematcher + LDPC decoder) on the GPU's LLRs" % (done, ok, dt))

    out["cpu_baseline"] = chain(t_all, 1)
    out["cpu_baseline_t1"] = chain(1, 1)
    out["cpu_baseline_decoder_only"] = {"t1": dec(1), "all_cores": dec(t_all)}
    # Courtesy figure: the reference also ships AVX-512 decoder / dematcher classes (ldpc_decoder_avx512.cpp), picked by its
    # "auto" factories on hosts that have the instruction set. Not the baseline (north_star names the AVX2 path).
    for T, tag in ((1, "avx512_t1"), (t_all, "avx512_all_cores")):
        r512 = O.r_pusch_decoder_bench_isa(T, cpus, seconds, llr4, w["mod"], w["nsym"], w["tbs"], max_iter, early_stop, 2)
        if r512 is None:
            out["cpu_baseline_decoder_only"][tag] = "the host (or the build of oracle/_ref) has no AVX-512 decoder"
            break
        dt5, done5, ok5 = r512
        out["cpu_baseline_decoder_only"][tag] = dict(value=done5 * w["tbs"] / dt5, unit="info_bits/s", cores=T, kind="reference",
                                                     sample="%d slots (%d TB CRC ok) in %.1f s; srsRAN pusch_decoder with the avx512 decoder / dematcher classes" % (done5, ok5, dt5))
    return out


# ------------------------------------------------------------------------------------------------ extra GPU legs
def sch_leg(ctx, miphy, torch, dev, name, bg, mod, nsym, tb_bytes, n_tb, max_iter, sigma, seed):
    """One SCH configuration through the transport-block level path (prepared plan: rate dematch + LDPC decode + TB assembly): n_tb
    transport blocks encoded on the device, BPSK-AWGN LLRs on the reference's int8 scale, per-kernel HIP-event times."""
    sg = miphy.sch_segmentation(tb_bytes, bg)
    C, G = sg.nof_cbs, nsym * mod
    rng = np.random.default_rng(seed)
    n_u = min(n_tb, 8)
    tb_u = rng.integers(0, 256, (n_u, tb_bytes), dtype=np.uint8)
    td = np.zeros(n_u, dtype=miphy.PdschTbDesc)
    for u in range(n_u):
        td[u] = (bg, 0, mod, 1, 0, nsym, tb_bytes, u * tb_bytes, u * G)
    cw = torch.zeros(n_u * G, dtype=torch.uint8, device=dev)
    ctx.pdsch_encode_batch(td, torch.from_numpy(tb_u.reshape(-1)).to(dev), cw)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    idx = torch.arange(n_tb, device=dev) % n_u
    y = (1.0 - 2.0 * cw.reshape(n_u, G)[idx].to(torch.float32)) + sigma * torch.randn(n_tb, G, device=dev, generator=g)
    llr = torch.clamp(torch.round(torch.clamp(4.0 * y, -20, 20) * 6.0), -120, 120).to(torch.int8).reshape(-1)
    del y
    tbd = np.zeros(n_tb, dtype=miphy.PuschTbDesc)
    for t in range(n_tb):
        tbd[t] = (bg, 0, mod, 1, 1, 0, max_iter, 0, nsym, tb_bytes, t * C, t * G, t * tb_bytes)
    soft = torch.zeros(n_tb * C * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev)
    msgs = torch.zeros(n_tb * C * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)
    crc = torch.zeros(n_tb * C, dtype=torch.uint8, device=dev)
    out = torch.zeros(n_tb * tb_bytes, dtype=torch.uint8, device=dev)
    res = torch.zeros(n_tb * miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)
    plan = ctx.pusch_decode_plan(tbd)
    plan.enable_timing(16)
    ms = ev_ms(torch, lambda: plan.run(llr, soft, msgs, crc, out, res), 5)
    tm = plan.read_timing()
    r = res.cpu().numpy().view(miphy.PuschResult)
    ok = int((r["tb_crc_ok"] != 0).sum())
    same = bool(torch.equal(out.reshape(n_tb, tb_bytes), torch.from_numpy(tb_u).to(dev)[idx])) if ok == n_tb else False
    fused = plan.info()[1]
    plan.close()
    in_len = min(sg.N, max((22 if bg == 1 else 10) * sg.Z + 2 * sg.Z, -(-(G // C + sg.nof_filler_bits) // sg.Z) * sg.Z))
    # dematch inside the decoder: rate-matched LLRs in + soft-buffer image (N per codeblock) out, no intermediate re-read
    alg = n_tb * (G + C * (sg.N + sg.K // 8 + 4)) if fused else n_tb * C * (in_len + sg.K // 8 + 4)
    return {"config": name, "transport_blocks": n_tb, "codeblocks": n_tb * C, "Z": sg.Z, "decoder_in_len": in_len, "ldpc_iterations": max_iter,
            "ms_per_launch": ms, "kernel_ms": tm, "us_per_codeblock": ms * 1e3 / (n_tb * C), "info_bits_per_s": n_tb * tb_bytes * 8 / (ms * 1e-3),
            "ldpc_decode_algorithmic_GBps": alg / (tm["ldpc_decode"] * 1e-3) / 1e9, "ldpc_decode_hbm_frac": alg / (tm["ldpc_decode"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "tb_crc_ok": ok, "transport_blocks_recovered": same}


def polar_leg(ctx, miphy, torch, dev, n=16384, A=40):
    """BASELINE configs[3]: PDCCH polar encode (CRC24C + interleaver + polar + rate matching), reference-style SSC decode and
    CRC-aided SCL-8, aggregation levels 1-16; codewords/s and algorithmic GB/s (E in + K/8 out per codeword)."""
    rng = np.random.default_rng(0)
    pay = torch.from_numpy(rng.integers(0, 2, (n, A), dtype=np.uint8)).to(dev)
    rnti_h = rng.integers(1, 65536, n).astype(np.uint16)
    rnti = torch.from_numpy(rnti_h.view(np.int16)).to(dev)
    rows = []
    for AL in (1, 2, 4, 8, 16):
        E, K = 108 * AL, A + 24
        out = torch.zeros(n * E, dtype=torch.uint8, device=dev)
        t_enc = ev_ms(torch, lambda: ctx.pdcch_encode_batch(A, E, n, pay, rnti, out), 5)
        sigma = {1: 0.75, 2: 1.0, 4: 1.4, 8: 2.0, 16: 2.8}[AL]
        g = torch.Generator(device=dev)
        g.manual_seed(AL)
        y = (1.0 - 2.0 * out.to(torch.float32)) + sigma * torch.randn(n * E, device=dev, generator=g)
        llr = torch.clamp(torch.round(y * (2.0 / sigma ** 2) * 4), -120, 120).to(torch.int8)
        code = miphy.PolarCode(K, E, 9, 0)
        msg = torch.zeros(n * K, dtype=torch.uint8, device=dev)
        t_ssc = ev_ms(torch, lambda: ctx.polar_decode_batch(code, n, llr, msg), 5)
        ok = torch.zeros(n, dtype=torch.uint8, device=dev)
        t_scl = ev_ms(torch, lambda: ctx.polar_decode_list_batch(code, 8, 1, n, llr, rnti, msg, ok), 3)
        got = msg.reshape(n, K)[:, :A]
        good = (ok != 0) & (got == pay).all(dim=1)
        byt = n * (E + (K + 7) // 8)
        rows.append({"aggregation_level": AL, "K": K, "E": E, "encode_Mcw_per_s": n / t_enc / 1e3, "ssc_decode_Mcw_per_s": n / t_ssc / 1e3,
                     "scl8_decode_Mcw_per_s": n / t_scl / 1e3, "encode_GBps": byt / t_enc / 1e6, "ssc_GBps": byt / t_ssc / 1e6, "scl8_GBps": byt / t_scl / 1e6,
                     "scl8_bler": float(1.0 - good.float().mean().item()), "sigma": sigma})
    return {"codewords_per_call": n, "payload_bits": A, "rows": rows}


def load_stamped(name):
    """profiles/<round>_<name>.json, only if it was collected on this build of the kernels."""
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (PROFILE_ROUND, name))
    try:
        j = json.load(open(path))
    except (OSError, ValueError):
        return None, {"file": os.path.relpath(path, ROOT), "state": "absent"}
    stamp = {"file": os.path.relpath(path, ROOT), "collected_on_commit": j.get("git_commit"), "kernel_source_sha": j.get("kernel_source_sha")}
    if j.get("kernel_source_sha") != kernel_source_sha():
        stamp["state"] = "stale (kernel sources changed since the counters were collected): not quoted"
        return None, stamp
    stamp["state"] = "current"
    return j, stamp


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(args))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):  # under a launcher the launcher's world size is the truth
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running %d ranks" % (args.gpus, world, world), file=sys.stderr)
    # Rehearsal on a machine with fewer GPUs than ranks: every rank on cuda:0, gloo for the barrier, the MAX reduction and the
    # scatter / gather (RCCL refuses two ranks on one device). Measured multi-GPU runs never set it; the line carries the flag.
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
    from miphy import shard
    ctx = miphy.Context(local_rank)
    if args.ldpc_force:
        miphy.lib().miphy_debug_force_ldpc_kernel(args.ldpc_force)
    w = pusch_workload()
    S = args.slots
    # Segmentation of the transport block through the library's own host logic (ldpc.h:128-207 restated in csrc/sch.hip).
    sg = miphy.sch_segmentation(w["tbs"] // 8, w["bg"])
    C, Z, N, K, F = sg.nof_cbs, sg.Z, sg.N, sg.K, sg.nof_filler_bits
    G = w["nsym"] * w["mod"]
    tb_bytes = w["tbs"] // 8
    unit = w["nof_layers"] * w["mod"]
    n_short = C - (G // unit) % C
    seg_E = [unit * ((G // unit) // C) if c < n_short else unit * (-(-(G // unit) // C)) for c in range(C)]
    assert sum(seg_E) == G
    dec_in_len = [min(N, max((22 + 2) * Z, -(-(seg_E[c] + F) // Z) * Z)) for c in range(C)]
    n_unique = 4
    nsc = w["nprb"] * 12
    grids_tx, tbs_u = build_tx_grids(ctx, miphy, torch, dev, w, n_unique, seed=1234 + rank)
    slot_src = np.arange(S) % n_unique  # transport block carried by slot s (slot s is slot-in-frame s % 20)

    # ---- transmit side + channel, once, outside the timed region: OFDM modulation of the S grids on the device and AWGN
    stream = torch.cuda.current_stream()
    ofdm_args = (4096, 144, 1.0 / 64, 3.5e9)  # dft size, window offset, scale, centre frequency of the receiver
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
    llr_d = torch.zeros(S * G, dtype=torch.int8, device=dev)                                  # codeword LLRs produced by the demodulator
    soft_d = torch.zeros(S * C * miphy.HARQ_CB_STRIDE, dtype=torch.int8, device=dev)          # HARQ soft buffers
    msgs_d = torch.zeros(S * C * miphy.HARQ_MSG_STRIDE, dtype=torch.uint8, device=dev)        # decoded codeblock messages
    crc_d = torch.zeros(S * C, dtype=torch.uint8, device=dev)                                 # codeblock CRC flags
    tb_d = torch.zeros(S * tb_bytes, dtype=torch.uint8, device=dev)                           # transport blocks (the output)
    res_d = torch.zeros(S * miphy.PuschResult.itemsize, dtype=torch.uint8, device=dev)        # pusch_decoder_result records

    def make_plans(bounds, early_stop=None):
        plans = []
        early_stop = args.early_stop if early_stop is None else early_stop
        for a, b in bounds:
            td = np.zeros(b - a, dtype=miphy.PuschTbDesc)
            for i, s in enumerate(range(a, b)):
                td[i] = (w["bg"], w["rv"], w["mod"], w["nof_layers"], 1, early_stop, args.max_iter, w["Nref"], w["nsym"], tb_bytes, s * C, s * G,
                         s * tb_bytes)
            p = ctx.pusch_decode_plan(td)
            p.enable_timing(max(64, args.steps + 8))
            plans.append(p)
        return plans

    # ---- front end of the slot: 1 rx port, 1 layer, DM-RS type 1 in symbol 2 with two CDM groups without data (like
    # pusch_processor_benchmark.cpp:104-105), all 273 PRB allocated.
    ce_d = torch.zeros(S * nsc, dtype=torch.complex64, device=dev)  # compact estimate: one row per (slot, port), see miphy.h ce_compact
    sc_d = torch.zeros(S * 5, dtype=torch.float32, device=dev)
    cjobs = np.zeros(S, dtype=miphy.PuschChestJob)
    djobs = np.zeros(S, dtype=miphy.PuschDemodJob)
    rb_words = [0xFFFFFFFFFFFFFFFF] * 4 + [(1 << (w["nprb"] - 256)) - 1]
    for s in range(S):
        j = cjobs[s]
        j["numerology"], j["slot_in_frame"], j["scrambling_id"], j["scaling"] = 1, s % 20, DMRS_SCR_ID, DMRS_SCALING
        j["nof_tx_layers"], j["nof_rx_ports"], j["first_symbol"], j["nof_symbols"] = 1, 1, 0, 14
        j["rx_ports"] = [0, 1, 2, 3]
        j["symbols_mask"], j["grid_nof_prb"], j["ce_compact"] = 1 << 2, w["nprb"], 1
        j["rb_mask"] = rb_words
        j["grid_offset"], j["ce_offset"], j["scalars_offset"] = s * 14 * nsc, s * nsc, s * 5
        q = djobs[s]
        q["rnti"], q["n_id"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = RNTI, N_ID, w["mod"], 1, 0, 14
        q["dmrs_type"], q["nof_cdm_groups_without_data"], q["ce_nof_symbols"], q["ce_compact"] = 1, 2, 14, 1
        q["rx_ports"] = [0, 1, 2, 3]
        q["dmrs_symbols_mask"], q["grid_nof_prb"], q["nof_llr"] = 1 << 2, w["nprb"], G
        q["rb_mask"] = rb_words
        q["grid_offset"], q["ce_offset"], q["scalars_offset"], q["llr_offset"] = s * 14 * nsc, s * nsc, s * 5, s * G
    assert miphy.pusch_demod_nof_llr(djobs[0]) == G
    cjobs_d = torch.from_numpy(cjobs.view(np.uint8)).to(dev)
    djobs_d = torch.from_numpy(djobs.view(np.uint8)).to(dev)

    front = ["ofdm_demod", "dmrs_chest", "pusch_demod"]
    back = ["rate_dematch", "ldpc_decode", "tb_assemble"]  # inside miphy_pusch_decode_plan_run, timed by the library's own events
    stages = front + back
    ev = {k: [] for k in front}

    # One step = the whole batch of S slots. With --chunks G > 1 the batch is cut into G groups of slots that alternate between
    # two HIP streams; every step still processes all S slots and the timed region ends with a device-wide sync.
    G_ch = max(1, min(args.chunks, S))
    bounds = [(S * i // G_ch, S * (i + 1) // G_ch) for i in range(G_ch)]
    streams = [stream] if (G_ch == 1 or args.chunk_streams == 1) else [torch.cuda.Stream(), torch.cuda.Stream()]
    plans = make_plans(bounds)

    RS = miphy.PuschResult.itemsize

    def step(timed, src=None):
        src = samples_d if src is None else src
        for ci, (a, b) in enumerate(bounds):
            st = streams[ci % len(streams)]
            e = [torch.cuda.Event(enable_timing=True) for _ in range(len(front) + 1)] if timed else None
            if timed:
                e[0].record(st)
            ctx.ofdm_demodulate_slots(ocfg, ojobs_d[a * 24:b * 24], src, grid_d, st)
            if timed:
                e[1].record(st)
            ctx.dmrs_pusch_estimate_batch(cjobs_d[a * 152:b * 152], grid_d, ce_d, sc_d, st, max_ports=1, max_layers=1)
            if timed:
                e[2].record(st)
            ctx.pusch_demodulate_batch(djobs_d[a * 120:b * 120], grid_d, ce_d, sc_d, llr_d, st)
            if timed:
                e[3].record(st)
                for i, k in enumerate(front):
                    ev[k].append((e[i], e[i + 1]))
            plans[ci].run(llr_d, soft_d, msgs_d, crc_d, tb_d, res_d[a * RS:b * RS], st)  # result records of this chunk's transport blocks

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    for p in plans:
        p.read_timing()  # drop the warm-up samples
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
    dt_rank = dt
    per_rank_ms = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dt = float(t.item())
        per_rank_ms = {"min": float(tmin.item()) / args.steps * 1e3, "max": dt / args.steps * 1e3, "this_rank": dt_rank / args.steps * 1e3}
    back_ms = {k: 0.0 for k in back}
    for p in plans:
        tm = p.read_timing()
        for k in back:
            back_ms[k] += tm[k]

    # ---- correctness of what was just computed (not timed). (0) every slot of the step: TB CRC verdict and transport-block bytes on the
    # device. (1) Demodulator: the oracle, fed with the GPU's own resource grid, channel estimate and noise variance of slot 0, must
    # give the same LLRs bit for bit. (2) Decoder: the oracle decoder on the GPU's LLRs must give the same codeblock messages and
    # transport block.
    exp_tb = torch.from_numpy(np.stack(tbs_u)).to(dev)[torch.from_numpy(slot_src).to(dev)]
    res_h = res_d.cpu().numpy().view(miphy.PuschResult)
    tb_ok_h = res_h["tb_crc_ok"] != 0
    tb_same = (tb_d.reshape(S, tb_bytes) == exp_tb).all(dim=1).cpu().numpy()
    nof_tb_good = int((tb_ok_h & tb_same).sum())
    all_ok = nof_tb_good == S
    checked = min(S, 4 if world > 1 else 8)
    llr_h = llr_d[:checked * G].cpu().numpy().reshape(checked, G)
    msgs_h = msgs_d[:checked * C * miphy.HARQ_MSG_STRIDE].cpu().numpy().reshape(checked, C, miphy.HARQ_MSG_STRIDE)[:, :, :K // 8]
    g0 = grid_d[:14 * nsc].cpu().numpy().reshape(1, 14, nsc)
    h0 = np.ascontiguousarray(np.broadcast_to(ce_d[:nsc].cpu().numpy().reshape(1, 1, nsc), (1, 14, nsc)))  # the reference's layout: a copy per symbol
    dm = np.zeros(14, np.uint8)
    dm[2] = 1
    o_llr, _, _ = O.o_pusch_demodulate(RNTI, N_ID, w["mod"], 0, 14, dm, 0, 2, np.ones(w["nprb"], np.uint8), g0, h0, float(sc_d[2].item()))
    demod_ok = bool(np.array_equal(o_llr, llr_h[0]))
    ok_slots = 0
    for s in range(checked):
        od = O.OraclePuschDecoder(w["bg"], w["mod"], w["Nref"], w["nof_layers"], w["nsym"], tb_bytes)
        ok, tb, _ = od.decode(llr_h[s], 0, True, args.max_iter, bool(args.early_stop))
        same = np.array_equal(od.cb_msgs.reshape(C, -1), msgs_h[s])
        # the GPU's verdict must be the oracle's (a transport block the oracle cannot decode either is parity, not a failure of the path)
        ok_slots += int(same and bool(ok) == bool(tb_ok_h[s]) and (not ok or np.array_equal(tb, tbs_u[slot_src[s]])))
    if not demod_ok:
        ok_slots = -1
    # every rank checks its own slots; the line (and the exit code of every rank) carries the verdict of all of them
    parity_all_ranks = ok_slots == checked and all_ok
    if world > 1:
        v = torch.tensor([1 if parity_all_ranks else 0], dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        parity_all_ranks = bool(v.item())

    # ---- single-slot latency of the same pipeline (one slot = 38 codeblocks: the real-time unit of work), not part of `value`
    lat_us = lat_stage_us = None
    if rank == 0 and not args.no_latency:
        p1 = make_plans([(0, 1)])[0]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        for r in range(reps + 3):
            if r == 3:
                ev0.record(stream)
            ctx.ofdm_demodulate_slots(ocfg, ojobs_d[:24], samples_d, grid_d, stream)
            ctx.dmrs_pusch_estimate_batch(cjobs_d[:152], grid_d, ce_d, sc_d, stream)
            ctx.pusch_demodulate_batch(djobs_d[:120], grid_d, ce_d, sc_d, llr_d, stream)
            p1.run(llr_d, soft_d, msgs_d, crc_d, tb_d, res_d, stream)
        ev1.record(stream)
        torch.cuda.synchronize()
        lat_us = ev0.elapsed_time(ev1) / reps * 1e3
        # the same pipeline once more with an event after every stage (the events cost a few microseconds themselves: the total above is
        # measured without them)
        marks = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(reps)]
        for r in range(reps):
            marks[r][0].record(stream)
            ctx.ofdm_demodulate_slots(ocfg, ojobs_d[:24], samples_d, grid_d, stream)
            marks[r][1].record(stream)
            ctx.dmrs_pusch_estimate_batch(cjobs_d[:152], grid_d, ce_d, sc_d, stream)
            marks[r][2].record(stream)
            ctx.pusch_demodulate_batch(djobs_d[:120], grid_d, ce_d, sc_d, llr_d, stream)
            marks[r][3].record(stream)
            p1.run(llr_d, soft_d, msgs_d, crc_d, tb_d, res_d, stream)
            marks[r][4].record(stream)
        torch.cuda.synchronize()
        lat_stage_us = {k: float(np.mean([m[i].elapsed_time(m[i + 1]) for m in marks])) * 1e3
                        for i, k in enumerate(["ofdm_demod", "dmrs_chest", "scrambling_and_demod", "decode_and_tb_assembly"])}
        p1.close()

    # ---- per-launch durations (HIP events on the launching stream); a step has G_ch launches of each kernel
    kernel_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in ev[k]])) * G_ch for k in front}
    kernel_ms.update(back_ms)
    total_slots = S * args.steps * world
    value = total_slots * w["tbs"] / dt
    # Algorithmic bytes per launch: decode = LLRs in + K/8 out + 4 B iterations per codeblock; dematch = E in + N
    # out per codeblock; OFDM demod = 61440*8 in + 14*3276*8 out per slot-port; estimator = DM-RS REs in (n_dmrs * 13104 B) +
    # 14*3276*8 out per (slot, port, layer); demodulator = 16 B in + mod B out per data RE and port; TB assembly = K/8 per codeblock in
    # + TB bytes out.
    # When the plan dematches inside the decoder (first transmissions, rv 0) the decoder kernel carries the dematcher's figure instead of its own
    # input read: rate-matched LLRs in + soft-buffer image out (N per codeblock, pusch_decoder_impl.cpp:176-181 keeps it for the next
    # retransmission) + K/8 + 4 B out; the "rate_dematch" stage is then EMPTY (no launch: the decoder also writes the codeblock CRC flags).
    dematch_in_decoder = all(p.info()[1] for p in plans)
    alg = {"ldpc_decode": S * (G + C * (N + K // 8 + 4)) if dematch_in_decoder else S * sum(dec_in_len[c] + K // 8 + 4 for c in range(C)),
           "rate_dematch": 0 if dematch_in_decoder else S * (G + C * N),
           "ofdm_demod": S * (slot_samples * 8 + 14 * nsc * 8),
           "dmrs_chest": S * (1 * (nsc // 2) * 8 + nsc * 8),          # DM-RS REs in, one estimate row out (compact form)
           "pusch_demod": S * (w["nsym"] * 8 + nsc * 8 + G),          # data REs + the estimate row in, LLRs out
           "tb_assemble": S * (C * (K // 8) + tb_bytes)}
    gbs = {k: alg[k] / (kernel_ms[k] * 1e-3) / 1e9 for k in stages}
    dom = max(kernel_ms, key=kernel_ms.get)
    # HBM traffic per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes over this same command:
    # tools/profile_round.sh writes profiles/r02_traffic.json with the hash of the kernel sources it was collected on).
    traffic, tstamp = {}, None
    tj, tstamp = load_stamped("traffic")
    if tj:
        for k, v in tj["kernels"].items():
            traffic[k] = v["hbm_bytes_per_launch"] * S / tj["slots_per_gpu_per_step"]
    # VALU-issue roofline of the decoder (it is not an HBM kernel): wave64 VALU instructions per codeblock from the SQ counters
    # (profiles/r02_pmc_sq.json, same stamping), against (a) the guide's peak -- a wave64 VALU instruction occupies its SIMD-32 for 2
    # cycles once two or more waves are resident (MI355X_MICROARCH.md:54,473) -- and (b) the issue cost measured for the decoder's
    # own instruction mix with tools/valu_probe (profiles/r02_valu_probe.txt), which is what the kernel can actually reach.
    valu = None
    sj, sstamp = load_stamped("pmc_sq")
    if sj and "ldpc_decode" in sj.get("kernels", {}):
        kk = sj["kernels"]["ldpc_decode"]
        per_cb = kk["SQ_INSTS_VALU"] / kk["codeblocks_per_launch"]
        ach = per_cb * S * C / (kernel_ms["ldpc_decode"] * 1e-3) / 1e9
        peak = NOF_SIMDS * CLOCK_GHZ / 2.0
        valu = {"kernel": "ldpc_decode", "bound": "valu_issue", "achieved": ach, "peak": peak, "unit": "G wave-instr/s", "frac": ach / peak,
                "peak_note": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md:54,473)",
                "valu_wave_instructions_per_codeblock": per_cb, "source": sstamp}
        mix = sj.get("measured_mix_cycles_per_wave_instruction")
        if mix:
            valu["measured_issue_cost_cycles"] = mix
            valu["frac_of_measured_issue_rate"] = ach / (NOF_SIMDS * CLOCK_GHZ / mix)
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
        "config": {"workload": "273-PRB 30kHz PUSCH, 256QAM R=948/1024, 1 layer, 38 CB/slot BG1 Z=384, TBS 319784; per slot, time-domain "
                               "samples to transport block: OFDM demod (4096-pt, 1 port) + DM-RS channel estimate + equalise/soft-demap/descramble + "
                               "rate-dematch + LDPC decode (%d it, early_stop=%d) + TB assembly/CRC24A; slots synthesised by the transmit chain + AWGN"
                               % (args.max_iter, args.early_stop),
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
        "single_slot_stage_us": lat_stage_us,
        "parity_check": "%d/%d slots: LLRs, codeblocks and CRC verdict identical to the oracle; all %d transport blocks of the last step CRC-ok and equal to the "
                        "transmitted ones: %s" % (ok_slots, checked, S, all_ok),
        "transport_blocks_recovered": nof_tb_good,
        "kernel_source_sha": kernel_source_sha(),
    }
    # Roofline of the dominant kernel. The decoder is bound by VALU issue, not by HBM (SQ counters, roofline_valu): with current counters the
    # line says so and carries the issue-rate figure; the HBM figure of the contract (algorithmic bytes / launch time against 8 TB/s) stays
    # beside it as hbm_frac. Without counters of this build only the HBM figure is known.
    roof = {"kernel": dom, "bound": "hbm", "achieved": gbs[dom], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs[dom] / HBM_PEAK_GBS,
            "traffic": traffic.get(dom), "traffic_source": tstamp, "algorithmic_bytes": alg[dom], "dematch_in_decoder": dematch_in_decoder,
            "hbm_achieved_GBps": gbs[dom], "hbm_frac": gbs[dom] / HBM_PEAK_GBS}
    if dom == "ldpc_decode":
        roof["bound"] = "valu_issue"
        if valu:
            roof.update({"achieved": valu["achieved"], "peak": valu["peak"], "unit": valu["unit"], "frac": valu["frac"],
                         "frac_of_measured_issue_rate": valu.get("frac_of_measured_issue_rate"), "peak_note": valu["peak_note"]})
        else:
            roof["note"] = ("no SQ counters of this kernel build under profiles/ (tools/profile_round.sh collects them): achieved / peak / frac are the HBM "
                            "figures, although the kernel is bound by VALU issue")
    out["roofline"] = roof
    if per_rank_ms:
        out["per_rank_ms"] = per_rank_ms
    if world > 1:
        out["collective_backend"] = {"name": "gloo (rehearsal)" if rehearsal else "rccl (torch.distributed nccl backend)", "world_size_seen": dist.get_world_size(),
                                     "use": "barrier + MAX/MIN of the elapsed time + MIN of the parity verdict; no collective in the data path"}
    out["parity_all_ranks"] = parity_all_ranks
    if rehearsal:
        out["rehearsal_shared_gpu"] = True
        out["config"]["parallelism"] += " (REHEARSAL: all ranks share one GPU over gloo; not a scaling measurement)"

    # ---- single-ingest leg ("RCCL only for the batch scatter/gather"): rank 0 holds the codeword LLRs of world x S_in
    # slots, scatters them (grouped point-to-point sends, one per xGMI peer), every rank runs rate-dematch + LDPC decode + TB assembly
    # on its share, the result records are all-gathered. Timed separately; the no-collective weak-scaling number stays `value`.
    if args.ingest == "scatter":
        S_in = max(1, min(args.ingest_slots, S))
        units = world * S_in
        payload = llr_d.reshape(S, G)[torch.arange(units, device=dev) % S].contiguous() if rank == 0 else None
        pin = make_plans([(0, S_in)])[0]
        rs = miphy.PuschResult.itemsize

        def ingest(timed_parts=False):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            a = time.perf_counter()
            mine = shard.scatter_units(payload, units, 0, (G,), torch.int8, dev) if world > 1 else payload
            torch.cuda.synchronize()
            b = time.perf_counter()
            pin.run(mine.reshape(-1), soft_d, msgs_d, crc_d, tb_d, res_d, stream)
            rec = res_d[:S_in * rs].reshape(S_in, rs)
            allrec = shard.gather_results(rec, units) if world > 1 else rec
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            c = time.perf_counter()
            return b - a, c - a, allrec

        ingest()
        runs = [ingest() for _ in range(3)]
        t_sc, t_all = min(r[0] for r in runs), min(r[1] for r in runs)
        tt = torch.tensor([t_sc, t_all], dtype=torch.float64, device="cpu" if rehearsal or world == 1 else dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rec_h = runs[-1][2].cpu().numpy().reshape(-1).view(miphy.PuschResult)
        out["ingest_scatter"] = {"slots": units, "slots_per_rank": S_in, "scatter_ms": float(tt[0]) * 1e3, "total_ms": float(tt[1]) * 1e3,
                                 "scatter_GBps": (units - S_in) * G / float(tt[0]) / 1e9, "info_bits_per_s": units * w["tbs"] / float(tt[1]),
                                 "tb_crc_ok": int((rec_h["tb_crc_ok"] != 0).sum()), "backend": "gloo (rehearsal)" if rehearsal else "rccl",
                                 "stages": "scatter LLR slabs from rank 0 -> rate dematch + LDPC decode + TB assembly per rank -> all-gather result records"}
        if int((rec_h["tb_crc_ok"] != 0).sum()) != units:
            ok_slots = -2
        pin.close()

    legs_ok = True
    if rank == 0 and not args.no_extra and (world == 1 or args.extra_multi):
        legs = {}
        legs["pusch_16qam_r658_273prb"] = sch_leg(ctx, miphy, torch, dev, "273 PRB 16QAM R=658/1024: TBS 108552, 13 CB BG1 Z=384, E=13104 (15 layers)",
                                                  1, 4, 273 * 156, 108552 // 8, 1024, args.max_iter, 0.32, 7)
        legs["bg1_z384_rate_one_third"] = sch_leg(ctx, miphy, torch, dev, "BASELINE configs[0]: single codeblock BG1 Z=384, K=8448, full length N=25344 "
                                                  "(rate 1/3, 46 layers)", 1, 2, 12672, 1050, 8192, args.max_iter, 0.7, 8)
        legs["polar_pdcch"] = polar_leg(ctx, miphy, torch, dev)
        if not args.no_cpu:
            legs["polar_pdcch"]["cpu_reference"] = BL.polar_cpu_leg(ctx, miphy, torch, dev, 0.4)
        # a slot that mixes allocation sizes and MCS, same chain, with the reference's chain on the host beside it
        legs["pusch_mixed_slot"], ok_leg = BL.mixed_slot_leg(ctx, miphy, torch, dev, min(S, 1024), args.max_iter, args.snr_db, 777, args.cpu_seconds, not args.no_cpu)
        legs_ok &= ok_leg
        # retransmissions: the rate dematcher combining into full-length soft buffers, the decoder over all layers
        legs["pusch_harq_retransmissions"], ok_leg = BL.harq_retx_leg(ctx, miphy, torch, dev, 1024, args.max_iter)
        legs_ok &= ok_leg
        # the transmit half of north_star: PDSCH processor + OFDM modulator
        legs["pdsch_tx_chain"], ok_leg = BL.pdsch_tx_leg(ctx, miphy, torch, dev, w, min(S, 1024), args.max_iter, args.cpu_seconds, not args.no_cpu, HBM_PEAK_GBS)
        legs_ok &= ok_leg
        legs["pusch_4_rx_ports"] = ports_leg(ctx, miphy, torch, dev, w, grids_tx, tbs_u, 4, min(S, 256), args.max_iter, 27.0, 4321)
        if G_ch == 1 and not args.early_stop:
            # The same step with the decoder stopping at the first iteration whose codeblock CRC matches: the gNB's default
            # (pusch_dec_enable_early_stop = true, at most args.max_iter iterations); `value` stays the fixed-iteration figure of BASELINE.md.
            saved, plans[:] = plans[:], make_plans(bounds, early_stop=1)
            ms_es = ev_ms(torch, lambda: step(False), 5)
            torch.cuda.synchronize()
            rec = res_d.cpu().numpy().view(miphy.PuschResult)
            legs["pusch_early_stop"] = {"config": "the headline step with early stop (srsRAN default: at most %d iterations, stop on codeblock CRC)" % args.max_iter,
                                        "ms_per_step": ms_es, "info_bits_per_s": S * w["tbs"] / (ms_es * 1e-3), "slots_per_s": S / (ms_es * 1e-3),
                                        "ldpc_iterations_mean": float(rec["iters_mean"].mean()), "tb_crc_ok": int((rec["tb_crc_ok"] != 0).sum()),
                                        "transport_blocks_recovered": bool(np.array_equal(tb_d.cpu().numpy().reshape(S, tb_bytes)[:8],
                                                                                          np.stack([tbs_u[slot_src[i]] for i in range(8)])))}
            for q in plans:
                q.close()
            plans[:] = saved
        # One PDU per call through the HOST-descriptor entry point (what the srsRAN adapters issue per PUSCH PDU): the composed
        # miphy_pusch_process_batch on slot 0's grid -- estimator, demodulator, decode, assembly; the calls only enqueue.
        pdu = np.zeros(1, dtype=miphy.PuschPdu)
        q = pdu[0]
        q["numerology"], q["slot_in_frame"], q["rnti"], q["n_id"], q["dmrs_scrambling_id"] = 1, 0, RNTI, N_ID, DMRS_SCR_ID
        q["tb_bytes"], q["harq_cb_index"], q["mod"], q["nof_rx_ports"], q["start_symbol"], q["nof_symbols"] = tb_bytes, 0, w["mod"], 1, 0, 14
        q["bg"], q["rv"], q["new_data"], q["rx_ports"], q["use_early_stop"], q["nof_ldpc_iterations"] = w["bg"], 0, 1, [0, 1, 2, 3], args.early_stop, args.max_iter
        q["dmrs_symbols_mask"], q["grid_nof_prb"], q["rb_mask"], q["grid_offset"], q["tb_offset"] = 1 << 2, w["nprb"], rb_words, 0, 0
        sc1 = torch.zeros(20, dtype=torch.float32, device=dev)
        one = lambda: ctx.pusch_process_batch(pdu, grid_d, soft_d, msgs_d, crc_d, tb_d, res_d, sc1, stream)
        tb_d[:tb_bytes].zero_()
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        ok1 = bool(np.array_equal(tb_d[:tb_bytes].cpu().numpy(), tbs_u[slot_src[0]]))
        t1 = time.perf_counter()
        for _ in range(50):
            one()
        torch.cuda.synchronize()
        queued_us = (time.perf_counter() - t1) / 50 * 1e6
        t1 = time.perf_counter()
        for _ in range(50):
            one()
            torch.cuda.synchronize()
        synced_us = (time.perf_counter() - t1) / 50 * 1e6
        legs["host_descriptor_api_one_pdu"] = {"config": "miphy_pusch_process_batch, one 273-PRB PDU per call, host descriptors (the adapters' path)",
                                               "us_per_call_queued_back_to_back": queued_us, "us_per_call_with_synchronise": synced_us,
                                               "transport_block_recovered": ok1}
        out["legs"] = legs
        # PCIe-inclusive rate (never `value`): the S slots of time-domain samples from pinned host memory, the step, the transport
        # blocks back to pinned host memory, back to back on one stream.
        h_in = torch.empty(S * slot_samples, dtype=torch.complex64).pin_memory()
        h_in.copy_(samples_d)
        h_out = torch.empty(S * tb_bytes, dtype=torch.uint8).pin_memory()

        def pcie_step():
            samples_d.copy_(h_in, non_blocking=True)
            step(False)
            h_out.copy_(tb_d, non_blocking=True)

        ms_all = ev_ms(torch, pcie_step, 3)
        ms_h2d = ev_ms(torch, lambda: samples_d.copy_(h_in, non_blocking=True), 3)
        ms_d2h = ev_ms(torch, lambda: h_out.copy_(tb_d, non_blocking=True), 3)
        ms_step = dt / args.steps * 1e3
        out["pcie_inclusive"] = {"bits_per_s": S * w["tbs"] / (ms_all * 1e-3), "ms_h2d_plus_step_plus_d2h": ms_all, "h2d_ms": ms_h2d, "d2h_ms": ms_d2h,
                                 "h2d_GBps": S * slot_samples * 8 / ms_h2d / 1e6, "d2h_GBps": S * tb_bytes / ms_d2h / 1e6,
                                 "overlapped_bound_bits_per_s": S * w["tbs"] / (max(ms_h2d, ms_step, ms_d2h) * 1e-3),
                                 "note": "serial H2D + step + D2H measured; the overlapped bound is the slowest of the three stages"}
        out["pcie_inclusive_bits_per_s"] = out["pcie_inclusive"]["bits_per_s"]
        # The same with the upload of step i + 1 under the compute of step i: two device sample buffers, a copy stream, events both ways
        # (the transport blocks go back on the compute stream: 0.7 ms). Measured, not a bound.
        if G_ch == 1:
            samples_b = torch.empty_like(samples_d)
            bufs = [samples_d, samples_b]
            copy_s = torch.cuda.Stream()
            ready = [torch.cuda.Event(), torch.cuda.Event()]
            done = [torch.cuda.Event(), torch.cuda.Event()]
            for e_ in done:
                e_.record(stream)

            def pipelined(k_steps):
                for i in range(k_steps):
                    b_ = i % 2
                    with torch.cuda.stream(copy_s):
                        copy_s.wait_event(done[b_])  # the step that last read this buffer has finished
                        bufs[b_].copy_(h_in, non_blocking=True)
                        ready[b_].record(copy_s)
                    stream.wait_event(ready[b_])
                    step(False, bufs[b_])
                    h_out.copy_(tb_d, non_blocking=True)
                    done[b_].record(stream)

            pipelined(2)
            torch.cuda.synchronize()
            kp = 8
            t0p = time.perf_counter()
            pipelined(kp)
            torch.cuda.synchronize()
            ms_pipe = (time.perf_counter() - t0p) / kp * 1e3
            ok_pipe = bool(torch.equal(tb_d.reshape(S, tb_bytes), exp_tb))
            out["pcie_inclusive"]["overlapped_measured_bits_per_s"] = S * w["tbs"] / (ms_pipe * 1e-3)
            out["pcie_inclusive"]["overlapped_ms_per_step"] = ms_pipe
            out["pcie_inclusive"]["overlapped_transport_blocks_ok"] = ok_pipe
            del samples_b
        del h_in, h_out
        if G_ch == 1:
            # compressed-IQ ingest: BFP-9 payloads instead of time samples
            def step_from_grid():
                ctx.dmrs_pusch_estimate_batch(cjobs_d, grid_d, ce_d, sc_d, stream, max_ports=1, max_layers=1)
                ctx.pusch_demodulate_batch(djobs_d, grid_d, ce_d, sc_d, llr_d, stream)
                plans[0].run(llr_d, soft_d, msgs_d, crc_d, tb_d, res_d, stream)

            step(False)  # the receiver's grids of the step (what a fronthaul would have delivered in the frequency domain)
            out["pcie_inclusive_ofh_bfp9"], ok_leg = BL.ofh_ingest_leg(ctx, miphy, torch, dev, step_from_grid, grid_d, S, w["nprb"], w["tbs"], tb_d, tb_bytes, exp_tb)
            legs_ok &= ok_leg
    if rank == 0 and not args.no_cpu:
        samples4 = samples_d[:4 * slot_samples].cpu().numpy().reshape(4, slot_samples)
        out.update(cpu_legs(w, samples4, llr_h[:4], ofdm_args, args.max_iter, bool(args.early_stop), args.cpu_seconds))
        if "cpu_baseline_t1" in out:
            out["cpu_baseline_all_cores"] = out["cpu_baseline"]
    def graph_latency():
        """The eight launches of one slot captured once in a HIP graph and replayed (the library only enqueues: nothing in the calls
        waits for the stream, so they can be captured as they are). Runs LAST: a failed capture must not disturb the measurements."""
        reps, ev0, ev1 = 20, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            td1 = np.zeros(1, dtype=miphy.PuschTbDesc)
            td1[0] = (w["bg"], w["rv"], w["mod"], w["nof_layers"], 1, args.early_stop, args.max_iter, w["Nref"], w["nsym"], tb_bytes, 0, 0, 0)
            pg = ctx.pusch_decode_plan(td1)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                cs = torch.cuda.current_stream()
                ctx.ofdm_demodulate_slots(ocfg, ojobs_d[:24], samples_d, grid_d, cs)
                ctx.dmrs_pusch_estimate_batch(cjobs_d[:152], grid_d, ce_d, sc_d, cs)
                ctx.pusch_demodulate_batch(djobs_d[:120], grid_d, ce_d, sc_d, llr_d, cs)
                pg.run(llr_d, soft_d, msgs_d, crc_d, tb_d, res_d, cs)
            tb_d[:tb_bytes].zero_()
            for r in range(reps + 3):
                if r == 3:
                    ev0.record(stream)
                graph.replay()
            ev1.record(stream)
            torch.cuda.synchronize()
            us = ev0.elapsed_time(ev1) / reps * 1e3
            ok = np.array_equal(tb_d[:tb_bytes].cpu().numpy(), tbs_u[slot_src[0]])  # the replayed graph must deliver the transport block
            pg.close()
            return us if ok else None
        except Exception as e:  # capture is an extra: the line is printed without it
            print("bench.py: HIP graph capture of the single-slot pipeline failed: %s" % e, file=sys.stderr)
            return None

    if rank == 0 and world == 1 and not args.no_latency:
        out["single_slot_latency_hip_graph_us"] = graph_latency()
    if world > 1:
        dist.barrier()  # the other ranks wait here while rank 0 runs the CPU baseline legs (after the timed region)
    if rank == 0:
        print(json.dumps(out))
    if not parity_all_ranks or not legs_ok:
        print("PARITY FAILURE in bench (rank %d: %d of %d slots identical to the oracle, all transport blocks ok: %s, legs ok: %s)" % (rank, ok_slots, checked, all_ok, legs_ok),
              file=sys.stderr)
        sys.exit(3)
    for p in plans:
        p.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
