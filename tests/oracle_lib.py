"""ctypes bindings for the CPU oracle (oracle/libphy_oracle.so) and, when built, for the reference
itself (oracle/_ref/libref_capi.so).  TEST INFRASTRUCTURE: imported by tests/, bench.py (cpu_baseline) and
__graft_entry__.smoke() only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(vp)


class Segmentation(C.Structure):
    _fields_ = [
        ("tbs", C.c_uint), ("nof_tb_crc_bits", C.c_uint), ("nof_cbs", C.c_uint), ("Z", C.c_uint), ("K", C.c_uint),
        ("N", C.c_uint), ("cb_info_bits", C.c_uint), ("nof_cb_crc_bits", C.c_uint), ("nof_filler_bits", C.c_uint),
        ("zero_pad", C.c_uint), ("nof_short_segments", C.c_uint), ("E", C.c_uint * 52), ("cw_offset", C.c_uint * 52),
        ("crc_poly", C.c_uint),
    ]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, "libphy_oracle.so")
        if not os.path.exists(so):
            build_oracle()
        _oracle = C.CDLL(so)
        _oracle.orc_crc_bits.restype = C.c_uint32
        _oracle.orc_crc_packed.restype = C.c_uint32
    return _oracle


_ref = None


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref_capi.so"))


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref_capi.so"))
        _ref.ref_crc_bits.restype = C.c_uint32
        _ref.ref_ldpc_decoder_create.restype = vp
        _ref.ref_pusch_decoder_create.restype = vp
        _ref.ref_ldpc_decode_time.restype = C.c_double
        _ref.ref_pusch_chain_bench.restype = C.c_double
        _ref.ref_pusch_decoder_bench.restype = C.c_double
    return _ref


BG_K = {1: 22, 2: 10}
BG_NS = {1: 66, 2: 50}
BG_NF = {1: 68, 2: 52}
ALL_Z = sorted(a * 2 ** j for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * 2 ** j <= 384)
CRC24A, CRC24B, CRC24C, CRC16, CRC11, CRC6 = range(6)
CRC_ORDER = [24, 24, 24, 16, 11, 6]


# ---------------------------------------------------------------------------------------------- oracle wrappers
def o_crc_bits(poly, bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    return int(oracle().orc_crc_bits(poly, _p(bits), C.c_uint(bits.size)))


def o_ldpc_encode(bg, Z, msg, out_len):
    msg = np.ascontiguousarray(msg, dtype=np.uint8)
    assert msg.size == BG_K[bg] * Z
    out = np.zeros(out_len, dtype=np.uint8)
    rc = oracle().orc_ldpc_encode(bg, Z, _p(msg), _p(out), C.c_uint(out_len))
    assert rc == 0, rc
    return out


def o_ldpc_decode(bg, Z, llr, nof_filler=0, crc_poly=-1, max_iter=6, want_soft=False, out_init=None):
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    K = BG_K[bg] * Z
    out = np.zeros((K + 7) // 8, dtype=np.uint8) if out_init is None else out_init.copy()
    soft = np.zeros(BG_NF[bg] * Z, dtype=np.int8) if want_soft else None
    it = oracle().orc_ldpc_decode(bg, Z, _p(llr), C.c_uint(llr.size), C.c_uint(nof_filler), crc_poly,
                                  C.c_uint(max_iter), _p(out), _p(soft) if want_soft else None)
    assert it >= 0, it
    return (it, out, soft) if want_soft else (it, out)


def o_rate_match(rv, mod, Nref, nof_filler, cb, E):
    cb = np.ascontiguousarray(cb, dtype=np.uint8)
    out = np.zeros(E, dtype=np.uint8)
    rc = oracle().orc_ldpc_rate_match(rv, mod, C.c_uint(Nref), C.c_uint(nof_filler), _p(cb), C.c_uint(cb.size), _p(out), C.c_uint(E))
    assert rc == 0
    return out


def o_rate_dematch(rv, mod, Nref, nof_filler, new_data, llr, softbuf):
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    out = np.ascontiguousarray(softbuf, dtype=np.int8).copy()
    rc = oracle().orc_ldpc_rate_dematch(rv, mod, C.c_uint(Nref), C.c_uint(nof_filler), int(new_data), _p(llr),
                                        C.c_uint(llr.size), _p(out), C.c_uint(out.size))
    assert rc == 0
    return out


def o_segmentation(tbs, bg, mod, nof_layers, nof_ch_symbols):
    s = Segmentation()
    rc = oracle().orc_ldpc_segmentation(C.c_uint(tbs), bg, mod, C.c_uint(nof_layers), C.c_uint(nof_ch_symbols), C.byref(s))
    assert rc == 0, rc
    return s


def o_pdsch_encode(bg, rv, mod, Nref, nof_layers, nof_ch_symbols, tb):
    tb = np.ascontiguousarray(tb, dtype=np.uint8)
    cw = np.zeros(nof_ch_symbols * mod, dtype=np.uint8)
    rc = oracle().orc_pdsch_encode(bg, rv, mod, C.c_uint(Nref), C.c_uint(nof_layers), C.c_uint(nof_ch_symbols), _p(tb),
                                   C.c_uint(tb.size), _p(cw))
    assert rc > 0, rc
    return cw


class OraclePuschDecoder:
    """Stateful (HARQ) wrapper around orc_pusch_decode."""

    def __init__(self, bg, mod, Nref, nof_layers, nof_ch_symbols, tb_bytes):
        self.args = (bg, mod, Nref, nof_layers, nof_ch_symbols, tb_bytes)
        s = o_segmentation(tb_bytes * 8, bg, mod, nof_layers, nof_ch_symbols)
        self.seg = s
        self.softbuf = np.zeros(s.nof_cbs * s.N, dtype=np.int8)
        self.cb_crc = np.zeros(s.nof_cbs, dtype=np.uint8)
        self.cb_msgs = np.zeros(s.nof_cbs * ((s.K + 7) // 8), dtype=np.uint8)

    def decode(self, llrs, rv, new_data, max_iter=6, early_stop=True):
        bg, mod, Nref, nl, nsym, tb_bytes = self.args
        llrs = np.ascontiguousarray(llrs, dtype=np.int8)
        tb = np.zeros(tb_bytes, dtype=np.uint8)
        mm = (C.c_int * 2)()
        ok = oracle().orc_pusch_decode(bg, rv, mod, C.c_uint(Nref), C.c_uint(nl), C.c_uint(nsym), C.c_uint(tb_bytes),
                                       int(new_data), _p(llrs), C.c_uint(max_iter), int(early_stop), _p(self.softbuf),
                                       _p(self.cb_crc), _p(self.cb_msgs), _p(tb), mm)
        assert ok >= 0
        return bool(ok), tb, (mm[0], mm[1])


class OfdmCfg(C.Structure):
    _fields_ = [("numerology", C.c_uint), ("bw_rb", C.c_uint), ("dft_size", C.c_uint), ("window_offset", C.c_uint),
                ("scale", C.c_float), ("center_freq_hz", C.c_double)]


def o_dft(x, inverse=False):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.zeros_like(x)
    oracle().orc_dft(C.c_uint(x.size), int(inverse), _p(x), _p(out))
    return out


def o_ofdm_slot_size(cfg, slot_index):
    oracle().orc_ofdm_slot_size.restype = C.c_uint
    return int(oracle().orc_ofdm_slot_size(C.byref(cfg), C.c_uint(slot_index)))


def o_ofdm_demod_slot(cfg, slot_index, samples):
    samples = np.ascontiguousarray(samples, dtype=np.complex64)
    assert samples.size == o_ofdm_slot_size(cfg, slot_index)
    grid = np.zeros((14, cfg.bw_rb * 12), dtype=np.complex64)
    oracle().orc_ofdm_demod_slot(C.byref(cfg), C.c_uint(slot_index), _p(samples), _p(grid))
    return grid


def o_ofdm_mod_slot(cfg, slot_index, grid):
    grid = np.ascontiguousarray(grid, dtype=np.complex64)
    out = np.zeros(o_ofdm_slot_size(cfg, slot_index), dtype=np.complex64)
    oracle().orc_ofdm_mod_slot(C.byref(cfg), C.c_uint(slot_index), _p(grid), _p(out))
    return out


def _chest_args(numerology, slot, type2, scr_id, n_scid, scaling, symbols_mask, rb_mask, first, nof, nl, grid):
    symbols_mask = np.ascontiguousarray(symbols_mask, dtype=np.uint8)
    rb_mask = np.ascontiguousarray(rb_mask, dtype=np.uint8)
    grid = np.ascontiguousarray(grid, dtype=np.complex64)
    nports, nsym, nsc = grid.shape
    assert nsym == 14 and nsc == rb_mask.size * 12 and symbols_mask.size == 14
    ce = np.ones((nl, nports, first + nof, nsc), dtype=np.complex64)
    sc = np.zeros((nports, nl, 5), dtype=np.float32)
    args = (C.c_uint(numerology), C.c_uint(slot), int(type2), C.c_uint(scr_id), int(n_scid), C.c_float(scaling), _p(symbols_mask),
            _p(rb_mask), C.c_uint(rb_mask.size), C.c_uint(first), C.c_uint(nof), C.c_uint(nl), C.c_uint(nports), _p(grid), _p(ce), _p(sc))
    return args, ce, sc, (symbols_mask, rb_mask, grid)


def o_dmrs_pusch_estimate(*a):
    args, ce, sc, keep = _chest_args(*a)
    rc = oracle().orc_dmrs_pusch_estimate(*args)
    assert rc == 0, rc
    return ce, sc


def o_gold(c_init, offset, nbits):
    out = np.zeros(nbits, dtype=np.uint8)
    oracle().orc_gold_sequence(C.c_uint(c_init), C.c_uint(offset), C.c_uint(nbits), _p(out))
    return out


def o_polar_encode_chain(K, E, nMax, ibil, msg):
    msg = np.ascontiguousarray(msg, dtype=np.uint8)
    out, alloc, enc = np.zeros(E, np.uint8), np.zeros(1024, np.uint8), np.zeros(1024, np.uint8)
    N = oracle().orc_polar_encode_chain(C.c_uint(K), C.c_uint(E), C.c_uint(nMax), int(ibil), _p(msg), _p(out), _p(alloc), _p(enc))
    assert N > 0, N
    return out, alloc[:N], enc[:N]


def o_polar_interleave(bits, K, rx):
    """polar_interleaver::interleave (TX: rx = 0, RX: rx = 1) of K bits."""
    x = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(K, np.uint8)
    oracle().orc_polar_interleave(_p(x), _p(out), C.c_uint(K), int(rx))
    return out


def o_polar_decode_chain(K, E, nMax, ibil, llr):
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    msg, dem, u = np.zeros(K, np.uint8), np.zeros(1024, np.int8), np.zeros(1024, np.uint8)
    N = oracle().orc_polar_decode_chain(C.c_uint(K), C.c_uint(E), C.c_uint(nMax), int(ibil), _p(llr), _p(msg), _p(dem), _p(u))
    assert N > 0, N
    return msg, dem[:N], u[:N]


def o_polar_sc_textbook(K, E, nMax, ibil, llr):
    """Plain successive cancellation written from the definition (no pruning, no list). Returns (message, zero_seen)."""
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    msg = np.zeros(K, np.uint8)
    z = C.c_int(0)
    N = oracle().orc_polar_sc_textbook(C.c_uint(K), C.c_uint(E), C.c_uint(nMax), int(ibil), _p(llr), _p(msg), C.byref(z))
    assert N > 0, N
    return msg, bool(z.value)


def o_polar_scl_decode(K, E, nMax, ibil, L, crc_mode, rnti, llr):
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    msg = np.zeros(K, np.uint8)
    ok = C.c_int(0)
    pm = oracle().orc_polar_scl_decode(C.c_uint(K), C.c_uint(E), C.c_uint(nMax), int(ibil), C.c_uint(L), int(crc_mode), C.c_uint(rnti),
                                       _p(llr), _p(msg), C.byref(ok))
    assert pm >= 0, pm
    return msg, bool(ok.value), pm


def o_pdcch_encode(payload, rnti, E):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(E, np.uint8)
    rc = oracle().orc_pdcch_encode(_p(payload), C.c_uint(payload.size), C.c_uint(rnti), C.c_uint(E), _p(out))
    assert rc == 0
    return out


def o_pbch_encode(N_id, ssb_idx, L_max, hrf, sfn, k_ssb, payload):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(864, np.uint8)
    rc = oracle().orc_pbch_encode(C.c_uint(N_id), C.c_uint(ssb_idx), C.c_uint(L_max), int(hrf), C.c_uint(sfn), C.c_uint(k_ssb), _p(payload), _p(out))
    assert rc == 0
    return out


# ---------------------------------------------------------------------------------------------- reference wrappers
IMPL = {"generic": 0, "avx2": 1, "avx512": 2, "auto": 3}


def r_crc_bits(poly, bits, lut=False):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    return int(ref().ref_crc_bits(poly, _p(bits), C.c_uint(bits.size), int(lut)))


def r_ldpc_encode(bg, Z, msg, out_len, impl="avx2"):
    msg = np.ascontiguousarray(msg, dtype=np.uint8)
    out = np.zeros(out_len, dtype=np.uint8)
    rc = ref().ref_ldpc_encode(bg, Z, _p(msg), C.c_uint(msg.size), _p(out), C.c_uint(out_len), IMPL[impl])
    assert rc == 0
    return out


class RefLdpcDecoder:
    def __init__(self, impl="avx2"):
        self.h = vp(ref().ref_ldpc_decoder_create(IMPL[impl]))
        assert self.h.value, "reference decoder '%s' unavailable on this CPU" % impl

    def decode(self, bg, Z, llr, nof_filler=0, crc_poly=-1, max_iter=6):
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        K = BG_K[bg] * Z
        out = np.zeros((K + 7) // 8, dtype=np.uint8)
        it = ref().ref_ldpc_decode(self.h, bg, Z, _p(llr), C.c_uint(llr.size), C.c_uint(nof_filler), C.c_uint(24), crc_poly,
                                   C.c_uint(max_iter), _p(out))
        return it, out

    def time_batch(self, bg, Z, llrs, in_len, n_cb, nof_filler, crc_poly, max_iter, reps):
        llrs = np.ascontiguousarray(llrs, dtype=np.int8)
        K = BG_K[bg] * Z
        out = np.zeros(n_cb * ((K + 7) // 8), dtype=np.uint8)
        t = ref().ref_ldpc_decode_time(self.h, bg, Z, _p(llrs), C.c_uint(in_len), C.c_uint(n_cb), C.c_uint(nof_filler), crc_poly,
                                       C.c_uint(max_iter), C.c_uint(reps), _p(out))
        return t, out

    def __del__(self):
        try:
            ref().ref_ldpc_decoder_destroy(self.h)
        except Exception:
            pass


def r_rate_match(bg, Z, rv, mod, Nref, nof_filler, cb, E):
    cb = np.ascontiguousarray(cb, dtype=np.uint8)
    out = np.zeros(E, dtype=np.uint8)
    ref().ref_ldpc_rate_match(bg, Z, rv, mod, C.c_uint(Nref), C.c_uint(nof_filler), _p(cb), C.c_uint(cb.size), _p(out), C.c_uint(E))
    return out


def r_rate_dematch(bg, Z, rv, mod, Nref, nof_filler, new_data, llr, softbuf, impl="avx2"):
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    out = np.ascontiguousarray(softbuf, dtype=np.int8).copy()
    rc = ref().ref_ldpc_rate_dematch(bg, Z, rv, mod, C.c_uint(Nref), C.c_uint(nof_filler), int(new_data), _p(llr),
                                     C.c_uint(llr.size), _p(out), C.c_uint(out.size), IMPL[impl])
    assert rc == 0
    return out


def r_pdsch_encode(bg, rv, mod, Nref, nof_layers, nof_ch_symbols, tb, impl="avx2"):
    tb = np.ascontiguousarray(tb, dtype=np.uint8)
    cw = np.zeros(nof_ch_symbols * mod, dtype=np.uint8)
    rc = ref().ref_pdsch_encode(bg, rv, mod, C.c_uint(Nref), C.c_uint(nof_layers), C.c_uint(nof_ch_symbols), _p(tb),
                                C.c_uint(tb.size), _p(cw), C.c_uint(cw.size), IMPL[impl])
    assert rc == 0
    return cw


class RefPuschDecoder:
    def __init__(self, impl="avx2"):
        self.h = vp(ref().ref_pusch_decoder_create(IMPL[impl]))
        assert self.h.value

    def decode_sequence(self, bg, mod, Nref, nof_layers, nof_ch_symbols, tb_bytes, rvs, llrs, max_iter=6, early_stop=True):
        """llrs: [nof_tx, cw_len]."""
        llrs = np.ascontiguousarray(llrs, dtype=np.int8)
        nof_tx, cw_len = llrs.shape
        rv_arr = (C.c_int * nof_tx)(*rvs)
        tb = np.zeros((nof_tx, tb_bytes), dtype=np.uint8)
        ok = (C.c_int * nof_tx)()
        mm = (C.c_int * (2 * nof_tx))()
        n = ref().ref_pusch_decode(self.h, bg, mod, C.c_uint(Nref), C.c_uint(nof_layers), C.c_uint(nof_ch_symbols),
                                   C.c_uint(tb_bytes), C.c_uint(nof_tx), rv_arr, _p(llrs), C.c_uint(cw_len), C.c_uint(max_iter),
                                   int(early_stop), _p(tb), ok, mm)
        assert n > 0, n
        return [bool(x) for x in ok], tb, [(mm[2 * i], mm[2 * i + 1]) for i in range(nof_tx)]

    def __del__(self):
        try:
            ref().ref_pusch_decoder_destroy(self.h)
        except Exception:
            pass


def host_cpus():
    """CPUs this process may run on, and the CPU budget of its cgroup (cpu.max quota / period) when there is one."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return cpus, quota


def r_pusch_chain_bench(nthreads, cpus, seconds, stage, samples, nof_prb, mod, tbs_bits, rnti, n_id, dmrs_scr_id, dft_size, window_offset,
                        scale, center_freq_hz, max_iter, early_stop):
    """Reference receive chain on `nthreads` pinned threads (ref_capi.cpp::ref_pusch_chain_bench). samples: [nslots][slot_samples]
    complex64, slot s = slot-in-frame s. stage 1: OFDM demodulation + pusch_processor, 0: pusch_processor only.
    Returns (elapsed_s, slots_done, tb_ok)."""
    samples = np.ascontiguousarray(samples, dtype=np.complex64)
    nslots, slot_samples = samples.shape
    cp = (C.c_int * nthreads)(*[int(cpus[t % len(cpus)]) if cpus else -1 for t in range(nthreads)])
    done = (C.c_uint64 * nthreads)()
    ok = (C.c_uint64 * nthreads)()
    dt = ref().ref_pusch_chain_bench(C.c_uint(nthreads), cp, C.c_double(seconds), int(stage), _p(samples), C.c_uint(nslots), C.c_uint(slot_samples),
                                     C.c_uint(nof_prb), int(mod), C.c_uint(tbs_bits), C.c_uint(rnti), C.c_uint(n_id), C.c_uint(dmrs_scr_id),
                                     C.c_uint(dft_size), C.c_uint(window_offset), C.c_float(scale), C.c_double(center_freq_hz), C.c_uint(max_iter),
                                     int(early_stop), done, ok)
    return float(dt), int(sum(done)), int(sum(ok))


def r_pusch_decoder_bench(nthreads, cpus, seconds, llrs, mod, nof_ch_symbols, tbs_bits, max_iter, early_stop):
    """Reference pusch_decoder (AVX2 dematcher + decoder) on `nthreads` pinned threads. llrs: [nslots][cw_len] int8."""
    llrs = np.ascontiguousarray(llrs, dtype=np.int8)
    nslots, cw_len = llrs.shape
    cp = (C.c_int * nthreads)(*[int(cpus[t % len(cpus)]) if cpus else -1 for t in range(nthreads)])
    done = (C.c_uint64 * nthreads)()
    ok = (C.c_uint64 * nthreads)()
    dt = ref().ref_pusch_decoder_bench(C.c_uint(nthreads), cp, C.c_double(seconds), _p(llrs), C.c_uint(nslots), C.c_uint(cw_len), int(mod),
                                       C.c_uint(nof_ch_symbols), C.c_uint(tbs_bits), C.c_uint(max_iter), int(early_stop), done, ok)
    return float(dt), int(sum(done)), int(sum(ok))


def r_tbs_calculate(nof_symb_sh, nof_dmrs_prb, mod_bits, rate_x1024, nof_layers, n_prb, nof_oh_prb=0):
    """The reference's tbs_calculator_calculate (TS 38.214 5.1.3.2)."""
    f = ref().ref_tbs_calculate
    f.restype = C.c_uint
    return int(f(C.c_uint(nof_symb_sh), C.c_uint(nof_dmrs_prb), C.c_uint(nof_oh_prb), int(mod_bits), C.c_float(rate_x1024), C.c_uint(nof_layers), C.c_uint(n_prb)))


def r_ldpc_base_graph(rate_x1024, tbs_bits):
    return int(ref().ref_ldpc_base_graph(C.c_float(rate_x1024), C.c_uint(tbs_bits)))


def r_pusch_chain_bench_multi(nthreads, cpus, seconds, stage, samples, grid_prb, pdus, dmrs_scr_id, dft_size, window_offset, scale, center_freq_hz, max_iter,
                              early_stop, isa=1):
    """Reference receive chain of slots that carry several PUSCH PDUs (ref_capi.cpp::ref_pusch_chain_bench_multi): OFDM demodulation once per
    slot, pusch_processor per PDU. pdus: rows of (rb_start, nof_prb, mod bits, TBS bits, base graph, rnti, n_id, R x 1024).
    Returns (elapsed_s, slots_done, transport blocks CRC-ok)."""
    samples = np.ascontiguousarray(samples, dtype=np.complex64)
    nslots, slot_samples = samples.shape
    pd = np.ascontiguousarray(np.asarray(pdus, dtype=np.uint32).reshape(-1, 8))
    cp = (C.c_int * nthreads)(*[int(cpus[t % len(cpus)]) if cpus else -1 for t in range(nthreads)])
    done = (C.c_uint64 * nthreads)()
    ok = (C.c_uint64 * nthreads)()
    f = ref().ref_pusch_chain_bench_multi
    f.restype = C.c_double
    dt = f(C.c_uint(nthreads), cp, C.c_double(seconds), int(stage), _p(samples), C.c_uint(nslots), C.c_uint(slot_samples), C.c_uint(grid_prb),
           C.c_uint(pd.shape[0]), _p(pd), C.c_uint(dmrs_scr_id), C.c_uint(dft_size), C.c_uint(window_offset), C.c_float(scale), C.c_double(center_freq_hz),
           C.c_uint(max_iter), int(early_stop), int(isa), done, ok)
    return float(dt), int(sum(done)), int(sum(ok))


def r_pusch_decoder_bench_isa(nthreads, cpus, seconds, llrs, mod, nof_ch_symbols, tbs_bits, max_iter, early_stop, isa):
    """ref_pusch_decoder_bench with the reference's decoder / dematcher classes of one instruction set: isa 1 = avx2, 2 = avx512.
    Returns None when the host (or the build) lacks it."""
    llrs = np.ascontiguousarray(llrs, dtype=np.int8)
    nslots, cw_len = llrs.shape
    cp = (C.c_int * nthreads)(*[int(cpus[t % len(cpus)]) if cpus else -1 for t in range(nthreads)])
    done = (C.c_uint64 * nthreads)()
    ok = (C.c_uint64 * nthreads)()
    f = ref().ref_pusch_decoder_bench_isa
    f.restype = C.c_double
    dt = f(C.c_uint(nthreads), cp, C.c_double(seconds), _p(llrs), C.c_uint(nslots), C.c_uint(cw_len), int(mod), C.c_uint(nof_ch_symbols), C.c_uint(tbs_bits),
           C.c_uint(max_iter), int(early_stop), int(isa), done, ok)
    return None if dt < 0 else (float(dt), int(sum(done)), int(sum(ok)))


def r_pdsch_chain_bench(nthreads, cpus, seconds, with_ofdm, tbs, nof_prb, mod, tbs_bits, rnti, n_id, dmrs_scr_id, dft_size, scale, center_freq_hz):
    """Reference transmit chain (ref_capi.cpp::ref_pdsch_chain_bench): pdsch_processor::process of one full-band PDU per slot and, with_ofdm,
    ofdm_slot_modulator::modulate of its grid. tbs: [nslots][tbs_bits / 8] uint8. Returns (elapsed_s, slots_done)."""
    tbs = np.ascontiguousarray(tbs, dtype=np.uint8)
    nslots = tbs.shape[0]
    cp = (C.c_int * nthreads)(*[int(cpus[t % len(cpus)]) if cpus else -1 for t in range(nthreads)])
    done = (C.c_uint64 * nthreads)()
    chk = C.c_float()
    f = ref().ref_pdsch_chain_bench
    f.restype = C.c_double
    dt = f(C.c_uint(nthreads), cp, C.c_double(seconds), int(with_ofdm), _p(tbs), C.c_uint(nslots), C.c_uint(nof_prb), int(mod), C.c_uint(tbs_bits), C.c_uint(rnti),
           C.c_uint(n_id), C.c_uint(dmrs_scr_id), C.c_uint(dft_size), C.c_float(scale), C.c_double(center_freq_hz), done, C.byref(chk))
    return float(dt), int(sum(done))


def r_polar_chain_bench(nthreads, cpus, seconds, stage, A, E, payloads, llrs):
    """Reference PDCCH polar chains (ref_capi.cpp::ref_polar_chain_bench): stage 0 = pdcch_encoder::encode, stage 1 = rate dematcher + SSC
    decoder + deallocator. payloads [ncw][A] bits, llrs [ncw][E] int8. Returns (elapsed_s, codewords done)."""
    payloads = np.ascontiguousarray(payloads, dtype=np.uint8)
    llrs = np.ascontiguousarray(llrs, dtype=np.int8)
    ncw = payloads.shape[0]
    cp = (C.c_int * nthreads)(*[int(cpus[t % len(cpus)]) if cpus else -1 for t in range(nthreads)])
    done = (C.c_uint64 * nthreads)()
    f = ref().ref_polar_chain_bench
    f.restype = C.c_double
    dt = f(C.c_uint(nthreads), cp, C.c_double(seconds), int(stage), C.c_uint(A), C.c_uint(E), _p(payloads), _p(llrs), C.c_uint(ncw), done)
    return float(dt), int(sum(done))


def r_dft(x, inverse=False):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.zeros_like(x)
    rc = ref().ref_dft(C.c_uint(x.size), int(inverse), _p(x), _p(out))
    assert rc == 0
    return out


def r_ofdm_demod_slot(cfg, slot_index, samples):
    samples = np.ascontiguousarray(samples, dtype=np.complex64)
    grid = np.zeros((14, cfg.bw_rb * 12), dtype=np.complex64)
    rc = ref().ref_ofdm_demod_slot(C.c_uint(cfg.numerology), C.c_uint(cfg.bw_rb), C.c_uint(cfg.dft_size), C.c_uint(cfg.window_offset),
                                   C.c_float(cfg.scale), C.c_double(cfg.center_freq_hz), C.c_uint(slot_index), _p(samples),
                                   C.c_uint(samples.size), _p(grid))
    assert rc == 0, rc
    return grid


def r_ofdm_mod_slot(cfg, slot_index, grid, nsamples):
    grid = np.ascontiguousarray(grid, dtype=np.complex64)
    out = np.zeros(nsamples, dtype=np.complex64)
    rc = ref().ref_ofdm_mod_slot(C.c_uint(cfg.numerology), C.c_uint(cfg.bw_rb), C.c_uint(cfg.dft_size), C.c_float(cfg.scale),
                                 C.c_double(cfg.center_freq_hz), C.c_uint(slot_index), _p(grid), _p(out), C.c_uint(nsamples))
    assert rc == 0, rc
    return out


def r_dmrs_pusch_estimate(*a):
    args, ce, sc, keep = _chest_args(*a)
    rc = ref().ref_dmrs_pusch_estimate(*args)
    assert rc == 0, rc
    return ce, sc


def r_polar_encode_chain(K, E, nMax, ibil, msg):
    msg = np.ascontiguousarray(msg, dtype=np.uint8)
    out, alloc, enc = np.zeros(E, np.uint8), np.zeros(1024, np.uint8), np.zeros(1024, np.uint8)
    N = ref().ref_polar_encode_chain(C.c_uint(K), C.c_uint(E), C.c_uint(nMax), int(ibil), _p(msg), _p(out), _p(alloc), _p(enc))
    return out, alloc[:N], enc[:N]


def r_polar_decode_chain(K, E, nMax, ibil, llr):
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    msg, dem, u = np.zeros(K, np.uint8), np.zeros(1024, np.int8), np.zeros(1024, np.uint8)
    N = ref().ref_polar_decode_chain(C.c_uint(K), C.c_uint(E), C.c_uint(nMax), int(ibil), _p(llr), _p(msg), _p(dem), _p(u))
    return msg, dem[:N], u[:N]


def r_pdcch_encode(payload, rnti, E):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(E, np.uint8)
    ref().ref_pdcch_encode(_p(payload), C.c_uint(payload.size), C.c_uint(rnti), C.c_uint(E), _p(out))
    return out


def r_pbch_encode(N_id, ssb_idx, L_max, hrf, sfn, k_ssb, payload):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(864, np.uint8)
    ref().ref_pbch_encode(C.c_uint(N_id), C.c_uint(ssb_idx), C.c_uint(L_max), int(hrf), C.c_uint(sfn), C.c_uint(k_ssb), _p(payload), _p(out))
    return out


# ------------------------------------------------------------------------------------------------ PUSCH demodulator
def nr_modulate(bits, mod):
    """TS 38.211 5.1 mapper (numpy, float32): bits one per element -> complex64 symbols. mod in {1 (pi/2-BPSK), 2, 4, 6, 8}."""
    b = np.asarray(bits, dtype=np.float32).reshape(-1, mod)
    s = 1.0 - 2.0 * b
    if mod == 1:
        i = np.arange(b.shape[0])
        z = (s[:, 0] + 1j * s[:, 0]) / np.sqrt(2.0)
        return (z * np.exp(1j * (np.pi / 2) * (i % 2))).astype(np.complex64)
    if mod == 2:
        return ((s[:, 0] + 1j * s[:, 1]) / np.sqrt(2.0)).astype(np.complex64)
    if mod == 4:
        return ((s[:, 0] * (2 - s[:, 2]) + 1j * s[:, 1] * (2 - s[:, 3])) / np.sqrt(10.0)).astype(np.complex64)
    if mod == 6:
        return ((s[:, 0] * (4 - s[:, 2] * (2 - s[:, 4])) + 1j * s[:, 1] * (4 - s[:, 3] * (2 - s[:, 5]))) / np.sqrt(42.0)).astype(np.complex64)
    return ((s[:, 0] * (8 - s[:, 2] * (4 - s[:, 4] * (2 - s[:, 6]))) + 1j * s[:, 1] * (8 - s[:, 3] * (4 - s[:, 5] * (2 - s[:, 7])))) /
            np.sqrt(170.0)).astype(np.complex64)


def o_demodulate_soft(mod, symbols, noise_vars):
    sym = np.ascontiguousarray(symbols, dtype=np.complex64)
    nv = np.ascontiguousarray(noise_vars, dtype=np.float32)
    out = np.zeros(sym.size * mod, dtype=np.int8)
    oracle().orc_demodulate_soft(int(mod), C.c_uint(sym.size), _p(sym), _p(nv), _p(out))
    return out


def r_demodulate_soft(mod, symbols, noise_vars):
    sym = np.ascontiguousarray(symbols, dtype=np.complex64)
    nv = np.ascontiguousarray(noise_vars, dtype=np.float32)
    out = np.zeros(sym.size * mod, dtype=np.int8)
    assert ref().ref_demodulate_soft(int(mod), C.c_uint(sym.size), _p(sym), _p(nv), _p(out)) == 0
    return out


def r_modulate(mod, bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(bits.size // mod, dtype=np.complex64)
    assert ref().ref_modulate(int(mod), C.c_uint(out.size), _p(bits), _p(out)) == 0
    return out


def pusch_nof_re(start, nof, dmrs_mask, type2, cdm, rb_mask):
    per_dmrs = 12 - (cdm * (4 if type2 else 6))
    nprb = int(np.count_nonzero(rb_mask))
    return sum(nprb * (per_dmrs if dmrs_mask[s] else 12) for s in range(start, start + nof))


def _demod_args(rnti, n_id, mod, start, nof, dmrs_mask, type2, cdm, rb_mask, grid, ce, noise_var):
    dm = np.ascontiguousarray(dmrs_mask, dtype=np.uint8)
    rb = np.ascontiguousarray(rb_mask, dtype=np.uint8)
    g = np.ascontiguousarray(grid, dtype=np.complex64)  # [ports][14][nsc]
    h = np.ascontiguousarray(ce, dtype=np.complex64)    # [ports][ce_syms][nsc]
    assert g.ndim == 3 and g.shape[1] == 14 and h.ndim == 3 and h.shape[0] == g.shape[0] and h.shape[2] == g.shape[2]
    n = pusch_nof_re(start, nof, dm, type2, cdm, rb)
    llr = np.zeros(n * mod, dtype=np.int8)
    args = [C.c_uint(rnti), C.c_uint(n_id), int(mod), C.c_uint(start), C.c_uint(nof), _p(dm), int(type2), C.c_uint(cdm), _p(rb),
            C.c_uint(rb.size), C.c_uint(g.shape[0]), _p(g), _p(h), C.c_uint(h.shape[1]), C.c_float(noise_var), _p(llr)]
    return args, llr, n, (dm, rb, g, h)


def o_pusch_demodulate(*a):
    args, llr, n, keep = _demod_args(*a)
    eq = np.zeros(n, dtype=np.complex64)
    nv = np.zeros(n, dtype=np.float32)
    got = oracle().orc_pusch_demodulate(*args, _p(eq), _p(nv))
    assert got == llr.size, (got, llr.size)
    return llr, eq, nv


def o_pusch_demodulate_ex(*a, placeholders=(), want_evm=True):
    """orc_pusch_demodulate with repetition placeholders (sorted RE indices) and the EVM. Returns (llr, evm)."""
    args, llr, n, keep = _demod_args(*a)
    ph = np.ascontiguousarray(placeholders, dtype=np.uint16)
    evm = C.c_float(0)
    got = oracle().orc_pusch_demodulate_ex(*args, None, None, _p(ph) if ph.size else None, C.c_uint(ph.size), C.byref(evm) if want_evm else None)
    assert got == llr.size, (got, llr.size)
    return llr, float(evm.value)


def r_pusch_demodulate_ex(*a, placeholders=()):
    args, llr, n, keep = _demod_args(*a)
    ph = np.ascontiguousarray(placeholders, dtype=np.uint16)
    evm = C.c_float(0)
    assert ref().ref_pusch_demodulate_ex(*args, C.c_uint(llr.size), _p(ph) if ph.size else None, C.c_uint(ph.size), C.byref(evm)) == 0
    return llr, float(evm.value)


def r_pusch_demodulate(*a):
    args, llr, n, keep = _demod_args(*a)
    assert ref().ref_pusch_demodulate(*args, C.c_uint(llr.size)) == 0
    return llr


# ------------------------------------------------------------------------------------------------ PDSCH modulator + DM-RS
def o_modulate(mod, bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(bits.size // mod, dtype=np.complex64)
    oracle().orc_modulate(int(mod), C.c_uint(out.size), _p(bits), _p(out))
    return out


def _reserved_args(reserved, nprb_grid):
    """reserved: list of (prb_mask bytes [nprb_grid], re_mask 12 bits, symbols 14 bits)."""
    n = len(reserved)
    pm = np.zeros((max(n, 1), nprb_grid), dtype=np.uint8)
    rm = np.zeros(max(n, 1), dtype=np.uint16)
    sm = np.zeros(max(n, 1), dtype=np.uint16)
    for i, (p, r, s) in enumerate(reserved):
        pm[i], rm[i], sm[i] = p, r, s
    return n, pm, rm, sm


def o_pdsch_modulate(rnti, n_id, scaling, nof_layers, mods, cws, start, nof, dmrs_mask, type2, cdm, bwp_start, bwp_size, prb_list, reserved,
                     ports, nprb_grid, grid):
    """grid: complex64 [nof_grid_ports][14][nsc], updated in place (only the mapped REs). Returns the number of REs per layer."""
    mod = (C.c_int * 2)(int(mods[0]), int(mods[1] if len(mods) > 1 else mods[0]))
    cw0 = np.ascontiguousarray(cws[0], dtype=np.uint8)
    cw1 = np.ascontiguousarray(cws[1] if len(cws) > 1 else np.zeros(1, np.uint8), dtype=np.uint8)
    dm = np.ascontiguousarray(dmrs_mask, dtype=np.uint8)
    pl = np.ascontiguousarray(prb_list, dtype=np.uint16)
    n, pm, rm, sm = _reserved_args(reserved, nprb_grid)
    pt = np.ascontiguousarray(ports, dtype=np.uint8)
    assert grid.dtype == np.complex64 and grid.flags.c_contiguous
    return oracle().orc_pdsch_modulate(C.c_uint(rnti), C.c_uint(n_id), C.c_float(scaling), C.c_uint(nof_layers), mod, _p(cw0), C.c_uint(cw0.size),
                                       _p(cw1), C.c_uint(cw1.size if len(cws) > 1 else 0), C.c_uint(start), C.c_uint(nof), _p(dm), int(type2),
                                       C.c_uint(cdm), C.c_uint(bwp_start), C.c_uint(bwp_size), _p(pl), C.c_uint(pl.size), C.c_uint(n), _p(pm),
                                       _p(rm), _p(sm), _p(pt), C.c_uint(nprb_grid), _p(grid))


def r_pdsch_modulate(rnti, n_id, scaling, nof_layers, mods, cws, start, nof, dmrs_mask, type2, cdm, bwp_start, bwp_size, vrb_mask, interleaved,
                     reserved, ports, nprb_grid, nof_grid_ports):
    """Returns (grid [nof_grid_ports][14][nsc], prb_list in mapping order)."""
    mod = (C.c_int * 2)(int(mods[0]), int(mods[1] if len(mods) > 1 else mods[0]))
    cw0 = np.ascontiguousarray(cws[0], dtype=np.uint8)
    cw1 = np.ascontiguousarray(cws[1] if len(cws) > 1 else np.zeros(1, np.uint8), dtype=np.uint8)
    dm = np.ascontiguousarray(dmrs_mask, dtype=np.uint8)
    vm = np.ascontiguousarray(vrb_mask, dtype=np.uint8)
    n, pm, rm, sm = _reserved_args(reserved, nprb_grid)
    pt = np.ascontiguousarray(ports, dtype=np.uint8)
    grid = np.zeros((nof_grid_ports, 14, nprb_grid * 12), dtype=np.complex64)
    pl = np.zeros(275, dtype=np.uint16)
    npl = C.c_uint(0)
    rc = ref().ref_pdsch_modulate(C.c_uint(rnti), C.c_uint(n_id), C.c_float(scaling), C.c_uint(nof_layers), mod, _p(cw0), C.c_uint(cw0.size), _p(cw1),
                                  C.c_uint(cw1.size if len(cws) > 1 else 0), C.c_uint(start), C.c_uint(nof), _p(dm), int(type2), C.c_uint(cdm),
                                  C.c_uint(bwp_start), C.c_uint(bwp_size), _p(vm), int(interleaved), C.c_uint(n), _p(pm), _p(rm), _p(sm), _p(pt),
                                  C.c_uint(nprb_grid), C.c_uint(nof_grid_ports), _p(grid), _p(pl), C.byref(npl))
    assert rc == 0
    return grid, pl[:npl.value].copy()


def o_dmrs_pdsch_map(slot_in_frame, ref_point, type2, scr_id, n_scid, amplitude, symbols_mask, rb_mask, ports, grid):
    sm = np.ascontiguousarray(symbols_mask, dtype=np.uint8)
    rb = np.ascontiguousarray(rb_mask, dtype=np.uint8)
    pt = np.ascontiguousarray(ports, dtype=np.uint8)
    assert grid.dtype == np.complex64 and grid.flags.c_contiguous
    return oracle().orc_dmrs_pdsch_map(C.c_uint(slot_in_frame), C.c_uint(ref_point), int(type2), C.c_uint(scr_id), int(n_scid), C.c_float(amplitude),
                                       _p(sm), _p(rb), C.c_uint(rb.size), C.c_uint(pt.size), _p(pt), _p(grid))


def r_dmrs_pdsch_map(numerology, slot_index, ref_point, type2, scr_id, n_scid, amplitude, symbols_mask, rb_mask, ports, nof_grid_ports):
    sm = np.ascontiguousarray(symbols_mask, dtype=np.uint8)
    rb = np.ascontiguousarray(rb_mask, dtype=np.uint8)
    pt = np.ascontiguousarray(ports, dtype=np.uint8)
    grid = np.zeros((nof_grid_ports, 14, rb.size * 12), dtype=np.complex64)
    assert ref().ref_dmrs_pdsch_map(C.c_uint(numerology), C.c_uint(slot_index), C.c_uint(ref_point), int(type2), C.c_uint(scr_id), int(n_scid),
                                    C.c_float(amplitude), _p(sm), _p(rb), C.c_uint(rb.size), C.c_uint(pt.size), _p(pt), C.c_uint(nof_grid_ports),
                                    _p(grid)) == 0
    return grid


def r_prb_indices(bwp_start, bwp_size, vrb_mask, interleaved):
    vm = np.ascontiguousarray(vrb_mask, dtype=np.uint8)
    pl = np.zeros(275, dtype=np.uint16)
    n = C.c_uint(0)
    assert ref().ref_prb_indices(C.c_uint(bwp_start), C.c_uint(bwp_size), _p(vm), int(interleaved), _p(pl), C.byref(n)) == 0
    return pl[:n.value].copy()


def pdsch_nof_re(prb_list, start, nof, dmrs_mask, type2, cdm, bwp_start, bwp_size, reserved):
    """Data REs per layer of a PDSCH allocation (brute force, mirrors pdsch_modulator_impl.cpp:102-160)."""
    n = 0
    for sy in range(start, start + nof):
        for rb in prb_list:
            for k in range(12):
                ex = bool(dmrs_mask[sy]) and (((k % 6) < 2 * cdm) if type2 else ((k % 2) < cdm)) and bwp_start <= rb < bwp_start + bwp_size
                for (pm, rm, sm) in reserved:
                    ex = ex or (bool(pm[rb]) and bool((rm >> k) & 1) and bool((sm >> sy) & 1))
                n += 0 if ex else 1
    return n


# ---------------------------------------------------------------------- rx_softbuffer_pool traces
POOL_RESERVE, POOL_DROP, POOL_RELEASE, POOL_RUN_SLOT = 0, 1, 2, 3


def pool_trace(seed, n_ops, nof_handles=6, nof_rnti=3, nof_harq=2, max_cbs=7, start_slot=20400, period=20480):
    """Random caller behaviour against a softbuffer pool: columns (op, slot, rnti, harq_id, nof_codeblocks, handle). A handle is
    one scope holding a unique_rx_softbuffer. Slots advance and wrap around the system-frame period."""
    rng = np.random.default_rng(seed)
    ops = np.zeros((n_ops, 6), np.int64)
    slot = start_slot
    for i in range(n_ops):
        op = int(rng.choice([POOL_RESERVE, POOL_DROP, POOL_RELEASE, POOL_RUN_SLOT], p=[0.4, 0.25, 0.15, 0.2]))
        if op == POOL_RUN_SLOT:
            slot = (slot + int(rng.integers(1, 6))) % period
        ops[i] = (op, slot, 0x4600 * int(rng.integers(0, nof_rnti)), int(rng.integers(0, nof_harq)), int(rng.integers(1, max_cbs + 1)),
                  int(rng.integers(0, nof_handles)))
    return ops


def r_pool_run(ops, max_softbuffers, max_nof_codeblocks, expire_timeout_slots, numerology=1):
    """The reference pool driven by a trace -> per op (softbuffer ordinal or -1, nof_codeblocks) for reservations, (0, 0) otherwise."""
    L = ref()
    L.ref_pool_create.restype = vp
    h = vp(L.ref_pool_create(66 * 384, max_softbuffers, max_nof_codeblocks, expire_timeout_slots, numerology))
    out = np.zeros((len(ops), 2), np.int64)
    try:
        for i, (op, slot, rnti, harq, ncb, hd) in enumerate(ops.tolist()):
            if op == POOL_RESERVE:
                n = C.c_uint(0)
                o = L.ref_pool_reserve(h, slot, rnti, harq, ncb, hd, C.byref(n))
                out[i] = (o, n.value if o >= 0 else 0)
            elif op == POOL_DROP:
                L.ref_pool_drop(h, hd)
            elif op == POOL_RELEASE:
                L.ref_pool_release(h, hd)
            else:
                L.ref_pool_run_slot(h, slot)
    finally:
        L.ref_pool_destroy(h)
    return out


# ---------------------------------------------------------------------- Open Fronthaul IQ (de)compression
OFH_NONE, OFH_BFP = 0, 1  # srsran::ofh::compression_type


def ofh_payload_bytes(nof_prb, w, comp=OFH_BFP):
    return nof_prb * ((1 if comp == OFH_BFP else 0) + 3 * w)


def o_ofh_iq_decompress(payload, nof_prb, w, simd=True, comp=OFH_BFP):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    assert payload.size >= ofh_payload_bytes(nof_prb, w, comp)
    out = np.zeros(nof_prb * 12, dtype=np.complex64)
    oracle().orc_ofh_iq_decompress(int(comp), _p(payload), C.c_uint(nof_prb), C.c_uint(w), int(simd), _p(out))
    return out


def o_ofh_iq_compress(x, nof_prb, w, iq_scaling=1.0, comp=OFH_BFP):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    assert x.size == nof_prb * 12
    payload = np.zeros(ofh_payload_bytes(nof_prb, w, comp), dtype=np.uint8)
    oracle().orc_ofh_iq_compress(int(comp), _p(x), C.c_uint(nof_prb), C.c_uint(w), C.c_float(iq_scaling), _p(payload))
    return payload


def r_ofh_iq_decompress(payload, nof_prb, w, impl="avx2", comp=OFH_BFP):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(nof_prb * 12, dtype=np.complex64)
    assert ref().ref_ofh_iq_decompress(int(comp), impl.encode(), _p(payload), C.c_uint(nof_prb), C.c_uint(w), _p(out)) == 0
    return out


def r_ofh_iq_compress(x, nof_prb, w, iq_scaling=1.0, impl="avx2", comp=OFH_BFP):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    payload = np.zeros(ofh_payload_bytes(nof_prb, w, comp), dtype=np.uint8)
    assert ref().ref_ofh_iq_compress(int(comp), impl.encode(), _p(x), C.c_uint(nof_prb), C.c_uint(w), C.c_float(iq_scaling), _p(payload)) == 0
    return payload


# ---------------------------------------------------------------------- PDCCH processor
def o_pdcch_process(slot_in_frame, rnti, n_id_data, n_rnti, n_id_dmrs, ref_point, data_dB, dmrs_dB, payload, AL, start, duration, rb_mask, grid):
    """grid: complex64 [14][nsc] of one port, updated in place. Returns the number of data REs."""
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    rb = np.ascontiguousarray(rb_mask, dtype=np.uint8)
    assert grid.dtype == np.complex64 and grid.flags.c_contiguous and grid.shape == (14, rb.size * 12)
    return oracle().orc_pdcch_process(C.c_uint(slot_in_frame), C.c_uint(rnti), C.c_uint(n_id_data), C.c_uint(n_rnti), C.c_uint(n_id_dmrs), C.c_uint(ref_point),
                                      C.c_float(data_dB), C.c_float(dmrs_dB), _p(pl), C.c_uint(pl.size), C.c_uint(AL), C.c_uint(start), C.c_uint(duration), _p(rb),
                                      C.c_uint(rb.size), _p(grid))


def r_pdcch_process(mapping, bwp_start, bwp_size, start, duration, freq_resources, reg_bundle, interleaver, shift, numerology, slot_index, rnti, n_id_dmrs,
                    n_id_data, n_rnti, cce_index, AL, dmrs_dB, data_dB, payload, nprb_grid):
    """The reference processor. Returns (grid [14][nsc], rb_mask bytes [nprb_grid] from its CCE-to-PRB mapping)."""
    fr = np.ascontiguousarray(freq_resources, dtype=np.uint8)
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    grid = np.zeros((14, nprb_grid * 12), dtype=np.complex64)
    rb = np.zeros(nprb_grid, dtype=np.uint8)
    rc = ref().ref_pdcch_process(int(mapping), C.c_uint(bwp_start), C.c_uint(bwp_size), C.c_uint(start), C.c_uint(duration), _p(fr), C.c_uint(fr.size),
                                 C.c_uint(reg_bundle), C.c_uint(interleaver), C.c_uint(shift), C.c_uint(numerology), C.c_uint(slot_index), C.c_uint(rnti),
                                 C.c_uint(n_id_dmrs), C.c_uint(n_id_data), C.c_uint(n_rnti), C.c_uint(cce_index), C.c_uint(AL), C.c_float(dmrs_dB),
                                 C.c_float(data_dB), _p(pl), C.c_uint(pl.size), C.c_uint(nprb_grid), _p(grid), _p(rb))
    assert rc == 0, rc
    return grid, rb


def pdcch_cases(rng, n):
    """Random valid PDCCH configurations: (mapping, bwp_start, bwp_size, start, duration, freq_resources, reg_bundle, interleaver, shift, cce_index, AL)."""
    out = []
    while len(out) < n:
        mapping = int(rng.integers(0, 3))
        duration = int(rng.integers(1, 4))
        AL = int(rng.choice([1, 2, 4, 8, 16]))
        if mapping == 0:  # CORESET 0: 24, 48 or 96 PRBs at the start of the BWP, interleaved with bundle 6, interleaver 2
            size = int(rng.choice([24, 48, 96]))
            bwp_start, bwp_size = int(rng.integers(0, 20)), size
            nfr = size // 6
            fr = np.ones(nfr, np.uint8)
            reg_bundle, interleaver, shift = 6, 2, int(rng.integers(0, 1008))
        else:
            nfr = int(rng.integers(2, 17))
            fr = (rng.uniform(size=nfr) < 0.8).astype(np.uint8)
            fr[0] = 1
            bwp_start, bwp_size = int(rng.integers(0, 30)), nfr * 6 + int(rng.integers(0, 6))
            reg_bundle = int(rng.choice([2, 6] if duration < 3 else [3, 6])) if mapping == 2 else 6
            interleaver, shift = int(rng.choice([2, 3, 6])), int(rng.integers(0, 275))
        nreg = int(fr.sum()) * 6 * duration
        ncce = nreg // 6
        if ncce < AL:
            continue
        if mapping == 2 and (nreg % (reg_bundle * interleaver)) != 0:
            continue
        if mapping == 0 and (nreg % (6 * 2)) != 0:
            continue
        cce_index = AL * int(rng.integers(0, ncce // AL))
        out.append((mapping, bwp_start, bwp_size, int(rng.integers(0, 3)), duration, fr, reg_bundle, interleaver, shift, cce_index, AL))
    return out


# ---------------------------------------------------------------------- SS/PBCH block processor
def o_ssb_process(N_id, ssb_idx, L_max, hrf, sfn, k_ssb, payload, k0, l0, beta_pss_dB, nprb_grid, grid):
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    assert pl.size == 32 and grid.dtype == np.complex64 and grid.shape == (14, nprb_grid * 12) and grid.flags.c_contiguous
    return oracle().orc_ssb_process(C.c_uint(N_id), C.c_uint(ssb_idx), C.c_uint(L_max), int(hrf), C.c_uint(sfn), C.c_uint(k_ssb), _p(pl), C.c_uint(k0), C.c_uint(l0),
                                    C.c_float(beta_pss_dB), C.c_uint(nprb_grid), _p(grid))


def r_ssb_process(numerology, sfn, slot_in_frame, N_id, beta_pss, ssb_idx, L_max, common_scs_khz, subcarrier_offset, offset_to_pointA, pattern_case, payload,
                  nprb_grid):
    """The reference SSB processor. Returns (rc, grid [14][nsc], l_start, k_start); rc -2 when the slot does not carry the block."""
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    grid = np.zeros((14, nprb_grid * 12), dtype=np.complex64)
    l0, k0 = C.c_uint(0), C.c_uint(0)
    rc = ref().ref_ssb_process(C.c_uint(numerology), C.c_uint(sfn), C.c_uint(slot_in_frame), C.c_uint(N_id), C.c_float(beta_pss), C.c_uint(ssb_idx), C.c_uint(L_max),
                               C.c_uint(common_scs_khz), C.c_uint(subcarrier_offset), C.c_uint(offset_to_pointA), int(pattern_case), _p(pl), C.c_uint(nprb_grid),
                               _p(grid), C.byref(l0), C.byref(k0))
    return rc, grid, l0.value, k0.value


def ssb_cases(rng, n):
    """Random valid SS/PBCH configurations (FR1 pattern cases A, B, C): (numerology, sfn, slot, N_id, beta, ssb_idx, L_max, scs, k_ssb, offset, case)."""
    out = []
    first = {0: [2, 8, 16, 22, 30, 36, 44, 50], 1: [4, 8, 16, 20, 32, 36, 44, 48], 2: [2, 8, 16, 22, 30, 36, 44, 50]}  # TS 38.213 4.1, L_max <= 8
    while len(out) < n:
        case = int(rng.integers(0, 3))
        mu = 0 if case == 0 else 1
        L_max = int(rng.choice([4, 8]))
        ssb_idx = int(rng.integers(0, L_max))
        slot_hrf = first[case][ssb_idx] // 14
        hrf = int(rng.integers(0, 2))
        slots_per_hrf = 5 << mu
        out.append((mu, int(rng.integers(0, 1024)), hrf * slots_per_hrf + slot_hrf, int(rng.integers(0, 1008)), float(rng.choice([0.0, 3.0, -3.0])), ssb_idx, L_max,
                    15 if mu == 0 else 30, int(rng.integers(0, 12 if mu == 0 else 24)) & ~(0 if mu == 0 else 1), int(rng.integers(0, 40)) * (1 if mu == 0 else 2), case))
    return out


# ---------------------------------------------------------------------- NZP-CSI-RS generator
# (row, nof_ports, nof k_ref, cdm, allowed densities [0 even, 1 odd, 2 one, 3 three], k alignment, uses l1)
CSI_RS_ROWS = [(1, 1, 1, 0, [3], 1, 0), (2, 1, 1, 0, [0, 1, 2], 1, 0), (3, 2, 1, 1, [0, 1, 2], 2, 0), (4, 4, 1, 1, [2], 4, 0), (5, 4, 1, 1, [2], 2, 0),
               (6, 8, 4, 1, [2], 2, 0), (7, 8, 2, 1, [2], 2, 0), (8, 8, 2, 2, [2], 2, 0)]


def csi_rs_cases(rng, n):
    out = []
    for i in range(n):
        row, nports, nk, cdm, dens, align, _ = CSI_RS_ROWS[i % len(CSI_RS_ROWS)]
        if row == 1:
            k = [int(rng.integers(0, 4))]
        elif row == 4:
            k = [int(rng.choice([0, 4, 8]))]
        else:
            k = sorted(int(x) for x in rng.choice(np.arange(0, 11, 2), nk, replace=False))
        start_rb, nof_rb = int(rng.integers(0, 20)), int(rng.integers(4, 60))
        out.append((row, nports, k, cdm, int(rng.choice(dens)), start_rb, nof_rb, int(rng.integers(0, 12)), int(rng.integers(0, 20)), int(rng.integers(0, 1024)),
                    float(rng.choice([1.0, 0.5, 1.4125]))))
    return out


def r_csi_rs_map(numerology, slot_index, start_rb, nof_rb, row, k_ref, l0, l1, cdm, density, scr_id, amplitude, nof_ports, nprb_grid):
    kr = (C.c_uint * len(k_ref))(*k_ref)
    grid = np.zeros((nof_ports, 14, nprb_grid * 12), dtype=np.complex64)
    bes = (C.c_uint * 3)()
    rm, sm = np.zeros(16, np.uint16), np.zeros(16, np.uint16)
    assert ref().ref_csi_rs_map(C.c_uint(numerology), C.c_uint(slot_index), C.c_uint(start_rb), C.c_uint(nof_rb), C.c_uint(row), kr, C.c_uint(len(k_ref)), C.c_uint(l0),
                                C.c_uint(l1), C.c_uint(cdm), C.c_uint(density), C.c_uint(scr_id), C.c_float(amplitude), C.c_uint(nof_ports), C.c_uint(nprb_grid),
                                _p(grid), bes, _p(rm), _p(sm)) == 0
    return grid, (bes[0], bes[1], bes[2]), rm, sm


def o_csi_rs_map(slot, scr_id, amplitude, start_rb, nof_rb, bes, row, cdm, density, ports, re_mask, symbol_mask, nprb_grid, grid):
    pt = np.ascontiguousarray(ports, dtype=np.uint8)
    rm, sm = np.ascontiguousarray(re_mask, dtype=np.uint16), np.ascontiguousarray(symbol_mask, dtype=np.uint16)
    assert grid.dtype == np.complex64 and grid.flags.c_contiguous
    return oracle().orc_csi_rs_map(C.c_uint(slot), C.c_uint(scr_id), C.c_float(amplitude), C.c_uint(start_rb), C.c_uint(nof_rb), C.c_uint(bes[0]), C.c_uint(bes[1]),
                                   C.c_uint(bes[2]), C.c_uint(row), C.c_uint(cdm), C.c_uint(density), C.c_uint(pt.size), _p(pt), _p(rm), _p(sm), C.c_uint(nprb_grid),
                                   _p(grid))


# ---------------------------------------------------------------------------------------------- UL-SCH demultiplexing (UCI on PUSCH)
def o_ulsch_demultiplex(mod, nof_layers, nof_prb, start, nof, G_rvd, dmrs_type, dmrs_mask, cdm, G, O, llr=None):
    """G = (G_ack, G_csi1, G_csi2) encoded bits, O = (O_ack, O_csi1, O_csi2) information bits. Returns (n_in, n_sch, streams or None,
    placeholder RE indices) or None where the reference asserts (fields that do not fit)."""
    cap = nof_prb * 12 * 14 * mod * nof_layers
    sch, ack, c1, c2 = np.zeros(cap, np.int8), np.zeros(max(G[0], 1), np.int8), np.zeros(max(G[1], 1), np.int8), np.zeros(max(G[2], 1), np.int8)
    ph = np.zeros(nof_prb * 12 * 14, np.uint16)
    n_sch, n_ph = C.c_uint(0), C.c_uint(0)
    lin = None if llr is None else np.ascontiguousarray(llr, dtype=np.int8)
    n = oracle().orc_ulsch_demultiplex(int(mod), C.c_uint(nof_layers), C.c_uint(nof_prb), C.c_uint(start), C.c_uint(nof), C.c_uint(G_rvd), int(dmrs_type),
                                       C.c_uint(dmrs_mask), C.c_uint(cdm), C.c_uint(G[0]), C.c_uint(G[1]), C.c_uint(G[2]), C.c_uint(O[0]), C.c_uint(O[1]),
                                       C.c_uint(O[2]), _p(lin) if lin is not None else None, _p(sch), _p(ack), _p(c1), _p(c2), C.byref(n_sch), _p(ph),
                                       C.byref(n_ph))
    if n < 0:
        return None
    streams = None if llr is None else (sch[:n_sch.value].copy(), ack[:G[0]].copy(), c1[:G[1]].copy(), c2[:G[2]].copy())
    return n, n_sch.value, streams, ph[:n_ph.value].copy()


def r_ulsch_demultiplex(mod, nof_layers, nof_prb, start, nof, G_rvd, dmrs_type, dmrs_mask, cdm, G, O, llr, n_sch):
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    sch, ack, c1, c2 = np.zeros(max(n_sch, 1), np.int8), np.zeros(max(G[0], 1), np.int8), np.zeros(max(G[1], 1), np.int8), np.zeros(max(G[2], 1), np.int8)
    ph = np.zeros(nof_prb * 12 * 14, np.uint16)
    n_ph = C.c_uint(0)
    rc = ref().ref_ulsch_demultiplex(int(mod), C.c_uint(nof_layers), C.c_uint(nof_prb), C.c_uint(start), C.c_uint(nof), C.c_uint(G_rvd), int(dmrs_type == 2),
                                     C.c_uint(dmrs_mask), C.c_uint(cdm), C.c_uint(G[0]), C.c_uint(G[1]), C.c_uint(G[2]), C.c_uint(O[0]), C.c_uint(O[1]),
                                     C.c_uint(O[2]), _p(llr), C.c_uint(llr.size), _p(sch), C.c_uint(n_sch), _p(ack), _p(c1), _p(c2), _p(ph), C.byref(n_ph))
    assert rc == 0
    return (sch[:n_sch].copy(), ack[:G[0]].copy(), c1[:G[1]].copy(), c2[:G[2]].copy()), ph[:n_ph.value].copy()


def ulsch_cases(rng, n):
    """Random but valid UCI-on-PUSCH configurations: (mod, layers, nprb, start, nof, G_rvd, dmrs_type, dmrs_mask, cdm, G, O)."""
    out = []
    while len(out) < n:
        mod = int(rng.choice([1, 2, 4, 6, 8]))
        nprb = int(rng.integers(1, 60))
        start = int(rng.integers(0, 3))
        nof = int(rng.integers(6, 15 - start))
        dm = 0
        for l in sorted(set(int(x) for x in rng.integers(start, start + nof, int(rng.integers(1, 4))))):
            dm |= 1 << l
        if dm == ((1 << nof) - 1) << start:
            continue
        dtype, cdm = (1, int(rng.integers(1, 3))) if rng.random() < 0.7 else (2, int(rng.integers(1, 4)))
        bpr = mod
        re_sym = nprb * 12
        O_ack = int(rng.choice([0, 1, 2, 5, 11]))
        G_ack = 0 if O_ack == 0 else bpr * int(rng.integers(1, max(2, re_sym // 2)))
        G_rvd = 0
        if O_ack <= 2 and rng.random() < 0.6:
            G_rvd = bpr * int(rng.integers(max(1, G_ack // bpr), max(2, G_ack // bpr + re_sym // 2)))
        O_c1 = int(rng.choice([0, 0, 1, 4, 11]))
        G_c1 = 0 if O_c1 == 0 else bpr * int(rng.integers(1, max(2, re_sym)))
        O_c2 = int(rng.choice([0, 0, 0, 1, 7]))
        G_c2 = 0 if O_c2 == 0 else bpr * int(rng.integers(1, max(2, re_sym)))
        case = (mod, 1, nprb, start, nof, G_rvd, dtype, dm, cdm, (G_ack, G_c1, G_c2), (O_ack, O_c1, O_c2))
        if o_ulsch_demultiplex(*case) is None:
            continue
        out.append(case)
    return out


# ---------------------------------------------------------------------------------------------- zero-forcing equalizer on its own
def _equalize(fn, ch_symbols, ch_estimates, noise_var, tx_scaling):
    """ch_symbols complex64 [ports][nof_re]; ch_estimates complex64 [layers][ports][nof_re] -> (eq complex64 [layers][nof_re], noise vars)."""
    y = np.ascontiguousarray(ch_symbols, dtype=np.complex64)
    h = np.ascontiguousarray(ch_estimates, dtype=np.complex64)
    nl, npt, nre = h.shape
    assert y.shape == (npt, nre)
    z = np.zeros((nl, nre), np.complex64)
    nv = np.zeros((nl, nre), np.float32)
    rc = fn(C.c_uint(nre), C.c_uint(npt), C.c_uint(nl), _p(y), _p(h), C.c_float(noise_var), C.c_float(tx_scaling), _p(z), _p(nv))
    return (z, nv) if rc == 0 else None


def o_channel_equalize(ch_symbols, ch_estimates, noise_var, tx_scaling):
    return _equalize(oracle().orc_channel_equalize, ch_symbols, ch_estimates, noise_var, tx_scaling)


def r_channel_equalize(ch_symbols, ch_estimates, noise_var, tx_scaling):
    return _equalize(ref().ref_channel_equalize, ch_symbols, ch_estimates, noise_var, tx_scaling)


def equalizer_case(rng, nre, npt, nl, snr_db=20.0, dead=()):
    """A random flat-ish channel, QAM-like symbols through it plus noise; `dead` lists resource elements whose estimates are zeroed."""
    h = (rng.standard_normal((nl, npt, nre)) + 1j * rng.standard_normal((nl, npt, nre))).astype(np.complex64) * np.float32(0.7)
    x = (rng.integers(0, 4, (nl, nre)) * 2 - 3 + 1j * (rng.integers(0, 4, (nl, nre)) * 2 - 3)).astype(np.complex64) / np.float32(np.sqrt(10))
    nvar = float(10 ** (-snr_db / 10))
    y = np.einsum("lpr,lr->pr", h, x).astype(np.complex64)
    y = y + (np.sqrt(nvar / 2) * (rng.standard_normal((npt, nre)) + 1j * rng.standard_normal((npt, nre)))).astype(np.complex64)
    for i in dead:
        h[:, :, i] = 0
    return y.astype(np.complex64), h, nvar, x
