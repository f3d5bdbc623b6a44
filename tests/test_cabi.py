"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/miphy.h declares.
No compute call is made (there is no GPU here); struct layouts in the Python binding are checked against the header."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "miphy.h")
LIB = os.path.join(ROOT, "srsran_project_23.5_amd", "libmiphy.so")


def declared_functions():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(miphy_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_hot_path():
    fns = declared_functions()
    for must in ("miphy_create", "miphy_ldpc_decode_batch", "miphy_ldpc_encode_batch", "miphy_ldpc_rate_match_batch",
                 "miphy_ldpc_rate_dematch_batch", "miphy_crc_batch", "miphy_dft_batch", "miphy_ofdm_demodulate_slots",
                 "miphy_ofdm_modulate_slots", "miphy_dmrs_pusch_estimate_batch", "miphy_polar_encode_batch",
                 "miphy_polar_decode_batch", "miphy_pdcch_encode_batch"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    exported = set(re.findall(r" T (miphy_[a-z0-9_]+)", out))
    missing = [f for f in declared_functions() if f not in exported]
    assert not missing, missing
    # device code for gfx950 is embedded
    blob = open(LIB, "rb").read()
    assert b"gfx950" in blob


def test_library_loads_and_reports_version():
    import torch  # noqa: F401  (brings the HIP runtime the library links against)
    lib = ctypes.CDLL(LIB)
    lib.miphy_version.restype = ctypes.c_int
    assert lib.miphy_version() == 1
    lib.miphy_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.miphy_last_error(), bytes)


def test_host_only_entry_points():
    """Entry points that need no device: OFDM slot size (ofdm_slot_demodulator::get_slot_size) and polar code parameters."""
    import miphy
    cfg = miphy.OfdmConfig(1, 273, 4096, 144, 1.0, 0.0, 3.5e9)
    assert cfg.slot_size(0) == 61440 and cfg.slot_size(1) == 61440
    cfg = miphy.OfdmConfig(1, 106, 2048, 72, 1.0, 0.0, 3.5e9)
    assert cfg.slot_size(0) == 30720
    assert miphy.OfdmConfig(0, 52, 1024, 0, 1.0, 0.0, 2.6e9).slot_size(0) == 15360
    assert miphy.PolarCode(56, 864, 9, 0).info() == (9, 512, 0)   # PBCH
    assert miphy.PolarCode(36, 108, 9, 0).info()[1] == 128
    assert miphy.PolarCode(20, 100, 10, 1).info()[2] == 3           # parity-check bits for K <= 25
    with pytest.raises(RuntimeError):
        miphy.PolarCode(30, 100, 9, 0).info()


def test_binding_struct_sizes_match_header():
    """Compile a tiny C program against include/miphy.h and compare sizeof() with the numpy/ctypes mirrors."""
    import miphy
    import tempfile
    src = r'''
#include "miphy.h"
#include <stdio.h>
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(miphy_ldpc_dec_desc), sizeof(miphy_ldpc_rdm_desc), sizeof(miphy_ldpc_enc_desc),
         sizeof(miphy_crc_desc), sizeof(miphy_ofdm_job), sizeof(miphy_ofdm_config), sizeof(miphy_pusch_chest_job), sizeof(miphy_polar_code));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(miphy_pusch_tb_desc), sizeof(miphy_pusch_result), sizeof(miphy_pdsch_tb_desc),
         sizeof(miphy_sch_segmentation), sizeof(miphy_pusch_demod_job), sizeof(miphy_re_pattern), sizeof(miphy_pdsch_mod_job), sizeof(miphy_dmrs_pdsch_job));
  printf("%zu %zu %zu %zu\n", sizeof(miphy_pusch_pdu), sizeof(miphy_pdsch_pdu), sizeof(miphy_harq_pool_config), sizeof(miphy_harq_buffer_info));
  printf("%zu %zu\n", sizeof(miphy_ofh_iq_job), sizeof(miphy_pdcch_pdu));
  printf("%zu %zu\n", sizeof(miphy_ssb_pdu), sizeof(miphy_csi_rs_job));
  printf("%zu %zu %zu\n", sizeof(miphy_ulsch_demux_job), sizeof(miphy_pusch_uci), sizeof(miphy_equalizer_job));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe], text=True).split()]
    mine = [miphy.LdpcDecDesc.itemsize, miphy.LdpcRdmDesc.itemsize, miphy.LdpcEncDesc.itemsize, miphy.CrcDesc.itemsize,
            miphy.OfdmJob.itemsize, ctypes.sizeof(miphy.OfdmConfig), miphy.PuschChestJob.itemsize, ctypes.sizeof(miphy.PolarCode),
            miphy.PuschTbDesc.itemsize, miphy.PuschResult.itemsize, miphy.PdschTbDesc.itemsize, ctypes.sizeof(miphy.binding.SchSegmentation),
            miphy.PuschDemodJob.itemsize, miphy.RePattern.itemsize, miphy.PdschModJob.itemsize, miphy.DmrsPdschJob.itemsize,
            miphy.PuschPdu.itemsize, miphy.PdschPdu.itemsize, ctypes.sizeof(miphy.HarqPoolConfig), ctypes.sizeof(miphy.HarqBufferInfo),
            miphy.OfhIqJob.itemsize, miphy.PdcchPdu.itemsize, miphy.SsbPdu.itemsize, miphy.CsiRsJob.itemsize,
            miphy.UlschDemuxJob.itemsize, miphy.PuschUci.itemsize, miphy.EqualizerJob.itemsize]
    assert sizes == mine, (sizes, mine)


def test_host_segmentation_matches_oracle():
    """Host-side a6 logic (ldpc.h:128-207, ldpc_segmenter_impl.cpp:104-141) against the oracle for TS 38.214-like TB sizes."""
    import miphy
    import oracle_lib as O
    for bg, tbs_bits in ((2, 24), (2, 320), (2, 3848), (1, 3824), (1, 3848), (1, 8424), (1, 8448), (1, 42016), (1, 83976), (1, 319784),
                         (2, 9984), (1, 52 * 8424 - 24), (2, 3840 - 16), (2, 3840)):
        s = miphy.sch_segmentation(tbs_bits // 8, bg)
        o = O.o_segmentation(tbs_bits // 8 * 8, bg, 2, 1, 2 * 52 * 156)
        for k in ("nof_cbs", "Z", "K", "N", "nof_filler_bits", "nof_tb_crc_bits", "nof_cb_crc_bits", "cb_info_bits", "zero_pad"):
            assert getattr(s, k) == getattr(o, k), (bg, tbs_bits, k)


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU: no oracle, no CPU path."""
    import torch
    import miphy
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        miphy.Context(0)
    # and nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, "srsran_project_23.5_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "phy_oracle" not in txt and "oracle_lib" not in txt and "libref_capi" not in txt, os.path.join(dp, fn)
