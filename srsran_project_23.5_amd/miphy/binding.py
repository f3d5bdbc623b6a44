"""ctypes binding of include/miphy.h.  No torch types cross the C ABI: tensors are passed as raw device pointers."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIPHY_LIBRARY: another build of the same library (tools/ab_bench.py compares kernel variants this way)
lib_path = os.environ.get("MIPHY_LIBRARY") or os.path.join(os.path.dirname(_HERE), "libmiphy.so")

CRC24A, CRC24B, CRC24C, CRC16, CRC11 = range(5)
CRC_NONE = 255


class LibraryNotBuilt(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libmiphy.so (torch must already be imported so that its HIP runtime is the one in the process)."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise LibraryNotBuilt("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C "
                                  "srsran_project_23.5_amd`. There is no CPU fallback." % lib_path)
        import torch  # noqa: F401  (loads libamdhip64 first; the C ABI itself has no torch dependency)
        l = C.CDLL(lib_path)
        l.miphy_last_error.restype = C.c_char_p
        l.miphy_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.miphy_destroy.argtypes = [C.c_void_p]
        l.miphy_ldpc_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("miphy_ldpc_rate_match_batch", "miphy_ldpc_encode_batch"):
            getattr(l, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_ldpc_rate_dematch_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]
        l.miphy_dft_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("miphy_ofdm_demodulate_slots", "miphy_ofdm_modulate_slots"):
            getattr(l, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_ofdm_slot_size.argtypes = [C.c_void_p, C.c_uint32]
        l.miphy_ofdm_slot_size.restype = C.c_uint32
        l.miphy_dmrs_pusch_estimate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p]
        l.miphy_port_channel_estimate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_void_p]
        l.miphy_polar_code_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_polar_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_polar_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_pdcch_encode_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_pusch_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 7
        l.miphy_pdsch_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_pusch_decode_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        l.miphy_pusch_decode_plan_run.argtypes = [C.c_void_p] * 8
        l.miphy_pusch_decode_plan_destroy.argtypes = [C.c_void_p]
        l.miphy_pusch_decode_plan_destroy.restype = None
        l.miphy_pusch_decode_plan_enable_timing.argtypes = [C.c_void_p, C.c_uint32]
        l.miphy_pusch_decode_plan_read_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_pusch_decode_plan_info.argtypes = [C.c_void_p, C.c_void_p]
        l.miphy_pusch_decode_plan_nof_launches.argtypes = [C.c_void_p]
        l.miphy_pusch_decode_plan_nof_launches.restype = C.c_uint32
        l.miphy_sch_segmentation_info.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        l.miphy_pdsch_process_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        l.miphy_pdsch_process_plan_run.argtypes = [C.c_void_p] * 4
        l.miphy_pdsch_process_plan_destroy.argtypes = [C.c_void_p]
        l.miphy_pdsch_process_plan_destroy.restype = None
        l.miphy_ldpc_decode_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        l.miphy_ldpc_decode_plan_run.argtypes = [C.c_void_p] * 5
        l.miphy_ldpc_decode_plan_nof_launches.argtypes = [C.c_void_p]
        l.miphy_ldpc_decode_plan_nof_launches.restype = C.c_uint32
        l.miphy_ldpc_decode_plan_destroy.argtypes = [C.c_void_p]
        l.miphy_ldpc_decode_plan_destroy.restype = None
        l.miphy_debug_ldpc_kernels_used.argtypes = [C.c_int]
        l.miphy_debug_ldpc_kernels_used.restype = C.c_uint32
        l.miphy_polar_decode_list_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_pbch_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        l.miphy_crc_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_pusch_demodulate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]
        l.miphy_pdsch_modulate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_dmrs_pdsch_map_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        l.miphy_pdsch_mod_nof_re.argtypes = [C.c_void_p]
        l.miphy_pdsch_mod_nof_re.restype = C.c_uint32
        l.miphy_pusch_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 8
        l.miphy_pusch_demod_nof_llr.argtypes = [C.c_void_p]
        l.miphy_pdsch_pdu_nof_re.argtypes = [C.c_void_p]
        l.miphy_pdsch_pdu_nof_re.restype = C.c_uint32
        l.miphy_pdsch_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_ofh_iq_decompress_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        l.miphy_ofh_iq_compress_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        l.miphy_pdcch_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_ssb_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        l.miphy_csi_rs_map_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        l.miphy_harq_pool_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        l.miphy_harq_pool_destroy.argtypes = [C.c_void_p]
        l.miphy_harq_pool_destroy.restype = None
        l.miphy_harq_pool_reserve.argtypes = [C.c_void_p] + [C.c_uint32] * 4 + [C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        for name in ("miphy_harq_pool_lock", "miphy_harq_pool_unlock", "miphy_harq_pool_release"):
            getattr(l, name).argtypes = [C.c_void_p, C.c_int32]
        l.miphy_harq_pool_run_slot.argtypes = [C.c_void_p, C.c_uint32]
        l.miphy_harq_pool_info.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        l.miphy_harq_pool_free_codeblocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        l.miphy_harq_pool_arrays.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 3
        l.miphy_pusch_demod_nof_llr.restype = C.c_uint32
        l.miphy_pusch_process_batch_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 10
        l.miphy_pusch_demodulate_batch_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32] + [C.c_void_p] * 7
        l.miphy_ulsch_demux_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_ulsch_placeholders.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        l.miphy_ulsch_demultiplex_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        l.miphy_channel_equalize_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32] + [C.c_void_p] * 5
        l.miphy_polar_block_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("miphy_ofdm_demodulate_symbols", "miphy_ofdm_modulate_symbols"):
            getattr(l, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.miphy_ofdm_symbol_size.argtypes = [C.c_void_p, C.c_uint32]
        l.miphy_ofdm_symbol_size.restype = C.c_uint32
        _lib = l
    return _lib


# Mirrors miphy_ldpc_dec_desc.
LdpcDecDesc = np.dtype([("bg", np.uint8), ("crc_poly", np.uint8), ("Z", np.uint16), ("max_iter", np.uint16),
                        ("nof_filler_bits", np.uint16), ("in_len", np.uint32), ("flags", np.uint32),
                        ("llr_offset", np.uint64), ("out_offset", np.uint64)], align=True)
assert LdpcDecDesc.itemsize == 32
# Mirrors miphy_ldpc_rdm_desc.
LdpcRdmDesc = np.dtype([("bg", np.uint8), ("rv", np.uint8), ("mod", np.uint8), ("new_data", np.uint8), ("Z", np.uint16),
                        ("nof_filler_bits", np.uint16), ("Nref", np.uint32), ("E", np.uint32), ("in_offset", np.uint64),
                        ("out_offset", np.uint64)], align=True)
assert LdpcRdmDesc.itemsize == 32
# Mirrors miphy_ldpc_enc_desc.
LdpcEncDesc = np.dtype([("bg", np.uint8), ("reserved0", np.uint8), ("Z", np.uint16), ("out_len", np.uint32),
                        ("in_offset", np.uint64), ("out_offset", np.uint64)], align=True)
assert LdpcEncDesc.itemsize == 24
# Mirrors miphy_ofdm_job.
OfdmJob = np.dtype([("samples_offset", np.uint64), ("grid_offset", np.uint64), ("slot_index", np.uint32), ("grid_empty", np.uint32)],
                   align=True)
assert OfdmJob.itemsize == 24


class OfdmConfig(C.Structure):
    """Mirrors miphy_ofdm_config (= srsran::ofdm_demodulator_configuration / ofdm_modulator_configuration)."""
    _fields_ = [("numerology", C.c_uint32), ("bw_rb", C.c_uint32), ("dft_size", C.c_uint32),
                ("nof_samples_window_offset", C.c_uint32), ("scale", C.c_float), ("reserved", C.c_float),
                ("center_freq_hz", C.c_double)]

    def slot_size(self, slot_index):
        return int(lib().miphy_ofdm_slot_size(C.byref(self), slot_index))

    def symbol_size(self, symbol_index):
        """Samples (cyclic prefix included) of OFDM symbol `symbol_index` of the subframe."""
        return int(lib().miphy_ofdm_symbol_size(C.byref(self), symbol_index))


def ofdm_symbol_size(cfg, symbol_index):
    return cfg.symbol_size(symbol_index)


# Mirrors miphy_pusch_chest_job.
PuschChestJob = np.dtype([("numerology", np.uint32), ("slot_in_frame", np.uint32), ("scrambling_id", np.uint32), ("scaling", np.float32),
                          ("n_scid", np.uint8), ("nof_tx_layers", np.uint8), ("nof_rx_ports", np.uint8), ("first_symbol", np.uint8),
                          ("nof_symbols", np.uint8), ("rx_ports", np.uint8, 4), ("ce_compact", np.uint8), ("reserved", np.uint8, 2), ("symbols_mask", np.uint16),
                          ("grid_nof_prb", np.uint16), ("rb_mask", np.uint64, 5), ("grid_offset", np.uint64), ("ce_offset", np.uint64),
                          ("scalars_offset", np.uint64), ("rb_mask2", np.uint64, 5), ("pilots_offset", np.uint64), ("hop_symbol", np.uint8),
                          ("re_odd_mask", np.uint8), ("reserved2", np.uint8, 6)], align=True)
assert PuschChestJob.itemsize == 152, PuschChestJob.itemsize
assert PuschChestJob.fields["rb_mask"][1] == 32 and PuschChestJob.fields["symbols_mask"][1] == 28


# Mirrors miphy_pusch_demod_job.
PuschDemodJob = np.dtype([("rnti", np.uint32), ("n_id", np.uint32), ("mod", np.uint8), ("nof_rx_ports", np.uint8), ("start_symbol", np.uint8),
                          ("nof_symbols", np.uint8), ("dmrs_type", np.uint8), ("nof_cdm_groups_without_data", np.uint8),
                          ("ce_nof_symbols", np.uint8), ("ce_compact", np.uint8), ("rx_ports", np.uint8, 4), ("dmrs_symbols_mask", np.uint16),
                          ("grid_nof_prb", np.uint16), ("nof_llr", np.uint32), ("rb_mask", np.uint64, 5), ("grid_offset", np.uint64),
                          ("ce_offset", np.uint64), ("scalars_offset", np.uint64), ("llr_offset", np.uint64), ("placeholders_offset", np.uint32),
                          ("nof_placeholders", np.uint32), ("evm_offset", np.uint64)], align=True)
assert PuschDemodJob.itemsize == 120 and PuschDemodJob.fields["rb_mask"][1] == 32, PuschDemodJob.itemsize


def pusch_demod_nof_llr(job):
    """Codeword length (data REs x bits per symbol) of one PuschDemodJob record (host computation in the library)."""
    a = np.ascontiguousarray(np.asarray(job, dtype=PuschDemodJob).reshape(1))
    return int(lib().miphy_pusch_demod_nof_llr(a.ctypes.data_as(C.c_void_p)))


# Mirrors miphy_re_pattern / miphy_pdsch_mod_job / miphy_dmrs_pdsch_job.
RePattern = np.dtype([("prb_mask", np.uint64, 5), ("re_mask", np.uint16), ("symbols", np.uint16), ("pad", np.uint32)], align=True)
PdschModJob = np.dtype([("rnti", np.uint32), ("n_id", np.uint32), ("scaling", np.float32), ("mod", np.uint8), ("port", np.uint8),
                        ("start_symbol", np.uint8), ("nof_symbols", np.uint8), ("dmrs_type", np.uint8), ("nof_cdm_groups_without_data", np.uint8),
                        ("nof_reserved", np.uint8), ("reserved0", np.uint8), ("dmrs_symbols_mask", np.uint16), ("grid_nof_prb", np.uint16),
                        ("bwp_start_rb", np.uint16), ("bwp_size_rb", np.uint16), ("nof_bits", np.uint32), ("rb_mask", np.uint64, 5),
                        ("reserved", RePattern, 4), ("cw_offset", np.uint64), ("grid_offset", np.uint64)], align=True)
assert RePattern.itemsize == 48 and PdschModJob.itemsize == 280 and PdschModJob.fields["rb_mask"][1] == 32, PdschModJob.itemsize
DmrsPdschJob = np.dtype([("slot_in_frame", np.uint32), ("reference_point_k_rb", np.uint32), ("scrambling_id", np.uint32), ("amplitude", np.float32),
                         ("dmrs_type", np.uint8), ("n_scid", np.uint8), ("nof_ports", np.uint8), ("reserved0", np.uint8), ("ports", np.uint8, 12),
                         ("symbols_mask", np.uint16), ("grid_nof_prb", np.uint16), ("pad", np.uint32), ("rb_mask", np.uint64, 5),
                         ("grid_offset", np.uint64)], align=True)
assert DmrsPdschJob.itemsize == 88 and DmrsPdschJob.fields["rb_mask"][1] == 40, DmrsPdschJob.itemsize


def pdsch_mod_nof_re(job):
    """Data REs of one PdschModJob record (host computation in the library)."""
    a = np.ascontiguousarray(np.asarray(job, dtype=PdschModJob).reshape(1))
    return int(lib().miphy_pdsch_mod_nof_re(a.ctypes.data_as(C.c_void_p)))


# Mirrors miphy_ofh_iq_job.
OfhIqJob = np.dtype([("payload_offset", np.uint64), ("grid_offset", np.uint64), ("nof_prb", np.uint32), ("data_width", np.uint16),
                     ("compression", np.uint16)], align=True)
OFH_COMPRESSION_NONE, OFH_COMPRESSION_BFP = 0, 1  # srsran::ofh::compression_type
assert OfhIqJob.itemsize == 24

# Mirrors miphy_pdsch_pdu.
PdschPdu = np.dtype([("slot_in_frame", np.uint32), ("rnti", np.uint32), ("n_id", np.uint32), ("dmrs_scrambling_id", np.uint32),
                     ("tbs_lbrm_bytes", np.uint32), ("tb_bytes", np.uint32), ("ratio_pdsch_dmrs_to_sss_dB", np.float32),
                     ("ratio_pdsch_data_to_sss_dB", np.float32), ("bg", np.uint8), ("rv", np.uint8), ("mod", np.uint8), ("port", np.uint8),
                     ("start_symbol", np.uint8), ("nof_symbols", np.uint8), ("nof_cdm_groups_without_data", np.uint8), ("n_scid", np.uint8),
                     ("ref_point_prb0", np.uint8), ("nof_reserved", np.uint8), ("dmrs_symbols_mask", np.uint16), ("grid_nof_prb", np.uint16),
                     ("bwp_start_rb", np.uint16), ("bwp_size_rb", np.uint16), ("pad", np.uint16), ("rb_mask", np.uint64, 5),
                     ("reserved", RePattern, 4), ("tb_offset", np.uint64), ("grid_offset", np.uint64)], align=True)
assert PdschPdu.itemsize == 304 and PdschPdu.fields["rb_mask"][1] == 56, PdschPdu.itemsize


def pdsch_pdu_nof_re(pdu):
    """Data REs of one PdschPdu record (pdsch_processor_impl::compute_nof_data_re; host computation in the library)."""
    a = np.ascontiguousarray(np.asarray(pdu, dtype=PdschPdu).reshape(1))
    return int(lib().miphy_pdsch_pdu_nof_re(a.ctypes.data_as(C.c_void_p)))


# Mirrors miphy_csi_rs_job.
CsiRsJob = np.dtype([("slot_in_frame", np.uint32), ("scrambling_id", np.uint32), ("amplitude", np.float32), ("start_rb", np.uint16), ("nof_rb", np.uint16),
                     ("rb_begin", np.uint16), ("rb_end", np.uint16), ("rb_stride", np.uint16), ("grid_nof_prb", np.uint16), ("mapping_row", np.uint8),
                     ("cdm", np.uint8), ("freq_density", np.uint8), ("nof_ports", np.uint8), ("ports", np.uint8, 16), ("re_mask", np.uint16, 16),
                     ("symbol_mask", np.uint16, 16), ("grid_offset", np.uint64)], align=True)
assert CsiRsJob.itemsize == 120 and CsiRsJob.fields["grid_offset"][1] == 112, CsiRsJob.itemsize

# Mirrors miphy_pdcch_pdu.
PdcchPdu = np.dtype([("slot_in_frame", np.uint32), ("rnti", np.uint32), ("n_id_pdcch_data", np.uint32), ("n_rnti", np.uint32), ("n_id_pdcch_dmrs", np.uint32),
                     ("reference_point_k_rb", np.uint32), ("data_power_offset_dB", np.float32), ("dmrs_power_offset_dB", np.float32),
                     ("payload_size", np.uint16), ("aggregation_level", np.uint8), ("start_symbol", np.uint8), ("duration", np.uint8), ("port", np.uint8),
                     ("grid_nof_prb", np.uint16), ("rb_mask", np.uint64, 5), ("payload_offset", np.uint64), ("grid_offset", np.uint64),
                     ("work_offset", np.uint64)], align=True)
assert PdcchPdu.itemsize == 104 and PdcchPdu.fields["rb_mask"][1] == 40, PdcchPdu.itemsize

# Mirrors miphy_pusch_pdu.
PuschPdu = np.dtype([("numerology", np.uint32), ("slot_in_frame", np.uint32), ("rnti", np.uint32), ("n_id", np.uint32),
                     ("dmrs_scrambling_id", np.uint32), ("Nref", np.uint32), ("tb_bytes", np.uint32), ("harq_cb_index", np.uint32),
                     ("n_scid", np.uint8), ("mod", np.uint8), ("nof_rx_ports", np.uint8), ("start_symbol", np.uint8), ("nof_symbols", np.uint8),
                     ("bg", np.uint8), ("rv", np.uint8), ("new_data", np.uint8), ("rx_ports", np.uint8, 4), ("use_early_stop", np.uint8),
                     ("reserved0", np.uint8), ("nof_ldpc_iterations", np.uint16), ("dmrs_symbols_mask", np.uint16), ("grid_nof_prb", np.uint16),
                     ("pad", np.uint32), ("rb_mask", np.uint64, 5), ("grid_offset", np.uint64), ("tb_offset", np.uint64)], align=True)
assert PuschPdu.itemsize == 112 and PuschPdu.fields["rb_mask"][1] == 56, PuschPdu.itemsize


# Mirrors miphy_pusch_uci.
PuschUci = np.dtype([("nof_harq_ack_bits", np.uint32), ("nof_csi_part1_bits", np.uint32), ("nof_csi_part2_bits", np.uint32),
                     ("nof_enc_harq_ack_bits", np.uint32), ("nof_enc_csi_part1_bits", np.uint32), ("nof_enc_csi_part2_bits", np.uint32),
                     ("nof_harq_ack_rvd", np.uint32), ("has_codeword", np.uint32), ("harq_ack_offset", np.uint64), ("csi_part1_offset", np.uint64),
                     ("csi_part2_offset", np.uint64)], align=True)
assert PuschUci.itemsize == 56


class PolarCode(C.Structure):
    """Mirrors miphy_polar_code (the arguments of srsran::polar_code::set)."""
    _fields_ = [("K", C.c_uint32), ("E", C.c_uint32), ("nMax", C.c_uint32), ("ibil", C.c_uint32)]

    def info(self):
        n, N, npc = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().miphy_polar_code_info(C.byref(self), C.byref(n), C.byref(N), C.byref(npc)))
        return n.value, N.value, npc.value


# Mirrors miphy_pusch_tb_desc / miphy_pusch_result / miphy_pdsch_tb_desc / miphy_sch_segmentation.
PuschTbDesc = np.dtype([("bg", np.uint8), ("rv", np.uint8), ("mod", np.uint8), ("nof_layers", np.uint8), ("new_data", np.uint8),
                        ("use_early_stop", np.uint8), ("nof_ldpc_iterations", np.uint16), ("Nref", np.uint32),
                        ("nof_ch_symbols", np.uint32), ("tb_bytes", np.uint32), ("harq_cb_index", np.uint32),
                        ("llr_offset", np.uint64), ("tb_offset", np.uint64)], align=True)
assert PuschTbDesc.itemsize == 40
PuschResult = np.dtype([("tb_crc_ok", np.int32), ("nof_codeblocks_total", np.uint32), ("iters_min", np.uint32),
                        ("iters_max", np.uint32), ("iters_mean", np.float32), ("nof_decoded", np.uint32)], align=True)
assert PuschResult.itemsize == 24
PdschTbDesc = np.dtype([("bg", np.uint8), ("rv", np.uint8), ("mod", np.uint8), ("nof_layers", np.uint8), ("Nref", np.uint32),
                        ("nof_ch_symbols", np.uint32), ("tb_bytes", np.uint32), ("tb_offset", np.uint64),
                        ("codeword_offset", np.uint64)], align=True)
assert PdschTbDesc.itemsize == 32
HARQ_CB_STRIDE = 66 * 384
HARQ_MSG_STRIDE = 1056


class SchSegmentation(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("nof_cbs", "Z", "K", "N", "nof_filler_bits", "nof_tb_crc_bits", "nof_cb_crc_bits",
                                          "cb_info_bits", "zero_pad")]


def sch_segmentation(tb_bytes, bg):
    """Host-side segmentation parameters (ldpc::compute_nof_codeblocks / compute_lifting_size / compute_codeblock_size)."""
    s = SchSegmentation()
    check(lib().miphy_sch_segmentation_info(tb_bytes, bg, C.byref(s)))
    return s


# Mirrors miphy_pbch_msg.
PbchMsg = np.dtype([("N_id", np.uint32), ("ssb_idx", np.uint32), ("L_max", np.uint32), ("hrf", np.uint32), ("sfn", np.uint32),
                    ("k_ssb", np.uint32), ("payload", np.uint8, 32)], align=True)
assert PbchMsg.itemsize == 56
# Mirrors miphy_ssb_pdu.
SsbPdu = np.dtype([("msg", PbchMsg), ("ssb_first_subcarrier", np.uint32), ("ssb_first_symbol", np.uint32), ("beta_pss_dB", np.float32),
                   ("grid_nof_prb", np.uint16), ("nof_ports", np.uint8), ("ports", np.uint8, 4), ("pad", np.uint8), ("grid_offset", np.uint64)], align=True)
assert SsbPdu.itemsize == 88 and SsbPdu.fields["grid_offset"][1] == 80, SsbPdu.itemsize


# Mirrors miphy_ulsch_demux_job.
UlschDemuxJob = np.dtype([("mod", np.uint8), ("nof_layers", np.uint8), ("start_symbol", np.uint8), ("nof_symbols", np.uint8), ("dmrs_type", np.uint8),
                          ("nof_cdm_groups_without_data", np.uint8), ("dmrs_symbols_mask", np.uint16), ("nof_prb", np.uint16), ("reserved", np.uint16),
                          ("nof_harq_ack_rvd", np.uint32), ("nof_enc_harq_ack_bits", np.uint32), ("nof_enc_csi_part1_bits", np.uint32),
                          ("nof_enc_csi_part2_bits", np.uint32), ("nof_harq_ack_bits", np.uint32), ("nof_csi_part1_bits", np.uint32),
                          ("nof_csi_part2_bits", np.uint32), ("in_offset", np.uint64), ("sch_offset", np.uint64), ("harq_ack_offset", np.uint64),
                          ("csi_part1_offset", np.uint64), ("csi_part2_offset", np.uint64)], align=True)
assert UlschDemuxJob.itemsize == 80, UlschDemuxJob.itemsize


def ulsch_demux_sizes(job):
    """(codeword LLRs in, UL-SCH LLRs out) of a UlschDemuxJob record (host function)."""
    a, b = C.c_uint32(), C.c_uint32()
    j = np.ascontiguousarray(np.asarray(job).reshape(1))
    check(lib().miphy_ulsch_demux_sizes(C.c_void_p(j.ctypes.data), C.byref(a), C.byref(b)))
    return a.value, b.value


def ulsch_placeholders(job):
    """RE indices of the repetition placeholders (ulsch_demultiplex::get_placeholders), host function."""
    j = np.ascontiguousarray(np.asarray(job).reshape(1))
    out = np.zeros(275 * 12 * 14, np.uint16)
    n = C.c_uint32()
    check(lib().miphy_ulsch_placeholders(C.c_void_p(j.ctypes.data), C.c_void_p(out.ctypes.data), out.size, C.byref(n)))
    return out[:n.value].copy()


# MIPHY_POLAR_OP_* of miphy_polar_block_batch.
(POLAR_OP_ALLOCATE, POLAR_OP_ENCODE, POLAR_OP_RATE_MATCH, POLAR_OP_RATE_DEMATCH, POLAR_OP_DECODE, POLAR_OP_DEALLOCATE, POLAR_OP_INTERLEAVE_TX,
 POLAR_OP_INTERLEAVE_RX) = range(8)

# Mirrors miphy_equalizer_job.
EqualizerJob = np.dtype([("nof_re", np.uint32), ("nof_rx_ports", np.uint8), ("nof_tx_layers", np.uint8), ("reserved", np.uint8, 2), ("noise_var", np.float32),
                         ("tx_scaling", np.float32), ("ch_symbols_offset", np.uint64), ("ch_estimates_offset", np.uint64), ("eq_symbols_offset", np.uint64),
                         ("eq_noise_vars_offset", np.uint64)], align=True)
assert EqualizerJob.itemsize == 48, EqualizerJob.itemsize

# Mirrors miphy_crc_desc.
CrcDesc = np.dtype([("bit_offset", np.uint64), ("nbits", np.uint32), ("poly", np.uint32)], align=True)
assert CrcDesc.itemsize == 16


def check(rc):
    if rc != 0:
        raise RuntimeError("miphy error %d: %s" % (rc, lib().miphy_last_error().decode()))


def _stream_ptr(stream):
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _dptr(t):
    """Device pointer of a torch tensor (must be a contiguous CUDA/HIP tensor)."""
    if not t.is_cuda:
        raise ValueError("miphy expects device (HBM-resident) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


class Context:
    """One per host thread / GPU (miphy_create)."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("miphy needs a HIP device (MI355X); no CPU fallback exists")
        self.device = device
        torch.cuda.set_device(device)
        h = C.c_void_p()
        check(lib().miphy_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().miphy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _descs(descs, dtype):
        """descs: numpy structured array (host) or torch uint8 device tensor holding the same bytes."""
        import torch
        if isinstance(descs, np.ndarray):
            assert descs.dtype == dtype, (descs.dtype, dtype)
            descs = np.ascontiguousarray(descs)
            return descs, descs.size, C.c_void_p(descs.ctypes.data), 0
        assert descs.dtype == torch.uint8 and descs.numel() % dtype.itemsize == 0
        return descs, descs.numel() // dtype.itemsize, _dptr(descs), 1

    # ------------------------------------------------------------------ LDPC decoder
    def ldpc_decode_batch(self, descs, llr, out_bits, iters, stream=None, limits=None):
        """limits: optional (max_Z, max_in_len) for device-resident descriptors."""
        import torch
        descs, n, ptr, on_dev = self._descs(descs, LdpcDecDesc)
        assert iters.dtype == torch.int32 and iters.numel() >= n
        lim = None
        if limits is not None:
            lim = (C.c_uint32 * 2)(int(limits[0]), int(limits[1]))
        check(lib().miphy_ldpc_decode_batch(self.h, ptr, on_dev, n, _dptr(llr), _dptr(out_bits), _dptr(iters),
                                            lim, _stream_ptr(stream)))

    # ------------------------------------------------------------------ LDPC rate (de)matching, encoder, CRC
    def ldpc_rate_dematch_batch(self, descs, llr_in, softbuf, stream=None, max_E=None):
        """max_E: optional bound on the rate-matched length for device-resident descriptors."""
        descs, n, ptr, on_dev = self._descs(descs, LdpcRdmDesc)
        lim = (C.c_uint32 * 1)(int(max_E)) if max_E is not None else None
        check(lib().miphy_ldpc_rate_dematch_batch(self.h, ptr, on_dev, n, _dptr(llr_in), _dptr(softbuf), lim, _stream_ptr(stream)))

    def ldpc_rate_match_batch(self, descs, cb_in, out, stream=None):
        descs, n, ptr, on_dev = self._descs(descs, LdpcRdmDesc)
        check(lib().miphy_ldpc_rate_match_batch(self.h, ptr, on_dev, n, _dptr(cb_in), _dptr(out), _stream_ptr(stream)))

    def ldpc_encode_batch(self, descs, msg_in, cb_out, stream=None):
        descs, n, ptr, on_dev = self._descs(descs, LdpcEncDesc)
        check(lib().miphy_ldpc_encode_batch(self.h, ptr, on_dev, n, _dptr(msg_in), _dptr(cb_out), _stream_ptr(stream)))

    def crc_batch(self, descs, data, checksums, stream=None):
        descs, n, ptr, on_dev = self._descs(descs, CrcDesc)
        check(lib().miphy_crc_batch(self.h, ptr, on_dev, n, _dptr(data), _dptr(checksums), _stream_ptr(stream)))

    # ------------------------------------------------------------------ DFT / OFDM
    def dft_batch(self, size, inverse, n, x, out, stream=None):
        check(lib().miphy_dft_batch(self.h, size, int(inverse), n, _dptr(x), _dptr(out), _stream_ptr(stream)))

    def ofdm_demodulate_slots(self, cfg, jobs, samples, grid, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, OfdmJob)
        check(lib().miphy_ofdm_demodulate_slots(self.h, C.byref(cfg), ptr, on_dev, n, _dptr(samples), _dptr(grid), _stream_ptr(stream)))

    def ofdm_modulate_slots(self, cfg, jobs, grid, samples, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, OfdmJob)
        check(lib().miphy_ofdm_modulate_slots(self.h, C.byref(cfg), ptr, on_dev, n, _dptr(grid), _dptr(samples), _stream_ptr(stream)))

    # ------------------------------------------------------------------ PDSCH modulator / PDSCH DM-RS
    def pdsch_modulate_batch(self, jobs, codewords, grid, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, PdschModJob)
        check(lib().miphy_pdsch_modulate_batch(self.h, ptr, on_dev, n, _dptr(codewords), _dptr(grid), _stream_ptr(stream)))

    def dmrs_pdsch_map_batch(self, jobs, grid, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, DmrsPdschJob)
        check(lib().miphy_dmrs_pdsch_map_batch(self.h, ptr, on_dev, n, _dptr(grid), _stream_ptr(stream)))

    def pdsch_process_batch(self, pdus, tb_in, grid, stream=None):
        """pdsch_processor::process for a batch of PDUs (host descriptors): transport blocks -> REs of the resource grid."""
        assert isinstance(pdus, np.ndarray) and pdus.dtype == PdschPdu
        pdus = np.ascontiguousarray(pdus)
        check(lib().miphy_pdsch_process_batch(self.h, C.c_void_p(pdus.ctypes.data), pdus.size, _dptr(tb_in), _dptr(grid), _stream_ptr(stream)))

    def csi_rs_map_batch(self, jobs, grid, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, CsiRsJob)
        check(lib().miphy_csi_rs_map_batch(self.h, ptr, on_dev, n, _dptr(grid), _stream_ptr(stream)))

    def ssb_process_batch(self, pdus, grid, stream=None):
        """ssb_processor::process (after the position look-up) for a batch of SS/PBCH blocks (host descriptors)."""
        assert isinstance(pdus, np.ndarray) and pdus.dtype == SsbPdu
        pdus = np.ascontiguousarray(pdus)
        check(lib().miphy_ssb_process_batch(self.h, C.c_void_p(pdus.ctypes.data), pdus.size, _dptr(grid), _stream_ptr(stream)))

    def pdcch_process_batch(self, pdus, payloads, grid, stream=None):
        """pdcch_processor::process (after the CCE-to-PRB mapping) for a batch of PDUs (host descriptors): DCI payload bits -> grid REs."""
        assert isinstance(pdus, np.ndarray) and pdus.dtype == PdcchPdu
        pdus = np.ascontiguousarray(pdus)
        check(lib().miphy_pdcch_process_batch(self.h, C.c_void_p(pdus.ctypes.data), pdus.size, _dptr(payloads), _dptr(grid), _stream_ptr(stream)))

    # ------------------------------------------------------------------ Open Fronthaul IQ (de)compression (U-plane payloads <-> grid rows)
    def ofh_iq_decompress_batch(self, jobs, payload, grid, simd_arithmetic=True, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, OfhIqJob)
        check(lib().miphy_ofh_iq_decompress_batch(self.h, ptr, on_dev, n, _dptr(payload), _dptr(grid), int(simd_arithmetic), _stream_ptr(stream)))

    def ofh_iq_compress_batch(self, jobs, grid, payload, iq_scaling=1.0, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, OfhIqJob)
        check(lib().miphy_ofh_iq_compress_batch(self.h, ptr, on_dev, n, _dptr(grid), float(iq_scaling), _dptr(payload), _stream_ptr(stream)))

    # ------------------------------------------------------------------ PUSCH demodulator (equalise + soft-demap + descramble)
    def pusch_demodulate_batch_ex(self, jobs, grid, ce, scalars, llr, placeholders=None, evm_sums=None, stream=None):
        """With the repetition placeholders (device uint16) and / or the per-symbol EVM sums (device float32, 14 per job at evm_offset)."""
        jobs, n, ptr, on_dev = self._descs(jobs, PuschDemodJob)
        check(lib().miphy_pusch_demodulate_batch_ex(self.h, ptr, on_dev, n, _dptr(grid), _dptr(ce), _dptr(scalars), _dptr(llr),
                                                    _dptr(placeholders) if placeholders is not None else None,
                                                    _dptr(evm_sums) if evm_sums is not None else None, _stream_ptr(stream)))

    def pusch_demodulate_batch(self, jobs, grid, ce, scalars, llr, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, PuschDemodJob)
        check(lib().miphy_pusch_demodulate_batch(self.h, ptr, on_dev, n, _dptr(grid), _dptr(ce), _dptr(scalars), _dptr(llr), _stream_ptr(stream)))

    # ------------------------------------------------------------------ DM-RS PUSCH channel estimator
    def dmrs_pusch_estimate_batch(self, jobs, grid, ce, scalars, stream=None, max_ports=0, max_layers=0):
        """max_ports / max_layers: optional bound for device-resident jobs (sizes the launch; 0 = unknown)."""
        jobs, n, ptr, on_dev = self._descs(jobs, PuschChestJob)
        if on_dev:
            on_dev |= (int(max_ports) << 8) | (int(max_layers) << 12)
        check(lib().miphy_dmrs_pusch_estimate_batch(self.h, ptr, on_dev, n, _dptr(grid), _dptr(ce), _dptr(scalars), _stream_ptr(stream)))

    def port_channel_estimate_batch(self, jobs, grid, pilots, ce, scalars, stream=None):
        """port_channel_estimator::compute: pilots given by the caller (device complex64), optional intra-slot hopping."""
        jobs, n, ptr, on_dev = self._descs(jobs, PuschChestJob)
        check(lib().miphy_port_channel_estimate_batch(self.h, ptr, on_dev, n, _dptr(grid), _dptr(pilots), _dptr(ce), _dptr(scalars), _stream_ptr(stream)))

    # ------------------------------------------------------------------ polar chains / PDCCH encoder
    def polar_encode_batch(self, code, n, msg, rm_out, allocated_tap=None, encoded_tap=None, stream=None):
        check(lib().miphy_polar_encode_batch(self.h, C.byref(code), n, _dptr(msg), _dptr(rm_out),
                                             _dptr(allocated_tap) if allocated_tap is not None else None,
                                             _dptr(encoded_tap) if encoded_tap is not None else None, _stream_ptr(stream)))

    def polar_decode_batch(self, code, n, llr, msg_out, dematched_tap=None, decoded_u_tap=None, stream=None):
        check(lib().miphy_polar_decode_batch(self.h, C.byref(code), n, _dptr(llr), _dptr(msg_out),
                                             _dptr(dematched_tap) if dematched_tap is not None else None,
                                             _dptr(decoded_u_tap) if decoded_u_tap is not None else None, _stream_ptr(stream)))

    def pdcch_encode_batch(self, A, E, n, payload, rnti, out, stream=None):
        check(lib().miphy_pdcch_encode_batch(self.h, A, E, n, _dptr(payload), _dptr(rnti), _dptr(out), _stream_ptr(stream)))

    # ------------------------------------------------------------------ transport-block level shared channel
    def pusch_decode_batch(self, tbs, llrs, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, stream=None):
        """tbs: numpy PuschTbDesc array (host). results: torch uint8 tensor of n * PuschResult.itemsize bytes."""
        assert isinstance(tbs, np.ndarray) and tbs.dtype == PuschTbDesc
        tbs = np.ascontiguousarray(tbs)
        check(lib().miphy_pusch_decode_batch(self.h, C.c_void_p(tbs.ctypes.data), tbs.size, _dptr(llrs), _dptr(harq_softbits),
                                             _dptr(harq_msgs), _dptr(harq_crc_ok), _dptr(tb_out), _dptr(results), _stream_ptr(stream)))

    def ulsch_demultiplex_batch(self, jobs, llr_in, sch, harq_ack, csi1, csi2, stream=None):
        assert isinstance(jobs, np.ndarray) and jobs.dtype == UlschDemuxJob
        jobs = np.ascontiguousarray(jobs)
        check(lib().miphy_ulsch_demultiplex_batch(self.h, C.c_void_p(jobs.ctypes.data), jobs.size, _dptr(llr_in), _dptr(sch), _dptr(harq_ack),
                                                  _dptr(csi1), _dptr(csi2), _stream_ptr(stream)))

    def channel_equalize_batch(self, jobs, ch_symbols, ch_estimates, eq_symbols, eq_noise_vars, stream=None):
        """channel_equalizer::equalize (zero forcing) for a batch of EqualizerJob records."""
        jobs, n, ptr, on_dev = self._descs(jobs, EqualizerJob)
        check(lib().miphy_channel_equalize_batch(self.h, ptr, on_dev, n, _dptr(ch_symbols), _dptr(ch_estimates), _dptr(eq_symbols), _dptr(eq_noise_vars),
                                                 _stream_ptr(stream)))

    def polar_block_batch(self, code, op, param, n, x, out, stream=None):
        check(lib().miphy_polar_block_batch(self.h, C.byref(code) if code is not None else None, op, param, n, _dptr(x), _dptr(out), _stream_ptr(stream)))

    def ofdm_demodulate_symbols(self, cfg, jobs, samples, grid, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, OfdmJob)
        check(lib().miphy_ofdm_demodulate_symbols(self.h, C.byref(cfg), ptr, on_dev, n, _dptr(samples), _dptr(grid), _stream_ptr(stream)))

    def ofdm_modulate_symbols(self, cfg, jobs, grid, samples, stream=None):
        jobs, n, ptr, on_dev = self._descs(jobs, OfdmJob)
        check(lib().miphy_ofdm_modulate_symbols(self.h, C.byref(cfg), ptr, on_dev, n, _dptr(grid), _dptr(samples), _stream_ptr(stream)))

    def pusch_decode_plan(self, tbs):
        """Prepared miphy_pusch_decode_batch (descriptors uploaded once): returns a PuschDecodePlan."""
        return PuschDecodePlan(self, tbs)

    def pusch_process_batch(self, pdus, grid, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, scalars, stream=None):
        """pdus: numpy PuschPdu array (host). results: torch uint8 tensor of n * PuschResult.itemsize bytes; scalars: float32 n x 20."""
        assert isinstance(pdus, np.ndarray) and pdus.dtype == PuschPdu
        pdus = np.ascontiguousarray(pdus)
        check(lib().miphy_pusch_process_batch(self.h, C.c_void_p(pdus.ctypes.data), pdus.size, _dptr(grid), _dptr(harq_softbits), _dptr(harq_msgs),
                                              _dptr(harq_crc_ok), _dptr(tb_out), _dptr(results), _dptr(scalars), _stream_ptr(stream)))

    def pusch_process_batch_ex(self, pdus, uci, grid, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, scalars, uci_llr, evm, stream=None):
        """PDUs with multiplexed UCI (uci: numpy PuschUci array or None) and the EVM (evm: float32 device tensor of n or None)."""
        assert isinstance(pdus, np.ndarray) and pdus.dtype == PuschPdu
        pdus = np.ascontiguousarray(pdus)
        up = None
        if uci is not None:
            assert isinstance(uci, np.ndarray) and uci.dtype == PuschUci and uci.size == pdus.size
            uci = np.ascontiguousarray(uci)
            up = C.c_void_p(uci.ctypes.data)
        check(lib().miphy_pusch_process_batch_ex(self.h, C.c_void_p(pdus.ctypes.data), up, pdus.size, _dptr(grid), _dptr(harq_softbits), _dptr(harq_msgs),
                                                 _dptr(harq_crc_ok), _dptr(tb_out), _dptr(results), _dptr(scalars),
                                                 _dptr(uci_llr) if uci_llr is not None else None, _dptr(evm) if evm is not None else None,
                                                 _stream_ptr(stream)))

    def pdsch_encode_batch(self, tbs, tb_in, codeword_out, stream=None):
        assert isinstance(tbs, np.ndarray) and tbs.dtype == PdschTbDesc
        tbs = np.ascontiguousarray(tbs)
        check(lib().miphy_pdsch_encode_batch(self.h, C.c_void_p(tbs.ctypes.data), tbs.size, _dptr(tb_in), _dptr(codeword_out),
                                             _stream_ptr(stream)))

    def pbch_encode_batch(self, msgs, out, stream=None):
        assert isinstance(msgs, np.ndarray) and msgs.dtype == PbchMsg
        msgs = np.ascontiguousarray(msgs)
        check(lib().miphy_pbch_encode_batch(self.h, C.c_void_p(msgs.ctypes.data), msgs.size, _dptr(out), _stream_ptr(stream)))

    def polar_decode_list_batch(self, code, list_size, crc_mode, n, llr, rnti, msg_out, crc_ok_out, metric_out=None, stream=None):
        check(lib().miphy_polar_decode_list_batch(self.h, C.byref(code), list_size, crc_mode, n, _dptr(llr),
                                                  _dptr(rnti) if rnti is not None else None, _dptr(msg_out), _dptr(crc_ok_out),
                                                  _dptr(metric_out) if metric_out is not None else None, _stream_ptr(stream)))


class PdschProcessPlan:
    """miphy_pdsch_process_plan_*: PDU validation, segmentation and descriptor uploads once; run() is the launches of the transmit chain only."""

    def __init__(self, ctx, pdus):
        assert isinstance(pdus, np.ndarray) and pdus.dtype == PdschPdu
        pdus = np.ascontiguousarray(pdus)
        self.ctx, self.n = ctx, pdus.size
        h = C.c_void_p()
        check(lib().miphy_pdsch_process_plan_create(ctx.h, C.c_void_p(pdus.ctypes.data), pdus.size, C.byref(h)))
        self.h = h

    def run(self, tb_in, grid, stream=None):
        check(lib().miphy_pdsch_process_plan_run(self.h, _dptr(tb_in), _dptr(grid), _stream_ptr(stream)))

    def close(self):
        if getattr(self, "h", None):
            lib().miphy_pdsch_process_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LdpcDecodePlan:
    """miphy_ldpc_decode_plan_*: codeblock descriptors validated, sorted into launch classes and uploaded once; run() is launches only."""

    def __init__(self, ctx, descs):
        assert isinstance(descs, np.ndarray) and descs.dtype == LdpcDecDesc
        descs = np.ascontiguousarray(descs)
        self.ctx, self.n = ctx, descs.size
        h = C.c_void_p()
        check(lib().miphy_ldpc_decode_plan_create(ctx.h, C.c_void_p(descs.ctypes.data), descs.size, C.byref(h)))
        self.h = h

    def run(self, llr, out_bits, iters, stream=None):
        check(lib().miphy_ldpc_decode_plan_run(self.h, _dptr(llr), _dptr(out_bits), _dptr(iters), _stream_ptr(stream)))

    def nof_launches(self):
        return int(lib().miphy_ldpc_decode_plan_nof_launches(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib().miphy_ldpc_decode_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PuschDecodePlan:
    """miphy_pusch_decode_plan_*: segmentation and descriptor upload once, run() is launches only."""

    def __init__(self, ctx, tbs):
        assert isinstance(tbs, np.ndarray) and tbs.dtype == PuschTbDesc
        tbs = np.ascontiguousarray(tbs)
        self.ctx, self.n = ctx, tbs.size
        h = C.c_void_p()
        check(lib().miphy_pusch_decode_plan_create(ctx.h, C.c_void_p(tbs.ctypes.data), tbs.size, C.byref(h)))
        self.h = h

    def run(self, llrs, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, stream=None):
        check(lib().miphy_pusch_decode_plan_run(self.h, _dptr(llrs), _dptr(harq_softbits), _dptr(harq_msgs), _dptr(harq_crc_ok), _dptr(tb_out),
                                                _dptr(results), _stream_ptr(stream)))

    def info(self):
        """(codeblocks, dematch-inside-the-decoder flag, largest number of variable nodes)."""
        a = (C.c_uint32 * 3)()
        check(lib().miphy_pusch_decode_plan_info(self.h, a))
        return int(a[0]), bool(a[1]), int(a[2])

    def nof_launches(self):
        """LDPC decoder launches per run (= launch classes of the batch)."""
        return int(lib().miphy_pusch_decode_plan_nof_launches(self.h))

    def enable_timing(self, max_runs=64):
        check(lib().miphy_pusch_decode_plan_enable_timing(self.h, max_runs))

    def read_timing(self):
        """Mean per-kernel milliseconds of the runs since the last call: {"rate_dematch", "ldpc_decode", "tb_assemble", "runs"}."""
        ms, runs = (C.c_float * 3)(), C.c_uint32()
        check(lib().miphy_pusch_decode_plan_read_timing(self.h, ms, C.byref(runs)))
        return {"rate_dematch": float(ms[0]), "ldpc_decode": float(ms[1]), "tb_assemble": float(ms[2]), "runs": int(runs.value)}

    def close(self):
        if getattr(self, "h", None):
            lib().miphy_pusch_decode_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------- HARQ softbuffer pool
class HarqPoolConfig(C.Structure):
    """Mirrors miphy_harq_pool_config (rx_softbuffer_pool_config + the slot period + the extent of one softbuffer)."""
    _fields_ = [("max_softbuffers", C.c_uint32), ("max_nof_codeblocks", C.c_uint32), ("expire_timeout_slots", C.c_uint32),
                ("nof_slots_wrap", C.c_uint32), ("max_codeblocks_per_buffer", C.c_uint32)]


class HarqBufferInfo(C.Structure):
    _fields_ = [("state", C.c_uint32), ("rnti", C.c_uint32), ("harq_id", C.c_uint32), ("nof_codeblocks", C.c_uint32),
                ("first_cb", C.c_uint32), ("expire_slot", C.c_uint32)]


HARQ_AVAILABLE, HARQ_RESERVED, HARQ_LOCKED, HARQ_RELEASED = 0, 1, 2, 3


class HarqPool:
    """miphy_harq_pool: the reference's rx_softbuffer_pool over device-resident HARQ arrays. ctx=None gives a
    bookkeeping-only pool (reservation state machine without device memory; used by the host-logic tests)."""

    def __init__(self, ctx, max_softbuffers, max_nof_codeblocks, expire_timeout_slots, numerology=1, max_codeblocks_per_buffer=0):
        self.cfg = HarqPoolConfig(max_softbuffers, max_nof_codeblocks, expire_timeout_slots, 10240 << numerology, max_codeblocks_per_buffer)
        self.ctx = ctx  # keeps the context alive
        h = C.c_void_p()
        check(lib().miphy_harq_pool_create(ctx.h if ctx is not None else None, C.byref(self.cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().miphy_harq_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, slot, rnti, harq_id, nof_codeblocks):
        """-> (softbuffer index or -1, first codeblock slot = harq_cb_index of the transport-block descriptor)."""
        b, f = C.c_int32(), C.c_uint32()
        check(lib().miphy_harq_pool_reserve(self.h, slot, rnti, harq_id, nof_codeblocks, C.byref(b), C.byref(f)))
        return b.value, f.value

    def lock(self, buffer):
        check(lib().miphy_harq_pool_lock(self.h, buffer))

    def unlock(self, buffer):
        check(lib().miphy_harq_pool_unlock(self.h, buffer))

    def release(self, buffer):
        check(lib().miphy_harq_pool_release(self.h, buffer))

    def run_slot(self, slot):
        check(lib().miphy_harq_pool_run_slot(self.h, slot))

    def info(self, buffer):
        out = HarqBufferInfo()
        check(lib().miphy_harq_pool_info(self.h, buffer, C.byref(out)))
        return out

    def free_codeblocks(self):
        out = C.c_uint32()
        check(lib().miphy_harq_pool_free_codeblocks(self.h, C.byref(out)))
        return out.value

    def arrays(self):
        """Device arrays as torch uint8 / int8 views: (softbits [ncb, 66*384] int8, msgs [ncb, 1056] uint8, crc_ok [ncb] uint8)."""
        import torch
        ptrs = [C.c_void_p() for _ in range(3)]
        check(lib().miphy_harq_pool_arrays(self.h, *[C.byref(p) for p in ptrs]))
        ncb = self.cfg.max_softbuffers * (self.cfg.max_codeblocks_per_buffer or 52)
        dev = torch.device("cuda", self.ctx.device)

        def view(ptr, nbytes, dtype):
            class _Holder:  # __cuda_array_interface__ view over memory the pool owns
                pass
            hld = _Holder()
            hld.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr.value, False), "version": 2}
            t = torch.as_tensor(hld, device=dev)
            return t.view(dtype)
        soft = view(ptrs[0], ncb * HARQ_CB_STRIDE, torch.int8).view(ncb, HARQ_CB_STRIDE)
        msgs = view(ptrs[1], ncb * HARQ_MSG_STRIDE, torch.uint8).view(ncb, HARQ_MSG_STRIDE)
        crc = view(ptrs[2], ncb, torch.uint8)
        return soft, msgs, crc
