"""miphy -- Python host-side binding of the MI355X-native 5G NR upper-PHY hot path (libmiphy.so, C ABI in include/miphy.h).

PyTorch is used only as plumbing (device memory, streams, torch.distributed); all compute is in the hand-written HIP
kernels behind the C ABI.  There is NO CPU fallback: importing this package without the built library, or calling it
without a GPU, raises.
"""
from .binding import (Context, LdpcDecDesc, LdpcRdmDesc, LdpcEncDesc, CrcDesc, OfdmJob, OfdmConfig, PuschChestJob, PuschDemodJob, PuschPdu, pusch_demod_nof_llr, PdschModJob, PdcchPdu, SsbPdu, CsiRsJob, OfhIqJob, OFH_COMPRESSION_NONE, OFH_COMPRESSION_BFP, PdschPdu, pdsch_pdu_nof_re, DmrsPdschJob, RePattern, pdsch_mod_nof_re, PolarCode, PbchMsg, PuschTbDesc, PuschResult, PuschDecodePlan, LdpcDecodePlan, PdschProcessPlan, PuschUci, UlschDemuxJob, EqualizerJob, ofdm_symbol_size, ulsch_demux_sizes, ulsch_placeholders, PdschTbDesc, sch_segmentation, HarqPool, HarqPoolConfig, HarqBufferInfo,
                      HARQ_AVAILABLE, HARQ_RESERVED, HARQ_LOCKED, HARQ_RELEASED, HARQ_CB_STRIDE, HARQ_MSG_STRIDE, lib, lib_path, LibraryNotBuilt, CRC24A, CRC24B, CRC24C, CRC16, CRC11,
                      CRC_NONE)
from . import ldpc

__all__ = ["Context", "LdpcDecDesc", "LdpcRdmDesc", "LdpcEncDesc", "CrcDesc", "OfdmJob", "OfdmConfig", "PuschChestJob", "PuschDemodJob", "PuschPdu", "pusch_demod_nof_llr", "PdschModJob", "PdcchPdu", "SsbPdu", "CsiRsJob", "OfhIqJob", "OFH_COMPRESSION_NONE", "OFH_COMPRESSION_BFP", "PdschPdu", "pdsch_pdu_nof_re", "DmrsPdschJob", "RePattern", "pdsch_mod_nof_re", "PolarCode", "PbchMsg", "PuschTbDesc", "PuschResult", "PuschDecodePlan", "LdpcDecodePlan", "PdschProcessPlan", "PuschUci", "UlschDemuxJob", "EqualizerJob", "ofdm_symbol_size", "ulsch_demux_sizes", "ulsch_placeholders", "PdschTbDesc", "sch_segmentation", "HarqPool", "HarqPoolConfig", "HarqBufferInfo", "HARQ_AVAILABLE", "HARQ_RESERVED", "HARQ_LOCKED", "HARQ_RELEASED", "HARQ_CB_STRIDE",
           "HARQ_MSG_STRIDE", "lib", "lib_path", "LibraryNotBuilt", "ldpc", "CRC24A", "CRC24B", "CRC24C", "CRC16",
           "CRC11", "CRC_NONE"]
