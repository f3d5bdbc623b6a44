"""Host-side helpers mirroring the reference's LDPC operator interfaces on top of the batched C ABI.

`LdpcDecoder.decode` has the argument meaning of srsran::ldpc_decoder::decode
(include/srsran/phy/upper/channel_coding/ldpc/ldpc_decoder.h:73-74): LLRs in, packed message out, optional CRC for
early stopping, returns the iteration count or None ("nullopt").  It is a batch of one over miphy_ldpc_decode_batch.
"""
import numpy as np

from .binding import CRC_NONE, Context, LdpcDecDesc

BG_K = {1: 22, 2: 10}
BG_N_SHORT = {1: 66, 2: 50}
BG_N_FULL = {1: 68, 2: 52}


def make_dec_descs(n, bg, Z, in_len, crc_poly=CRC_NONE, max_iter=6, nof_filler_bits=0, llr_stride=None, out_stride=None):
    """Uniform batch: n codeblocks laid out back to back."""
    K = BG_K[bg] * Z
    llr_stride = in_len if llr_stride is None else llr_stride
    out_stride = (K + 7) // 8 if out_stride is None else out_stride
    d = np.zeros(n, dtype=LdpcDecDesc)
    d["bg"], d["crc_poly"], d["Z"], d["max_iter"] = bg, crc_poly, Z, max_iter
    d["nof_filler_bits"], d["in_len"] = nof_filler_bits, in_len
    d["llr_offset"] = np.arange(n, dtype=np.uint64) * np.uint64(llr_stride)
    d["out_offset"] = np.arange(n, dtype=np.uint64) * np.uint64(out_stride)
    return d


class LdpcDecoder:
    """Drop-in shaped like srsran::ldpc_decoder (one codeblock per call; use Context.ldpc_decode_batch for throughput)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def decode(self, llr, bg, Z, crc_poly=CRC_NONE, max_iter=6, nof_filler_bits=0, out=None):
        import torch
        K = BG_K[bg] * Z
        llr = llr.contiguous()
        if out is None:
            out = torch.zeros((K + 7) // 8, dtype=torch.uint8, device=llr.device)
        iters = torch.zeros(1, dtype=torch.int32, device=llr.device)
        descs = make_dec_descs(1, bg, Z, llr.numel(), crc_poly, max_iter, nof_filler_bits)
        self.ctx.ldpc_decode_batch(descs, llr, out, iters)
        it = int(iters.item())
        return (it if it > 0 else None), out
