"""GPU: miphy_ulsch_demultiplex_batch (one closed-form classification per resource element) against the oracle's serial scan, which is
pinned against the reference (tests/test_ulsch_demux.py): the UL-SCH data stream incl. the all-zero elements where HARQ-ACK punctures
reserved resource elements, and the three UCI streams; a heterogeneous batch."""
import numpy as np
import pytest

import oracle_lib as O
from test_ulsch_demux import _job

pytestmark = pytest.mark.gpu


def test_demultiplex_batch_matches_oracle(ctx):
    import torch
    import miphy
    rng = np.random.default_rng(614)
    cases = O.ulsch_cases(rng, 96)
    jobs = np.zeros(len(cases), dtype=miphy.UlschDemuxJob)
    ins, exp, off_in, off = [], [], 0, [0, 0, 0, 0]
    for i, case in enumerate(cases):
        n_in, n_sch, _, _ = O.o_ulsch_demultiplex(*case)
        llr = rng.integers(-120, 121, n_in).astype(np.int8)
        llr[llr == 0] = 1
        _, _, streams, _ = O.o_ulsch_demultiplex(*case, llr=llr)
        jobs[i] = _job(miphy, case)
        jobs[i]["in_offset"], jobs[i]["sch_offset"], jobs[i]["harq_ack_offset"] = off_in, off[0], off[1]
        jobs[i]["csi_part1_offset"], jobs[i]["csi_part2_offset"] = off[2], off[3]
        ins.append(llr)
        exp.append((tuple(off), streams))
        off_in += n_in
        for k in range(4):
            off[k] += streams[k].size
    d_in = torch.from_numpy(np.concatenate(ins)).cuda()
    outs = [torch.full((max(off[k], 1),), 77, dtype=torch.int8, device="cuda") for k in range(4)]
    ctx.ulsch_demultiplex_batch(jobs, d_in, *outs)
    torch.cuda.synchronize()
    got = [o.cpu().numpy() for o in outs]
    for i, (o0, streams) in enumerate(exp):
        for k in range(4):
            assert np.array_equal(got[k][o0[k]:o0[k] + streams[k].size], streams[k]), (i, cases[i], k)


def test_demultiplex_full_size_allocation(ctx):
    """The largest allocation of the path (273 PRB, 14 symbols, 256QAM, HARQ-ACK on reserved elements + both CSI parts) against the oracle,
    and the size-independent property: every input LLR lands in exactly one stream (or is punctured), so the multiset of non-zero
    inputs equals the multiset of non-zero outputs."""
    import torch
    import miphy
    rng = np.random.default_rng(615)
    mod, nprb = 8, 273
    re_sym = nprb * 12
    # (mod, layers, nprb, start, nof, G_rvd, dmrs_type, dmrs_mask, cdm, (G_ack, G_csi1, G_csi2), (O_ack, O_csi1, O_csi2))
    case = (mod, 1, nprb, 0, 14, mod * 700, 1, (1 << 2) | (1 << 11), 2, (mod * 650, mod * 900, mod * 1500), (2, 20, 40))
    n_in, n_sch, _, _ = O.o_ulsch_demultiplex(*case)
    assert n_in == re_sym * 12 * mod
    llr = rng.integers(1, 121, n_in).astype(np.int8) * rng.choice(np.array([-1, 1], np.int8), n_in)
    _, _, streams, _ = O.o_ulsch_demultiplex(*case, llr=llr)
    job = np.zeros(1, dtype=miphy.UlschDemuxJob)
    job[0] = _job(miphy, case)
    outs = [torch.full((max(s.size, 1),), 77, dtype=torch.int8, device="cuda") for s in streams]
    ctx.ulsch_demultiplex_batch(job, torch.from_numpy(llr).cuda(), *outs)
    torch.cuda.synchronize()
    got = [o.cpu().numpy()[:s.size] for o, s in zip(outs, streams)]
    for k in range(4):
        assert np.array_equal(got[k], streams[k]), k
    allout = np.concatenate(got)
    assert np.array_equal(np.sort(allout[allout != 0]), np.sort(llr))  # no LLR lost or duplicated (inputs are all non-zero)
    # HARQ-ACK sits on reserved elements: each of its LLRs leaves an all-zero LLR in the stream it punctures (SCH or CSI part 2)
    assert (allout == 0).sum() == case[9][0] and (got[0] == 0).sum() + (got[3] == 0).sum() == case[9][0]
