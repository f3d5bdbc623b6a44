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
