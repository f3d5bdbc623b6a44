"""CPU: the reservation state machine of miphy_harq_pool (bookkeeping-only pool, no device) against recorded traces of the
reference's rx_softbuffer_pool (tests/golden/harq_pool.npz, written by oracle/gen_golden.py) and, where oracle/_ref is built,
against the reference pool driven live by fresh random traces."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "srsran_project_23.5_amd"))
import miphy  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "harq_pool.npz")


def miphy_pool_run(ops, max_softbuffers, max_nof_codeblocks, expire_timeout_slots, numerology=1, pool=None):
    """The same caller behaviour as oracle_lib.r_pool_run: a reservation is reserve + lock (what constructing the
    unique_rx_softbuffer does), leaving the scope is unlock, release is release."""
    pool = pool or miphy.HarqPool(None, max_softbuffers, max_nof_codeblocks, expire_timeout_slots, numerology)
    held, seen = {}, []
    out = np.zeros((len(ops), 2), np.int64)
    for i, (op, slot, rnti, harq, ncb, hd) in enumerate(ops.tolist()):
        if op == O.POOL_RESERVE:
            if hd in held:
                pool.unlock(held.pop(hd))
            b, first = pool.reserve(slot, rnti, harq, ncb)
            if b < 0:
                out[i] = (-1, 0)
                continue
            assert first == b * 52
            pool.lock(b)
            held[hd] = b
            if b not in seen:
                seen.append(b)
            out[i] = (seen.index(b), pool.info(b).nof_codeblocks)
        elif op == O.POOL_DROP:
            if hd in held:
                pool.unlock(held.pop(hd))
        elif op == O.POOL_RELEASE:
            if hd in held:
                pool.release(held.pop(hd))
        else:
            pool.run_slot(slot)
    return out


def test_against_recorded_reference_traces():
    g = np.load(GOLDEN)
    for i in range(int(g["n"])):
        ms, mc, ex = g["cfg_%d" % i].tolist()
        got = miphy_pool_run(g["ops_%d" % i], ms, mc, ex)
        assert np.array_equal(got, g["res_%d" % i]), "trace %d: first difference at op %d" % (i, int(np.argmax((got != g["res_%d" % i]).any(axis=1))))
        assert (g["res_%d" % i][:, 0] < 0).any() and (g["res_%d" % i][:, 0] >= 0).any()  # the trace exercises refusals too


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_against_live_reference_pool():
    for seed, (ms, mc, ex) in enumerate([(4, 20, 8), (2, 9, 3), (8, 16, 40), (3, 200, 1), (5, 12, 0)]):
        ops = O.pool_trace(1000 + seed, 4000)
        assert np.array_equal(miphy_pool_run(ops, ms, mc, ex), O.r_pool_run(ops, ms, mc, ex))


def test_state_machine_and_errors():
    p = miphy.HarqPool(None, 2, 10, 4)
    b, first = p.reserve(10, 0x4601, 1, 4)
    assert (b, first) == (0, 0) and p.info(b).state == miphy.HARQ_RESERVED and p.free_codeblocks() == 6
    assert p.info(b).expire_slot == 14
    p.lock(b)
    with pytest.raises(RuntimeError, match="not reserved"):
        p.lock(b)
    assert p.reserve(11, 0x4601, 1, 4)[0] == -1          # locked: no new reservation, no retransmission
    p.unlock(b)
    assert p.reserve(11, 0x4601, 1, 4) == (0, 0)         # same size: same softbuffer, budget untouched
    assert p.free_codeblocks() == 6 and p.info(0).expire_slot == 15
    assert p.reserve(11, 0x4602, 0, 7)[0] == -1          # budget: 6 left
    assert p.info(1).state == miphy.HARQ_AVAILABLE and p.info(1).rnti == 0x4602  # the identifier sticks (reference behaviour)
    b2, first2 = p.reserve(13, 0x4602, 0, 6)
    assert (b2, first2) == (1, 52) and p.free_codeblocks() == 0
    p.run_slot(14)
    assert p.info(0).state == miphy.HARQ_RESERVED
    p.run_slot(15)                                        # expiry slot reached
    assert p.info(0).state == miphy.HARQ_AVAILABLE and p.free_codeblocks() == 4
    p.release(b2)
    with pytest.raises(RuntimeError, match="neither reserved nor locked"):
        p.release(b2)
    assert p.info(b2).state == miphy.HARQ_RELEASED and p.free_codeblocks() == 4
    assert p.reserve(16, 0x4602, 0, 6) == (1, 52)        # a released softbuffer is reused with its codeblocks
    p.release(1)
    p.run_slot(16)
    assert p.info(1).state == miphy.HARQ_AVAILABLE and p.free_codeblocks() == 10
    with pytest.raises(RuntimeError, match="out of range"):
        p.lock(2)
    with pytest.raises(RuntimeError, match="exceed the softbuffer extent"):
        p.reserve(16, 1, 1, 53)
    with pytest.raises(RuntimeError, match="outside the period"):
        p.run_slot(20480)
    with pytest.raises(RuntimeError, match="bookkeeping-only"):
        p.arrays()
    # expiry across the wrap of the slot counter
    q = miphy.HarqPool(None, 1, 10, 8)
    assert q.reserve(20478, 7, 0, 1)[0] == 0 and q.info(0).expire_slot == 6
    q.run_slot(20479), q.run_slot(5)
    assert q.info(0).state == miphy.HARQ_RESERVED
    q.run_slot(6)
    assert q.info(0).state == miphy.HARQ_AVAILABLE
