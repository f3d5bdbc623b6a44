"""CPU: how list size 1 of the list decoder is tied to the reference.
(1) A successive-cancellation decoder written from the definition (oracle/phy_oracle.c::orc_polar_sc_textbook: the code tree
    recursion with the reference's LLR algebra, no node pruning, no list) gives the message of the reference's simplified decoder
    (orc_polar_decode_chain, pinned against the reference in test_oracle_vs_ref.py) on every codeword in which no information leaf
    sees an LLR of exactly zero; a zero is a tie which SSC (threshold on a rate-1 node's input) and SC (threshold at the leaf) break differently --
    that divergence is principled, counted here, and the only one.
(2) The list decoder's restatement with L = 1 (orc_polar_scl_decode, which prunes all-frozen blocks) equals the textbook SC always:
    its decisions do not depend on the path metric.
L > 1 has no reference counterpart and stays parity unpinned."""
import numpy as np
import pytest

import oracle_lib as O


def polar_cases():
    cases = []
    for A in (12, 40, 70, 140):
        for AL in (1, 2, 4, 8, 16):
            if A + 24 < 108 * AL:
                cases.append((A + 24, 108 * AL, 9, 0))
    cases.append((56, 864, 9, 0))
    for K, E in ((18, 60), (20, 100), (25, 300), (31, 64), (40, 100), (100, 200), (200, 1000), (500, 1500), (1023, 2000), (64, 8192), (300, 400), (22, 500),
                 (19, 29)):
        for ibil in (0, 1):
            cases.append((K, E, 10, ibil))
    return cases


def stimuli(K, E, nMax, ibil, rng, nb=6):
    msgs = rng.integers(0, 2, (nb, K), dtype=np.uint8)
    llrs = np.zeros((nb, E), np.int8)
    for i in range(nb):
        tx = O.o_polar_encode_chain(K, E, nMax, ibil, msgs[i])[0]
        if i < 2:
            llrs[i] = (1 - 2 * tx.astype(np.int16)) * 10  # noiseless, like polar_chain_test.cpp:192-195 (scaled)
        elif i < 4:
            y = (1.0 - 2.0 * tx) + [0.7, 1.0][i - 2] * rng.standard_normal(E)
            llrs[i] = np.round(np.clip(4 * y, -20, 20) / 20 * 120)
        else:
            v = rng.integers(-120, 121, E)
            v[v == 0] = 1
            v[rng.random(E) < 0.05] = 127
            v[rng.random(E) < 0.05] = -127
            llrs[i] = v
    return msgs, llrs


def test_textbook_sc_equals_reference_style_ssc_unless_a_tie_occurs():
    rng = np.random.default_rng(2025)
    total = ties = differing = 0
    for K, E, nMax, ibil in polar_cases():
        msgs, llrs = stimuli(K, E, nMax, ibil, rng)
        for i in range(llrs.shape[0]):
            sc, zero = O.o_polar_sc_textbook(K, E, nMax, ibil, llrs[i])
            ssc = O.o_polar_decode_chain(K, E, nMax, ibil, llrs[i])[0]
            total += 1
            ties += int(zero)
            if not np.array_equal(sc, ssc):
                differing += 1
                assert zero, ("SC and SSC differ although no zero LLR was met", K, E, nMax, ibil, i)
            if i < 2:
                assert np.array_equal(sc, msgs[i]) and np.array_equal(ssc, msgs[i])  # noiseless: both recover the message
            # list size 1 of the list decoder's restatement: same decisions as the textbook SC, always
            l1 = O.o_polar_scl_decode(K, E, nMax, ibil, 1, 0, 0, llrs[i])[0]
            assert np.array_equal(l1, sc), (K, E, nMax, ibil, i)
    assert total - ties > total // 3, (total, ties)  # most codewords are tie-free, so the equality above has teeth
    print("codewords %d, with a zero LLR at an information leaf %d, SC != SSC on %d of those" % (total, ties, differing))
