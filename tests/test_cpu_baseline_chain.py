"""CPU: the reference receive chain that bench.py times as `cpu_baseline` (oracle/ref_capi.cpp::ref_pusch_chain_bench: the
reference's ofdm_slot_demodulator + pusch_processor, one instance per pinned thread, pusch_processor_benchmark.cpp:576-632 style)
recovers slots produced by the ORACLE transmit chain (SCH encoder, scrambler + 256QAM mapper, DM-RS, OFDM modulator): an end-to-end
pin of the restated transmit side against the reference's receiver at the headline configuration (273 PRB, 256QAM R=948/1024)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")

NPRB, MOD, TBS, RNTI, N_ID, SCR = 273, 8, 319784, 0x4601, 935, 1


def _tx_slot(slot, rng):
    nsc, nre = NPRB * 12, NPRB * 156
    tb = rng.integers(0, 256, TBS // 8, dtype=np.uint8)
    cw = O.o_pdsch_encode(1, 0, MOD, 0, 1, nre, tb)
    grid = np.zeros((1, 14, nsc), np.complex64)
    dm = np.zeros(14, np.uint8)
    dm[2] = 1
    n = O.o_pdsch_modulate(RNTI, N_ID, 1.0, 1, [MOD], [cw], 0, 14, dm, 0, 2, 0, NPRB, np.arange(NPRB), [], [0], NPRB, grid)
    assert n == nre
    O.o_dmrs_pdsch_map(slot, 0, 0, SCR, 0, 10 ** (3 / 20), dm, np.ones(NPRB, np.uint8), [0], grid)
    cfg = O.OfdmCfg(1, NPRB, 4096, 0, 1.0 / 64, 3.5e9)
    x = O.o_ofdm_mod_slot(cfg, slot % 2, grid[0])
    sigma = 10 ** (-33 / 20)
    x = x + (sigma * 0.7071 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    return tb, x.astype(np.complex64)


@pytest.mark.timeout(300)
def test_reference_chain_decodes_oracle_slots():
    rng = np.random.default_rng(5)
    slots = [_tx_slot(s, rng) for s in range(2)]
    samples = np.stack([x for _, x in slots])
    cpus, _ = O.host_cpus()
    for stage in (1, 0):
        dt, done, ok = O.r_pusch_chain_bench(2, cpus, 1.0, stage, samples, NPRB, MOD, TBS, RNTI, N_ID, SCR, 4096, 144, 1.0 / 64, 3.5e9, 6, 0)
        assert done >= 2 and ok == done, (stage, done, ok)
        assert 0.9 < dt < 30
    # decoder-only leg on LLRs of the right length (hard +-10: every codeblock decodes)
    cw = O.o_pdsch_encode(1, 0, MOD, 0, 1, NPRB * 156, slots[0][0])
    llr = ((1 - 2 * cw.astype(np.int8)) * 10).astype(np.int8)[None, :]
    dt, done, ok = O.r_pusch_decoder_bench(2, cpus, 0.5, llr, MOD, NPRB * 156, TBS, 6, 0)
    assert done >= 2 and ok == done
