"""Re-runs the seeded GPU parity tests with shifted seeds: every np.random.default_rng(seed) in the tests becomes
default_rng(seed + 1000 * k) for k = 1..K, so the same oracle comparisons see fresh random cases (allocations, noise, sizes
drawn from the seed). One process, sequential (the GPU box allows few processes on the card).
usage (GPU box): python tools/fuzz_seeds.py [K] [pytest node ids ...]"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DEFAULT = ["tests/test_ldpc_decode_gpu.py", "tests/test_ldpc_chain_gpu.py", "tests/test_sch_gpu.py", "tests/test_pusch_demod_gpu.py", "tests/test_pdsch_mod_gpu.py",
           "tests/test_pdsch_proc_gpu.py", "tests/test_polar_gpu.py", "tests/test_ofh_iq_gpu.py", "tests/test_harq_pool_gpu.py", "tests/test_chest_gpu.py",
           "tests/test_ofdm_gpu.py", "tests/test_pusch_proc_gpu.py", "tests/test_pdcch_proc_gpu.py", "tests/test_ssb_proc_gpu.py", "tests/test_csi_rs_gpu.py",
           "tests/test_equalizer_gpu.py", "tests/test_ulsch_demux_gpu.py", "tests/test_pusch_uci_gpu.py"]


# Scenario tests: besides their parity assertions they assert a particular outcome of a borderline transmission (first attempt fails,
# a retransmission recovers), which holds for their own seed only; the parity they check is covered by the tests above.
SCENARIO = ["tests/test_harq_pool_gpu.py::test_pool_backed_harq", "tests/test_pusch_proc_gpu.py::test_retransmission_through_the_processor"]


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    files = sys.argv[2:] or [f for f in DEFAULT if os.path.exists(os.path.join(ROOT, f))]
    files = [os.path.relpath(f, ROOT) if os.path.isabs(f) else f for f in files]
    orig = np.random.default_rng
    failed = []
    for k in range(1, K + 1):
        np.random.default_rng = lambda seed=None, _k=k: orig(None if seed is None else seed + 1000 * _k)
        os.chdir(ROOT)
        rc = pytest.main(["-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + ["--deselect=" + t for t in SCENARIO] + list(files))
        print("seed shift %d: exit %d" % (1000 * k, rc), flush=True)
        if rc != 0:
            failed.append(k)
    np.random.default_rng = orig
    print("FUZZ", "FAILED shifts %s" % failed if failed else "PASSED", K, "shifts")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
