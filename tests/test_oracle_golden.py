"""CPU: the C restatement replayed against fixtures frozen from the reference build (tests/golden/make_golden.py).
Works without /root/reference and without oracle/_ref: inputs are regenerated from the seed, expected outputs are the
SHA-256 digests of what the reference's own C produced."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402


def test_oracle_matches_frozen_reference_outputs(oracle):
    gold = np.load(os.path.join(HERE, "golden", "reference_digests.npz"))
    cases = MG.gen_cases(oracle, np.random.default_rng(20260924))
    seen = {}
    for fam, args in cases:
        k = seen.get(fam, 0)
        assert MG.digest(MG.run_case(oracle, fam, args)) == str(gold[fam][k]), (fam, k, {a: v for a, v in args.items() if np.isscalar(v)})
        seen[fam] = k + 1
    assert {f: n for f, n in seen.items()} == {f: len(gold[f]) for f in gold.files}
    assert sum(seen.values()) > 1500
