"""The oracle must reproduce the committed golden result batches byte for byte."""
import pytest

import _golden


@pytest.mark.parametrize("name", _golden.names())
def test_oracle_reproduces_golden(host, oracle, name):
    tasks, arena, cases = _golden.load(host, name)
    assert cases
    for pname, (params, expect) in cases.items():
        got = oracle.pair_batch(params, tasks, nthreads=2)
        assert got.tobytes() == expect.tobytes(), (name, pname)
