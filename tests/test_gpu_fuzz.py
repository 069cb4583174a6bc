"""GPU tier: bounded runs of the two fuzzers (tests/fuzz_path.py: the whole hot path against the CPU oracle on random small
volumes incl. the opt-in sparse field; tests/fuzz_slab.py: the Z-slab job with 2-4 rank threads against the single-GPU path,
three passes per job; tests/fuzz_sort.py: the unique stage on long sort segments) -- so that "0 mismatches" is a record of the test run, not a claim.  Longer runs by hand:
`python tests/fuzz_path.py SEED CASES`, `python tests/fuzz_slab.py SEED CASES`."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def test_path_fuzz_bounded(capsys):
    import fuzz_path
    assert fuzz_path.run(seed=3, cases=24) == 0, capsys.readouterr().out


def test_sort_fuzz_bounded(capsys):
    """tests/fuzz_sort.py: the unique stage's kernel on segments of hundreds to thousands of vertices (merge rounds, the two-half
    path, the clamped run, the library path for longer ones, ties across buckets)."""
    import fuzz_sort
    assert fuzz_sort.run(seed=5, cases=15) == 0, capsys.readouterr().out


def test_slab_fuzz_bounded(capsys):
    import fuzz_slab
    assert fuzz_slab.run(seed=3, cases=20) == 0, capsys.readouterr().out
