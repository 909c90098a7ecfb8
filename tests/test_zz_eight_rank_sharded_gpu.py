"""The eight-rank sharded-optimizer test, in a file of its own that pytest collects LAST: eight processes on one GPU over gloo is the one
harness of the suite whose result is not bit-reproducible from run to run (see the comments in tests/test_dp_gpu.py _worker_shard), and the
driver runs the suite with -x."""
import pytest
import torch

from test_dp_gpu import _run_ranks, _worker_shard


@pytest.mark.gpu
def test_sharded_optimizer_state_eight_ranks_on_one_gpu():
    """the ZeRO-2-style path at W = 8: buckets padded to 8 x 64 elements, rank r owning the r-th eighth of every bucket, the owned slice
    of the summed gradient, the all-reduced clip norm, the all-gather of the updated parameters, the sharded checkpoint (eight ranks
    over gloo on one GPU)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    res = _run_ranks(_worker_shard, 8, extra=("gloo",))
    bad = [r for r in res if r[1] != "ok"]
    # the ranks that failed by themselves first ("Connection closed by peer" is what the others see afterwards)
    assert len(res) == 8 and not bad, "\n".join(f"rank {r[0]}: {r[2][-1500:]}" for r in sorted(bad, key=lambda r: "Connection closed" in r[2]))
