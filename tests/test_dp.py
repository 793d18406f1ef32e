"""Data-parallel layer on CPU: world_size 2 over gloo, the oracle standing in for the HIP sampler."""
import pytest
import torch

from drakes_amd import dp
from dp_launch import run_world


def test_shard_range_partitions_rows():
    for gb in (0, 1, 5, 8, 1023, 1024):
        for w in (1, 2, 3, 4, 8):
            pieces = [dp.shard_range(gb, r, w) for r in range(w)]
            assert pieces[0][0] == 0 and sum(c for _, c in pieces) == gb
            assert all(pieces[k][0] + pieces[k][1] == pieces[k + 1][0] for k in range(w - 1))
            assert max(c for _, c in pieces) - min(c for _, c in pieces) <= 1
    with pytest.raises(ValueError):
        dp.shard_range(8, 2, 2)


def test_grad_bucket_views_and_clip():
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref.load_state_dict(lin.state_dict())
    b = dp.GradBucket(lin)
    x = torch.randn(4, 7)
    for _ in range(2):                                   # accumulation lands in the bucket
        lin(x).square().sum().backward()
        ref(x).square().sum().backward()
    for p, q in zip(lin.parameters(), ref.parameters()):
        assert torch.equal(p.grad, q.grad)
    assert b.allreduce_mean_() is b.flat                 # world size 1: no collective
    n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
    n = b.clip_grad_norm_(0.5)
    assert torch.allclose(n, n_ref, rtol=1e-6)
    for p, q in zip(lin.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-9)
    lin.zero_grad(set_to_none=True)
    b.zero_()                                            # re-attaches
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0 for p in lin.parameters())


@pytest.mark.timeout(900)
def test_world2_matches_world1(tmp_path):
    """Sample j has the same tokens at world size 1 and 2 (uneven 3+2 shards), and the mean of the two ranks'
    gradients is the one-process gradient on the whole batch."""
    one = run_world(1, 'cpu', str(tmp_path / 'w1'))[0]
    two = run_world(2, 'cpu', str(tmp_path / 'w2'))
    assert one['tokens'].shape == (5, 40) and int(one['tokens'].max()) <= 3
    for r in two:
        assert torch.equal(r['tokens'], one['tokens'])           # gathered rows, on every rank
        assert r['views_ok']
    assert [r['mine_first'] for r in two] == [0, 3]
    assert torch.equal(torch.cat([r['mine'] for r in two]), one['tokens'])
    assert torch.equal(two[0]['flat'], two[1]['flat'])            # all ranks hold the same clipped gradient
    assert not torch.equal(two[0]['local_flat'], two[1]['local_flat'])
    scale = float(one['flat'].abs().max())
    assert float((two[0]['flat'] - one['flat']).abs().max()) <= 1e-5 * scale
    assert abs(two[0]['norm'] - one['norm']) <= 1e-5 * one['norm']
    assert abs(0.5 * (two[0]['loss'] + two[1]['loss']) - one['loss']) <= 1e-5 * abs(one['loss'])
