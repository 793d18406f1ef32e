"""Data-parallel layer on the real HIP path: two ranks share cuda:0 (gpurun boxes have one GPU; two RCCL ranks
cannot share a device, so the collective runs over gloo on device tensors — dp.py is backend-agnostic)."""
import pytest
import torch

from dp_launch import run_world

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
@pytest.mark.parametrize('hip_dtype', ['fp32', 'bf16'])
def test_world2_matches_world1_on_gpu(tmp_path, hip_dtype):
    extra = ['--global-batch', 5, '--grad-batch', 4, '--length', 200, '--steps', 16, '--trunc', 4, '--stacks', 4,
             '--hip-dtype', hip_dtype]
    one = run_world(1, 'gpu', str(tmp_path / 'w1'), extra)[0]
    two = run_world(2, 'gpu', str(tmp_path / 'w2'), extra)
    assert one['tokens'].shape == (5, 200) and int(one['tokens'].max()) <= 3
    for r in two:
        assert torch.equal(r['tokens'], one['tokens'])            # bit-exact: row j is keyed by its global index
        assert r['views_ok']
    assert torch.equal(torch.cat([r['mine'] for r in two]), one['tokens'])
    assert torch.equal(two[0]['flat'], two[1]['flat'])
    # Gradients.  The convolution weights (99 % of the bucket) come from deterministic split-K reductions: a repeated
    # world-2 run gives the same bits, and world 2 differs from world 1 only by where the sum over rows is cut (two ranks
    # of 2 rows, then the mean, against one pass over 4 rows): fp32 summation order, 1e-6 of the largest gradient.  The
    # small vectors (LayerNorm affine, biases, time path, stem) still gather per-workgroup sums with fp32 atomics on the
    # fp32 path, so their order is run-dependent: 1e-5.
    import drakes_amd
    from drakes_amd import dp
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=200, steps=16, num_cnn_stacks=4, truncate_steps=4))
    conv = torch.zeros(one['flat'].numel(), dtype=torch.bool)
    off = 0
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        if name.startswith('backbone.convs.') and name.endswith('.weight'):
            conv[off:off + p.numel()] = True
        off += (p.numel() + 63) // 64 * 64
    assert off == one['flat'].numel() and int(conv.sum()) == 20 * 9 * 128 * 128
    scale = float(one['flat'].abs().max())
    d = (two[0]['flat'] - one['flat']).abs()
    print(f'world 2 vs world 1: conv weights {float(d[conv].max()) / scale:.2e}, other parameters {float(d[~conv].max()) / scale:.2e} of the largest gradient')
    assert float(d[conv].max()) <= 1e-6 * scale          # measured 6e-8
    assert float(d[~conv].max()) <= 1e-5 * scale         # measured 1.2e-7
    assert abs(two[0]['norm'] - one['norm']) <= 1e-4 * one['norm']
    again = run_world(2, 'gpu', str(tmp_path / 'w2b'), extra)
    assert torch.equal(again[0]['local_flat'][conv], two[0]['local_flat'][conv])      # bitwise reproducible
    assert torch.equal(again[0]['flat'][conv], two[0]['flat'][conv])


@pytest.mark.timeout(600)
def test_fine_tune_world2_matches_world1(tmp_path):
    """drakes_amd.fine_tune under data parallelism: two ranks with 2 rows each take the optimizer step one process takes
    with 4 rows (same global rows, same noise; the bucket is all-reduced once, after the last micro-step, then clipped)."""
    extra = ['--global-batch', 4, '--grad-batch', 4, '--length', 200, '--steps', 10, '--trunc', 3, '--stacks', 4, '--finetune']
    one = run_world(1, 'gpu', str(tmp_path / 'w1'), extra)[0]
    two = run_world(2, 'gpu', str(tmp_path / 'w2'), extra)
    lr = 1e-4
    assert torch.equal(two[0]['ft_delta'], two[1]['ft_delta'])               # both ranks took the same step
    d = (two[0]['ft_delta'] - one['ft_delta']).abs()
    print('mean |d|/lr', float(d.mean()) / lr, 'fraction beyond 0.1 lr', float((d > 0.1 * lr).float().mean()))
    assert float(d.mean()) <= 0.01 * lr and float((d > 0.1 * lr).float().mean()) <= 0.01
    assert abs(two[0]['ft_norm'] - one['ft_norm']) <= 1e-3 * one['ft_norm']
    mean_two = 0.5 * (two[0]['ft_losses'] + two[1]['ft_losses'])
    assert torch.allclose(mean_two, one['ft_losses'], rtol=1e-4, atol=1e-5)


@pytest.mark.timeout(300)
def test_rccl_world1_allreduce_of_grad_bucket():
    """The 'nccl' (RCCL) branches of dp.py, which every other DP test replaces by gloo: one rank, one GPU."""
    import os
    import subprocess
    import sys
    from dp_launch import HERE, free_port
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONUNBUFFERED='1')
    p = subprocess.run([sys.executable, os.path.join(HERE, 'rccl_world1.py')], env=env, capture_output=True, timeout=240)
    out = p.stdout.decode(errors='replace') + p.stderr.decode(errors='replace')
    assert p.returncode == 0, out[-3000:]
    assert 'RCCL world-1 ok ranks_seen=1' in out
