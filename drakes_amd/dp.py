"""Data parallelism for the two sharded paths: one process per GPU, RCCL over xGMI.

* `_sample`: sequences are independent (no cross-sequence term in reference
  diffusion_gosai_update.py:462-550), so the global batch is split by rows and NO collective is on
  the path.  The noise of row j is keyed by its GLOBAL index (sample_offset), so sample j has the
  same tokens at world size 1, 2, 4, 8.
* fine-tune step: both loss terms are batch means (reference finetune_reward_bp.py:83-84), so with
  equal shards the mean of the per-rank gradients is the single-process gradient.  All gradients
  live in ONE flat fp32 bucket (13.3 MB for the 3.33 M-parameter CNN): one all-reduce after the
  last accumulation micro-step, scale by 1/world, THEN clip (reference :89-93 clips before
  Adam.step), so every rank clips by the same norm.  One 13 MB ring all-reduce over xGMI is
  latency-bound; splitting it would only add launches.

world_size == 1 runs the same code with the collective skipped.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None, device=None):
    """Joins the process group torchrun described (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, device).  Without WORLD_SIZE in the environment: (0, 1, device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if device is None:
        device = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
    device = torch.device(device)
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if device.type == 'cuda' else 'gloo'     # 'nccl' is RCCL on ROCm
        kw = {'device_id': device} if backend == 'nccl' else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_range(global_batch, rank, world):
    """(first global row, row count) of `rank`: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f'rank {rank} outside world {world}')
    base, rem = divmod(global_batch, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def sample_sharded(model, global_batch, *, seed, num_steps=None, eps=1e-5, gather=False):
    """This rank's rows of a `global_batch`-row `_sample` call.  `seed` must be the same on all ranks.

    Returns (tokens int64 [count, L], first_row).  gather=True returns all rows on every rank
    ([global_batch, L], 0) through one all_gather of int8 tokens (205 KB at B=1024)."""
    r, w = rank(), world()
    first, count = shard_range(global_batch, r, w)
    x = model._sample(num_steps=num_steps, eps=eps, eval_sp_size=count, seed=seed, sample_offset=first)
    if not gather or w == 1:
        return x, first
    base = -(-global_batch // w)                     # all_gather wants equal pieces: pad to the largest shard
    piece = torch.zeros(base, x.shape[1], dtype=torch.int8, device=x.device)
    piece[:count] = x.to(torch.int8)
    if dist.get_backend() == 'nccl':
        out = torch.empty(w * base, x.shape[1], dtype=torch.int8, device=x.device)
        dist.all_gather_into_tensor(out, piece)
    else:                                            # gloo (tests): list form, staged on the host
        parts = [torch.empty(base, x.shape[1], dtype=torch.int8) for _ in range(w)]
        dist.all_gather(parts, piece.cpu())
        out = torch.cat(parts).to(x.device)
    rows = [out[k * base:k * base + shard_range(global_batch, k, w)[1]] for k in range(w)]
    return torch.cat(rows).to(torch.int64), 0


class GradBucket:
    """All trainable gradients of `module` as views into one flat fp32 tensor.

    autograd accumulates into an existing .grad in place, so after `attach()` every backward
    lands in the bucket and the exchange is a single all-reduce of `flat`."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError('GradBucket wants fp32 parameters on one device')
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64          # 256-byte aligned slices
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.attach()

    def attach(self):
        """(Re)points every p.grad at its slice; call after anything that set grads to None."""
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero_(self):
        self.flat.zero_()
        if any(p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off
               for p, off in zip(self.params, self.offsets)):
            self.attach()

    def allreduce_mean_(self, force=False):
        """Sum over ranks, divide by the world size.  No-op at world size 1 unless `force` (then the collective runs
        anyway, when a group exists: the one-GPU test of the RCCL branch)."""
        w = world()
        if w > 1 or (force and dist.is_initialized()):
            if dist.get_backend() == 'nccl' or self.flat.device.type == 'cpu':
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)          # RCCL ring over xGMI
            else:                       # gloo rehearsal with device tensors (tests): stage on the host
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                self.flat.copy_(host)
            self.flat.mul_(1.0 / w)
        return self.flat

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ on the bucket (reference finetune_reward_bp.py:89): one norm over
        the flat tensor (padding is zero), coefficient clamp(max_norm / (norm + 1e-6), max=1)."""
        total = torch.linalg.vector_norm(self.flat, 2)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total
