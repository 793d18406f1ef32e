"""One RCCL rank on cuda:0 (started by tests/test_gpu_dp.py): joins an 'nccl' group of world size 1 and pushes the
fine-tune gradient bucket through the RCCL branches of drakes_amd.dp (GradBucket.allreduce_mean_, the all_gather of
sample_sharded).  A gpurun box has one GPU, and two RCCL ranks cannot share a device, so world 1 is what can run
there; it still loads librccl, creates the communicator and launches the collective kernels."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import drakes_amd                                    # noqa: E402
from drakes_amd import dp                            # noqa: E402


def main():
    faulthandler.dump_traceback_later(120, exit=True)
    rank, world, device = dp.init(backend='nccl', device='cuda:0')
    assert world == 1 and not dist.is_initialized()
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)      # dp.init skips the group at world 1
    assert dist.get_backend() == 'nccl'
    model = drakes_amd.Diffusion(drakes_amd.default_config(length=40, steps=4, num_cnn_stacks=1)).to(device)
    bucket = dp.GradBucket(model)
    g = torch.Generator(device='cpu').manual_seed(3)
    bucket.flat.copy_(torch.randn(bucket.flat.numel(), generator=g))
    before = bucket.flat.clone()
    bucket.allreduce_mean_(force=True)                # dist.all_reduce over RCCL, then * 1/world
    torch.cuda.synchronize()
    assert torch.equal(bucket.flat, before), 'a world-1 all-reduce must return its input'
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    piece = torch.arange(12, dtype=torch.int8, device=device).reshape(3, 4)
    out = torch.empty_like(piece)
    dist.all_gather_into_tensor(out, piece)           # the gather=True branch of sample_sharded
    assert torch.equal(out, piece)
    dist.barrier()
    dist.destroy_process_group()
    print(f'RCCL world-1 ok ranks_seen={int(ones.item())} bucket={bucket.flat.numel()}', flush=True)


if __name__ == '__main__':
    main()
