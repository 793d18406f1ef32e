import os, sys, torch
sys.path.insert(0, os.getcwd())
from drakes_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
w = torch.randn(128, 5, 9, device=dev) * 0.3
wt = ops.pack_stem_weight(w); b = torch.randn(128, device=dev)
import torch.nn.functional as F
for B, L in ((1, 1), (3, 200), (2, 77), (5, 1024), (256, 1024)):
    x = torch.randint(0, 5, (B, L), device=dev)
    soft = F.one_hot(x, 5).float()
    a = ops.stem_fwd(x, wt, b); c = ops.stem_fwd(soft, wt, b)
    print(B, L, 'token kernel == one-hot soft kernel bitwise:', torch.equal(a, c), flush=True)
    if B == 256:
        for name, inp in (('tok', x), ('soft', soft)):
            for _ in range(3): ops.stem_fwd(inp, wt, b)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for s, e in evs:
                s.record(); ops.stem_fwd(inp, wt, b); e.record()
            torch.cuda.synchronize()
            print(name, sorted(s.elapsed_time(e) for s, e in evs)[5] * 1e3, 'us')
