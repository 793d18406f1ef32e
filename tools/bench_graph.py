"""fp32 `_sample` with and without the captured step graph: python tools/bench_graph.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, drakes_amd
torch.manual_seed(0)
m = drakes_amd.Diffusion(drakes_amd.default_config()).to('cuda:0').eval()
for B in [int(a) for a in sys.argv[1:]] or [8, 64, 256, 1024]:
    res = {}
    for graph in (False, True):
        m.hip_graph = graph
        m._graph = None
        m._sample(eval_sp_size=B, seed=1)
        ts = []
        for r in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m._sample(eval_sp_size=B, seed=2 + r)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[graph] = min(ts)
    print(f'B={B}: plain loop {res[False]*1e3:.1f} ms ({B/res[False]:.0f} seq/s), hipGraph {res[True]*1e3:.1f} ms ({B/res[True]:.0f} seq/s), x{res[False]/res[True]:.2f}', flush=True)
