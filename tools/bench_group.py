"""Times the bf16 forward of a long batch (BASELINE.json configs[4]: B=256, L=1024) launch by launch: the grouped
launches (drk_layer_group_bf16) under several plans against one launch per layer.  python tools/bench_group.py [B] [L]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import drakes_amd
from drakes_amd import fastpath, ops
from drakes_amd.det_init import det_state_dict


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3     # us


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith('--')]
    B = int(pos[0]) if len(pos) > 0 else 256
    L = int(pos[1]) if len(pos) > 1 else 1024
    dev = torch.device('cuda:0')
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=L)).to(dev)
    m.load_state_dict(det_state_dict(0))
    bb = m.backbone
    x = torch.randint(0, 4, (B, L), device=dev)
    x[:, ::3] = 4
    tbias = bb.time_bias(torch.zeros(1, device=dev))
    out = {'B': B, 'L': L, 'drk_version': drakes_amd._lib.load().drk_version()}
    if '--pmc' in sys.argv:                      # under rocprofv3 --pmc: three forwards of the shipped plan, nothing else
        w0 = {}
        for _ in range(3):
            fastpath.forward_bf16_groups(bb, x, tbias, w0)
        torch.cuda.synchronize()
        return
    work = {}
    out['layers_us'] = timed(lambda: fastpath.forward_bf16_layers(bb, x, tbias, 0, work))
    plans = {'auto': fastpath.group_plan(bb, L), 'h4x16': fastpath.group_plan(bb, L, 4, 16),
             'split444': [(0, 4, 1, 18), (4, 4, 1, 18), (8, 4, 4, 16), (12, 4, 4, 16), (16, 4, 4, 16)],
             'split84': [(0, 4, 1, 18), (4, 4, 1, 18), (8, 8, 4, 16), (16, 4, 4, 16)],
             'h2x18': fastpath.group_plan(bb, L, 2, 18)}
    for name, plan in plans.items():
        w2 = {}
        out[f'groups_{name}_us'] = timed(lambda: fastpath.forward_bf16_groups(bb, x, tbias, w2, plan))
        # launch by launch (the first with the stem, the last with the head)
        launches, _, _ = fastpath.group_launches(bb, x, tbias, w2, plan)
        per = []
        for g, args in launches:
            per.append((g, round(timed(lambda: drakes_amd._lib.call('drk_layer_group_bf16', *args)), 1)))
        out[f'groups_{name}_launches'] = per
    print(json.dumps(out))


if __name__ == '__main__':
    main()
