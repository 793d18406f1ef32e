"""Starts `world` copies of tests/dp_worker.py (one per rank, as torchrun would) and loads what they wrote."""
import os
import socket
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def run_world(world, mode, out_dir, extra=(), timeout=170):
    os.makedirs(out_dir, exist_ok=True)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='2', HSA_ENABLE_IPC_MODE_LEGACY='0',
                   GLOO_SOCKET_IFNAME='lo', PYTHONUNBUFFERED='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), '--mode', mode, '--out', out_dir,
                                       *map(str, extra)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=timeout)[0].decode(errors='replace'))
    finally:
        for p in procs:                       # our own children, by PID
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f'rank {r} exited {p.returncode}:\n{logs[r][-3000:]}'
    return [torch.load(os.path.join(out_dir, f'rank{r}.pt'), weights_only=True) for r in range(world)]
