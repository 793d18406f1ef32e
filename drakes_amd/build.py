"""Builds drakes_amd/libdrakes_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m drakes_amd.build [--force]

The library is built IN-TREE so that it travels with the source snapshot to the GPU box;
it is git-ignored.  Nothing is built at import time: drakes_amd._lib raises if it is missing.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdrakes_hip.so')

# file -> extra flags.  The fp32 parity path is built without fma contraction so that
# elementwise arithmetic follows the reference's op-by-op rounding.
SOURCES = {
    'drk_step.hip': ['-ffp-contract=off'],
    'drk_cnn_f32.hip': ['-ffp-contract=off'],
    'drk_sample_bf16.hip': ['-ffp-contract=off'],
    'drk_train_f32.hip': ['-ffp-contract=off'],
    'drk_layer_bf16.hip': ['-ffp-contract=off'],
    'drk_train_bf16.hip': ['-ffp-contract=off'],
    'drk_conv_x3.hip': ['-ffp-contract=off'],
    'drk_train_chain.hip': ['-ffp-contract=off'],
    'drk_train_fused_bf16.hip': ['-ffp-contract=off'],
    'drk_group_bf16.hip': ['-ffp-contract=off'],
}
COMMON = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def sources():
    return [os.path.join(CSRC, f) for f in SOURCES]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'drakes_hip.h'))
    objs = []
    for f, extra in SOURCES.items():
        src = os.path.join(CSRC, f)
        obj = os.path.join(CSRC, f.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + COMMON + extra + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
    print(LIB)
