import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from dp_launch import run_world
extra = ['--global-batch', 5, '--grad-batch', 4, '--length', 200, '--steps', 16, '--trunc', 4, '--stacks', 4]
for w in (1, 2):
    t = time.time()
    try:
        r = run_world(w, 'gpu', f'/tmp/dpdbg{w}', extra, timeout=160)
        print('world', w, 'ok', time.time() - t, r[0]['norm'], flush=True)
    except BaseException as e:
        print('world', w, 'FAILED', time.time() - t, repr(e)[:6000], flush=True)
        break
