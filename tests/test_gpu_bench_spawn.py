"""`python bench.py --gpus 2` with no launcher around it starts its own ranks (before touching the GPU) and rank 0 prints the one
JSON line; on a 1-GPU box the two ranks share the device and rendezvous over gloo (bench.py reports the backend it used)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--small', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['value'] > 0
    import torch
    if torch.cuda.device_count() >= 2:
        assert out['dist_backend'] == 'nccl' and out['rccl_ranks'] == 2
    else:
        assert out['dist_backend'] == 'gloo' and out['rccl_ranks'] is None
    # after the timed steps every rank holds bit-identical weights (same start, same all-reduced gradients, same Adam), and the
    # communication stream has its hardware queue: three compute streams under world > 1
    assert out['replicas_identical'] is True and out['compute_streams'] == 3
    for k in ('roofline', 'forward_only'):
        assert k in out
    assert out['roofline']['frac'] <= 1.0


def test_bench_rejects_a_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--small'], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr
