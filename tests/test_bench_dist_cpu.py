"""bench.py's N > 1 machinery on CPU: world_size-2 gloo, stubbed per-rank step.

Drives the same functions `bench.py --gpus N` runs on every rank -- the timed region, the
product's gather (deepvariant_amd/dist.gather_call_outputs) inside the step and the
MAX-over-ranks reduction -- with a step that needs no GPU.  The sharding rule under test
is the reference's (deepvariant/make_examples_core.py:879-888).
"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  sys.path.insert(0, ROOT)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import bench
  from deepvariant_amd import dist as dvd
  regions = dvd.regions_for_rank(list(range(9)), rank, world)
  n = 3 * len(regions)                       # 3 candidates per region: shards differ in size
  ids = torch.arange(n, dtype=torch.int64) + rank * (1 << 24)
  calls = {'n': 0}

  def local_step():
    calls['n'] += 1
    p = torch.zeros((n, 3))
    p[:, 0] = (ids % 97).float() / 100.0
    p[:, 2] = 1.0 - p[:, 0]
    return p

  step, state = bench.make_gather_step(local_step, ids, world, torch.device('cpu'))
  elapsed, probs = bench.timed_steps(step, dist.barrier, warmup=1, steps=3)
  elapsed, total = bench.reduce_elapsed(elapsed + rank, n, world, torch.device('cpu'))
  all_p, all_i = state['all']
  q.put((rank, calls['n'], n, elapsed, total, all_i.tolist(), all_p[:, 0].tolist(), probs.shape[0]))
  dist.destroy_process_group()


def test_bench_rank_functions_world2():
  world = 2
  port = _free_port()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  results = sorted(q.get(timeout=180) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  sizes = [r[2] for r in results]
  assert sizes == [15, 12]                   # regions 0,2,4,6,8 / 1,3,5,7
  for rank, calls, n, elapsed, total, ids, p0, n_local in results:
    assert calls == 4                        # 1 warm-up + exactly 3 timed steps
    assert total == 27.0 and n_local == n    # whole-job item count, local probs returned
    assert elapsed >= 1.0                    # MAX over ranks (rank 1 reported +1 s)
    want = [i + r * (1 << 24) for r, s in enumerate(sizes) for i in range(s)]
    assert ids == want                       # every rank holds every rank's candidates
    for i, v in zip(ids, p0):
      assert abs(v - (i % 97) / 100.0) < 1e-6
  assert abs(results[0][3] - results[1][3]) < 1e-9
