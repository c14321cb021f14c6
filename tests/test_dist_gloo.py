"""N > 1 path on CPU: world_size-2 gloo (no GPU needed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from deepvariant_amd import dist as dvd
  regions = list(range(11))
  mine = dvd.regions_for_rank(regions, rank, world)
  # each rank "classifies" its regions: 2 candidates per region
  ids = torch.tensor([r * 10 + k for r in mine for k in range(2)], dtype=torch.int64)
  probs = torch.stack([torch.tensor([0.1, 0.2, 0.7]) * 0 + (i % 7) / 10.0
                       for i in ids.tolist()]).float()
  all_p, all_i = dvd.gather_call_outputs(probs, ids)
  q.put((rank, mine, all_i.tolist(), all_p[:, 0].tolist()))
  dist.destroy_process_group()


def test_shards_and_all_gather_world2():
  world = 2
  port = _free_port()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  results = [q.get(timeout=120) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  results.sort()
  # the reference's round-robin rule
  assert results[0][1] == [0, 2, 4, 6, 8, 10] and results[1][1] == [1, 3, 5, 7, 9]
  # every rank holds every candidate exactly once, with the right payload
  for _, _, ids, p0 in results:
    assert sorted(ids) == sorted(r * 10 + k for r in range(11) for k in range(2))
    for i, v in zip(ids, p0):
      assert abs(v - (i % 7) / 10.0) < 1e-6
  assert results[0][2] == results[1][2]


def _records_worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from deepvariant_amd import dist as dvd
  if rank == 0:
    mine = [bytes([i % 251]) * (1 + (7 * i) % 300) for i in range(57)] + [b'']
  else:
    mine = [b'rank1-%d' % i for i in range(3)]
  got = dvd.gather_records(mine, max_chunk_bytes=1000)      # several payload chunks
  empty = dvd.gather_records([])                            # nobody has anything
  q.put((rank, got, empty))
  dist.destroy_process_group()


def test_gather_records_world2():
  """Variable-length records, uneven counts, an empty record, chunked payload."""
  world = 2
  port = _free_port()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_records_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  results = sorted(q.get(timeout=120) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  want0 = [bytes([i % 251]) * (1 + (7 * i) % 300) for i in range(57)] + [b'']
  want1 = [b'rank1-%d' % i for i in range(3)]
  for _, got, empty in results:
    assert got == [want0, want1]
    assert empty == [[], []]


def _big_records_worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from deepvariant_amd import dist as dvd
  mine = [bytes([(rank * 37 + i) % 251]) * (50 + (11 * i) % 200) for i in range(400 + 30 * rank)] + [b'']
  assert sum(map(len, mine)) > dvd._SHM_MIN_BYTES          # pylint: disable=protected-access
  import glob
  before = set(glob.glob('/dev/shm/dvamd-*'))
  shm = dvd.gather_records(mine)                   # host ranks of one node: through /dev/shm
  os.environ['DV_NO_SHM_EXCHANGE'] = '1'
  tcp = dvd.gather_records(mine, max_chunk_bytes=7000)     # the collective, several chunks
  dist.barrier()
  left = set(glob.glob('/dev/shm/dvamd-*')) - before
  q.put((rank, shm == tcp, [len(x) for x in shm], shm[rank] == mine, sorted(left)))
  dist.destroy_process_group()


def test_gather_records_through_shared_memory_equals_the_collective_world3():
  world = 3
  port = _free_port()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_big_records_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  results = sorted(q.get(timeout=120) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for _, same, counts, own, left in results:
    assert same and own
    assert counts == [401, 431, 461]
    assert left == []                              # the exchange directory is gone


def _shm_write_fails_worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world, timeout=__import__('datetime').timedelta(seconds=60))
  import builtins
  import glob
  from deepvariant_amd import dist as dvd
  mine = [bytes([(rank * 37 + i) % 251]) * 120 for i in range(300)]
  real_open = builtins.open
  def failing_open(path, mode='r', *a, **k):           # rank 1 cannot write its file (ENOSPC on a 64 MB /dev/shm)
    if rank == 1 and 'w' in mode and '/dev/shm/dvamd-' in str(path):
      raise OSError(28, 'No space left on device')
    return real_open(path, mode, *a, **k)
  before = set(glob.glob('/dev/shm/dvamd-*'))
  builtins.open = failing_open
  try:
    got = dvd.gather_records(mine)               # every rank falls back to the collective TOGETHER
  finally:
    builtins.open = real_open
  dist.barrier()
  q.put((rank, [len(x) for x in got], got[rank] == mine, sorted(set(glob.glob('/dev/shm/dvamd-*')) - before)))
  dist.destroy_process_group()


def test_a_rank_that_cannot_write_to_shared_memory_takes_everybody_to_the_collective():
  """a one-sided failure inside the /dev/shm exchange (ENOSPC) used to leave the other ranks in a
  barrier until the backend's timeout; every phase now ends in an agreement, and the job completes."""
  world = 3
  port = _free_port()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_shm_write_fails_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  results = sorted(q.get(timeout=120) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for _, counts, own, left in results:
    assert counts == [300, 300, 300] and own and left == []


def _failing_worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from deepvariant_amd import dist as dvd
  try:
    dvd.gather_records([b'x'] * (rank + 1), failed=(rank == 1))
    q.put((rank, 'returned'))
  except dvd.PeerFailed as e:
    q.put((rank, str(e)))
  dist.destroy_process_group()


def test_a_failed_rank_releases_its_peers_at_the_record_exchange():
  """make_examples --gpus N: a rank whose region loop raised joins the first exchange with a failure flag, so
  the healthy ranks leave with PeerFailed at once instead of waiting for the backend's timeout."""
  world = 2
  port = _free_port()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  results = dict(q.get(timeout=120) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert results[0] == results[1] == 'rank 1 failed before the record exchange'
