"""Multi-GPU plumbing: one process per GPU, shards by genomic interval, one
final all-gather of the call outputs (RCCL over xGMI on MI355X; `gloo` in the
CPU tests).

The reference's only real parallelism is the same sharding rule
(`deepvariant/make_examples_core.py:879-888`: region i belongs to task
`i % num_shards`), realised there as N independent processes plus files
(`scripts/run_deepvariant.py:457-462`).  Here rank r of an N-GPU node takes the
regions of task r, encodes and classifies them on its own GPU with no
inter-GPU traffic, and the per-candidate results (3 probabilities + a
candidate id, ~20 bytes) are exchanged once.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def regions_for_rank(regions: Sequence, rank: int, world_size: int) -> List:
  """make_examples_core.py:879-888: `i % num_shards == task_id`."""
  return [r for i, r in enumerate(regions) if i % world_size == rank]


def exchange_counts(n: int, device, group=None) -> List[int]:
  """Every rank's candidate count (one small all-gather)."""
  world = dist.get_world_size(group)
  mine = torch.tensor([n], dtype=torch.int64, device=device)
  counts = torch.zeros(world, dtype=torch.int64, device=device)
  dist.all_gather_into_tensor(counts, mine, group=group)
  return [int(c) for c in counts.tolist()]


def gather_call_outputs(probs: torch.Tensor, ids: torch.Tensor, group=None,
                        counts: Sequence[int] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
  """All ranks receive every rank's (probabilities [n_i, 3], ids [n_i]).

  Counts differ per rank, so they are exchanged first and the payload is padded
  to the maximum; two collectives in total, no reduction.  A caller that
  repeats the gather with fixed shard sizes passes `counts` (from
  `exchange_counts`) and pays for the payload collective only -- no host
  synchronisation inside the step.
  """
  world = dist.get_world_size(group)
  if counts is None:
    counts = exchange_counts(probs.shape[0], probs.device, group)
  max_n = max(counts)
  k = probs.shape[1]
  send = torch.zeros((max_n, k + 2), dtype=torch.float32, device=probs.device)
  send[:probs.shape[0], :k] = probs
  # int64 ids travel as two fp32-exact 24-bit halves (ids < 2^48)
  send[:probs.shape[0], k] = (ids & 0xFFFFFF).to(torch.float32)
  send[:probs.shape[0], k + 1] = (ids >> 24).to(torch.float32)
  recv = torch.empty((world * max_n, k + 2), dtype=torch.float32,
                     device=probs.device)
  dist.all_gather_into_tensor(recv, send, group=group)
  recv = recv.view(world, max_n, k + 2)
  out_p, out_i = [], []
  for r in range(world):
    c = counts[r]
    out_p.append(recv[r, :c, :k])
    out_i.append(recv[r, :c, k].to(torch.int64) +
                 (recv[r, :c, k + 1].to(torch.int64) << 24))
  return torch.cat(out_p), torch.cat(out_i)


_SHM_MIN_BYTES = 16384
_shm_seq = [0]


def _gather_records_shm(records, lengths, sizes, group):
  """The payload of gather_records between host ranks of ONE node: every rank writes its records into a file
  under /dev/shm, every rank reads every file.  Returns None -- on EVERY rank together, so that the caller's
  fall-back to the collective is taken by all of them -- unless every rank sees the same boot id, the directory
  rank 0 made and room for everybody's payload, and every write and every read succeeds: each phase ends in an
  all_reduce(MIN) of a success flag instead of a bare barrier (a rank that hit ENOSPC must not leave the others
  waiting for it until the backend's timeout).

  Why: gloo's TCP all-gather of a 129 KB payload between 8 ranks sharing one MI355X box took 4.4-5.3 s on its
  first use and 0.85 s afterwards (16 ranks: 16.6 s; a 6 KB message: 1 ms) -- more than the ranks' whole region
  loop.  The RCCL path (one rank per GPU, device tensors) does not come here."""
  import os
  import shutil
  import numpy as np
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  try:
    with open('/proc/sys/kernel/random/boot_id') as f:
      boot = f.read().strip()
  except OSError:
    boot = ''
  _shm_seq[0] += 1
  token = [None]
  if rank == 0:
    path = '/dev/shm/dvamd-%d-%d' % (os.getpid(), _shm_seq[0])
    try:
      os.makedirs(path, exist_ok=False)
      token[0] = (boot, path)
    except OSError:
      token[0] = ('', '')
  dist.broadcast_object_list(token, src=0, group=group)
  boot0, path = token[0]

  def agreed(ok: bool) -> bool:
    """True only when EVERY rank says ok: no rank leaves this function on its own."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1

  ok = bool(boot) and boot == boot0 and bool(path) and os.path.isdir(path)
  if ok:
    # everybody's payload must fit: Docker's default /dev/shm is 64 MB, a whole-genome gather is hundreds
    try:
      st = os.statvfs(path)
      need = sum(8 * n + total for n, total in sizes)
      ok = st.f_bavail * st.f_frsize >= need + (1 << 20)
    except OSError:
      ok = False
  if not agreed(ok):
    if rank == 0 and path:
      shutil.rmtree(path, ignore_errors=True)
    return None
  out = None
  try:
    wrote = True
    try:
      with open(os.path.join(path, '%d.bin' % rank), 'wb') as f:
        f.write(np.asarray(lengths.numpy(), np.int64).tobytes())
        f.write(b''.join(records))
    except OSError:                  # ENOSPC after all, a vanished directory: all ranks take the collective together
      wrote = False
    if not agreed(wrote):            # (also the barrier: every file is complete before anybody reads)
      return None
    got = []
    try:
      for r in range(world):
        n, total = sizes[r]
        with open(os.path.join(path, '%d.bin' % r), 'rb') as f:
          blob = f.read()
        if len(blob) != 8 * n + total:
          raise OSError('record exchange: rank %d wrote %d bytes, %d announced' % (r, len(blob), 8 * n + total))
        lens = np.frombuffer(blob, np.int64, count=n).tolist()
        at = 8 * n
        recs = []
        for k in lens:
          recs.append(blob[at:at + k])
          at += k
        got.append(recs)
    except OSError:
      got = None
    if agreed(got is not None):      # everybody has read (or everybody falls back): the files may go
      out = got
  finally:
    if rank == 0:
      shutil.rmtree(path, ignore_errors=True)
  return out


class PeerFailed(RuntimeError):
  """Another rank reported a failure at the record exchange (its own exception is raised there)."""


def gather_records(records: Sequence[bytes], device=None, group=None,
                   max_chunk_bytes: int = 1 << 28, failed: bool = False) -> List[List[bytes]]:
  """All ranks receive every rank's list of serialised records (CallVariantsOutput protos of the
  fused route), rank by rank in the order they were produced.

  Record lengths and bytes travel as two padded all-gathers (lengths int64, payload uint8; the
  payload in chunks of at most `max_chunk_bytes` per rank so that the staging buffers stay
  bounded whatever the run's size).  `device`: where the staging tensors live -- a CUDA device
  under RCCL, None (CPU) under gloo.

  `failed=True`: this rank could not produce its records.  The first exchange (counts) carries the
  flag, so every rank leaves the collective at once with `PeerFailed` naming the ranks -- instead of the
  healthy ranks waiting in an all-gather the failed one never joins until the backend's timeout.

  Host ranks (`device=None`, gloo) of ONE node move a payload of more than 16 KB per rank through files under
  /dev/shm instead of the TCP collective (`_gather_records_shm`; DV_NO_SHM_EXCHANGE keeps the collective);
  DV_DIST_DEBUG prints the seconds each phase took on stderr."""
  import os as _os, sys as _sys, time as _time
  _dbg = _os.environ.get('DV_DIST_DEBUG') is not None
  _t = [_time.perf_counter()]

  def _mark(what):
    if _dbg:
      now = _time.perf_counter()
      print('[dv-dist rank %d] %s %.3f s' % (dist.get_rank(group), what, now - _t[0]), file=_sys.stderr, flush=True)
      _t[0] = now
  world = dist.get_world_size(group)
  lengths = torch.tensor([len(r) for r in records], dtype=torch.int64)
  total = int(lengths.sum()) if len(records) else 0
  mine = torch.tensor([-1 if failed else len(records), total], dtype=torch.int64, device=device)
  sizes = torch.zeros((world, 2), dtype=torch.int64, device=device)
  dist.all_gather_into_tensor(sizes.view(-1), mine, group=group)
  sizes = sizes.cpu().tolist()
  _mark('counts exchanged')
  bad = [r for r in range(world) if sizes[r][0] < 0]
  if bad:
    raise PeerFailed('rank%s %s failed before the record exchange' % ('s' if len(bad) > 1 else '', ', '.join(map(str, bad))))
  max_n = max(s[0] for s in sizes)
  max_total = max(s[1] for s in sizes)
  if max_n == 0:
    return [[] for _ in range(world)]
  if device is None and max_total > _SHM_MIN_BYTES and _os.environ.get('DV_NO_SHM_EXCHANGE') is None:
    # host ranks (gloo): when they all sit on one node -- ranks that share a GPU always do -- the payload goes
    # through /dev/shm files and only its bookkeeping through the collective (see _gather_records_shm)
    shared = _gather_records_shm(records, lengths, sizes, group)
    if shared is not None:
      _mark('payload exchanged through /dev/shm (%d bytes per rank at most)' % max_total)
      return shared
  send_len = torch.zeros(max_n, dtype=torch.int64, device=device)
  send_len[:len(records)] = lengths.to(send_len.device)
  recv_len = torch.empty(world * max_n, dtype=torch.int64, device=device)
  dist.all_gather_into_tensor(recv_len, send_len, group=group)
  recv_len = recv_len.view(world, max_n).cpu()
  _mark('lengths exchanged (%d per rank)' % max_n)
  payload = torch.frombuffer(bytearray(b''.join(records)), dtype=torch.uint8) if total else torch.zeros(0, dtype=torch.uint8)
  blobs = [bytearray() for _ in range(world)]
  for lo in range(0, max_total, max_chunk_bytes):
    width = min(max_chunk_bytes, max_total - lo)
    send = torch.zeros(width, dtype=torch.uint8, device=device)
    part = payload[lo:lo + width]
    send[:part.numel()] = part.to(send.device)
    recv = torch.empty(world * width, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, width).cpu()
    for r in range(world):
      have = max(0, min(width, sizes[r][1] - lo))
      blobs[r] += recv[r, :have].numpy().tobytes()
  _mark('payload exchanged (%d bytes per rank)' % max_total)
  out = []
  for r in range(world):
    lens = recv_len[r, :sizes[r][0]].tolist()
    recs, at = [], 0
    blob = bytes(blobs[r])
    for n in lens:
      recs.append(blob[at:at + n])
      at += n
    out.append(recs)
  return out
