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
  healthy ranks waiting in an all-gather the failed one never joins until the backend's timeout."""
  world = dist.get_world_size(group)
  lengths = torch.tensor([len(r) for r in records], dtype=torch.int64)
  total = int(lengths.sum()) if len(records) else 0
  mine = torch.tensor([-1 if failed else len(records), total], dtype=torch.int64, device=device)
  sizes = torch.zeros((world, 2), dtype=torch.int64, device=device)
  dist.all_gather_into_tensor(sizes.view(-1), mine, group=group)
  sizes = sizes.cpu().tolist()
  bad = [r for r in range(world) if sizes[r][0] < 0]
  if bad:
    raise PeerFailed('rank%s %s failed before the record exchange' % ('s' if len(bad) > 1 else '', ', '.join(map(str, bad))))
  max_n = max(s[0] for s in sizes)
  max_total = max(s[1] for s in sizes)
  if max_n == 0:
    return [[] for _ in range(world)]
  send_len = torch.zeros(max_n, dtype=torch.int64, device=device)
  send_len[:len(records)] = lengths.to(send_len.device)
  recv_len = torch.empty(world * max_n, dtype=torch.int64, device=device)
  dist.all_gather_into_tensor(recv_len, send_len, group=group)
  recv_len = recv_len.view(world, max_n).cpu()
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
