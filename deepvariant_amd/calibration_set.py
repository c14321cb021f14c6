"""The calibration set of a checkpoint: a FIXED set of synthetic pileups per model input shape.

The classifier's shift calibration (dv_model_calibrate, include/dvhip.h) measures the per-channel mean of the fp16
pipeline's error on a few hundred images.  Rounds 4-5 took those images from the run (the first >= 64 examples a
forward brought), which made a candidate's probabilities depend on what else was in the run, on the batch size and on
the rank layout.  The reference's call_variants is a pure function of (checkpoint, image)
(deepvariant/call_variants.py:904-932).  Here the images are a property of the MODEL SHAPE alone: `draw(shape)`
generates the same pileups on every rank, in every run -- synthetic reads (synth.py, fixed seed) through the HIP
encoder, bit-exact by construction -- so the corrections are a pure function of (weights, shape) and every process
that loads a checkpoint computes (or reads from the cache next to it) the same numbers.

  C <= 7         synthetic 30x Illumina pile-ups, the first C of make_examples' WGS channel list
  C = 8, 10      synthetic PacBio HiFi 35x (6 + haplotype + 5mC), + the two alt-aligned diff channels at C = 10
  C = 9          synthetic ONT R10.4 50x (6 + haplotype) + the two alt-aligned diff channels
  other          no set: the model stays uncalibrated (plain fp16), identically on every rank
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from deepvariant_amd import _lib
from deepvariant_amd import dv_types as T
from deepvariant_amd import synth

SET_VERSION = 1          # bump when the generator changes: cached corrections are keyed on it
SET_SEED = 600613        # never one of the seeds tests or bench.py evaluate on
DEFAULT_IMAGES = 256


def _illumina_options(h: int, w: int, c: int) -> T.PileupImageOptions:
  o = synth.illumina_options(7, height=h, width=w)
  o.channels = list(T.PILEUP_CHANNELS_WITH_INSERT_SIZE)[:c]
  o.num_channels = c
  return o


def supported(shape: Tuple[int, int, int]) -> bool:
  h, w, c = shape
  return 1 <= c <= 10 and w % 2 == 1 and h >= 8


_BLOCK = 64              # images are generated in blocks of 64 candidates (seed + block number): draw(shape, n) is a
                         # prefix of draw(shape, m) for n <= m


_last = {}               # the last set drawn in this process (a second model of the same shape needs the same images)


def draw(shape: Tuple[int, int, int], n: int = DEFAULT_IMAGES, device: int = 0,
         seed: int = SET_SEED) -> Optional[torch.Tensor]:
  """-> CUDA uint8 [n, H, W, C] (the same bytes for the same arguments, always), or None when the shape has no set."""
  h, w, c = (int(v) for v in shape)
  if not supported((h, w, c)) or n < 1:
    return None
  key = (h, w, c, int(n), int(device), int(seed))
  if _last.get('key') == key:
    return _last['images']
  out = _draw(h, w, c, n, device, seed)
  _last.update(key=key, images=out)
  return out


def _draw(h, w, c, n, device, seed):
  from deepvariant_amd.pileup_image_native import _Encoder
  dev = torch.device('cuda', device)
  out = torch.empty((n, h, w, c), dtype=torch.uint8, device=dev)
  if c <= 7:
    opts = _illumina_options(h, w, c)
    enc = _Encoder(opts, w, device=device)
  else:
    kind = 'ont' if c == 9 else 'hifi'
    opts = synth.longread_options(kind)
    opts.height, opts.width = h, w
    enc = _Encoder(opts, w, device=device) if c == 8 else None
  for k, done in enumerate(range(0, n, _BLOCK)):
    m = min(_BLOCK, n - done)
    if c <= 7:                             # multi-allelic sites are off: one pileup per candidate
      batch = synth.make_illumina_batch(_BLOCK, seed=seed + 7919 * k, options=opts, multi_allelic=False)
      img, _ = enc.encode(batch, c)
      block = torch.from_numpy(np.ascontiguousarray(img.reshape(-1, h, w, c)[:_BLOCK])).to(dev)
    elif c == 8:                           # the drawn HiFi channels alone
      batch = synth.make_longread_batch(_BLOCK, kind, seed=seed + 7919 * k, options=opts)
      img, _ = enc.encode(batch, c)
      block = torch.from_numpy(np.ascontiguousarray(img.reshape(-1, h, w, c)[:_BLOCK])).to(dev)
    else:
      block = longread_examples(kind, _BLOCK, seed=seed + 7919 * k, device=device, options=opts)
    out[done:done + m] = block[:m]
  return out


def longread_examples(kind: str, n: int, seed=None, device: int = 0, options=None) -> torch.Tensor:
  """n examples at the released long-read models' input shapes ('hifi' -> [n,100,147,10], 'ont' -> [n,100,199,9]):
  dv_encode_batch over the reference-aligned and alt-aligned images + dv_merge_alt_channels -- the tensor
  bench.py's hifi35 / ont50 step hands to the classifier."""
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.pileup_image_native import _Encoder
  opts, batch, with_alt, c_enc, ct = synth.make_longread_workload(kind, n, seed=seed, options=options)
  h, w = opts.height, opts.width
  img_bytes = h * w * ct
  entries = (_lib.DvAltMergeEntry * max(len(with_alt), 1))()
  for k, i in enumerate(with_alt):
    entries[k].example, entries[k].first_row, entries[k].rows = i, 0, h
    entries[k].scratch_alt1, entries[k].scratch_alt2 = 2 * k, 2 * k + 1
  dev = torch.device('cuda', device)
  with torch.cuda.device(dev):
    dbatch = DeviceBatch(batch, dev)
    enc = _Encoder(opts, w, device=device)
    flat = torch.zeros(batch.n_items * img_bytes, dtype=torch.uint8, device=dev)
    rows = torch.empty(batch.n_items, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    dbatch.encode(enc, ct, flat, rows)
    _lib.check(_lib.lib().dv_merge_alt_channels(flat.data_ptr(), n * img_bytes, img_bytes, img_bytes, w, ct, c_enc,
                                                5, entries, len(with_alt), C.c_void_p(stream.cuda_stream)))
    torch.cuda.synchronize(dev)
  return flat[:n * img_bytes].view(n, h, w, ct).clone()
