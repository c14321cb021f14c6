"""Large-N CNN parity helpers (GPU box; test infrastructure, never imported by the product).

Images for the large-N tests drawn by the HIP encoder and kept on the GPU; the fp32 oracle on the GPU and the
tail statistics live in oracle/inception_gpu.py (re-exported here).  Used by tests/test_hip_cnn_tail.py,
tests/test_hip_calibration.py and tools/r5_cnn_tail.py.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from oracle import inception_ref as R   # noqa: E402,F401  (checker)
from oracle.inception_gpu import (check_gpu_oracle, fmt, oracle_probs_cpu, oracle_probs_gpu,   # noqa: E402,F401
                                  tail_stats)


def illumina_pileups_gpu(n, seed, chunk=8192, device='cuda'):
  """n encoder-drawn ILLUMINA30 pileups [n,100,221,7] uint8, resident on the GPU (HIP encoder)."""
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(7)
  enc = _Encoder(opts, opts.width)
  out = torch.empty((n, 100, 221, 7), dtype=torch.uint8, device=device)
  done = 0
  k = 0
  while done < n:
    m = min(chunk, n - done)
    batch = synth.make_illumina_batch(m, seed=seed + 7919 * k, options=opts, multi_allelic=False)
    img, _ = enc.encode(batch, 7)
    out[done:done + m] = torch.from_numpy(np.ascontiguousarray(img.reshape(-1, 100, 221, 7)[:m])).to(device)
    done += m
    k += 1
  return out


def longread_images_gpu(kind, n, seed=None, device='cuda'):
  """n examples of bench.py's hifi35 / ont50 workload ('hifi' -> [n,100,147,10], 'ont' -> [n,100,199,9]):
  HIP encoder + dv_merge_alt_channels, exactly the bench step's tensor."""
  import ctypes as CT
  import bench
  from deepvariant_amd import _lib
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.pileup_image_native import _Encoder
  opts, batch, with_alt, c_enc, ct = bench.make_longread_workload(kind, n, seed=seed)
  h, w = opts.height, opts.width
  img_bytes = h * w * ct
  entries = (_lib.DvAltMergeEntry * max(len(with_alt), 1))()
  for k, i in enumerate(with_alt):
    entries[k].example, entries[k].first_row, entries[k].rows = i, 0, h
    entries[k].scratch_alt1, entries[k].scratch_alt2 = 2 * k, 2 * k + 1
  dev = torch.device(device)
  dbatch = DeviceBatch(batch, dev)
  enc = _Encoder(opts, w)
  flat = torch.zeros(batch.n_items * img_bytes, dtype=torch.uint8, device=dev)
  rows = torch.empty(batch.n_items, dtype=torch.int32, device=dev)
  stream = torch.cuda.current_stream(dev)
  dbatch.encode(enc, ct, flat, rows)
  _lib.check(_lib.lib().dv_merge_alt_channels(flat.data_ptr(), n * img_bytes, img_bytes, img_bytes, w, ct, c_enc,
                                              5, entries, len(with_alt), CT.c_void_p(stream.cuda_stream)))
  torch.cuda.synchronize(dev)
  return flat[:n * img_bytes].view(n, h, w, ct).clone()


def hip_probs(model, images, chunk):
  outs = []
  for i in range(0, images.shape[0], chunk):
    outs.append(model(images[i:i + chunk]).cpu())
  return torch.cat(outs).numpy()


