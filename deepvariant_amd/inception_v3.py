"""call_variants' classifier on MI355X: a torch.nn.Module whose forward is the
hand-written HIP Inception-v3 of libdvhip.so (dv_model_*, include/dvhip.h).

Mirrors `deepvariant/keras_modeling.py:246-336` (`inceptionv3(...)`) at the
level call_variants uses it (deepvariant/call_variants.py:648-763,904-932):
build for an input shape, load weights, map a uint8 image batch to softmax
genotype probabilities.  PyTorch only owns device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np
import torch

from deepvariant_amd import _lib


class InceptionV3(torch.nn.Module):
  """`InceptionV3(input_shape=(H, W, C))`; forward(uint8 NHWC) -> probs [N,3]."""

  def __init__(self, input_shape: Tuple[int, int, int], num_classes: int = 3,
               max_batch: int = 256, device: int = 0):
    super().__init__()
    h, w, c = input_shape
    self.input_shape = (h, w, c)
    self.num_classes = num_classes
    self.max_batch = max_batch
    self.device_index = device
    desc = _lib.DvModelDesc(h, w, c, num_classes, max_batch)
    self._handle = C.c_void_p()
    _lib.check(_lib.lib().dv_model_create(C.byref(desc), device,
                                          C.byref(self._handle)))
    self.num_params = int(_lib.lib().dv_model_num_params(self._handle))
    self.flat_weights = None
    self._out_buffers = {}

  # ---- weights --------------------------------------------------------------
  def layer_table(self) -> List[Tuple[int, int, int, int, int]]:
    """(kh, kw, cin, cout, param_offset) per layer; last = Dense."""
    l = _lib.lib()
    out = []
    for i in range(l.dv_model_num_layers(self._handle)):
      kh, kw, ci, co = (C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32())
      off = C.c_int64()
      _lib.check(l.dv_model_layer_info(
          self._handle, i, C.byref(kh), C.byref(kw), C.byref(ci), C.byref(co),
          C.byref(off)))
      out.append((kh.value, kw.value, ci.value, co.value, off.value))
    return out

  def load_flat_weights(self, flat: np.ndarray):
    """conv HWIO + BN beta/mean/var per conv, then Dense kernel + bias
    (the layout documented at dv_model_load_weights)."""
    flat = np.ascontiguousarray(flat, dtype=np.float32)
    if flat.size != self.num_params:
      raise ValueError('expected %d parameters, got %d' %
                       (self.num_params, flat.size))
    _lib.check(_lib.lib().dv_model_load_weights(self._handle, flat.ctypes.data,
                                                flat.size))
    self.flat_weights = flat

  def apply_corrections(self, corrections: np.ndarray) -> None:
    """dv_model_apply_corrections: the shift corrections another model of the SAME weights measured
    (what `calibrate` returns)."""
    corr = np.ascontiguousarray(corrections, dtype=np.float32)
    _lib.check(_lib.lib().dv_model_apply_corrections(self._handle, corr.ctypes.data, corr.size))

  def _calibrate_or_share(self, images: torch.Tensor, share_key: Optional[str]) -> None:
    """One calibration per (job, GPU, set of weights): with a `share_key` the first process to get here measures and
    publishes the corrections in /dev/shm, the others apply them (host ranks sharing a GPU would otherwise repeat the
    same 0.17 s of device work one after the other)."""
    if not share_key:
      self.calibrate(images)
      return
    import os
    import tempfile
    import time
    import zlib
    base = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    stamp = zlib.crc32(np.ascontiguousarray(self.flat_weights[::1021]).tobytes()) & 0xffffffff
    path = os.path.join(base, 'dvamd-cal-%s-%08x-%dx%dx%d.f32' % ((share_key, stamp) + tuple(self.input_shape)))
    try:
      os.close(os.open(path + '.lock', os.O_CREAT | os.O_EXCL | os.O_WRONLY))
      winner = True
    except FileExistsError:
      winner = False
    if winner:
      import atexit
      atexit.register(lambda: [os.path.exists(f) and os.remove(f) for f in (path, path + '.lock')])
      corr = self.calibrate(images)
      tmp = '%s.tmp%d' % (path, os.getpid())
      corr.tofile(tmp)
      os.replace(tmp, path)
      return
    deadline = time.monotonic() + 60.0
    while not os.path.exists(path) and time.monotonic() < deadline:
      time.sleep(0.002)
    if os.path.exists(path):
      self.apply_corrections(np.fromfile(path, np.float32))
    else:                      # the publishing process died: measure here after all
      self.calibrate(images)

  def enable_auto_calibration(self, min_images: int = 64, max_images: int = 256, share_key: Optional[str] = None) -> None:
    """Model preparation inside a run: the FIRST forward that brings at least `min_images` examples calibrates the
    shifts on up to `max_images` of them (dv_model_calibrate) before it classifies; forwards before that (tiny
    inputs) run the uncalibrated fp16 model.  What call_variants and make_examples' fused route switch on after
    loading a checkpoint (`--calibration_examples`, 0 = off): deterministic for a given input, a few hundred
    milliseconds once per run (profiles/r05_cnn_tail.txt: why)."""
    self._auto_cal = (int(min_images), int(max_images), share_key) if max_images > 0 else None
    self.calibrated_on = 0

  def calibrate(self, images: torch.Tensor) -> np.ndarray:
    """dv_model_calibrate: moves every layer's fp32 shift by the per-channel mean of the fp16
    pipeline's error on `images` (CUDA uint8 [N,H,W,C], a few hundred examples drawn like the inputs
    the model will see).  Needs the weights given to load_flat_weights.  Returns the corrections
    (cout values per conv layer in layer order, then the logit corrections)."""
    if self.flat_weights is None:
      raise ValueError('calibrate() needs load_flat_weights() first')
    if images.dtype != torch.uint8 or not images.is_cuda or tuple(images.shape[1:]) != self.input_shape:
      raise ValueError('images must be a CUDA uint8 tensor [N, %d, %d, %d]' % self.input_shape)
    images = images.contiguous()
    n_corr = sum(co for _, _, _, co, _ in self.layer_table())
    corr = np.zeros(n_corr, np.float32)
    torch.cuda.synchronize(images.device)
    _lib.check(_lib.lib().dv_model_calibrate(
        self._handle, self.flat_weights.ctypes.data, self.flat_weights.size, images.data_ptr(),
        images.shape[0], corr.ctypes.data, corr.size))
    return corr

  def init_random(self, seed: int = 0):
    """Seeded He-normal kernels / randomised BN statistics (no checkpoint is
    available offline; the reference's own tests do the same,
    deepvariant/call_variants_test.py:109-127)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    flat = np.zeros(self.num_params, np.float32)
    table = self.layer_table()
    for kh, kw, ci, co, off in table[:-1]:
      n = kh * kw * ci * co
      flat[off:off + n] = rng.standard_normal(n) * np.sqrt(2.0 / (kh * kw * ci))
      flat[off + n:off + n + co] = rng.standard_normal(co) * 0.1
      flat[off + n + co:off + n + 2 * co] = rng.standard_normal(co) * 0.1
      flat[off + n + 2 * co:off + n + 3 * co] = rng.random(co) + 0.5
    kh, kw, ci, co, off = table[-1]
    # small, column-centred head weights: post-ReLU features are all positive, so random
    # columns would put class-dependent offsets of several units on the logits and the
    # softmax would saturate -- a parity check on saturated outputs sees nothing
    dense = rng.standard_normal((ci, co)) * 0.02
    dense -= dense.mean(axis=0, keepdims=True)
    flat[off:off + ci * co] = dense.reshape(-1)
    flat[off + ci * co:off + ci * co + co] = rng.standard_normal(co) * 0.1
    self.load_flat_weights(flat)
    return flat

  # ---- forward --------------------------------------------------------------
  def forward(self, images: torch.Tensor) -> torch.Tensor:
    if images.dtype != torch.uint8 or not images.is_cuda:
      raise ValueError('images must be a CUDA uint8 tensor [N, H, W, C]')
    if tuple(images.shape[1:]) != self.input_shape:
      # call_variants.py:704-733 raises on a shape mismatch as well
      raise ValueError('input shape %s != model shape %s' %
                       (tuple(images.shape[1:]), self.input_shape))
    images = images.contiguous()
    n = images.shape[0]
    auto = getattr(self, '_auto_cal', None)
    if auto is not None and n >= auto[0] and self.flat_weights is not None:
      self._auto_cal = None
      self.calibrated_on = min(n, auto[1])
      torch.cuda.current_stream(images.device).synchronize()
      self._calibrate_or_share(images[:self.calibrated_on], auto[2])
    # dv_model_infer replays the forward as a hipGraph keyed by (n, stream) -- the image and
    # output pointers travel through a device-side table, so fresh tensors replay the same
    # graph.  The output lives in a model-owned buffer per batch size; callers get their own
    # copy (n x 3 floats).
    out = self._out_buffers.get((n, images.device))
    if out is None:
      if len(self._out_buffers) >= 8:
        self._out_buffers.clear()
      out = torch.empty((n, self.num_classes), dtype=torch.float32, device=images.device)
      self._out_buffers[(n, images.device)] = out
    stream = torch.cuda.current_stream(images.device).cuda_stream
    _lib.check(_lib.lib().dv_model_infer(
        self._handle, images.data_ptr(), n, out.data_ptr(),
        C.c_void_p(stream)))
    return out.clone()

  def graph_stats(self) -> Tuple[int, int]:
    """(forwards captured into a new hipGraph, forwards replayed from the cache)."""
    cap, rep = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().dv_model_graph_stats(self._handle, C.byref(cap), C.byref(rep)))
    return cap.value, rep.value

  @property
  def conv_macs_per_example(self) -> int:
    return int(_lib.lib().dv_model_conv_macs(self._handle))

  def debug_tensor(self, index: int, n: int) -> np.ndarray:
    h, w, c = C.c_int32(), C.c_int32(), C.c_int32()
    l = _lib.lib()
    l.dv_model_debug_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p] + [C.c_void_p] * 3
    _lib.check(l.dv_model_debug_tensor(self._handle, index, n, None,
                                       C.byref(h), C.byref(w), C.byref(c)))
    # device layout is channel-blocked [N][C/8][H][W][8]; return NHWC
    raw = np.zeros((n, c.value // 8, h.value, w.value, 8), np.float16)
    _lib.check(l.dv_model_debug_tensor(self._handle, index, n, raw.ctypes.data,
                                       C.byref(h), C.byref(w), C.byref(c)))
    out = np.ascontiguousarray(raw.transpose(0, 2, 3, 1, 4)).reshape(
        n, h.value, w.value, c.value)
    return out  # padded plane: interior is out[:, halo:-halo, halo:-halo]

  def __del__(self):
    try:
      if self._handle:
        _lib.lib().dv_model_destroy(self._handle)
    except Exception:  # pylint: disable=broad-except
      pass
