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
    # precise mode (include/dvhip.h dv_model_is_precise): hi + lo activations through the 17x17 and 8x8 stages; the
    # default for inputs of more than 8 channels, DV_PRECISE=0 / 1 in the environment overrides at construction
    self.precise = bool(_lib.lib().dv_model_is_precise(self._handle))
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

  def calibrate_for_checkpoint(self, n_images: int = 256, cache_prefix: Optional[str] = None) -> Optional[np.ndarray]:
    """Model preparation, part of LOADING a checkpoint: calibrates the shifts (dv_model_calibrate) on the fixed
    synthetic calibration set of this input shape (calibration_set.draw: the same images on every rank, in every
    run), so that the probabilities stay a pure function of (checkpoint, image) -- batch size, example order,
    `--task` split and rank layout do not enter, as in the reference (deepvariant/call_variants.py:904-932).  What
    call_variants and make_examples' fused route do after loading weights (`--calibration_examples`, 0 = off).

    `cache_prefix` (the checkpoint's path): the corrections are kept next to it in a file named by the content
    hash of the weights, the shape, the set's version and size -- whoever finds the file applies it
    (dv_model_apply_corrections) instead of measuring again; written atomically, never locked, safe to share
    between jobs because the name says everything the content depends on.  Returns the corrections, or None when
    the shape has no calibration set (the model then stays plain fp16 -- on every rank alike)."""
    from deepvariant_amd import calibration_set
    import os
    import zlib
    if self.flat_weights is None:
      raise ValueError('calibrate_for_checkpoint() needs load_flat_weights() first')
    self.calibration = {'images': 0}
    if n_images <= 0 or not calibration_set.supported(self.input_shape):
      return None
    n_corr = sum(co for _, _, _, co, _ in self.layer_table())
    path = None
    if cache_prefix:
      stamp = zlib.crc32(self.flat_weights.tobytes()) & 0xffffffff
      path = '%s.dvcal-v%d-%08x-%dx%dx%d-n%d.f32' % ((cache_prefix, calibration_set.SET_VERSION, stamp) +
                                                  tuple(self.input_shape) + (n_images,))
      try:
        corr = np.fromfile(path, np.float32)
        if corr.size == n_corr and np.isfinite(corr).all():
          self.apply_corrections(corr)
          self.calibration = {'images': n_images, 'set_version': calibration_set.SET_VERSION, 'cached': True}
          return corr
      except OSError:
        pass
    images = calibration_set.draw(self.input_shape, n_images, device=self.device_index)
    corr = self.calibrate(images)
    self.calibration = {'images': n_images, 'set_version': calibration_set.SET_VERSION, 'cached': False}
    if path:
      try:
        tmp = '%s.tmp%d' % (path, os.getpid())
        corr.tofile(tmp)
        os.replace(tmp, path)
      except OSError:
        pass                                   # read-only model directory: every process measures for itself
    return corr

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
  def forward(self, images: torch.Tensor, rows_used: Optional[torch.Tensor] = None, rows_add: int = 0) -> torch.Tensor:
    """`rows_used` (CUDA int32 [N], optional; dv_model_infer_rows): the caller's promise that image i is all zero from
    row rows_used[i] + rows_add on -- the encoder's `rows` output + the reference band height for images it has just
    drawn; blank-row skipping then needs no scan of the images."""
    if images.dtype != torch.uint8 or not images.is_cuda:
      raise ValueError('images must be a CUDA uint8 tensor [N, H, W, C]')
    if tuple(images.shape[1:]) != self.input_shape:
      # call_variants.py:704-733 raises on a shape mismatch as well
      raise ValueError('input shape %s != model shape %s' %
                       (tuple(images.shape[1:]), self.input_shape))
    images = images.contiguous()
    n = images.shape[0]
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
    if rows_used is not None:
      if rows_used.dtype != torch.int32 or not rows_used.is_cuda or rows_used.numel() < n or not rows_used.is_contiguous():
        raise ValueError('rows_used must be a contiguous CUDA int32 tensor with one entry per image')
      _lib.check(_lib.lib().dv_model_infer_rows(
          self._handle, images.data_ptr(), n, out.data_ptr(), rows_used.data_ptr(), int(rows_add), C.c_void_p(stream)))
    else:
      _lib.check(_lib.lib().dv_model_infer(
          self._handle, images.data_ptr(), n, out.data_ptr(),
          C.c_void_p(stream)))
    return out.clone()

  def set_blank_skip(self, enabled: bool) -> None:
    """dv_model_set_blank_skip: False runs the dense stem (same probabilities, bit for bit)."""
    _lib.check(_lib.lib().dv_model_set_blank_skip(self._handle, 1 if enabled else 0))

  def blank_thresholds(self, n: int) -> Optional[np.ndarray]:
    """int32 [5, n]: the last forward's per-example thresholds (dv_model_blank_thresholds), or None when the model
    does not skip."""
    out = np.zeros((5, n), np.int32)
    rc = _lib.lib().dv_model_blank_thresholds(self._handle, n, out.ctypes.data)
    if rc == _lib.DV_ERR_UNSUPPORTED:
      return None
    _lib.check(rc)
    return out

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
