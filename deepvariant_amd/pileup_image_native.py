"""Drop-in for `deepvariant.python.pileup_image_native` on MI355X.

Same class, method names, argument meaning and error behaviour as the
reference's pybind module
(`deepvariant/python/pileup_image_native_pybind.cc:82-129`), implemented by
packing the proto-shaped arguments (deepvariant_amd.packing) and calling the
HIP encoder through the C ABI (`dv_encode_batch`, include/dvhip.h).  There is
no CPU path: without libdvhip.so / a GPU every method raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from deepvariant_amd import _lib
from deepvariant_amd import dv_types as T
from deepvariant_amd import packing


class _Encoder:
  """Owns one `dv_encoder*`."""

  def __init__(self, pic_options, width: int, device: int = 0):
    self.opts = packing.make_encoder_options(pic_options, width=width)
    self.handle = C.c_void_p()
    _lib.check(_lib.lib().dv_encoder_create(C.byref(self.opts), device,
                                            C.byref(self.handle)))

  def encode(self, batch: packing.PackedBatch, out_channels: int,
             stream=None, min_bytes: int = 0):
    b, keep = batch.to_ctypes()
    out = np.zeros(max(batch.out_bytes(out_channels), min_bytes), np.uint8)
    rows = np.zeros(max(batch.n_items, 1), np.int32)
    _lib.check(_lib.lib().dv_encode_batch(
        self.handle, C.byref(b), out_channels, out.ctypes.data,
        rows.ctypes.data, _lib.DV_MEM_HOST, stream))
    del keep
    return out, rows[:batch.n_items]

  def __del__(self):
    try:
      if self.handle:
        _lib.lib().dv_encoder_destroy(self.handle)
    except Exception:  # pylint: disable=broad-except
      pass


class PileupImageEncoderNative:
  """`PileupImageEncoderNative(options: PileupImageOptions)`.

  reference: deepvariant/pileup_image_native.h:125-200, .cc:111-123.
  """

  def __init__(self, options, device: int = 0):
    # CHECK((width % 2 == 1) && width >= 3) -- pileup_image_native.cc:114
    if not (options.width % 2 == 1 and options.width >= 3):
      raise ValueError('Width must be odd; found %d' % options.width)
    self._options = options
    self._device = device
    self._channel_enums = packing.channel_enums(options)
    # CHECK_LE(channel_enums_.size(), options_.num_channels()) -- :122
    if options.num_channels and len(self._channel_enums) > options.num_channels:
      raise ValueError('more channels than num_channels')
    self._need_aux = any(e in packing._READ_AUX_SLOT
                         for e in self._channel_enums)
    self._need_list_aux = any(e in packing._LIST_AUX_CHANNELS
                              for e in self._channel_enums)
    # (channel of base_aux0, base_aux1, base_aux2) or (): the per-base host-computed channels and the plane each reads
    self._need_seq_aux = packing.seq_aux_planes(self._channel_enums)
    self._need_ref_aux = any(e in packing._REF_AUX_CHANNELS for e in self._channel_enums)
    if sum(e in packing._LIST_AUX_CHANNELS for e in self._channel_enums) > 1:
      raise NotImplementedError(
          'allele_frequency / read_supports_variant_fuzzy / allele_sample_probability together in one channel '
          'set: the packed batch carries one host-computed pixel per (candidate, read)')
    self._encoders: Dict[int, _Encoder] = {}

  # ------------------------------------------------------------------ helpers
  def _encoder(self, width: int) -> _Encoder:
    if width not in self._encoders:
      self._encoders[width] = _Encoder(self._options, width, self._device)
    return self._encoders[width]

  @property
  def num_channels(self) -> int:
    return len(self._channel_enums)

  def all_channels_enum(self, alt_aligned_pileup: str) -> List[int]:
    """AllChannelsEnum -- pileup_image_native.cc:125-151."""
    out = list(self._channel_enums)
    if alt_aligned_pileup == 'diff_channels':
      out += [9, 10]
    elif alt_aligned_pileup == 'base_channels':
      out += [20, 21]
    return out

  def _list_aux(self, dv_call, alt_alleles, table, idx):
    if not self._need_list_aux:
      return None
    if 25 in self._channel_enums:
      return packing.fuzzy_support_pixels(self._options, dv_call, alt_alleles, table, idx)
    if 27 in self._channel_enums:      # allele_sample_probability (alleles in key order, see packing)
      return packing.allele_sample_probability_pixels(dv_call, table, idx)
    return packing.allele_frequency_pixels(self._options, dv_call, alt_alleles,
                                           table, idx)

  def _one_item(self, dv_call, ref_bases: str, reads: Sequence,
                image_start_pos: int, alt_alleles: Sequence[str], height: int,
                mean_coverage: float = 0.0, alignment_positions=None,
                channels_to_blank=None, non_uniform_threshold=None):
    width = len(ref_bases)
    table = packing.ReadTable.from_reads(
        reads, alignment_positions=alignment_positions,
        need_aux=self._need_aux, need_seq_aux=self._need_seq_aux)
    batch = packing.PackedBatch(table=table, width=width, use_ref_aux=self._need_ref_aux)
    ref_idx = batch.add_ref_window(ref_bases)
    idx = np.arange(len(reads), dtype=np.uint32)
    if non_uniform_threshold is not None:
      kept = packing.non_uniform_sample(dv_call, table, idx, height - self._options.reference_band_height,
                                        non_uniform_threshold, self._options.random_seed)
      if kept is not None:
        idx = idx[kept]
    codes = packing.support_codes(dv_call, alt_alleles, table, idx)
    groups = (packing.allele_groups(dv_call, table, idx)
              if getattr(self._options, 'sort_by_alt_allele_support', False)
              else None)
    batch.add_item(
        variant_start=dv_call.variant.start, image_start=image_start_pos,
        ref_idx=ref_idx, read_idx=idx, codes=codes, height=height, out_off=0,
        blank_mask=packing.blank_mask_for(self._channel_enums,
                                          channels_to_blank),
        mean_coverage=mean_coverage, groups=groups,
        list_aux=self._list_aux(dv_call, alt_alleles, table, idx))
    c = self.num_channels
    out, rows = self._encoder(width).encode(batch, c)
    return out.reshape(height, width, c), int(rows[0])

  # ------------------------------------------------------------- pybind API
  def encode_reference(self, ref_bases: str) -> np.ndarray:
    """-> uint8 [1, W, C]  (EncodeReference, pileup_image_native.cc:512-527)."""
    band = self._options.reference_band_height
    if band < 1:
      raise ValueError('encode_reference needs reference_band_height >= 1')
    img, _ = self._one_item(T.DeepVariantCall(), ref_bases, [], 0, [],
                            height=band + 1)
    return img[0:1].copy()

  def encode_read(self, dv_call, ref_bases: str, read, image_start_pos: int,
                  alt_alleles: Sequence[str], channels_to_blank=None
                  ) -> Optional[np.ndarray]:
    """-> uint8 [1, W, C] or None (EncodeRead, pileup_image_native.cc:477-510)."""
    band = self._options.reference_band_height
    img, kept = self._one_item(dv_call, ref_bases, [read], image_start_pos,
                               list(alt_alleles), height=band + 1,
                               channels_to_blank=channels_to_blank)
    if kept == 0:
      return None
    return img[band:band + 1].copy()

  def build_pileup_for_one_sample(self, dv_call, ref_bases: str, reads,
                                  image_start_pos: int, alt_alleles,
                                  sample_options, mean_coverage: float = 0.0,
                                  alignment_positions=None,
                                  channels_to_blank=None) -> np.ndarray:
    """-> uint8 [pileup_height, W, C] (rows of BuildPileupForOneSample,
    pileup_image_native.cc:297-447, in FillPileupArray's HWC order)."""
    if len(ref_bases) != self._options.width:
      raise ValueError('ref_bases.size() != width')  # CHECK_EQ, :308
    height = sample_options.pileup_height or self._options.height
    non_uniform = None
    if getattr(sample_options, 'use_non_uniform_downsampling', False):
      # every allele keeps a minimum of its supporters (:326-341); sampled on the host, see packing.non_uniform_sample
      non_uniform = int(sample_options.non_uniform_downsampling_threshold)
    img, _ = self._one_item(
        dv_call, ref_bases, list(reads), image_start_pos, list(alt_alleles),
        height=height, mean_coverage=mean_coverage,
        alignment_positions=alignment_positions,
        channels_to_blank=channels_to_blank, non_uniform_threshold=non_uniform)
    return img
