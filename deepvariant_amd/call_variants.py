"""MI355X `call_variants`: tf.Example TFRecords -> CallVariantsOutput TFRecords.

Mirrors the reference driver `deepvariant/call_variants.py` for the inference
path: `round_gls` (:248-285), the CVO record (`_create_cvo_proto` :353-399,
`write_variant_call` :288-333), the example reader (`get_dataset` /
`_parse_example` :449-538) and the batch loop (:766-1053), with the model
forward replaced by the HIP Inception-v3 (deepvariant_amd.inception_v3) and the
tf.data pipeline by a plain TFRecord reader.  The on-disk contract is unchanged
(SURVEY.md App. C): postprocess_variants consumes the output as is.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Iterable, List, Optional, Sequence

import numpy as np

from deepvariant_amd import protowire as pw
from deepvariant_amd import tfrecord

_GL_PRECISION = 10  # call_variants.py:83
_DEFAULT_BATCH = 1024  # --batch_size default, call_variants.py:104


def round_gls(gls, precision=None):
  """call_variants.py:248-285, verbatim semantics."""
  if abs(sum(gls) - 1) > 1e-6:
    raise ValueError(
        'Invalid genotype likelihoods do not sum to one: sum({}) = {}'.format(
            gls, sum(gls)))
  if precision is None:
    return gls
  min_ix = 0
  min_gl = gls[0]
  for ix, gl in enumerate(gls):
    if gl < min_gl:
      min_gl = gl
      min_ix = ix
  rounded_gls = [round(gl, precision) for gl in gls]
  rounded_gls[min_ix] = max(
      0.0,
      round(1 - sum(rounded_gls[:min_ix] + rounded_gls[min_ix + 1:]), precision))
  return rounded_gls


def create_cvo(encoded_variant: bytes, gls: Sequence[float],
               encoded_alt_allele_indices: bytes) -> bytes:
  """_create_cvo_proto (call_variants.py:353-399): the variant re-parsed from
  the example with call.info['MID'] = 'deepvariant' on its first call."""
  variant = pw.add_call_info_string(encoded_variant, 'MID', 'deepvariant')
  return pw.encode_call_variants_output(variant, encoded_alt_allele_indices, gls)


def sharded_paths(spec: str) -> List[str]:
  """`name@N.ext` -> N shard names; otherwise a glob / single file
  (third_party/nucleus/io/sharded_file_utils.py)."""
  m = re.match(r'^(.*)@(\d+)(.*)$', spec)
  if m:
    base, n, suffix = m.group(1), int(m.group(2)), m.group(3)
    return ['%s-%05d-of-%05d%s' % (base, i, n, suffix) for i in range(n)]
  paths = sorted(glob.glob(spec))
  return paths or [spec]


def read_examples(paths: Iterable[str]):
  """_parse_example (call_variants.py:462-487): the three features the model
  path consumes, plus the image shape."""
  for path in paths:
    for rec in tfrecord.read_tfrecords(path):
      ex = pw.decode_example(rec)
      yield (ex['image/encoded'][0], ex['variant/encoded'][0],
             ex['alt_allele_indices/encoded'][0], ex.get('image/shape'))


def example_info_shape(examples_path: str) -> Optional[List[int]]:
  """dv_utils.get_shape_and_channels_from_json (dv_utils.py:282-337)."""
  p = examples_path + '.example_info.json'
  if os.path.exists(p):
    with open(p) as f:
      return json.load(f)['shape']
  return None


def call_variants(examples: str, outfile: str, model, batch_size: int = _DEFAULT_BATCH,
                  max_batches: Optional[int] = None, writer_shards: int = 1,
                  allow_empty_examples: bool = False) -> int:
  """Runs `model` (deepvariant_amd.inception_v3.InceptionV3 with weights
  loaded) over every example and writes CallVariantsOutput TFRecords.

  `outfile` follows the reference's naming: `x.tfrecord.gz` becomes
  `x-0000i-of-0000K.tfrecord.gz` (call_variants.py:813-826).  Returns the
  number of records written.
  """
  import torch  # device memory + stream only
  paths = sharded_paths(examples)
  stem, ext = outfile, ''
  for e in ('.tfrecord.gz', '.tfrecord'):
    if outfile.endswith(e):
      stem, ext = outfile[:-len(e)], e
      break
  out_paths = ['%s-%05d-of-%05d%s' % (stem, i, writer_shards, ext)
               for i in range(writer_shards)]
  writers = [tfrecord.Writer(p) for p in out_paths]
  h, w, c = model.input_shape
  n_written = 0
  n_batches = 0
  buf_imgs, buf_meta = [], []

  def flush():
    nonlocal n_written
    if not buf_imgs:
      return
    x = torch.from_numpy(np.stack(buf_imgs)).to('cuda:%d' % model.device_index)
    probs = model(x).cpu().numpy().astype(np.float64)
    for p, (var, alt) in zip(probs, buf_meta):
      gls = round_gls([float(v) for v in p], precision=_GL_PRECISION)
      writers[n_written % writer_shards].write(create_cvo(var, gls, alt))
      n_written += 1
    buf_imgs.clear()
    buf_meta.clear()

  for image, variant, alt, shape in read_examples(paths):
    if shape is not None and list(shape) != [h, w, c]:
      # call_variants.py:704-733: input shape must match the model's
      raise ValueError('example shape %s != model shape %s' % (shape, [h, w, c]))
    buf_imgs.append(np.frombuffer(image, np.uint8).reshape(h, w, c))
    buf_meta.append((variant, alt))
    if len(buf_imgs) == batch_size:
      flush()
      n_batches += 1
      if max_batches is not None and n_batches >= max_batches:
        break
  flush()
  for wr in writers:
    wr.close()
  if n_written == 0 and not allow_empty_examples:
    # call_variants.py:793-808
    raise ValueError('No examples found in %s' % examples)
  return n_written
