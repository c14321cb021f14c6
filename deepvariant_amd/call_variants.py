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
from deepvariant_amd import sharded_file_utils
from deepvariant_amd import tfrecord

_GL_PRECISION = 10  # call_variants.py:83
_DEFAULT_BATCH = 1024  # --batch_size default, call_variants.py:104


def round_gls(gls, precision=None):
  """One candidate's genotype likelihoods rounded to `precision` decimals such that they
  still sum to one: the smallest entry (first on ties) absorbs the rounding residual,
  clamped at zero.  Same results as the reference's `round_gls`
  (deepvariant/call_variants.py:248-285; its pinned vectors are in
  tests/test_host_io_cpu.py).  np.float32 input is rounded in float32 like the reference,
  which feeds the model's float32 outputs through Python's `round`."""
  values = list(gls)
  total = sum(values)
  if abs(total - 1) > 1e-6:
    raise ValueError(
        'Invalid genotype likelihoods do not sum to one: sum({}) = {}'.format(gls, total))
  if precision is None:
    return gls
  lowest = min(range(len(values)), key=values.__getitem__)   # first index of the minimum
  out = [round(v, precision) for v in values]
  rest = sum(v for i, v in enumerate(out) if i != lowest)
  out[lowest] = max(0.0, round(1 - rest, precision))
  return out


def round_gls_batch(probs: np.ndarray, precision: int = _GL_PRECISION) -> np.ndarray:
  """`round_gls` for a whole batch [N, K] at once (the per-candidate Python loop costs more
  than the classifier at MI355X rates).  float32 probabilities stay float32 through the
  rounding -- NumPy's float32 `round` is what the reference applies to its float32
  predictions -- and are widened to the proto's doubles afterwards."""
  p = np.asarray(probs)
  if p.ndim != 2:
    raise ValueError('expected [N, K] probabilities')
  total = np.zeros(p.shape[0], p.dtype)
  for k in range(p.shape[1]):          # the reference's left-to-right sum
    total = total + p[:, k]
  bad = np.nonzero(np.abs(total - 1) > 1e-6)[0]
  if bad.size:
    row = p[bad[0]]
    raise ValueError('Invalid genotype likelihoods do not sum to one: sum({}) = {}'.format(
        list(row), total[bad[0]]))
  rows = np.arange(p.shape[0])
  lowest = np.argmin(p, axis=1)        # first minimum, like the reference's strict '<' scan
  out = np.round(p, precision)
  masked = out.copy()
  masked[rows, lowest] = 0
  rest = np.zeros(p.shape[0], p.dtype)
  for k in range(p.shape[1]):          # x + 0 is exact: same sum as skipping the entry
    rest = rest + masked[:, k]
  out[rows, lowest] = np.maximum(0.0, np.round(1 - rest, precision)).astype(p.dtype)
  return out.astype(np.float64)


def create_cvo(encoded_variant: bytes, gls: Sequence[float],
               encoded_alt_allele_indices: bytes) -> bytes:
  """_create_cvo_proto (call_variants.py:353-399): the variant re-parsed from
  the example with call.info['MID'] = 'deepvariant' on its first call."""
  variant = pw.add_call_info_string(encoded_variant, 'MID', 'deepvariant')
  return pw.encode_call_variants_output(variant, encoded_alt_allele_indices, gls)


def sharded_paths(spec: str) -> List[str]:
  """`name@N.ext` -> its N shard names; otherwise the files a pattern / comma list matches, or
  the name itself (sharded_file_utils.glob_list_sharded_file_patterns as call_variants.py:889 uses it)."""
  if sharded_file_utils.is_sharded_file_spec(spec):
    return sharded_file_utils.generate_sharded_filenames(spec)
  return sharded_file_utils.glob_list_sharded_file_patterns(spec) or [spec]


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


def is_sharded_filename(path: str) -> bool:
  """third_party/nucleus/io/sharded_file_utils.py:59,181-184 (`name-00003-of-00016[.ext]`)."""
  return sharded_file_utils.is_sharded_filename(path)


def call_variants(examples, outfile: str, model, batch_size: int = _DEFAULT_BATCH,
                  max_batches: Optional[int] = None, writer_shards: int = 1,
                  allow_empty_examples: bool = False, limit: int = 0) -> int:
  """Runs `model` (deepvariant_amd.inception_v3.InceptionV3 with weights
  loaded) over every example and writes CallVariantsOutput TFRecords.

  `outfile` follows the reference's naming: `x.tfrecord.gz` becomes
  `x-0000i-of-0000K.tfrecord.gz`; a name that is already one shard is written as is
  (call_variants.py:813-826).  `limit` > 0 stops after that many examples.  Returns the
  number of records written.
  """
  import torch  # device memory + stream only
  paths = list(examples) if isinstance(examples, (list, tuple)) else sharded_paths(examples)
  if is_sharded_filename(outfile):
    # "Output is already sharded, so dynamic sharding is disabled" (call_variants.py:813-817)
    out_paths = [outfile]
  else:
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
  staged = torch.empty((batch_size, h, w, c), dtype=torch.uint8,
                       device='cuda:%d' % model.device_index)

  def flush():
    nonlocal n_written
    if not buf_imgs:
      return
    # one device input buffer for the whole run: full batches then hit the model's captured
    # hipGraph (keyed by the buffer address) instead of re-capturing per batch
    k = len(buf_imgs)
    staged[:k].copy_(torch.from_numpy(np.stack(buf_imgs)))
    gls = round_gls_batch(model(staged[:k]).cpu().numpy(), _GL_PRECISION)
    for row, (var, alt) in zip(gls, buf_meta):
      writers[n_written % len(writers)].write(create_cvo(var, row.tolist(), alt))
      n_written += 1
    buf_imgs.clear()
    buf_meta.clear()

  n_read = 0
  for image, variant, alt, shape in read_examples(paths):
    if limit and n_read >= limit:      # --limit: "<= limit examples" (call_variants.py:199-201)
      break
    n_read += 1
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


# ---------------------------------------------------------------------------
# Command line: the reference's flag names (deepvariant/call_variants.py:88-224).
# Flags whose machinery is not part of this hot path are accepted when they are
# no-ops here and rejected -- never silently ignored -- when they would change
# the output.
# ---------------------------------------------------------------------------
def build_arg_parser():
  import argparse
  ap = argparse.ArgumentParser(
      prog='call_variants', allow_abbrev=False,
      description='MI355X call_variants: tf.Example TFRecords -> CallVariantsOutput TFRecords')
  boolean = dict(nargs='?', const='true', default='false')
  ap.add_argument('--examples', required=True)
  ap.add_argument('--outfile', required=True)
  ap.add_argument('--checkpoint', required=True,
                  help='TensorFlow checkpoint prefix / SavedModel directory of the Keras '
                       'InceptionV3 (call_variants.py:759-762), flat fp32 weights (.npy / .bin, '
                       'layout of dv_model_load_weights) or "random:<seed>"')
  ap.add_argument('--batch_size', type=int, default=_DEFAULT_BATCH)
  ap.add_argument('--max_batches', type=int, default=None)
  ap.add_argument('--num_readers', type=int, default=8)          # tf.data knob: no-op
  ap.add_argument('--include_debug_info', **boolean)
  ap.add_argument('--activation_layers', default='')
  ap.add_argument('--debugging_true_label_mode', **boolean)
  ap.add_argument('--execution_hardware', default='auto')
  ap.add_argument('--config_string', default=None)               # XLA/TPU knob: no-op
  ap.add_argument('--kmp_blocktime', default='0')                # MKL knob: no-op
  ap.add_argument('--writer_threads', type=int, default=0)
  ap.add_argument('--limit', type=int, default=0)
  ap.add_argument('--num_input_shards', type=int, default=0)
  ap.add_argument('--shm_prefix', default='')
  ap.add_argument('--stream_examples', **boolean)
  ap.add_argument('--allow_empty_examples', **boolean)
  ap.add_argument('--device', type=int, default=0)
  # not a reference flag: size of the fixed synthetic calibration set the fp16 classifier's shifts are calibrated on
  # when the checkpoint is loaded (InceptionV3.calibrate_for_checkpoint; a property of the checkpoint and the input
  # shape, never of the run's examples); 0 = the uncalibrated fp16 model
  ap.add_argument('--calibration_examples', type=int, default=256)
  return ap


def _flag_true(v) -> bool:
  return str(v).lower() in ('1', 'true', 't', 'yes')


def check_flags(args):
  """Raises ValueError for flag values this implementation cannot honour."""
  if args.execution_hardware not in ('auto', 'accelerator'):
    # call_variants.py:66-77: 'cpu' would run the TF graph on host cores
    raise ValueError('--execution_hardware=%s: this build has no CPU path '
                     '(use auto or accelerator)' % args.execution_hardware)
  for name in ('include_debug_info', 'debugging_true_label_mode', 'stream_examples'):
    if _flag_true(getattr(args, name)):
      raise ValueError('--%s is not supported by the MI355X call_variants' % name)
  if args.activation_layers:
    raise ValueError('--activation_layers is not supported')
  if args.shm_prefix:
    raise ValueError('--shm_prefix (stream_examples) is not supported')
  if args.batch_size < 1:
    raise ValueError('--batch_size must be positive')


def checkpoint_prefix(spec: str) -> Optional[str]:
  """The tensor-bundle prefix behind a `--checkpoint` value, or None: a TF checkpoint prefix
  (`.../ckpt-123`), its `.index` file, or a SavedModel directory (`variables/variables`)."""
  if spec.endswith('.index'):
    spec = spec[:-len('.index')]
  if os.path.exists(spec + '.index'):
    return spec
  if os.path.isdir(spec):
    for cand in (os.path.join(spec, 'variables', 'variables'), os.path.join(spec, 'variables')):
      if os.path.exists(cand + '.index'):
        return cand
  return None


def import_keras_checkpoint(prefix: str, in_channels: int, num_classes: int = 3,
                            allow_channel_mismatch: bool = False) -> np.ndarray:
  """A checkpoint written by the reference's Keras model (`model.save_weights` /
  `model.load_weights`, deepvariant/keras_modeling.py:304-335) -> the flat fp32 layout of
  `dv_model_load_weights`.  Variables are matched BY NAME (`layer_with_weights-N/...`, N in
  tf_keras' depth-sorted layer order -- deepvariant_amd/keras_layout.py) and every shape is
  checked, so a layout mistake is an error, never silently permuted weights.

  When the checkpoint was trained with a different number of input channels the reference
  copies the first kernel's common channels and leaves the rest at their random
  initialisation (`load_weights_to_model_with_different_channels`, :113-168); inference on
  that is meaningless, so it is an error here unless `allow_channel_mismatch`, in which case
  the missing channels are ZERO."""
  from deepvariant_amd import keras_layout, tf_checkpoint
  reader = tf_checkpoint.CheckpointReader(prefix)
  first = 'layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE'
  names = reader.get_variable_to_shape_map()
  if any(k.startswith('layer_with_weights-0/layer_with_weights-0/kernel') for k in names):
    raise ValueError('You are using an older DeepVariant Keras model architecture. '
                     'Please use a new model.')              # keras_modeling.py:186-192
  if first not in names or len(names[first]) != 4:
    raise ValueError('Unexpected model format.')              # keras_modeling.py:193
  ckpt_channels = names[first][2]
  if ckpt_channels != in_channels and not allow_channel_mismatch:
    raise ValueError('checkpoint has %d input channels, the examples have %d' %
                     (ckpt_channels, in_channels))
  src_entries, n_src = keras_layout.variable_layout(ckpt_channels, num_classes)
  src = np.zeros(n_src, np.float32)
  for name, shape, off in src_entries:
    if name not in names:
      raise ValueError('checkpoint lacks %s' % name)
    if tuple(names[name]) != tuple(shape):
      raise ValueError('%s has shape %s in the checkpoint, the InceptionV3 layout expects %s'
                       % (name, names[name], list(shape)))
    t = reader.get_tensor(name).astype(np.float32, copy=False)
    src[off:off + t.size] = t.reshape(-1)
  if ckpt_channels == in_channels:
    return src
  # first-conv channel surgery; everything behind the first kernel keeps its place
  k_src = 3 * 3 * ckpt_channels * 32
  k_dst = 3 * 3 * in_channels * 32
  dst = np.zeros(n_src - k_src + k_dst, np.float32)
  common = min(ckpt_channels, in_channels)
  dst[:k_dst].reshape(3, 3, in_channels, 32)[:, :, :common, :] = \
      src[:k_src].reshape(3, 3, ckpt_channels, 32)[:, :, :common, :]
  dst[k_dst:] = src[k_src:]
  return dst


def calibration_cache_prefix(spec: str):
  """Where a checkpoint's calibration corrections are cached (InceptionV3.calibrate_for_checkpoint): next to the
  checkpoint files; None for `random:<seed>` weights."""
  if spec.startswith('random:'):
    return None
  prefix = checkpoint_prefix(spec)
  return prefix if prefix is not None else spec


def load_flat_checkpoint(spec: str, model, allow_channel_mismatch: bool = False):
  """--checkpoint: a TensorFlow checkpoint / SavedModel directory of the reference's Keras
  model, `random:<seed>`, or a flat fp32 array in dv_model_load_weights order."""
  if spec.startswith('random:'):
    model.init_random(seed=int(spec.split(':', 1)[1]))
    return
  prefix = checkpoint_prefix(spec)
  if prefix is not None:
    flat = import_keras_checkpoint(prefix, model.input_shape[2], model.num_classes,
                                   allow_channel_mismatch)
    model.load_flat_weights(flat)
    return
  if os.path.isdir(spec):
    raise ValueError('%s holds no checkpoint (no variables.index)' % spec)
  flat = np.load(spec) if spec.endswith('.npy') else np.fromfile(spec, np.float32)
  model.load_flat_weights(np.ascontiguousarray(flat, np.float32))


def main(argv=None) -> int:
  args = build_arg_parser().parse_args(argv)
  check_flags(args)
  paths = sharded_paths(args.examples)
  if args.num_input_shards:
    paths = paths[:args.num_input_shards]        # call_variants.py:115-120
  shape = example_info_shape(paths[0]) if paths else None
  if shape is None:
    raise ValueError('%s.example_info.json is missing: the model shape comes from it '
                     '(call_variants.py:704-733)' % (paths[0] if paths else args.examples))
  from deepvariant_amd.inception_v3 import InceptionV3
  model = InceptionV3(tuple(shape), max_batch=min(args.batch_size, 8192), device=args.device)
  load_flat_checkpoint(args.checkpoint, model)
  model.calibrate_for_checkpoint(args.calibration_examples, cache_prefix=calibration_cache_prefix(args.checkpoint))
  n = call_variants(paths, args.outfile, model,
                    batch_size=args.batch_size, max_batches=args.max_batches,
                    writer_shards=max(1, min(args.writer_threads or 1, 16)),
                    allow_empty_examples=_flag_true(args.allow_empty_examples),
                    limit=max(0, args.limit))
  print('call_variants: wrote %d CallVariantsOutput records' % n)
  return 0


if __name__ == '__main__':
  import sys
  sys.exit(main())
