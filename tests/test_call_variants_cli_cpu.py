"""call_variants command line: the reference's flag names (call_variants.py:88-224) parse,
no-op flags are accepted, and flags that would change the output are rejected loudly."""
import pytest

from deepvariant_amd import call_variants as cv


def _parse(*extra):
  return cv.build_arg_parser().parse_args(
      ['--examples', 'ex@2.tfrecord.gz', '--outfile', 'out.tfrecord.gz',
       '--checkpoint', 'random:1'] + list(extra))


def test_reference_flags_parse_with_reference_defaults():
  a = _parse('--batch_size', '512', '--num_readers=4', '--kmp_blocktime', '0',
             '--writer_threads', '3', '--limit', '100', '--allow_empty_examples')
  cv.check_flags(a)
  assert (a.batch_size, a.writer_threads, a.limit) == (512, 3, 100)
  assert cv._flag_true(a.allow_empty_examples)
  assert _parse().batch_size == 1024            # call_variants.py:104
  assert cv.sharded_paths(a.examples) == ['ex-00000-of-00002.tfrecord.gz',
                                          'ex-00001-of-00002.tfrecord.gz']


@pytest.mark.parametrize('flags', [
    ['--execution_hardware', 'cpu'], ['--include_debug_info'], ['--stream_examples=true'],
    ['--activation_layers', 'mixed10'], ['--debugging_true_label_mode'], ['--shm_prefix', 'x'],
    ['--batch_size', '0']])
def test_unsupported_flags_are_rejected(flags):
  with pytest.raises(ValueError):
    cv.check_flags(_parse(*flags))


def test_checkpoint_argument_forms(tmp_path):
  """--checkpoint accepts what the reference's does (a TF checkpoint prefix or a SavedModel
  directory, call_variants.py:759-762); a directory without one and a corrupt index are
  errors, never a silent fallback."""
  class _Model:
    input_shape = (100, 221, 7)
    num_classes = 3
  (tmp_path / 'model.ckpt.index').write_bytes(b'')
  assert cv.checkpoint_prefix(str(tmp_path / 'model.ckpt')) == str(tmp_path / 'model.ckpt')
  with pytest.raises(ValueError, match='not a checkpoint index'):
    cv.load_flat_checkpoint(str(tmp_path / 'model.ckpt'), _Model())
  (tmp_path / 'saved' / 'variables').mkdir(parents=True)
  (tmp_path / 'saved' / 'variables' / 'variables.index').write_bytes(b'')
  assert cv.checkpoint_prefix(str(tmp_path / 'saved')).endswith('variables/variables')
  (tmp_path / 'empty').mkdir()
  with pytest.raises(ValueError, match='holds no checkpoint'):
    cv.load_flat_checkpoint(str(tmp_path / 'empty'), _Model())
