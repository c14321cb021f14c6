"""deepvariant_amd.sharded_file_utils on the vectors of
third_party/nucleus/io/sharded_file_utils_test.py:46-270."""
import os

import pytest

from deepvariant_amd import sharded_file_utils as io


@pytest.mark.parametrize('task_id,filespecs,expected', [
    (0, ['foo.txt'], [0, 'foo.txt']),
    (0, ['foo.txt', 'bar.txt'], [0, 'foo.txt', 'bar.txt']),
    (0, ['bar.txt', 'foo.txt'], [0, 'bar.txt', 'foo.txt']),
    (0, ['foo.txt', None], [0, 'foo.txt', None]),
    (0, ['foo.txt', ''], [0, 'foo.txt', '']),
    (0, ['foo@10.txt', None], [10, 'foo-00000-of-00010.txt', None]),
    (0, ['foo@10.txt', ''], [10, 'foo-00000-of-00010.txt', '']),
    (0, ['foo@10', None], [10, 'foo-00000-of-00010', None]),
    (1, ['foo@10', None], [10, 'foo-00001-of-00010', None]),
    (9, ['foo@10', None], [10, 'foo-00009-of-00010', None]),
    (0, ['foo@10', 'bar@10', 'baz@10'], [10, 'foo-00000-of-00010', 'bar-00000-of-00010', 'baz-00000-of-00010']),
    (9, ['foo@10', 'bar@10', 'baz@10'], [10, 'foo-00009-of-00010', 'bar-00009-of-00010', 'baz-00009-of-00010']),
])
def test_resolve_filespecs(task_id, filespecs, expected):
  assert io.resolve_filespecs(task_id, *filespecs) == expected


@pytest.mark.parametrize('task_id,outputs', [(10, ['foo@10']), (1, ['foo']), (0, ['foo@10', 'bad@11']),
                                             (0, ['foo', 'bad@11'])])
def test_resolve_filespecs_raises_with_bad_inputs(task_id, outputs):
  with pytest.raises(ValueError):
    io.resolve_filespecs(task_id, *outputs)


@pytest.mark.parametrize('filespec,expected', [
    ('foo.txt', ['foo.txt']),
    ('foo-00000-of-00010.txt', ['foo-00000-of-00010.txt']),
    ('foo@3.txt', ['foo-00000-of-00003.txt', 'foo-00001-of-00003.txt', 'foo-00002-of-00003.txt']),
    ('foo@3', ['foo-00000-of-00003', 'foo-00001-of-00003', 'foo-00002-of-00003']),
])
def test_maybe_generate_sharded_filenames(filespec, expected):
  assert io.maybe_generate_sharded_filenames(filespec) == expected


def test_parse_and_generate():
  assert io.parse_sharded_file_spec('/dir/foo/bar@3') == ('/dir/foo/bar', 3, '')
  assert io.parse_sharded_file_spec('/dir/foo/bar@3.txt') == ('/dir/foo/bar', 3, '.txt')
  with pytest.raises(io.ShardError):
    io.parse_sharded_file_spec('/dir/foo/bar@0')
  assert io.generate_sharded_filenames('/dir/foo/bar@3') == [
      '/dir/foo/bar-00000-of-00003', '/dir/foo/bar-00001-of-00003', '/dir/foo/bar-00002-of-00003']
  assert io.generate_sharded_filenames('/dir/foo/bar@3.txt') == [
      '/dir/foo/bar-00000-of-00003.txt', '/dir/foo/bar-00001-of-00003.txt', '/dir/foo/bar-00002-of-00003.txt']
  names = io.generate_sharded_filenames('/dir/foo/bar@100000')
  assert len(names) == 100000 and names[99999] == '/dir/foo/bar-099999-of-100000'
  for spec in ('/dir/foo/bar', '/dir/foo/bar@0'):
    with pytest.raises(io.ShardError):
      io.generate_sharded_filenames(spec)


@pytest.mark.parametrize('spec,expected', [('/dir/foo/bar@3', True), ('/dir/foo/bar@3,txt', True),
                                           ('/dir/foo/bar@123456', True), ('/dir/foo/bar@0', False),
                                           ('/dir/foo/bar', False)])
def test_is_sharded_file_spec(spec, expected):
  assert io.is_sharded_file_spec(spec) is expected


@pytest.mark.parametrize('name,expected', [
    ('/dir/foo/bar-00001-of-00003', True), ('/dir/foo/bar-00001-of-00003,txt', True),
    ('/dir/foo/bar-00000-of-12345', True), ('/dir/foo/bar-00000-of-00000', False), ('/dir/foo/bar', False),
    ('/dir/foo/bar-00001-of-10000/baz.txt', False), ('/dir/foo/bar-00001-of-10000.baz.txt', True)])
def test_is_sharded_filename(name, expected):
  assert io.is_sharded_filename(name) is expected


def test_patterns():
  assert io.generate_sharded_file_pattern('/dir/foo/bar', 3, '') == '/dir/foo/bar-?????-of-00003'
  assert io.generate_sharded_file_pattern('/dir/foo/bar', 3, '.txt') == '/dir/foo/bar-?????-of-00003.txt'
  assert io.generate_sharded_file_pattern('/dir/foo/bar', 1234567, '.txt') == '/dir/foo/bar-???????-of-1234567.txt'
  for spec, want in (('/dir/foo/bar', '/dir/foo/bar'), ('/dir/foo/bar@3.txt', '/dir/foo/bar-?????-of-00003.txt'),
                     ('/dir/foo/bar@3', '/dir/foo/bar-?????-of-00003'),
                     ('/dir/foo/bar@1000', '/dir/foo/bar-?????-of-01000'),
                     ('/dir/foo/bar@12345678', '/dir/foo/bar-????????-of-12345678')):
    assert io.normalize_to_sharded_file_pattern(spec) == want


@pytest.mark.parametrize('specs,expected_files', [
    ('no_spec', ['no_spec']),
    ('sharded@3', ['sharded-00000-of-00003', 'sharded-00001-of-00003', 'sharded-00002-of-00003']),
    ('*.ext', ['cat.ext', 'dog.ext']),
    ('fo?bar', ['foobar']),
    ('file1,file2,file3', ['file1', 'file2', 'file3']),
    ('mixed.*txt,mixed@1,mixed_file', ['mixed.1txt', 'mixed.2txt', 'mixed-00000-of-00001', 'mixed_file']),
    ('with_dups*', ['with_dups.1txt', 'with_dups.2txt', 'with_dups-00000-of-00001', 'with_dups']),
])
def test_glob_list_sharded_file_patterns(tmp_path, specs, expected_files):
  full = []
  for f in expected_files:
    (tmp_path / f).write_text('')
    full.append(str(tmp_path / f))
  full_specs = ','.join(str(tmp_path / s) for s in specs.split(','))
  assert io.glob_list_sharded_file_patterns(full_specs) == sorted(set(full))


@pytest.mark.parametrize('name,base,shard,n,suffix', [
    ('name3-00000-of-00001', 'name3', 0, 1, ''),
    ('name4-12-of-20.foo.bar', 'name4', 12, 20, '.foo.bar'),
    ('name5-123456-of-999999.baz', 'name5', 123456, 999999, '.baz'),
    ('dir/name6.xxx-01111-of-02222.yyy', 'dir/name6.xxx', 1111, 2222, '.yyy'),
    ('/dir/foo/bar-00001-of-10000.baz.txt', '/dir/foo/bar', 1, 10000, '.baz.txt'),
])
def test_parse_sharded_filename(name, base, shard, n, suffix):
  b, s, k, suf = io.parse_sharded_filename(name)
  assert (b, int(s), int(k), suf) == (base, shard, n, suffix)


def test_the_command_lines_use_it():
  from deepvariant_amd import call_variants, make_examples
  assert call_variants.sharded_paths('x.tfrecord@3.gz') == io.generate_sharded_filenames('x.tfrecord@3.gz')
  assert make_examples._shard('x.tfrecord@3.gz', 2) == ('x.tfrecord-00002-of-00003.gz', 3)
  assert call_variants.is_sharded_filename('cvo-00000-of-00001.tfrecord.gz')
