"""Sharded file names: `name@N[.ext]` specifications and `name-00003-of-00016[.ext]` files.

Same functions, results and errors as third_party/nucleus/io/sharded_file_utils.py (the
on-disk contract of make_examples / call_variants / postprocess_variants: SURVEY 8b); the
vectors of sharded_file_utils_test.py are in tests/test_sharded_file_utils_cpu.py."""
from __future__ import annotations

import glob
import os
import re
from typing import List, Tuple

# a spec is recognised by its PREFIX (`match`, as the reference does): base '@' count ['.' suffix]
_SPEC = re.compile(r'((.*)\@(\d*[1-9]\d*)(?:\.(.+))?)')
_FILE = re.compile(r'(.*)-(\d+)-of-(\d*[1-9]\d*)([^/]+)?$')


class ShardError(Exception):
  """Not a sharded specification / file name."""


def parse_sharded_file_spec(spec: str) -> Tuple[str, int, str]:
  """'gs://some/file@200.txt' -> ('gs://some/file', 200, '.txt')."""
  m = _SPEC.match(spec)
  if not m:
    raise ShardError('The file specification {0} is not a sharded file specification because it '
                     'did not match the regex {1}'.format(spec, _SPEC.pattern))
  return m.group(2), int(m.group(3)), '.' + m.group(4) if m.group(4) else ''


def _width(num_shards: int) -> int:
  return max(5, len(str(num_shards)))


def generate_sharded_filenames(spec: str) -> List[str]:
  base, n, suffix = parse_sharded_file_spec(spec)
  w = _width(n)
  return ['%s-%0*d-of-%0*d%s' % (base, w, i, w, n, suffix) for i in range(n)]


def generate_sharded_file_pattern(basename: str, num_shards: int, suffix: str) -> str:
  w = _width(num_shards)
  return '%s-%s-of-%0*d%s' % (basename, '?' * w, w, num_shards, suffix)


def normalize_to_sharded_file_pattern(spec_or_pattern: str) -> str:
  """A spec becomes its glob pattern; anything else passes through."""
  try:
    base, n, suffix = parse_sharded_file_spec(spec_or_pattern)
  except ShardError:
    return spec_or_pattern
  return generate_sharded_file_pattern(base, n, suffix)


def glob_list_sharded_file_patterns(comma_separated_patterns: str, sep: str = ',') -> List[str]:
  """Existing files matching any of the (spec | pattern | name)s, sorted, without duplicates."""
  found = set()
  for pattern in comma_separated_patterns.split(sep):
    found.update(os.fspath(f) for f in glob.glob(normalize_to_sharded_file_pattern(pattern)))
  return sorted(found)


def is_sharded_filename(filename: str) -> bool:
  return _FILE.match(filename) is not None


def is_sharded_file_spec(spec: str) -> bool:
  return _SPEC.match(spec) is not None


def sharded_filename(spec: str, i: int) -> str:
  return generate_sharded_filenames(spec)[i]


def parse_sharded_filename(filename: str) -> Tuple[str, str, str, str]:
  """'dir/name.x-01111-of-02222.y' -> ('dir/name.x', '01111', '02222', '.y')."""
  m = _FILE.match(filename)
  if not m:
    raise ShardError('The file specification {0} is not a sharded file specification because it '
                     'did not match the regex {1}'.format(filename, _FILE.pattern))
  return m.group(1), m.group(2), m.group(3), m.group(4) or ''


def resolve_filespecs(shard: int, *filespecs):
  """[number of shards (0 = unsharded), this shard's path for every filespec]; the first
  filespec is the master: all others must be sharded the same way (false values pass)."""
  if not filespecs:
    raise ValueError('filespecs must have at least one element.')
  master = filespecs[0]
  master_sharded = is_sharded_file_spec(master)
  n_master = 0
  if master_sharded:
    n_master = parse_sharded_file_spec(master)[1]
    if shard >= n_master or shard < 0:
      raise ValueError('Invalid shard={} value with master={} sharding'.format(shard, master))
  elif shard > 0:
    raise ValueError('Output is not sharded but shard > 0: {}'.format(shard))
  out = [n_master]
  for spec in filespecs:
    if not spec:
      out.append(spec)
      continue
    sharded = is_sharded_file_spec(spec)
    if sharded != master_sharded or (sharded and parse_sharded_file_spec(spec)[1] != n_master):
      raise ValueError('Master={} and {} have inconsistent sharding'.format(master, spec))
    out.append(sharded_filename(spec, shard) if sharded else spec)
  return out


def maybe_generate_sharded_filenames(filespec: str) -> List[str]:
  if not isinstance(filespec, str):
    raise TypeError('Invalid filespec: %s' % filespec)
  return generate_sharded_filenames(filespec) if is_sharded_file_spec(filespec) else [filespec]
