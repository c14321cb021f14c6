"""Loader for tests/golden/realigner_chr20.npz (made by tests/golden/make_golden.py realigner)
and the adapter that lets the CPU tests of the realigner's HOST logic count alleles with the
oracle (oracle/allelecounter_ref.py) -- the product counts on the device and has no CPU path."""
import os

import numpy as np

from deepvariant_amd import allelecounter as ac
from oracle import allelecounter_ref as AR
from tests import golden_io

FIXTURE = os.path.join(os.path.dirname(__file__), 'golden', 'realigner_chr20.npz')


class FixtureRef:
  """n_bases / get_bases over the stored stretch of chr20 (N outside it)."""

  def __init__(self, z):
    self._start = int(z['ref_start'][0])
    self._bases = bytes(z['ref_bases']).decode()
    self._n = int(z['n_contig_bases'][0])

  def n_bases(self, contig):
    if contig != 'chr20':
      raise KeyError(contig)
    return self._n

  def get_bases(self, contig, start, end):
    start, end = max(start, 0), min(end, self._n)
    out = []
    for p0, p1 in ((start, min(end, self._start)), ):
      out.append('N' * max(0, p1 - p0))
    a, b = max(start, self._start), min(end, self._start + len(self._bases))
    if b > a:
      out.append(self._bases[a - self._start:b - self._start])
    tail0 = max(start, self._start + len(self._bases))
    out.append('N' * max(0, end - tail0))
    return ''.join(out)


class StringRef:
  """InMemoryFastaReader([(chrom, 0, bases)])."""

  def __init__(self, contig, bases):
    self._contig, self._bases = contig, bases

  def n_bases(self, contig):
    if contig != self._contig:
      raise KeyError(contig)
    return len(self._bases)

  def get_bases(self, contig, start, end):
    return self._bases[start:end]


def load():
  with np.load(FIXTURE) as f:
    z = {k: f[k] for k in f.files}
  reads = golden_io.unpack_reads(z)
  sets = {k[5:]: [reads[i] for i in z[k]] for k in z if k.startswith('sets_')}
  return FixtureRef(z), sets


def golden_wgs_variants():
  """{(start, alts): dict} -- the Variant protos of golden.calling_examples (AD / DP / VAF of calls[0],
  the no-call genotype, the sample name)."""
  import json
  with np.load(FIXTURE) as f:
    lines = bytes(f['wgs_variants']).decode().split('\n')
  out = {}
  for line in lines:
    d = json.loads(line)
    out[(d['start'], tuple(d['alts']))] = d
  return out


def variant_facts(variant):
  """The same dict for a dv_types.Variant (after a trip through the wire format)."""
  from deepvariant_amd import protowire as pw
  v = pw.decode_variant(pw.encode_variant(variant))
  c = v.calls[0]
  return dict(start=v.start, end=v.end, ref=v.reference_bases, alts=list(v.alternate_bases), sample=c.call_set_name,
              genotype=list(c.genotype), AD=[x.int_value for x in c.info['AD'].values],
              DP=[x.int_value for x in c.info['DP'].values],
              VAF=[float(x.number_value).hex() for x in c.info['VAF'].values])


class OracleAlleleCounter:
  """oracle AlleleCounter behind deepvariant_amd.allelecounter.AlleleCounter's interface."""

  def __init__(self, ref_reader, reference_name, start, end, candidate_positions=(), min_mapping_quality=0,
               min_base_quality=0, keep_legacy_behavior=False, full_range=None, track_ref_reads=False):
    self._c = AR.AlleleCounter(ref_reader, reference_name, start, end, min_mapping_quality=min_mapping_quality,
                               min_base_quality=min_base_quality, keep_legacy_behavior=keep_legacy_behavior,
                               full_range=full_range)
    self._contig = reference_name

  def add(self, read, sample=''):
    self._c.add(read)

  def add_table(self, table):
    for read in table.to_reads(self._contig):
      self._c.add(read)

  def interval_length(self):
    return len(self._c.counts)

  def counts(self):
    out = []
    for c in self._c.counts:
      a = ac.AlleleCount(self._contig, c.position, c.ref_base)
      a.ref_supporting_read_count = c.ref_supporting_read_count
      a.read_alleles = {k: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for k, v in c.read_alleles.items()}
      out.append(a)
    return out


import contextlib


@contextlib.contextmanager
def oracle_allele_counter():
  """CPU tests of the realigner's host logic: `deepvariant_amd.allelecounter.AlleleCounter` (which counts on
  the device) is swapped for the oracle's counter while the block runs.  The product has no parameter
  for this; the swap is the test's business."""
  from deepvariant_amd import allelecounter
  old = allelecounter.AlleleCounter
  allelecounter.AlleleCounter = OracleAlleleCounter
  try:
    yield
  finally:
    allelecounter.AlleleCounter = old


def with_oracle_counter(fn):
  """Decorator form of oracle_allele_counter() for a whole CPU test."""
  import functools

  @functools.wraps(fn)
  def wrapped(*args, **kwargs):
    with oracle_allele_counter():
      return fn(*args, **kwargs)
  return wrapped
