"""deepvariant_amd/cram_reader.py (CRAM 3.0 on the host: containers, slices, rANS / gzip blocks,
data-series encodings, read features, reference-based reconstruction, mate links) against

  * the reference tree's NA12878 chr20:10.0-10.1 Mb slice, which exists there both as BAM and as
    CRAM (deepvariant/testdata/input/NA12878_S1.chr20.10_10p1mb.{bam,cram}: the reference's own
    make_examples test runs the same goldens from either, make_examples_test.py:330-369) -- every
    field of all 51,873 reads must be identical, and so must the packed tables;
  * nucleus' CRAM test files (third_party/nucleus/io/sam_test.py:250-300 CramReaderTests: one
    written against an external FASTA, one with embedded reference slices) and the SAM text they
    were made from.

Fixtures: tests/golden/cram.npz (make_golden.py cram) + na12878_100kb.npz (BAM, reference bases)."""
import dataclasses
import os

import numpy as np
import pytest

from deepvariant_amd import cram_reader
from deepvariant_amd import genomics_io
from deepvariant_amd import packing

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def files(tmp_path_factory):
  tmp = tmp_path_factory.mktemp('cram')
  out = {}
  with np.load(os.path.join(GOLDEN, 'cram.npz')) as z:
    for key, name in (('na12878_cram', 'na12878.cram'), ('na12878_crai', 'na12878.cram.crai'),
                      ('nucleus_embed_ref_0', 'embed0.cram'), ('nucleus_embed_ref_1', 'embed1.cram'),
                      ('nucleus_sam', 'test_cram.sam'), ('nucleus_fasta', 'test.fasta')):
      out[key] = str(tmp / name)
      with open(out[key], 'wb') as f:
        f.write(z[key].tobytes())
  with np.load(os.path.join(GOLDEN, 'na12878_100kb.npz')) as z:
    out['bam'] = str(tmp / 'na12878.bam')
    with open(out['bam'], 'wb') as f:
      f.write(z['bam'].tobytes())
    with open(out['bam'] + '.bai', 'wb') as f:
      f.write(z['bai'].tobytes())
    lo = int(z['ref_start'][0])
    out['fasta'] = str(tmp / 'chr20.fa')
    genomics_io.write_fasta(out['fasta'], [('chr20', 'N' * lo + z['ref_bases'].tobytes().decode())])
  return out


def _fields(r):
  return (r.fragment_name, r.read_number, r.number_reads, r._flag, r.alignment.position.reference_name,
          r.alignment.position.position, r.alignment.position.reverse_strand, r.alignment.mapping_quality,
          r.alignment.cigar, r.aligned_sequence, bytes(bytearray(r.aligned_quality)), r.fragment_length, r._mate_ok,
          r.proper_placement, r.duplicate_fragment, r.secondary_alignment, r.supplementary_alignment)


@pytest.mark.timeout(900)
def test_na12878_cram_equals_the_bam(files):
  ref = genomics_io.FastaReader(files['fasta'])
  names, from_cram = cram_reader.read_cram(files['na12878_cram'], ref.get_bases, 'chr20')
  bam_names, from_bam = genomics_io.read_bam(files['bam'], 'chr20')
  assert 'chr20' in names and names == bam_names == genomics_io.bam_contig_names(files['na12878_cram'])
  assert len(from_cram) == len(from_bam) == 51873
  for a, b in zip(from_cram, from_bam):
    assert _fields(a) == _fields(b)
  # a region query returns what the BAM's does (reads that overlap, file order), across container borders
  for lo, hi in ((10_019_500, 10_019_700), (10_050_000, 10_051_000), (9_999_000, 10_000_100)):
    got = cram_reader.read_cram(files['na12878_cram'], ref.get_bases, 'chr20', lo, hi)[1]
    want = genomics_io.read_bam(files['bam'], 'chr20', lo, hi)[1]
    assert [_fields(r) for r in got] == [_fields(r) for r in want] and len(got) > 20
  # the queries above went through the .crai next to the file; without it the container headers are
  # walked instead: same reads
  import shutil
  bare = files['na12878_cram'].replace('na12878.cram', 'no_index.cram')
  shutil.copy(files['na12878_cram'], bare)
  assert cram_reader.CramFile(files['na12878_cram'])._index() and cram_reader.CramFile(bare)._index() is None   # pylint: disable=protected-access
  got = cram_reader.read_cram(bare, ref.get_bases, 'chr20', 10_050_000, 10_051_000)[1]
  want = cram_reader.read_cram(files['na12878_cram'], ref.get_bases, 'chr20', 10_050_000, 10_051_000)[1]
  assert [_fields(r) for r in got] == [_fields(r) for r in want] and len(got) > 100


@pytest.mark.timeout(900)
def test_packed_table_from_cram_equals_the_native_bam_table(files):
  """ReadTable.from_cram (what make_examples feeds the region chain for --reads x.cram) against the
  native BAM reader on the same interval: same reads after the read requirements, same arrays."""
  ref = genomics_io.FastaReader(files['fasta'])
  lo, hi = 10_010_000, 10_030_000
  a = packing.ReadTable.from_cram(files['na12878_cram'], ref.get_bases, 'chr20', lo, hi, min_mapping_quality=5)
  b = packing.ReadTable.from_bam(files['bam'], 'chr20', lo, hi, min_mapping_quality=5)
  assert a.n_reads == b.n_reads > 5000
  for f in dataclasses.fields(packing.ReadTable):
    x, y = getattr(a, f.name), getattr(b, f.name)
    if isinstance(x, np.ndarray) and isinstance(y, np.ndarray):
      assert np.array_equal(x.astype(np.int64), y.astype(np.int64)), f.name
    elif f.name == 'keys':
      assert x == y


@pytest.mark.parametrize('key,embedded', [('nucleus_embed_ref_0', False), ('nucleus_embed_ref_1', True)])
def test_nucleus_cram_files(files, key, embedded):
  """sam_test.py CramReaderTests: header, iterate (names, sequences), query counts -- plus every SAM
  column of test_cram.sam.  The file with embedded reference slices is read WITHOUT a FASTA."""
  fetch = None if embedded else genomics_io.FastaReader(files['nucleus_fasta']).get_bases
  f = cram_reader.CramFile(files[key], fetch)
  assert f.header_text.startswith('@HD\tVN:1.3')
  assert f.contig_names == ['chrM', 'chr1', 'chr2']
  _, reads = cram_reader.read_cram(files[key], fetch)
  assert [r.fragment_name for r in reads] == ['cram1', 'cram2', 'cram3']
  assert [r.aligned_sequence for r in reads] == [
      'CCCTAACCCTAACCCTAACCCTAACCCTANNNNNN',
      'TAACCCTAACCCTAACCCTAACCCTAACCCTAACCCTAACCAAAACGAATCAAAAAAGAAAAACGAAAAAAAAA',
      'CACAGACGCTT']
  for contig, lo, hi, n in (('chr1', 0, 100, 3), ('chr2', 0, 121, 0)):
    assert len(cram_reader.read_cram(files[key], fetch, contig, lo, hi)[1]) == n
  sam = [line.rstrip('\n').split('\t') for line in open(files['nucleus_sam']) if not line.startswith('@')]
  for r, s in zip(f.records(), sam):
    cigar = ''.join('%d%s' % (n, op) for op, n in r.cigar)
    assert (r.name.decode(), r.flag, f.contig_names[r.ref_id], r.pos, r.mapq, cigar, r.mate_pos, r.tlen, r.seq,
            ''.join(chr(q + 33) for q in r.qual)) == (
                s[0], int(s[1]), s[2], int(s[3]), int(s[4]), s[5], int(s[7]), int(s[8]), s[9], s[10])


def test_cram_without_its_reference_fails_like_the_reference(files):
  """make_examples_test.py:551-568: --nouse_ref_for_cram on a CRAM written against an external
  reference -> 'Failed to parse BAM/CRAM file.'"""
  with pytest.raises(ValueError, match='Failed to parse BAM/CRAM file.'):
    cram_reader.read_cram(files['na12878_cram'], None, 'chr20', 10_000_000, 10_000_500)


def test_cram_against_the_wrong_reference_is_refused(files):
  """htslib refuses to decode a slice whose header MD5 does not match the reference it is given
  (a mismatching --ref would otherwise yield wrong read bases silently): one changed base inside the
  first slice's span is enough."""
  ref = genomics_io.FastaReader(files['fasta'])

  def tampered(contig, start, end):
    text = ref.get_bases(contig, start, end)
    if start <= 10_000_100 < end:
      k = 10_000_100 - start
      text = text[:k] + ('A' if text[k] != 'A' else 'C') + text[k + 1:]
    return text
  with pytest.raises(ValueError, match='MD5'):
    cram_reader.read_cram(files['na12878_cram'], tampered, 'chr20', 10_000_000, 10_001_000)
  # the untouched reference decodes
  assert cram_reader.read_cram(files['na12878_cram'], ref.get_bases, 'chr20', 10_000_000, 10_001_000)[1]


def test_rans_round_trip_on_synthetic_streams():
  """The order-0 / order-1 rANS 4x8 decoder against a straightforward encoder written here from the
  same published description (the real files above exercise it on 300 KB quality streams)."""
  rng = np.random.default_rng(7)
  for order in (0, 1):
    for n in (0, 1, 3, 4, 5, 17, 4096, 10_001):
      data = bytes(rng.choice(np.arange(33, 45, dtype=np.uint8), size=n, p=np.array([30, 20, 10, 8, 8, 6, 5, 4, 3, 3, 2, 1]) / 100))
      assert cram_reader._rans_decode(_rans_encode(data, order)) == data    # pylint: disable=protected-access


# ---- a plain rANS 4x8 encoder (test-side only)
def _norm_freqs(counts):
  total = sum(counts)
  f = [0] * 256
  if total == 0:
    return f
  for s, c in enumerate(counts):
    if c:
      f[s] = max(1, (c * 4096) // total)
  diff = 4096 - sum(f)
  top = max(range(256), key=lambda s: f[s])
  f[top] += diff
  assert f[top] > 0
  return f


def _write_table(f):
  out = bytearray()
  syms = [s for s in range(256) if f[s]]
  rle = 0
  for k, s in enumerate(syms):
    if rle:
      rle -= 1
    else:
      out.append(s)
      if k > 0 and syms[k - 1] == s - 1:
        run = 0
        while k + 1 + run < len(syms) and syms[k + 1 + run] == s + 1 + run:
          run += 1
        out.append(run)
        rle = run
    if f[s] >= 128:
      out += bytes([0x80 | (f[s] >> 8), f[s] & 0xFF])
    else:
      out.append(f[s])
  out.append(0)
  return bytes(out)


def _rans_encode(data, order):
  import struct
  n = len(data)
  head = bytes([order])
  if n == 0:
    return head + struct.pack('<II', 0, 0)
  low = 1 << 23
  if order == 0:
    counts = [0] * 256
    for b in data:
      counts[b] += 1
    f = _norm_freqs(counts)
    cum = [0] * 256
    x = 0
    for s in range(256):
      cum[s] = x
      x += f[s]
    states = [low] * 4
    out = bytearray()
    for k in range(n - 1, -1, -1):
      j, s = k & 3, data[k]
      x = states[j]
      x_max = ((low >> 12) << 8) * f[s]
      while x >= x_max:
        out.append(x & 0xFF)
        x >>= 8
      states[j] = ((x // f[s]) << 12) + (x % f[s]) + cum[s]
    body = _write_table(f) + struct.pack('<4I', *states) + bytes(reversed(out))
    return head + struct.pack('<II', len(body), n) + body
  q = n >> 2
  starts = [0, q, 2 * q, 3 * q]
  ends = [q, 2 * q, 3 * q, n]
  counts = {}
  for j in range(4):
    prev = 0
    for k in range(starts[j], ends[j]):
      counts.setdefault(prev, [0] * 256)[data[k]] += 1
      prev = data[k]
  tables = {c: _norm_freqs(v) for c, v in counts.items()}
  cums = {}
  for c, f in tables.items():
    cum, x = [0] * 256, 0
    for s in range(256):
      cum[s] = x
      x += f[s]
    cums[c] = cum
  states = [low] * 4
  out = bytearray()
  # encode backwards: the tail of stream 3 first, then the four streams in lockstep
  def put(j, k):
    s = data[k]
    ctx = data[k - 1] if k > starts[j] else 0
    f, cum = tables[ctx], cums[ctx]
    x = states[j]
    x_max = ((low >> 12) << 8) * f[s]
    while x >= x_max:
      out.append(x & 0xFF)
      x >>= 8
    states[j] = ((x // f[s]) << 12) + (x % f[s]) + cum[s]
  for k in range(n - 1, 4 * q - 1, -1):
    put(3, k)
  for i in range(q - 1, -1, -1):
    for j in (3, 2, 1, 0):
      put(j, starts[j] + i)
  table = bytearray()
  ctxs = sorted(tables)
  rle = 0
  for k, c in enumerate(ctxs):
    if rle:
      rle -= 1
    else:
      table.append(c)
      if k > 0 and ctxs[k - 1] == c - 1:
        run = 0
        while k + 1 + run < len(ctxs) and ctxs[k + 1 + run] == c + 1 + run:
          run += 1
        table.append(run)
        rle = run
    table += _write_table(tables[c])
  table.append(0)
  body = bytes(table) + struct.pack('<4I', *states) + bytes(reversed(out))
  return head + struct.pack('<II', len(body), n) + body


def test_make_examples_reads_a_cram_like_the_bam(files):
  """make_examples.RegionReads dispatches on the file's magic: --reads x.cram (+ --ref) yields, per
  calling region, the reads --reads x.bam yields; --nouse_ref_for_cram fails as the reference does."""
  from deepvariant_amd import dv_types as T
  from deepvariant_amd import make_examples as me
  ap = me.build_arg_parser()

  def regions_reads(reads_path, extra=()):
    args = ap.parse_args(['--ref', files['fasta'], '--reads', reads_path, '--examples', 'e'] + list(extra))
    reads_for = me.RegionReads(args)
    return [[(r.fragment_name, r.read_number, r.alignment.position.position, r.aligned_sequence,
              bytes(bytearray(r.aligned_quality)), r.alignment.cigar, r.fragment_length)
             for r in reads_for(T.Range('chr20', s, s + 1000))] for s in range(10_000_000, 10_004_000, 1000)]
  from_cram = regions_reads(files['na12878_cram'])
  assert from_cram == regions_reads(files['bam']) and all(len(x) > 100 for x in from_cram)
  assert me.absl_booleans(ap, ['--nouse_ref_for_cram']) == ['--use_ref_for_cram=false']
  with pytest.raises(ValueError, match='Failed to parse BAM/CRAM file.'):
    regions_reads(files['na12878_cram'], ['--use_ref_for_cram=false'])
