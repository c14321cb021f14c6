"""The native CRAM 3.0 decoder (dv_cram_read_region, deepvariant_amd/csrc/cram_reader.cpp) against

  * the native BAM reader on the reference tree's NA12878 slice, which exists both as BAM and as CRAM
    (deepvariant/testdata/input/NA12878_S1.chr20.10_10p1mb.{bam,cram}): every array of the packed
    table, whole contig and region queries, with the .crai and without, on 1 and 8 threads;
  * the Python decoder (deepvariant_amd/cram_reader.py, itself pinned by tests/test_cram_reader_cpu.py)
    on nucleus' CRAM test files (external and embedded reference);
  * files written by tests/cram_writer.py that take the paths the real files do not: every encoding
    (HUFFMAN with mixed code lengths, BETA, SUBEXP, GAMMA, BYTE_ARRAY_LEN / _STOP, EXTERNAL), every block
    codec (raw, gzip, bzip2, lzma, rANS order 0 / 1), every read feature, lossy names, absolute
    positions, detached / attached mates, multi-reference and unmapped slices, embedded references --
    decoded natively and by the Python twin, and compared with the reads that went in;
  * the reference's error behaviour: no reference for a CRAM that needs one, a reference whose MD5
    differs, an unknown contig, a 3.1 file.
"""
import dataclasses
import os
import random
import shutil

import numpy as np
import pytest

from tests import cram_writer
from tests.test_cram_reader_cpu import _rans_encode
from deepvariant_amd import _lib
from deepvariant_amd import cram_reader
from deepvariant_amd import genomics_io
from deepvariant_amd import packing

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
REQ = dict(min_mapping_quality=0, keep_duplicates=False, keep_failed_qc=False, keep_secondary=False,
           keep_supplementary=False, keep_improperly_placed=False)


@pytest.fixture(scope='module')
def files(tmp_path_factory):
  tmp = tmp_path_factory.mktemp('cram_native')
  out = {'tmp': str(tmp)}
  with np.load(os.path.join(GOLDEN, 'cram.npz')) as z:
    for key, name in (('na12878_cram', 'na12878.cram'), ('na12878_crai', 'na12878.cram.crai'),
                      ('nucleus_embed_ref_0', 'embed0.cram'), ('nucleus_embed_ref_1', 'embed1.cram'),
                      ('nucleus_fasta', 'test.fasta')):
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


def _same_tables(a, b):
  assert a.n_reads == b.n_reads
  for f in dataclasses.fields(packing.ReadTable):
    x, y = getattr(a, f.name), getattr(b, f.name)
    if isinstance(x, np.ndarray) and isinstance(y, np.ndarray):
      assert x.shape == y.shape and np.array_equal(x.astype(np.int64), y.astype(np.int64)), f.name
    elif f.name == 'keys':
      assert x == y


def _python_table(path, fetch, contig=None, start=0, end=1 << 62, **req):
  """The table the PYTHON decoder gives (what ReadTable.from_cram was before the native decoder)."""
  kw = dict(REQ)
  kw.update(req)
  oq = kw.pop('use_original_quality_scores', False)
  _, reads = cram_reader.read_cram(path, fetch, contig, start, end, use_original_quality_scores=oq)
  return packing.ReadTable.from_reads([r for r in reads if genomics_io.read_satisfies_requirements(r, **kw)])


@pytest.mark.parametrize('threads', [1, 8])
def test_na12878_cram_table_equals_the_bam_table(files, threads):
  ref = genomics_io.FastaReader(files['fasta'])
  a = packing.ReadTable.from_cram(files['na12878_cram'], ref.get_bases, 'chr20', n_threads=threads)
  b = packing.ReadTable.from_bam(files['bam'], 'chr20')
  assert a.n_reads > 51000
  _same_tables(a, b)
  # every optional read kept (duplicates, QC failures, improperly placed pairs ...): all 51,873 mapped reads
  a = packing.ReadTable.from_cram(files['na12878_cram'], ref.get_bases, 'chr20', n_threads=threads, keep_duplicates=True,
                                  keep_failed_qc=True, keep_secondary=True, keep_supplementary=True,
                                  keep_improperly_placed=True)
  b = packing.ReadTable.from_bam(files['bam'], 'chr20', keep_duplicates=True, keep_failed_qc=True, keep_secondary=True,
                                 keep_supplementary=True, keep_improperly_placed=True)
  assert a.n_reads == 51873
  _same_tables(a, b)


def test_region_queries_with_and_without_the_index(files):
  ref = genomics_io.FastaReader(files['fasta'])
  bare = os.path.join(files['tmp'], 'no_index.cram')
  shutil.copy(files['na12878_cram'], bare)
  for lo, hi in ((10_019_500, 10_019_700), (10_050_000, 10_051_000), (9_999_000, 10_000_100), (10_099_900, 10_200_000),
                 (5_000_000, 5_000_100)):
    want = packing.ReadTable.from_bam(files['bam'], 'chr20', lo, hi, min_mapping_quality=10)
    for path in (files['na12878_cram'], bare):
      _same_tables(packing.ReadTable.from_cram(path, ref.get_bases, 'chr20', lo, hi, min_mapping_quality=10), want)
  assert want.n_reads == 0      # the last interval lies outside the slice


@pytest.mark.parametrize('key,embedded', [('nucleus_embed_ref_0', False), ('nucleus_embed_ref_1', True)])
def test_nucleus_cram_files(files, key, embedded):
  """sam_test.py CramReaderTests' files: the one with embedded reference slices is read WITHOUT a FASTA."""
  fetch = None if embedded else genomics_io.FastaReader(files['nucleus_fasta']).get_bases
  kw = dict(keep_duplicates=True, keep_failed_qc=True, keep_secondary=True, keep_supplementary=True,
            keep_improperly_placed=True)
  t = packing.ReadTable.from_cram(files[key], fetch, **kw)
  assert [k.rsplit('/', 1)[0] for k in t.keys] == ['cram1', 'cram2', 'cram3']
  assert bytes(t.bases[:int(t.read_seq_off[1])]).decode() == 'CCCTAACCCTAACCCTAACCCTAACCCTANNNNNN'
  _same_tables(t, _python_table(files[key], fetch, **kw))
  for contig, lo, hi, n in (('chr1', 0, 100, 3), ('chr2', 0, 121, 0)):
    q = packing.ReadTable.from_cram(files[key], fetch, contig, lo, hi, **kw)
    assert q.n_reads == n
    _same_tables(q, _python_table(files[key], fetch, contig, lo, hi, **kw))
  # the header text through the C ABI
  import ctypes as C
  need = C.c_uint64()
  _lib.check(_lib.lib().dv_cram_header(files[key].encode(), None, 0, C.byref(need)))
  buf = C.create_string_buffer(need.value)
  _lib.check(_lib.lib().dv_cram_header(files[key].encode(), buf, need.value, C.byref(need)))
  assert buf.raw.decode() == cram_reader.CramFile(files[key]).header_text and buf.raw.startswith(b'@HD\tVN:1.3')


def test_error_behaviour(files):
  ref = genomics_io.FastaReader(files['fasta'])
  # make_examples_test.py:551-568: --nouse_ref_for_cram on a CRAM written against an external reference
  with pytest.raises(ValueError, match='Failed to parse BAM/CRAM file.'):
    packing.ReadTable.from_cram(files['na12878_cram'], None, 'chr20', 10_000_000, 10_000_500)

  def tampered(contig, start, end):
    text = ref.get_bases(contig, start, end)
    if start <= 10_000_100 < end:
      k = 10_000_100 - start
      text = text[:k] + ('A' if text[k] != 'A' else 'C') + text[k + 1:]
    return text
  with pytest.raises(ValueError, match='MD5'):
    packing.ReadTable.from_cram(files['na12878_cram'], tampered, 'chr20', 10_000_000, 10_001_000)
  with pytest.raises(ValueError, match='contig not in the CRAM header'):
    packing.ReadTable.from_cram(files['na12878_cram'], ref.get_bases, 'chrNone', 0, 1000)

  # an exception inside the reference callback comes back as itself, on the caller's thread
  def broken(contig, start, end):
    raise KeyError('no such contig in the FASTA: ' + contig)
  with pytest.raises(KeyError, match='no such contig'):
    packing.ReadTable.from_cram(files['na12878_cram'], broken, 'chr20', 10_000_000, 10_001_000)
  # CRAM 3.1 (other codecs) is refused, not misread
  newer = os.path.join(files['tmp'], 'v31.cram')
  data = bytearray(open(files['na12878_cram'], 'rb').read())
  data[5] = 1
  with open(newer, 'wb') as f:
    f.write(bytes(data))
  with pytest.raises(ValueError, match='3.1 is not supported'):
    packing.ReadTable.from_cram(newer, ref.get_bases, 'chr20', 10_000_000, 10_001_000)
  truncated = os.path.join(files['tmp'], 'cut.cram')
  with open(truncated, 'wb') as f:
    f.write(bytes(data[:200_000]).replace(b'CRAM\x03\x01', b'CRAM\x03\x00', 1))
  with pytest.raises(ValueError):
    packing.ReadTable.from_cram(truncated, ref.get_bases, 'chr20')
  # use_original_quality_scores on reads without OQ tags: an error, as from the BAM (bam_reader.cpp)
  with pytest.raises(ValueError, match='OQ'):
    packing.ReadTable.from_cram(files['na12878_cram'], ref.get_bases, 'chr20', 10_000_000, 10_001_000,
                                use_original_quality_scores=True)


# ---- synthetic files: the rest of the format ------------------------------------------------------
def _contigs(rng):
  def seq(n):
    s = [rng.choice('ACGT') for _ in range(n)]
    for _ in range(n // 60):
      s[rng.randrange(n)] = 'N'
    s[n // 3:n // 3 + 200] = [c.lower() for c in s[n // 3:n // 3 + 200]]      # soft-masked stretch, as FASTAs have
    return ''.join(s)
  return [('c1', seq(3000)), ('c2', seq(2000))]


def _random_read(rng, ref_id, contig, pos, name, flag=0):
  """A read at 1-based `pos` with a random CIGAR (every operator) and mismatches."""
  cigar, seq = [], []
  refp = pos - 1
  if rng.random() < 0.2:
    cigar.append(('H', rng.randint(1, 5)))
  if rng.random() < 0.3:
    n = rng.randint(1, 6)
    cigar.append(('S', n))
    seq += [rng.choice('ACGT') for _ in range(n)]
  n_blocks = rng.randint(1, 4)
  for b in range(n_blocks):
    n = rng.randint(8, 30)
    for k in range(n):
      base = contig[refp + k].upper()
      seq.append(rng.choice('ACGTN') if rng.random() < 0.12 else base)
    cigar.append(('M', n))
    refp += n
    if b + 1 < n_blocks:
      kind = rng.random()
      if kind < 0.3:
        m = rng.randint(1, 4)
        cigar.append(('I', m))
        seq += [rng.choice('ACGT') for _ in range(m)]
      elif kind < 0.6:
        m = rng.randint(1, 5)
        cigar.append(('D', m))
        refp += m
      elif kind < 0.75:
        m = rng.randint(10, 40)
        cigar.append(('N', m))
        refp += m
      elif kind < 0.85:
        cigar.append(('P', rng.randint(1, 3)))
        m = rng.randint(1, 2)           # P sits between insertions
        cigar.append(('I', m))
        seq += [rng.choice('ACGT') for _ in range(m)]
      else:
        m = rng.randint(1, 3)
        cigar.append(('I', m))
        seq += [rng.choice('ACGT') for _ in range(m)]
  if rng.random() < 0.3:
    n = rng.randint(1, 6)
    cigar.append(('S', n))
    seq += [rng.choice('ACGT') for _ in range(n)]
  if rng.random() < 0.1:
    cigar.append(('H', rng.randint(1, 5)))
  seq = ''.join(seq)
  return dict(name=name, flag=flag, ref_id=ref_id, pos=pos, mapq=rng.randint(0, 60), cigar=cigar, seq=seq,
              qual=[rng.randint(2, 41) for _ in seq], cf=0x1)


def _reads_for(rng, contigs, ref_id, n, lo, hi, prefix):
  """n reads (some paired inside the list, some detached, some with tags) sorted by position."""
  name_of = lambda k: '%s%d' % (prefix, k)      # noqa: E731
  reads = []
  for k in range(n):
    pos = rng.randint(lo, hi)
    r = _random_read(rng, ref_id, contigs[ref_id][1], pos, name_of(k))
    roll = rng.random()
    if roll < 0.1:
      r['flag'] |= 0x400
    elif roll < 0.15:
      r['flag'] |= 0x200
    elif roll < 0.2:
      r['flag'] |= 0x100
    elif roll < 0.25:
      r['flag'] |= 0x800
    if rng.random() < 0.5:
      r['flag'] |= 0x10
    if rng.random() < 0.25:      # HP of every integer type, OQ strings
      kind = rng.choice('cCsSiI')
      value = rng.randint(0, 2)
      fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I'}[kind]
      import struct
      r.setdefault('tags', []).append((b'HP' + kind.encode(), struct.pack('<' + fmt, value)))
      r['hp'] = value
    if rng.random() < 0.3:
      oq = bytes(rng.randint(35, 70) for _ in r['seq'])
      r.setdefault('tags', []).append((b'OQZ', oq + b'\0'))
    if rng.random() < 0.1:
      r.setdefault('tags', []).append((b'XAZ', b'hello\0'))
    if rng.random() < 0.15:      # qualities as read features instead of an array
      r['cf'] = 0
    reads.append(r)
  reads.sort(key=lambda r: r['pos'])
  # pairs: some reads get a mate later in the list (attached), some a mate elsewhere (detached)
  free = list(range(len(reads)))
  rng.shuffle(free)
  while len(free) >= 2 and rng.random() < 0.8:
    a, b = sorted((free.pop(), free.pop()))
    ra, rb = reads[a], reads[b]
    if 'tags' in ra or 'tags' in rb:
      continue
    rb['name'] = ra['name']
    ra['flag'] |= 0x1 | 0x40 | (0x2 if rng.random() < 0.7 else 0)
    rb['flag'] |= 0x1 | 0x80 | (ra['flag'] & 0x2)
    ra['cf'] |= 0x4
    ra['nf'] = b - a - 1
  for r in reads:
    if not r['flag'] & 0x1 and 'tags' not in r and rng.random() < 0.2:
      r['flag'] |= 0x1 | 0x40
      r['cf'] |= 0x2
      other = rng.random() < 0.5
      r['mf'] = rng.randint(0, 1)               # mate reverse / (never unmapped here)
      r['mate_ref'] = (1 - ref_id) if other else ref_id
      r['mate_pos'] = rng.randint(1, 1500)
      r['tlen'] = 0 if other else rng.randint(-800, 800)
  return reads


def _expected_rows(reads, contigs, contig_id, lo, hi, min_mapq=0, keep_all=False):
  """(key, 0-based start, cigar words, bases, qualities, mapq, reverse) of the reads a query must return."""
  ops = {'M': 1, 'I': 2, 'D': 3, 'N': 4, 'S': 5, 'H': 6, 'P': 7}
  rows = []
  for r in reads:
    if r['flag'] & 0x4 or r['ref_id'] != contig_id:
      continue
    if not keep_all and r['flag'] & (0x400 | 0x200 | 0x100 | 0x800):
      continue
    paired = bool(r['flag'] & 0x1)
    if not keep_all and paired and not r['flag'] & 0x2 and r.get('cf', 0) & 0x2 and r['mate_ref'] != r['ref_id']:
      continue
    if r['mapq'] < min_mapq:
      continue
    span = max(cram_writer._ref_len(r['cigar']), 1)     # pylint: disable=protected-access
    if not (hi > r['pos'] - 1 and lo < r['pos'] - 1 + span):
      continue
    merged = []
    for op, n in r['cigar']:
      if merged and merged[-1][0] == op:
        merged[-1][1] += n
      else:
        merged.append([op, n])
    rows.append(('%s/%d' % (r['name'], 0 if (not paired or r['flag'] & 0x40) else 1), r['pos'] - 1,
                 [(n << 4) | ops[op] for op, n in merged], r['seq'], r['qual'], r['mapq'], bool(r['flag'] & 0x10),
                 r.get('hp')))
  return rows


def _check_table(t, rows):
  assert t.n_reads == len(rows)
  assert t.keys == [r[0] for r in rows]
  assert t.read_pos.tolist() == [r[1] for r in rows]
  for i, r in enumerate(rows):
    assert t.cigar[int(t.read_cigar_off[i]):int(t.read_cigar_off[i + 1])].tolist() == r[2], (i, r[0])
    assert bytes(t.bases[int(t.read_seq_off[i]):int(t.read_seq_off[i + 1])]).decode() == r[3], (i, r[0])
    assert t.quals[int(t.read_seq_off[i]):int(t.read_seq_off[i + 1])].tolist() == r[4], (i, r[0])
    assert int(t.read_mapq[i]) == r[5] and bool(int(t.read_flags[i]) & 1) == r[6]
    assert int(t.read_hp[i]) == (r[7] if r[7] is not None else _lib.DV_HP_NONE)


BIT_CODECS = {
    'BF': ('huffman',), 'CF': ('huffman',), 'RI': ('huffman',), 'RL': ('beta', 0, 10), 'AP': ('subexp', 0, 3),
    'RG': ('gamma', 2), 'RN': ('stop', 0, 11), 'MF': ('huffman',), 'NS': ('external', 12), 'NP': ('external', 12),
    'TS': ('external', 13), 'NF': ('gamma', 1), 'TL': ('huffman',), 'FN': ('beta', 0, 8), 'FC': ('huffman',),
    'FP': ('subexp', 0, 2), 'BA': ('huffman',), 'QS': ('huffman',), 'BS': ('beta', 0, 2),
    'IN': ('len', ('gamma', 1), ('huffman',)), 'SC': ('stop', 9, 14), 'DL': ('gamma', 0), 'RS': ('subexp', 0, 4),
    'HC': ('beta', 0, 4), 'PD': ('beta', 0, 3), 'MQ': ('huffman',), 'BB': ('len', ('external', 15), ('external', 15)),
    'QQ': ('len', ('beta', 0, 9), ('huffman',)),
}
EXTERNAL_CODECS = {
    'BF': ('external', 1), 'CF': ('external', 2), 'RI': ('external', 3), 'RL': ('external', 4), 'AP': ('external', 5),
    'RG': ('external', 6), 'RN': ('stop', 0, 7), 'MF': ('external', 8), 'NS': ('external', 9), 'NP': ('external', 10),
    'TS': ('external', 11), 'NF': ('external', 12), 'TL': ('external', 13), 'FN': ('external', 14), 'FC': ('external', 15),
    'FP': ('external', 16), 'BA': ('external', 17), 'QS': ('external', 18), 'BS': ('external', 19),
    'IN': ('stop', 0, 20), 'SC': ('len', ('external', 21), ('external', 22)), 'DL': ('external', 23), 'RS': ('external', 24),
    'HC': ('external', 25), 'PD': ('external', 26), 'MQ': ('external', 27), 'BB': ('stop', 1, 28),
    'QQ': ('len', ('external', 29), ('external', 29)),
}
_METHODS = ['raw', 'gzip', 'bzip2', 'lzma', 'rans0', 'rans1']


def _fetch(contigs):
  table = dict(contigs)
  return lambda name, lo, hi: table[name][max(lo, 0):hi].upper()


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_bit_stream_codecs_and_every_read_feature(files, seed):
  """One slice per container, every series through a core-block codec where the format allows it
  (HUFFMAN alphabets with mixed code lengths, BETA, SUBEXP, GAMMA, BYTE_ARRAY_LEN over bit codecs), the core
  block gzip-compressed; stored names, delta positions; tags (HP of every integer type, OQ)."""
  rng = random.Random(seed)
  contigs = _contigs(rng)
  w = cram_writer.CramWriter(contigs, BIT_CODECS, {0: 'gzip', 11: 'rans1', 12: 'bzip2', 14: 'lzma', 15: 'rans0'}, _rans_encode)
  groups = [(_reads_for(rng, contigs, 0, 60, 1, 1400, 'a'), 0), (_reads_for(rng, contigs, 0, 40, 1300, 2700, 'b'), 0),
            (_reads_for(rng, contigs, 1, 50, 1, 1800, 'c'), 1)]
  for reads, rid in groups:
    w.add_container([(reads, rid)])
  path = os.path.join(files['tmp'], 'bits%d.cram' % seed)
  w.finish(path)
  fetch = _fetch(contigs)
  everything = [r for reads, _ in groups for r in reads]
  for contig_id, name in enumerate(('c1', 'c2')):
    for lo, hi in ((0, 1 << 40), (1350, 1500), (0, 10), (2900, 3000)):
      for kw in (dict(), dict(keep_duplicates=True, keep_failed_qc=True, keep_secondary=True, keep_supplementary=True,
                              keep_improperly_placed=True), dict(min_mapping_quality=30)):
        t = packing.ReadTable.from_cram(path, fetch, name, lo, hi, n_threads=3, **kw)
        _check_table(t, _expected_rows(everything, contigs, contig_id, lo, hi, kw.get('min_mapping_quality', 0),
                                       'keep_duplicates' in kw))
        _same_tables(t, _python_table(path, fetch, name, lo, hi, **kw))


@pytest.mark.parametrize('seed,read_names,ap_delta', [(11, True, False), (12, False, True), (13, False, False)])
def test_block_codecs_slices_lossy_names_and_multi_reference_containers(files, seed, read_names, ap_delta):
  """Every series EXTERNAL, the blocks cycling through raw / gzip / bzip2 / lzma / rANS order 0 / 1; containers
  of several slices, a multi-reference slice, a slice of unmapped reads, embedded references (decoded
  WITHOUT a FASTA); names dropped by the writer (generated: one per template); absolute positions."""
  rng = random.Random(seed)
  contigs = _contigs(rng)
  methods = {cid: _METHODS[(cid + seed) % len(_METHODS)] for cid in range(0, 40)}
  methods.update({cid: _METHODS[(cid + seed) % len(_METHODS)] for cid in range(200, 210)})
  w = cram_writer.CramWriter(contigs, EXTERNAL_CODECS, methods, _rans_encode, read_names=read_names, ap_delta=ap_delta)
  g1 = _reads_for(rng, contigs, 0, 40, 1, 900, 'a')
  g2 = _reads_for(rng, contigs, 0, 40, 800, 1900, 'b')
  g3 = _reads_for(rng, contigs, 0, 30, 1800, 2700, 'c')
  for g in (g1, g2, g3):          # several slices per container share one tag dictionary: none here
    for r in g:
      r.pop('tags', None)
      r.pop('hp', None)
  w.add_container([(g1, 0), (g2, 0)])
  w.add_container([(g3, 0)])
  # a multi-reference slice: reads of both contigs + an unmapped one in the middle
  m1 = _reads_for(rng, contigs, 0, 12, 2650, 2750, 'm')
  m2 = _reads_for(rng, contigs, 1, 12, 1, 300, 'n')
  for r in m1 + m2:
    if r.get('cf', 0) & 0x4:      # keep attached mates on one reference
      pass
  unmapped = dict(name='u0', flag=0x4, ref_id=-1, pos=0, mapq=0, cigar=[], seq='ACGTNACGT', qual=[30] * 9, cf=0x1)
  multi = m1 + [unmapped] + m2
  if not ap_delta:
    w.add_container([(multi, -2)])
  else:                           # delta positions need one reference per slice
    w.add_container([(m1, 0)])
    w.add_container([(m2, 1)])
  c2 = _reads_for(rng, contigs, 1, 40, 250, 1800, 'd')
  w.add_container([(c2, 1)])
  w.add_container([([dict(unmapped, name='u1'), dict(unmapped, name='u2')], -1)])
  path = os.path.join(files['tmp'], 'ext%d.cram' % seed)
  w.finish(path)
  fetch = _fetch(contigs)
  everything = g1 + g2 + g3 + m1 + m2 + c2
  bare = path.replace('.cram', '_noindex.cram')
  shutil.copy(path, bare)
  for contig_id, name in enumerate(('c1', 'c2')):
    for lo, hi in ((0, 1 << 40), (850, 1000), (2600, 2800), (200, 320)):
      want = _expected_rows(everything, contigs, contig_id, lo, hi)
      py = _python_table(path, fetch, name, lo, hi)
      for p in (path, bare):
        t = packing.ReadTable.from_cram(p, fetch, name, lo, hi, n_threads=4)
        if read_names:
          _check_table(t, want)
        else:                     # generated names: everything but the keys is what went in
          assert t.n_reads == len(want) and t.read_pos.tolist() == [r[1] for r in want]
          assert bytes(t.bases).decode() == ''.join(r[3] for r in want)
        _same_tables(t, py)
  if not read_names:
    # lossy names: the two mates of an attached pair share ONE generated name (htslib does the same), and
    # their read numbers tell them apart
    t = packing.ReadTable.from_cram(path, fetch, 'c1', keep_duplicates=True, keep_failed_qc=True, keep_secondary=True,
                                    keep_supplementary=True, keep_improperly_placed=True)
    names = [k.rsplit('/', 1)[0] for k in t.keys]
    pairs = sum(1 for r in g1 + g2 + g3 if r.get('cf', 0) & 0x4)
    assert pairs >= 1 and len(names) - len(set(names)) == pairs      # and no name twice otherwise
  # the same reads with the reference EMBEDDED in every slice: no FASTA needed, none consulted
  w2 = cram_writer.CramWriter(contigs, EXTERNAL_CODECS, methods, _rans_encode, read_names=True, ap_delta=True)
  w2.add_container([(g1, 0), (g2, 0)], embed=True, ref_required=False)
  w2.add_container([(c2, 1)], embed=True)
  emb = os.path.join(files['tmp'], 'embedded%d.cram' % seed)
  w2.finish(emb)
  if read_names:
    t = packing.ReadTable.from_cram(emb, None, 'c1', n_threads=2)
    _check_table(t, _expected_rows(g1 + g2, contigs, 0, 0, 1 << 40))
    _same_tables(t, _python_table(emb, None, 'c1'))
    _check_table(packing.ReadTable.from_cram(emb, None, 'c2'), _expected_rows(c2, contigs, 1, 0, 1 << 40))


def test_template_lengths_of_attached_mates_follow_htslib(files):
  """cram_decode_slice_xref: the leftmost read of a template gets +(rightmost end - leftmost start + 1), the
  others the negative; a detached read keeps its stored TS."""
  contig = ('c1', 'ACGT' * 200)
  mk = lambda name, pos, flag, cf, **kw: dict(name=name, flag=flag, ref_id=0, pos=pos, mapq=40, cigar=[('M', 20)],      # noqa: E731
                                             seq=contig[1][pos - 1:pos + 19], qual=[30] * 20, cf=cf, **kw)
  reads = [mk('p', 101, 0x1 | 0x2 | 0x40, 0x1 | 0x4, nf=1), mk('solo', 150, 0, 0x1), mk('p', 301, 0x1 | 0x2 | 0x80 | 0x10, 0x1),
           mk('far', 400, 0x1 | 0x2 | 0x40, 0x1 | 0x2, mf=1, mate_ref=0, mate_pos=700, tlen=321)]
  w = cram_writer.CramWriter([contig], EXTERNAL_CODECS, {}, _rans_encode)
  w.add_container([(reads, 0)])
  path = os.path.join(files['tmp'], 'tlen.cram')
  w.finish(path)
  t = packing.ReadTable.from_cram(path, _fetch([contig]), 'c1')
  assert t.keys == ['p/0', 'solo/0', 'p/1', 'far/0']
  assert t.read_frag_len.tolist() == [220, 0, -220, 321]
  assert [int(f) & 1 for f in t.read_flags] == [0, 0, 1, 0]
  _same_tables(t, _python_table(path, _fetch([contig]), 'c1'))


@pytest.mark.parametrize('offset', [-1, -2, -4, -(1 << 40), 1 << 62, (1 << 63) - 1])
def test_index_rows_that_point_outside_the_file_are_refused(files, offset):
  """A .crai is gzip text: its container offsets are checked against the file before anything is read through
  them (an offset of -1 ... -4 used to wrap the bounds check and crash the process)."""
  import gzip
  ref = genomics_io.FastaReader(files['fasta'])
  bad = os.path.join(files['tmp'], 'bad_index_%d.cram' % (offset & 0xffff))
  shutil.copy(files['na12878_cram'], bad)
  with gzip.open(files['na12878_crai'], 'rt') as f:
    rows = [l.rstrip('\n').split('\t') for l in f if l.strip()]
  rows[len(rows) // 2][3] = str(offset)
  with gzip.open(bad + '.crai', 'wt') as f:
    f.write(''.join('\t'.join(r) + '\n' for r in rows))
  with pytest.raises((ValueError, RuntimeError, _lib.DvError)):
    packing.ReadTable.from_cram(bad, ref.get_bases, 'chr20', 10_000_000, 10_100_000)


@pytest.mark.parametrize('method', ['rans0', 'rans1', 'bzip2'])
def test_constant_quality_blocks_expand_by_more_than_a_naive_bound(files, method):
  """A slice whose qualities are ONE value (binned / unavailable qualities, as real CRAMs hold): rANS 4x8 codes a
  symbol of frequency 4095 / 4096 in 3.5e-4 bits, so 400,000 quality bytes are a payload of a few dozen bytes --
  20,000x, beyond the 4096x-per-input-byte guard round 5 had put in front of the allocation."""
  rng = random.Random(5)
  contigs = _contigs(rng)
  ref = contigs[0][1].upper()
  reads = []
  for k in range(2000):
    pos = 1 + (k * 2600) // 2000
    reads.append(dict(name='q%d' % k, flag=0, ref_id=0, pos=pos, mapq=60, cigar=[('M', 200)],
                      seq=ref[pos - 1:pos - 1 + 200].replace('N', 'A'), qual=[30] * 200, cf=0x1))
  methods = {cid: 'gzip' for cid in range(0, 40)}
  methods[18] = method                      # QS
  w = cram_writer.CramWriter(contigs, EXTERNAL_CODECS, methods, _rans_encode)
  w.add_container([(reads, 0)])
  path = os.path.join(files['tmp'], 'constq_%s.cram' % method)
  w.finish(path)
  assert os.path.getsize(path) < 120_000            # 400,000 qualities + 400,000 bases in < 120 KB
  t = packing.ReadTable.from_cram(path, _fetch(contigs), 'c1', n_threads=2)
  assert t.n_reads == 2000 and t.quals.size == 400_000 and int(t.quals.min()) == 30 == int(t.quals.max())
  _same_tables(t, _python_table(path, _fetch(contigs), 'c1'))
