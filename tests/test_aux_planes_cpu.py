"""Per-base planes from aux tags in the native alignment-file decoders (include/dvhip.h ABI v6,
deepvariant_amd/csrc/aux_planes.h):

  * base modifications: MM / ML / MN -> Read.base_modifications['5mC' / '6mA'], nucleus' ParseBaseModifications
    (third_party/nucleus/io/sam_reader.cc:521-719).  Pinned by the nine known-answer cases of
    third_party/nucleus/io/sam_reader_test.cc:543-700 (Parse5mCAuxTagTest.*, Parse6mATagTest.*), run through the native
    BAM reader, the native CRAM reader and the Python restatement (genomics_io.parse_base_modifications); the two
    restatements are then compared on random tags that exercise the function's quirks;
  * the Ultima flow-space tags tp / t0 -> flow planes -> the Read objects' info -> the per-base channel pixels.
"""
import os

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import genomics_io as gio
from deepvariant_amd import packing


def _read(seq, name='read_name', start=1, reverse=False, info=None):
  r = T.Read(fragment_name=name, read_number=0, number_reads=1, aligned_sequence=seq,
             aligned_quality=bytes([30] * len(seq)),
             alignment=T.LinearAlignment(position=T.Position('chr1', start, reverse), mapping_quality=60,
                                         cigar=[T.CigarUnit(1, len(seq))]))
  for k, v in (info or {}).items():
    if isinstance(v, str):
      r.info[k] = T.ListValue(values=[T.Value(string_value=v)])
    elif isinstance(v, int):
      r.info[k] = T.ListValue(values=[T.Value(int_value=v)])
    else:
      r.info[k] = T.ListValue(values=[T.Value(int_value=int(x)) for x in v])
  return r


# third_party/nucleus/io/sam_reader_test.cc:543-700: (test name, sequence, reverse, MM, ML, MN, {modification: plane} | {}).
# The reference's tests look at the entries listed here (ThreeModifications also yields a 6mA entry, from its `T-a.`
# specification, which the test does not look at); {} means "no entry at all".
REFERENCE_CASES = [
    ('BasicCase', 'TCTCTCTCTCTCTCTCTCTC', False, 'C+m?,1,1,1,1,1', [1, 2, 3, 4, 5], None,
     {'5mC': [0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5]}),
    ('MultipleModifications', 'ACACACACACTCTCTCTCTC', False, 'A-a.,0,0,0,0,0;C+m?,1,1,1,1,1',
     [1, 1, 1, 1, 1, 2, 2, 2, 2, 2], None, {'5mC': [0, 0, 0, 2] * 5}),
    ('ThreeModifications', 'ACACACACACTCTCTCTCTC', False, 'A-a.,0,0,0,0,0;C+m?,1,1,1,1,1;T-a.,0,0,0,0,0',
     [1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1], None, {'5mC': [0, 0, 0, 2] * 5}),
    ('VariableMMDelta', 'CACAACAAACAAAAC', False, 'A-a.,0,0,0,0,0;C+m?,0,3', [0, 0, 0, 0, 0, 1, 2], None,
     {'5mC': [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2]}),
    ('ReverseStrandModifications', 'TTTTTGGGGG', True, 'C+m?,0,0,0,0,0', [1, 1, 1, 1, 1], None,
     {'5mC': [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]}),
    ('MismatchMNTag', 'CCCCCTTTTT', False, 'C+m?,0,0,0,0,0', [1, 1, 1, 1, 1], 11, {}),
    ('MatchMNTag', 'CCCCCTTTTT', False, 'C+m?,0,0,0,0,0', [1, 1, 1, 1, 1], 10, {'5mC': [1, 1, 1, 1, 1, 0, 0, 0, 0, 0]}),
    ('Parse5mCand6mA', 'ACCCAGGGTGGGTGGG', False, 'C+m?,0,0,0;A+a?,0,0;T-a?,0,0', [7, 8, 9, 1, 2, 3, 4], None,
     {'5mC': [0, 7, 8, 9] + [0] * 12, '6mA': [1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0]}),
]


def _same(got, want, name):
  if not want:
    assert got == {}, name
  for k, plane in want.items():
    assert got.get(k) == plane, (name, k)
  if name == 'ThreeModifications':
    assert got.get('6mA') == [0] * 10 + [1, 0] * 5, name      # (T-a. on the forward strand: every T)
  else:
    assert set(got) == set(want), name


def _case_reads():
  reads = []
  for k, (name, seq, reverse, mm, ml, mn, _) in enumerate(REFERENCE_CASES):
    info = {'MM': mm, 'ML': ml}
    if mn is not None:
      info['MN'] = mn
    reads.append(_read(seq, name=name, start=10 + 40 * k, reverse=reverse, info=info))
  return reads


def _planes(table, i):
  s0, s1 = int(table.read_seq_off[i]), int(table.read_seq_off[i + 1])
  out = {}
  if table.read_flags[i] & packing.DV_READ_HAS_5MC:
    out['5mC'] = table.mod_5mc[s0:s1].tolist()
  if table.read_flags[i] & packing.DV_READ_HAS_6MA:
    out['6mA'] = table.mod_6ma[s0:s1].tolist()
  return out


def test_reference_vectors_python_restatement():
  for r, case in zip(_case_reads(), REFERENCE_CASES):
    _same({k: list(v) for k, v in gio.parse_base_modifications(r).items()}, case[6], case[0])


def test_reference_vectors_native_bam(tmp_path):
  path = str(tmp_path / 'mods.bam')
  reads = _case_reads()
  gio.write_bam(path, [('chr1', 1000)], reads)
  table = packing.ReadTable.from_bam(path, 'chr1', 0, 1000, parse_base_modifications=True)
  assert table.n_reads == len(REFERENCE_CASES) and table.mod_5mc is not None and table.mod_6ma is not None
  for i, case in enumerate(REFERENCE_CASES):
    _same(_planes(table, i), case[6], case[0])
  # the Read objects made from the table carry what nucleus' reader would have put on them
  objs = table.to_reads('chr1')
  for r, case in zip(objs, REFERENCE_CASES):
    _same({k: list(v) for k, v in r.base_modifications.items()}, case[6], case[0])
  # not asked for: no planes, no flags
  plain = packing.ReadTable.from_bam(path, 'chr1', 0, 1000)
  assert plain.mod_5mc is None and not (plain.read_flags & 12).any()
  # and the Python reader with aux_fields agrees
  _, py = gio.read_bam(path, 'chr1', aux_fields=('MM', 'ML', 'MN'))
  for r, case in zip(py, REFERENCE_CASES):
    _same({k: list(v) for k, v in r.base_modifications.items()}, case[6], case[0])


def test_reference_vectors_native_cram(tmp_path):
  """The same tags through a CRAM's tag series (BAM-encoded values under a tag dictionary)."""
  import struct
  from tests import cram_writer as W
  from tests.test_cram_native_cpu import EXTERNAL_CODECS, _rans_encode
  records = []
  for r in _case_reads():
    tags = [(b'MMZ', r.info['MM'].values[0].string_value.encode() + b'\0'),
            (b'MLB', b'C' + struct.pack('<I', len(r.info['ML'].values)) + bytes(v.int_value for v in r.info['ML'].values))]
    if 'MN' in r.info:
      tags.append((b'MNi', struct.pack('<i', r.info['MN'].values[0].int_value)))
    n = len(r.aligned_sequence)
    records.append(dict(name=r.fragment_name, flag=0x10 if r.alignment.position.reverse_strand else 0, ref_id=0,
                        pos=r.alignment.position.position + 1, mapq=60, cigar=[('M', n)], seq=r.aligned_sequence,
                        qual=[30] * n, cf=0x1, tags=tags))
  contig = ('chr1', 'A' * 1000)
  w = W.CramWriter([contig], EXTERNAL_CODECS, {}, _rans_encode)
  # (one tag dictionary per container: the reads with and without MN go into containers of their own)
  for group in ([r for r in records if len(r['tags']) == 2], [r for r in records if len(r['tags']) == 3]):
    w.add_container([(group, 0)], embed=True)
  path = str(tmp_path / 'mods.cram')
  w.finish(path)
  table = packing.ReadTable.from_cram(path, None, 'chr1', 0, 1000, parse_base_modifications=True)
  assert table.n_reads == len(REFERENCE_CASES)
  by_name = {case[0]: case for case in REFERENCE_CASES}
  for i, key in enumerate(table.keys):
    case = by_name[key.rpartition('/')[0]]
    _same(_planes(table, i), case[6], case[0])


def _random_mod_read(rng, k):
  n = int(rng.integers(5, 60))
  seq = ''.join('ACGTN'[int(i)] for i in rng.choice(5, size=n, p=[.3, .3, .15, .2, .05]))
  specs, ml = [], []
  for _ in range(int(rng.integers(0, 5))):
    spec = ['C+m?', 'C+m', 'C+m.', 'A+a?', 'T-a.', 'A-a.', 'C+h?', 'G+o', 'C+76792', 'N+n?'][int(rng.integers(0, 10))]
    cnt = int(rng.integers(0, 7))
    deltas = [int(rng.integers(0, 4)) for _ in range(cnt)]
    specs.append(','.join([spec] + [str(d) for d in deltas]))
    ml += [int(v) for v in rng.integers(0, 256, size=cnt)]
  if ml and rng.random() < 0.15:
    ml = ml[:-int(rng.integers(1, len(ml) + 1))] or [7]          # ML shorter than MM announces
  info = {}
  if specs or rng.random() < 0.5:
    info['MM'] = ';'.join(specs) + (';' if rng.random() < 0.7 else '')
  if ml or rng.random() < 0.3:
    info['ML'] = ml
  if rng.random() < 0.2:
    info['MN'] = n if rng.random() < 0.6 else n + 1
  return _read(seq, name='r%03d' % k, start=5 + 70 * k, reverse=bool(rng.random() < 0.5), info=info)


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_native_reader_equals_the_python_restatement_on_random_tags(tmp_path, seed):
  rng = np.random.default_rng(seed)
  reads = [_random_mod_read(rng, k) for k in range(150)]
  path = str(tmp_path / 'fuzz.bam')
  gio.write_bam(path, [('chr1', 20000)], reads)
  table = packing.ReadTable.from_bam(path, 'chr1', 0, 20000, parse_base_modifications=True)
  assert table.n_reads == len(reads)
  n_mod = n_signed = 0
  for i, r in enumerate(reads):
    want = {k: list(v) for k, v in gio.parse_base_modifications(r).items()}
    assert _planes(table, i) == want, (seed, i, r.info, r.aligned_sequence, r.alignment.position.reverse_strand)
    n_mod += bool(want)
    n_signed += any(v > 127 for plane in want.values() for v in plane)
  assert n_mod > 20 and n_signed > 5


def test_flow_tags_reach_the_channel_planes(tmp_path):
  from tests import fuzz_inputs as FZ
  rng = np.random.default_rng(3)
  _, _, reads, _, _ = FZ.make_case(rng, 71, 40, with_ultima=True)
  for k, r in enumerate(reads):       # distinct names, plain matches, sorted positions: a file the reader keeps whole
    r.fragment_name, r.read_number, r.number_reads = 'u%02d' % k, 0, 1
    r.alignment.cigar = [T.CigarUnit(1, len(r.aligned_sequence))]
    r.alignment.position.position = 10 + 3 * k
    r.alignment.mapping_quality = 60
    r.supplementary_alignment = False
    if hasattr(r, '_dv_packed'):
      del r._dv_packed
  path = str(tmp_path / 'flow.bam')
  gio.write_bam(path, [('chr1', 5000)], reads)
  table = packing.ReadTable.from_bam(path, 'chr1', 0, 5000, parse_flow_tags=True)
  assert table.n_reads == len(reads) and table.flow_tp is not None and table.mod_5mc is None
  planes = packing.seq_aux_planes([1, 28, 29, 30])
  want = packing.ReadTable.from_reads(reads, need_seq_aux=planes)
  objs = table.to_reads('chr1')
  got = packing.ReadTable.from_reads(objs, need_seq_aux=planes)
  for name in ('base_aux0', 'base_aux1', 'base_aux2'):
    assert np.array_equal(getattr(got, name), getattr(want, name)), name
  assert sum('tp' in r.info for r in objs) == sum('tp' in r.info for r in reads) > 10
  assert sum('t0' in r.info for r in objs) == sum('t0' in r.info for r in reads) > 10
  sub = table.take(np.array([5, 1, 30]))
  assert np.array_equal(sub.flow_present, table.flow_present[[5, 1, 30]])
  s0, s1 = int(table.read_seq_off[5]), int(table.read_seq_off[6])
  assert np.array_equal(sub.flow_tp[:s1 - s0], table.flow_tp[s0:s1])


def test_make_examples_asks_for_the_tags_its_channels_need(tmp_path):
  """resolve_sam_aux_fields (make_examples_core.py:288-373): a base-modification channel in --channel_list makes the
  region reader parse MM / ML / MN; the reads it hands to the region chain -- objects or a table -- carry the planes."""
  from deepvariant_amd import make_examples as me
  path = str(tmp_path / 'mods.bam')
  gio.write_bam(path, [('chr1', 1000)], _case_reads())
  base = ['--ref', 'x', '--reads', path, '--examples', 'e', '--min_mapping_quality', '0']
  plain = me.RegionReads(me.build_arg_parser().parse_args(base))
  region = T.Range('chr1', 0, 1000)
  assert all(not r.base_modifications for r in plain(region)) and plain.table(region).mod_5mc is None
  args = me.build_arg_parser().parse_args(base + ['--channel_list', 'BASE_CHANNELS,base_methylation'])
  with_mods = me.RegionReads(args)
  got = {r.fragment_name: {k: list(v) for k, v in r.base_modifications.items()} for r in with_mods(region)}
  for case in REFERENCE_CASES:
    _same(got[case[0]], case[6], case[0])
  table = with_mods.table(T.Range('chr1', 0, 100))           # the first three reads
  assert table.n_reads == 3 and table.mod_5mc is not None and (table.read_flags & packing.DV_READ_HAS_5MC).all()
  # the packed batch hands the plane to the encoder as dv_batch::mod_5mc
  batch = packing.PackedBatch(table=table, width=21)
  c, keep = batch.to_ctypes()
  assert c.mod_5mc and c.mod_6ma
  del keep


def test_malformed_aux_blocks_do_not_break_the_decoder(tmp_path):
  """Aux blocks come from untrusted files: random bytes, truncated values, arrays that announce more elements than the
  record holds, MM strings without numbers.  The reader keeps what it parsed before the damage (as ParseAuxFields
  does), refuses what the reference aborts on (a position that is not a number: BAD_INPUT), and never reads outside
  the record -- every read comes back with planes of its own length."""
  import struct
  rng = np.random.default_rng(99)
  reads = []
  for k in range(400):
    n = int(rng.integers(1, 40))
    r = _read(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=n)), name='m%03d' % k, start=5 + 45 * k,
              reverse=bool(rng.integers(0, 2)))
    kind = int(rng.integers(0, 7))
    if kind == 0:
      raw = bytes(rng.integers(0, 256, size=int(rng.integers(0, 40))).astype(np.uint8))
    elif kind == 1:      # an array that announces far more elements than follow
      raw = b'MLBC' + struct.pack('<I', int(rng.integers(5, 1 << 31))) + b'\x01\x02'
    elif kind == 2:      # a string without its terminator, at the end of the record
      raw = b'MMZC+m?,0,0'
    elif kind == 3:      # valid MM, ML of another array type, MN of a small type
      raw = b'MMZC+m?,0;\0' + b'MLBs' + struct.pack('<Ih', 1, 300) + b'MNC' + bytes([n & 0xFF])
    elif kind == 4:      # tp as 32-bit integers (cut to int8), t0 longer than the read
      raw = b'tpBi' + struct.pack('<I3i', 3, 1, -2, 300) + b't0Z' + b'5' * (n + 7) + b'\0'
    elif kind == 5:      # unknown array subtype
      raw = b'MLBZ' + struct.pack('<I', 2) + b'ab'
    else:                # float array where integers are expected, hex string where a string is expected
      raw = b'MLBf' + struct.pack('<I2f', 2, 1.5, 2.5) + b'MMH4142\0'
    r._aux_raw = raw       # pylint: disable=protected-access
    reads.append(r)
  path = str(tmp_path / 'bad_aux.bam')
  gio.write_bam(path, [('chr1', 40000)], reads)
  table = packing.ReadTable.from_bam(path, 'chr1', 0, 40000, parse_base_modifications=True, parse_flow_tags=True)
  assert table.n_reads == len(reads)
  total = int(table.read_seq_off[-1])
  assert table.mod_5mc.size == table.mod_6ma.size == table.flow_tp.size == table.flow_t0.size == total
  # kind 4: values cut to a byte, the rest of the read zero; the long t0 cut to the read
  i = next(k for k, r in enumerate(reads) if r._aux_raw.startswith(b'tpBi') and len(r.aligned_sequence) >= 4)      # pylint: disable=protected-access
  s0, s1 = int(table.read_seq_off[i]), int(table.read_seq_off[i + 1])
  assert table.flow_tp[s0:s0 + 4].tolist() == [1, -2, 300 - 256, 0] and table.flow_present[i] == 3
  assert set(table.flow_t0[s0:s1].tolist()) == {ord('5') - 33}
  # what the reference aborts on is refused, with a message
  bad = _read('ACGTCC', name='bad', info={'MM': 'C+m?,x', 'ML': [1]})
  gio.write_bam(path, [('chr1', 40000)], [bad])
  with pytest.raises(Exception, match='not a number'):
    packing.ReadTable.from_bam(path, 'chr1', 0, 40000, parse_base_modifications=True)
  assert packing.ReadTable.from_bam(path, 'chr1', 0, 40000).n_reads == 1      # (not asked for: not looked at)


def test_methylation_from_a_bam_reaches_the_pixels(tmp_path):
  """End to end on the CPU: MM / ML tags in a BAM -> the native decoder's 5mC plane -> the packed batch of a calling
  region (table path: no Read objects) -> pile-ups with the base_methylation channel.  The device encoder is replaced
  by the oracle's packed adapter (no GPU here); the expectation is drawn by the oracle -- and by the reference build --
  from Read objects whose base_modifications the PYTHON restatement parsed."""
  from deepvariant_amd import make_examples_native as men
  from oracle import oracle as O
  from tests import fuzz_inputs as FZ
  from tests.test_reference_examples_cpu import OracleDeviceEncoder
  rng = np.random.default_rng(12)
  L = 600
  ref_seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=L))

  class Ref:
    def n_bases(self, contig):
      return L

    def get_bases(self, contig, start, end):
      return ref_seq[start:end]

  reads = []
  for k in range(90):
    n = int(rng.integers(60, 140))
    start = int(rng.integers(150, 330))
    seq = list(ref_seq[start:start + n])
    for _ in range(3):
      seq[int(rng.integers(0, n))] = 'ACGT'[int(rng.integers(0, 4))]
    seq = ''.join(seq)
    reverse = bool(rng.integers(0, 2))
    strand_seq = seq[::-1].translate(str.maketrans('ACGT', 'TGCA')) if reverse else seq
    n_c = strand_seq.count('C')
    info = {}
    if n_c >= 2 and rng.random() < 0.8:
      cnt = int(rng.integers(1, n_c + 1))
      info['MM'] = 'C+m?,' + ','.join('0' for _ in range(cnt)) + ';'
      info['ML'] = [int(v) for v in rng.integers(1, 256, size=cnt)]
    r = _read(seq, name='m%03d' % k, start=start, reverse=reverse, info=info)
    r.aligned_quality = bytes(rng.integers(20, 41, size=n).astype(np.uint8))
    reads.append(r)
  path = str(tmp_path / 'meth.bam')
  gio.write_bam(path, [('chr1', L)], reads)
  channels = list(T.PILEUP_DEFAULT_CHANNELS) + ['base_methylation']
  width, height = 41, 40
  pic = FZ.options(channels, width, height, min_mapq=0)
  pic.num_channels = len(channels)
  options = T.MakeExamplesOptions(pic_options=pic, sample_options=[
      T.SampleOptions(role='main', name='s', pileup_height=height, order=[0])])
  table = packing.ReadTable.from_bam(path, 'chr1', 0, L, parse_base_modifications=True)
  assert table.n_reads == len(reads) and (table.read_flags & packing.DV_READ_HAS_5MC).sum() > 40
  _, py_reads = gio.read_bam(path, 'chr1', aux_fields=('MM', 'ML', 'MN'))
  cands = []
  for pos in (230, 260, 290, 320):
    refb = ref_seq[pos]
    alt = [b for b in 'ACGT' if b != refb][0]
    near = [r for r in reads if r.alignment.position.position <= pos < r.alignment.position.position + len(r.aligned_sequence)]
    cands.append(T.DeepVariantCall(variant=T.Variant('chr1', pos, pos + 1, refb, [alt]),
                                   allele_support={alt: T.SupportingReads(['%s/0' % r.fragment_name for r in near[::3]])}))
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=Ref())
  gen._device_encoder = OracleDeviceEncoder(pic)       # pylint: disable=protected-access
  examples, shape = gen.encode_region(cands, [table], [0], [0.0], {}, role='main')
  assert shape == [height, width, len(channels)] and len(examples) == len(cands)
  from deepvariant_amd import protowire as pw
  hw = (width - 1) // 2
  meth = 0
  for ex, cand in zip(examples, cands):
    img = np.frombuffer(pw.decode_example(ex)['image/encoded'][0], np.uint8).reshape(shape)
    v = cand.variant
    overlapping = [r for r in py_reads                      # (file order: what InMemoryReader::Query yields)
                   if r.alignment.position.position < v.end + 5 and
                   r.alignment.position.position + len(r.aligned_sequence) > v.start - 5]
    window = ref_seq[v.start - hw:v.start + hw + 1]
    want = O.build_pileup(pic, cand, window, overlapping, v.start - hw, list(v.alternate_bases), pileup_height=height)
    assert np.array_equal(img, want), v.start
    if O.reference_available():
      with O.reference_backend():
        assert np.array_equal(img, O.build_pileup(pic, cand, window, overlapping, v.start - hw, list(v.alternate_bases),
                                                  pileup_height=height)), v.start
    meth += int((img[5:, :, len(channels) - 1] > 0).sum())
  assert meth > 200
