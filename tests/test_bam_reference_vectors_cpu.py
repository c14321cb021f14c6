"""The native BAM reader (dv_bam_read_region) against vectors the REFERENCE holds for its own
SamReader (third_party/nucleus/io/sam_reader_test.cc) -- not against this repo's Python
restatement:

* `SamReaderQueryTest` (:311-420) on nucleus' `test.bam` (+ .bai): the number of reads every
  query range returns, the exact-boundary behaviour of one named read, the effect of the read
  requirements.  nucleus with NO requirements also returns the one read flagged unmapped
  (it sits at its mate's position); DeepVariant always sets requirements and never
  `keep_unaligned`, and this reader drops such reads unconditionally -- so the "no
  requirements" counts are the reference's minus that read, the others are equal.
* `ReadBamFile.MatchesGolden` (:78-88): nucleus' conversion of the six records of `test.sam`,
  kept by the reference as `test.sam.golden.tfrecord` (Read protos).  The records are turned
  into BAM by the writer of tests/test_bam_native_cpu.py (plain SAMv1 field packing) and read
  back natively: name, read number, position, strand, mapping quality, CIGAR, bases,
  qualities, fragment length and end of the five mapped reads equal the golden protos.

Fixtures: tests/golden/nucleus_sam.npz = the bytes of the five files of
third_party/nucleus/testdata/ that sam_reader_test.cc runs on (reference test DATA, 18 KB
compressed; bundled by tests/golden/make_golden.py nucleus_sam), written back to a temporary
directory for the test session."""
import os
import struct

import numpy as np
import pytest

from deepvariant_amd import _lib, packing, protowire, tfrecord
from tests.test_bam_native_cpu import BAM_OPS, _bgzf

import tempfile

_BUNDLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'nucleus_sam.npz')
HERE = tempfile.mkdtemp(prefix='nucleus_sam_')
with np.load(_BUNDLE) as _z:
  for _name in ('test.bam', 'test.bam.bai', 'test.sam', 'test.sam.golden.tfrecord', 'test_oq.sam'):
    with open(os.path.join(HERE, _name), 'wb') as _f:
      _f.write(_z[_name.replace('.', '_')].tobytes())
BAM = os.path.join(HERE, 'test.bam')
EVERYTHING = dict(keep_duplicates=True, keep_supplementary=True, keep_secondary=True,
                  keep_failed_qc=True, keep_improperly_placed=True)


def _n(contig, start, end, **req):
  return packing.ReadTable.from_bam(BAM, contig, start, end, **req).n_reads


def test_simple_queries():
  # sam_reader_test.cc:311-327 (SimpleQueriesWork); -1 = the read flagged unmapped, see above
  assert _n('chr20', 9999999, 10000000, **EVERYTHING) == 45
  assert _n('chr20', 9999999, 10000100, **EVERYTHING) == 106 - 1
  assert _n('chr20', 999999, 10000000, **EVERYTHING) == 45
  assert _n('chr20', 999999, 100000000, **EVERYTHING) == 106 - 1
  assert _n('chr20', 999999, 2000000, **EVERYTHING) == 0
  assert _n('chr10', 9999999, 10000000, **EVERYTHING) == 0
  assert _n('chr1', 0, 100000000, **EVERYTHING) == 0


def test_read_requirements():
  # :386-412 (QueriedRespectsReadRequirements): 106 without requirements, 105 with the default
  # ones, 104 with min_mapping_quality = 38 (mapq column of the range: 1 x 0, 1 x 37, 104 x 60)
  assert _n('chr20', 9999999, 10000100) == 105
  assert _n('chr20', 9999999, 10000100, min_mapping_quality=38) == 104
  t = packing.ReadTable.from_bam(BAM, 'chr20', 9999999, 10000100)
  assert sorted(set(np.asarray(t.read_mapq).tolist())) == [37, 60]


def test_range_is_exactly_correct():
  # :329-383 (ThatRangeIsExactlyCorrect): the read spans [9999911, 10000010)
  name = 'HSQ1004:134:C0D8DACXX:4:1304:21341:94622'
  s, e = 9999911, 10000010

  def has(lo, hi):
    t = packing.ReadTable.from_bam(BAM, 'chr20', lo, hi, **EVERYTHING)
    return any(k.rsplit('/', 1)[0] == name for k in t.keys)

  assert has(s, e) and has(s + 1, e - 1) and has(s + 5, e + 5) and has(s - 5, e - 5)
  assert has(s - 10, s + 1) and not has(s - 10, s)
  assert has(e - 1, e + 10) and not has(e, e + 10)
  # ... and the indexed path (test.bam.bai is next to the file) agrees with the full scan
  os.environ['DV_BAM_NO_INDEX'] = '1'
  try:
    assert has(s - 10, s + 1) and not has(s - 10, s) and has(e - 1, e + 10) and not has(e, e + 10)
    assert _n('chr20', 9999999, 10000100) == 105
  finally:
    del os.environ['DV_BAM_NO_INDEX']


def _decode_read(buf):
  """nucleus.genomics.v1.Read (reads.proto:140-237), the fields the packed table carries."""
  r = dict(name='', number=0, n_reads=0, frag_len=0, seq='', qual=b'', aligned=False,
           contig='', pos=0, reverse=False, mapq=0, cigar=[], proper=False, supp=False)
  for f, wt, v in protowire.iter_fields(buf):
    if f == 4:
      r['name'] = bytes(v).decode()
    elif f == 5:
      r['proper'] = bool(v)
    elif f == 7:
      r['frag_len'] = protowire.to_signed64(v)
      if r['frag_len'] >= 1 << 31:
        r['frag_len'] -= 1 << 32
    elif f == 8:
      r['number'] = v
    elif f == 9:
      r['n_reads'] = v
    elif f == 13:
      r['supp'] = bool(v)
    elif f == 14:
      r['seq'] = bytes(v).decode()
    elif f == 15:
      r['qual'] = bytes(v)
    elif f == 11:
      r['aligned'] = True
      for f2, _, v2 in protowire.iter_fields(v):
        if f2 == 1:
          for f3, _, v3 in protowire.iter_fields(v2):
            if f3 == 1:
              r['contig'] = bytes(v3).decode()
            elif f3 == 2:
              r['pos'] = protowire.to_signed64(v3)
            elif f3 == 3:
              r['reverse'] = bool(v3)
        elif f2 == 2:
          r['mapq'] = v2
        elif f2 == 3:
          op = ln = 0
          for f3, _, v3 in protowire.iter_fields(v2):
            if f3 == 1:
              op = v3
            elif f3 == 2:
              ln = v3
          r['cigar'].append((ln << 4) | op)
  return r


def _sam_to_bam(sam_path, bam_path, keep_string_tags=()):
  """SAMv1 text -> BAM records (SAMv1 4.2); of the aux fields only the `Z` tags named in
  `keep_string_tags` are kept (the golden comparison ignores `info`, as the reference's does)."""
  contigs, recs = [], []
  for line in open(sam_path):
    line = line.rstrip('\n')
    if line.startswith('@SQ'):
      f = dict(x.split(':', 1) for x in line.split('\t')[1:])
      contigs.append((f['SN'], int(f['LN'])))
    if line.startswith('@') or not line:
      continue
    q, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = line.split('\t')[:11]
    aux = b''
    for field in line.split('\t')[11:]:
      tag, ty, value = field.split(':', 2)
      if ty == 'Z' and tag in keep_string_tags:
        aux += tag.encode() + b'Z' + value.encode() + b'\0'
      elif ty == 'i' and tag == 'NM':            # something else in front of / behind the tag
        aux += b'NMi' + struct.pack('<i', int(value))
    names = [c[0] for c in contigs]
    ref_id = names.index(rname) if rname != '*' else -1
    next_ref = ref_id if rnext == '=' else (names.index(rnext) if rnext != '*' else -1)
    ops, num = [], ''
    for ch in cigar if cigar != '*' else '':
      if ch.isdigit():
        num += ch
      else:
        ops.append((int(num), ch))
        num = ''
    l_seq = len(seq) if seq != '*' else 0
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq if seq != '*' else ''):
      packed[i >> 1] |= '=ACMGRSVTWYHKDBN'.index(ch) << (4 if i % 2 == 0 else 0)
    qb = bytes(ord(c) - 33 for c in qual) if qual != '*' else b'\xff' * l_seq
    cig = b''.join(struct.pack('<I', (n << 4) | BAM_OPS.index(op)) for n, op in ops)
    body = (struct.pack('<iiBBHHHiiii', ref_id, int(pos) - 1, len(q) + 1, int(mapq), 0, len(ops),
                        int(flag), l_seq, next_ref, int(pnext) - 1, int(tlen)) +
            q.encode() + b'\0' + cig + bytes(packed) + qb + aux)
    recs.append(struct.pack('<i', len(body)) + body)
  hdr = b'BAM\x01' + struct.pack('<i', 0) + struct.pack('<i', len(contigs))
  for name, ln in contigs:
    hdr += struct.pack('<i', len(name) + 1) + name.encode() + b'\0' + struct.pack('<i', ln)
  with open(bam_path, 'wb') as f:
    f.write(_bgzf(hdr + b''.join(recs), block=20000))


def test_conversion_matches_the_reference_golden_protos(tmp_path):
  golden = [_decode_read(rec) for rec in tfrecord.read_tfrecords(
      os.path.join(HERE, 'test.sam.golden.tfrecord'))]
  assert len(golden) == 6                               # TestIteration: SizeIs(6)
  mapped = [g for g in golden if g['aligned']]
  assert len(mapped) == 5
  bam = str(tmp_path / 'from_sam.bam')
  _sam_to_bam(os.path.join(HERE, 'test.sam'), bam)
  t = packing.ReadTable.from_bam(bam, None, 0, 1 << 40, **EVERYTHING)
  assert t.n_reads == len(mapped)
  for i, g in enumerate(mapped):
    assert t.keys[i] == '%s/%d' % (g['name'], g['number'])
    assert int(t.read_pos[i]) == g['pos']
    assert int(t.read_mapq[i]) == g['mapq']
    assert bool(t.read_flags[i] & packing.DV_READ_REVERSE) == g['reverse']
    assert bool(t.read_flags[i] & packing.DV_READ_SUPPLEMENTARY) == g['supp']
    assert int(t.read_frag_len[i]) == g['frag_len']
    c0, c1 = int(t.read_cigar_off[i]), int(t.read_cigar_off[i + 1])
    assert np.asarray(t.cigar[c0:c1]).tolist() == g['cigar']
    s0, s1 = int(t.read_seq_off[i]), int(t.read_seq_off[i + 1])
    assert bytes(np.asarray(t.bases[s0:s1])).decode() == g['seq']
    assert bytes(np.asarray(t.quals[s0:s1])) == g['qual']
    span = sum(w >> 4 for w in g['cigar'] if (w & 0xF) in (1, 3, 4, 8, 9))
    assert int(t.read_end[i]) == g['pos'] + span


def test_original_quality_scores_from_the_oq_tag(tmp_path):
  """SamReaderTest.TestAlignedQualityOQ / ...WhenTagIsNotPresent (sam_reader_test.cc:96-141):
  with use_original_base_quality_scores every quality of `test_oq.sam` is 'C' - 33 (its OQ tags
  are all 'C'), QUAL is ignored; `test.sam` has no OQ tags -- nucleus leaves aligned_quality
  EMPTY there, which the encoder cannot draw, so this reader reports it.  Native reader and
  Python restatement agree."""
  from deepvariant_amd import genomics_io
  bam = str(tmp_path / 'oq.bam')
  _sam_to_bam(os.path.join(HERE, 'test_oq.sam'), bam, keep_string_tags=('OQ', 'MD'))
  plain = packing.ReadTable.from_bam(bam, None, 0, 1 << 40, **EVERYTHING)
  oq = packing.ReadTable.from_bam(bam, None, 0, 1 << 40, use_original_quality_scores=True, **EVERYTHING)
  assert oq.n_reads == plain.n_reads == 5
  assert len(oq.quals) == len(plain.quals) and set(np.asarray(oq.quals).tolist()) == {ord('C') - 33}
  assert set(np.asarray(plain.quals).tolist()) != {ord('C') - 33}
  np.testing.assert_array_equal(np.asarray(oq.bases), np.asarray(plain.bases))
  _, py = genomics_io.read_bam(bam, use_original_quality_scores=True)
  assert [bytes(r.aligned_quality) for r in py] == [bytes([ord('C') - 33]) * len(r.aligned_sequence) for r in py]
  no_tags = str(tmp_path / 'no_oq.bam')
  _sam_to_bam(os.path.join(HERE, 'test.sam'), no_tags)
  with pytest.raises(_lib.DvError, match='OQ'):
    packing.ReadTable.from_bam(no_tags, None, 0, 1 << 40, use_original_quality_scores=True, **EVERYTHING)
  with pytest.raises(ValueError, match='OQ'):
    genomics_io.read_bam(no_tags, use_original_quality_scores=True)
