"""Native BAM -> packed read table (dv_bam_read_region) vs the Python
restatement of nucleus' SamReader semantics, on a synthetic BGZF BAM written here (every
flag combination the read requirements look at, every CIGAR op, HP tags of all integer
types behind Z / B / f tags, records straddling BGZF blocks), and on the reference's own
test BAMs when they are present in the container."""
import os
import struct
import zlib

import numpy as np
import pytest

from deepvariant_amd import _lib, genomics_io, packing

NT16 = '=ACMGRSVTWYHKDBN'
BAM_OPS = 'MIDNSHP=X'


def _bgzf(data: bytes, block: int) -> bytes:
  out = b''
  for i in list(range(0, len(data), block)) + [None]:
    chunk = b'' if i is None else data[i:i + block]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = c.compress(chunk) + c.flush()
    bsize = 12 + 6 + len(payload) + 8
    out += (b'\x1f\x8b\x08\x04' + b'\0' * 4 + b'\0\xff' + struct.pack('<H', 6) +
            b'BC' + struct.pack('<HH', 2, bsize - 1) + payload +
            struct.pack('<II', zlib.crc32(chunk), len(chunk)))
  return out


def _record(rng, ref_id, pos, name, flag, mapq, cigar, next_ref, tlen, aux=b''):
  l_seq = sum(n for n, op in cigar if op in 'MIS=X')
  seq = rng.integers(1, 16, size=l_seq)
  packed = bytearray((l_seq + 1) // 2)
  for i, v in enumerate(seq):
    packed[i >> 1] |= int(v) << (4 if i % 2 == 0 else 0)
  qual = bytes(rng.integers(0, 60, size=l_seq).astype(np.uint8))
  cig = b''.join(struct.pack('<I', (n << 4) | BAM_OPS.index(op)) for n, op in cigar)
  body = (struct.pack('<iiBBHHHiiii', ref_id, pos, len(name) + 1, mapq, 0, len(cigar), flag,
                      l_seq, next_ref, 0, tlen) + name.encode() + b'\0' + cig + bytes(packed) +
          qual + aux)
  return struct.pack('<i', len(body)) + body


def _write_bam(path, rng, n=400):
  refs = [('chrA', 100000), ('chrB', 50000)]
  hdr = b'BAM\x01' + struct.pack('<i', 0) + struct.pack('<i', len(refs))
  for name, ln in refs:
    hdr += struct.pack('<i', len(name) + 1) + name.encode() + b'\0' + struct.pack('<i', ln)
  recs = []
  flags = [0x0, 0x10, 0x1 | 0x2 | 0x40, 0x1 | 0x2 | 0x80 | 0x10, 0x1 | 0x40, 0x1 | 0x80 | 0x8,
           0x400, 0x200, 0x100, 0x800, 0x4, 0x1 | 0x2 | 0x40 | 0x400]
  hp_tags = [b'', b'HPc\x02', b'HPC\x01', b'HPs\x02\x00', b'HPS\x01\x00', b'HPi\x02\0\0\0',
             b'XAZabc\0HPC\x02', b'XBBs\x02\0\0\0\x01\0\x02\0XFf\0\0\x80?HPc\x01', b'HPZ1\0']
  pos = 100
  for i in range(n):
    flag = flags[int(rng.integers(0, len(flags)))]
    ref_id = -1 if flag & 0x4 else int(rng.integers(0, 2))
    n_ops = int(rng.integers(1, 6))
    cigar = []
    for k in range(n_ops):
      op = 'MMM=XIDNSHP'[int(rng.integers(0, 11))] if 0 < k < n_ops - 1 else 'MM=XSH'[int(rng.integers(0, 6))]
      cigar.append((int(rng.integers(1, 50)), op))
    if not any(op in 'M=X' for _, op in cigar):
      cigar.append((20, 'M'))
    next_ref = [ref_id, ref_id, 1 - ref_id if ref_id >= 0 else -1, -1][int(rng.integers(0, 4))]
    pos += int(rng.integers(0, 120))
    name = 'frag%03d' % int(rng.integers(0, n // 2))
    recs.append(_record(rng, ref_id, pos % 40000, name, flag, int(rng.integers(0, 61)), cigar,
                        next_ref, int(rng.integers(-900, 900)),
                        hp_tags[int(rng.integers(0, len(hp_tags)))]))
  with open(path, 'wb') as f:
    f.write(_bgzf(hdr + b''.join(recs), block=3000))   # records straddle blocks


def _python_table(path, contig, start, end, **req):
  _, reads = genomics_io.read_bam(path, contig, start, end)
  reads = [r for r in reads if genomics_io.read_satisfies_requirements(r, **req)]
  return packing.ReadTable.from_reads(reads)


def _assert_same(nat, py):
  assert nat.n_reads == py.n_reads
  for f in ('read_pos', 'read_seq_off', 'read_cigar_off', 'read_mapq', 'read_flags',
            'read_frag_len', 'read_hp', 'read_name_rank', 'bases', 'quals', 'cigar', 'read_end'):
    np.testing.assert_array_equal(np.asarray(getattr(nat, f)), np.asarray(getattr(py, f)), err_msg=f)
  assert nat.keys == py.keys


@pytest.mark.parametrize('req', [
    dict(), dict(min_mapping_quality=20), dict(keep_duplicates=True, keep_supplementary=True),
    dict(keep_secondary=True, keep_failed_qc=True, keep_improperly_placed=True)])
def test_native_equals_python_reader_on_synthetic_bam(tmp_path, req):
  path = str(tmp_path / 'synthetic.bam')
  _write_bam(path, np.random.default_rng(11))
  for contig, start, end in (('chrA', 0, 1 << 40), ('chrB', 5000, 20000), ('chrA', 39990, 39995)):
    nat = packing.ReadTable.from_bam(path, contig, start, end, n_threads=3, **req)
    _assert_same(nat, _python_table(path, contig, start, end, **req))
  assert packing.ReadTable.from_bam(path, 'chrA', 0, 1 << 40, **req).n_reads > 20


def test_errors_are_reported(tmp_path):
  bad = tmp_path / 'not_a.bam'
  bad.write_bytes(b'hello world, definitely not BGZF')
  with pytest.raises(_lib.DvError, match='BGZF'):
    packing.ReadTable.from_bam(str(bad))
  with pytest.raises(_lib.DvError, match='cannot open'):
    packing.ReadTable.from_bam(str(tmp_path / 'missing.bam'))
  path = str(tmp_path / 's.bam')
  _write_bam(path, np.random.default_rng(1), n=20)
  with pytest.raises(_lib.DvError, match='contig'):
    packing.ReadTable.from_bam(path, 'chrZ')


REF_INPUT = '/root/reference/deepvariant/testdata/input'


@pytest.mark.skipif(not os.path.isdir(REF_INPUT), reason='reference testdata not in this container')
@pytest.mark.parametrize('bam,contig,start,end,mapq', [
    ('NA12878_S1.chr20.10_10p1mb.bam', 'chr20', 9_999_000, 10_012_000, 5),
    ('test_pacbio.chr20_100kbp_at_9mb.bam', 'chr20', 8_900_000, 9_200_000, 1)])
def test_native_equals_python_reader_on_reference_bams(bam, contig, start, end, mapq):
  path = os.path.join(REF_INPUT, bam)
  nat = packing.ReadTable.from_bam(path, contig, start, end, min_mapping_quality=mapq)
  _assert_same(nat, _python_table(path, contig, start, end, min_mapping_quality=mapq))
  assert nat.n_reads in (6014, 281)


@pytest.mark.skipif(not os.path.isdir(REF_INPUT), reason='reference testdata not in this container')
@pytest.mark.parametrize('bam,regions,mapq', [
    ('NA12878_S1.chr20.10_10p1mb.bam',
     [(9_999_000, 10_012_000), (10_050_000, 10_051_000), (0, 1 << 40), (10_099_900, 10_200_000),
      (5, 10)], 5),
    ('test_pacbio.chr20_100kbp_at_9mb.bam',
     [(8_900_000, 9_200_000), (9_050_000, 9_050_100), (9_099_000, 9_099_500)], 1)])
def test_bai_indexed_read_equals_full_scan(bam, regions, mapq, monkeypatch):
  """With <bam>.bai next to the file only the BGZF members the index points at are read
  (reg2bins + linear index, SAMv1 5.2); the result must be the full scan's, read for read."""
  path = os.path.join(REF_INPUT, bam)
  assert os.path.exists(path + '.bai')
  for start, end in regions:
    monkeypatch.delenv('DV_BAM_NO_INDEX', raising=False)
    indexed = packing.ReadTable.from_bam(path, 'chr20', start, end, min_mapping_quality=mapq)
    monkeypatch.setenv('DV_BAM_NO_INDEX', '1')
    full = packing.ReadTable.from_bam(path, 'chr20', start, end, min_mapping_quality=mapq)
    _assert_same(indexed, full)


def _one_record_bam(path, body):
  hdr = (b'BAM\x01' + struct.pack('<i', 0) + struct.pack('<i', 1) + struct.pack('<i', 5) +
         b'chrA\0' + struct.pack('<i', 100000))
  with open(path, 'wb') as f:
    f.write(_bgzf(hdr + struct.pack('<i', len(body)) + body, block=60000))


def test_untrusted_record_sizes_are_rejected_not_followed(tmp_path):
  """A BAM is untrusted input: a 32-bit wrap in the record-size sum (l_seq = 0xAAAAAAAB makes
  32 + l_read_name + 4 n_cigar + (l_seq+1)/2 + l_seq come out as ~33) must be an error, not a
  multi-gigabyte read; so must a record whose SEQ is '*' (l_seq = 0) or shorter than what its
  CIGAR consumes -- the encoder indexes bases by CIGAR offsets without bounds checks."""
  cig = struct.pack('<I', (50 << 4) | 0)   # 50M
  def body(l_seq, seq_bytes):
    return (struct.pack('<iiBBHHHiiii', 0, 100, 2, 60, 0, 1, 0, l_seq, -1, 0, 0) + b'r\0' + cig +
            seq_bytes)
  cases = {
      'wrap': (body(0xAAAAAAAB - (1 << 32), b'\x11' * 40), 'corrupt BAM record'),
      'star': (body(0, b''), 'different length'),
      'short': (body(10, b'\x11' * 5 + b'\x20' * 10), 'different length'),
  }
  for name, (rec, msg) in cases.items():
    path = str(tmp_path / (name + '.bam'))
    _one_record_bam(path, rec)
    with pytest.raises(_lib.DvError, match=msg):
      packing.ReadTable.from_bam(path, 'chrA', 0, 1000)
  # the well-formed twin of 'short' is accepted
  ok = str(tmp_path / 'ok.bam')
  _one_record_bam(ok, body(50, b'\x11' * 25 + b'\x20' * 50))
  assert packing.ReadTable.from_bam(ok, 'chrA', 0, 1000).n_reads == 1


def test_bgzf_isize_is_capped(tmp_path):
  """BGZF blocks inflate to at most 64 KiB; a larger ISIZE is a corrupt file, not a 4 GB resize."""
  chunk = b'BAM\x01' + b'\0' * 8
  c = zlib.compressobj(6, zlib.DEFLATED, -15)
  payload = c.compress(chunk) + c.flush()
  bsize = 12 + 6 + len(payload) + 8
  blk = (b'\x1f\x8b\x08\x04' + b'\0' * 4 + b'\0\xff' + struct.pack('<H', 6) + b'BC' +
         struct.pack('<HH', 2, bsize - 1) + payload + struct.pack('<II', zlib.crc32(chunk), 0xF0000000))
  path = str(tmp_path / 'big.bam')
  with open(path, 'wb') as f:
    f.write(blk)
  with pytest.raises(_lib.DvError, match='64 KiB'):
    packing.ReadTable.from_bam(path, 'chrA', 0, 1000)


NUCLEUS_TESTDATA = '/root/reference/third_party/nucleus/testdata'


@pytest.mark.skipif(not os.path.isdir(NUCLEUS_TESTDATA), reason='reference testdata not in this container')
def test_known_answers_of_the_reference_sam_reader_tests():
  """Counts and coordinates the REFERENCE's own tests hold for its test.bam
  (third_party/nucleus/io/sam_reader_test.cc:311-418: SimpleQueriesWork,
  ThatRangeIsExactlyCorrect, QueriedRespectsReadRequirements) -- an anchor for the native
  reader that does not go through this repo's Python reader."""
  path = os.path.join(NUCLEUS_TESTDATA, 'test.bam')
  everything = dict(keep_duplicates=True, keep_supplementary=True, keep_secondary=True, keep_failed_qc=True,
                    keep_improperly_placed=True)
  # default ReadRequirements: 105 of the 106 records (the unmapped one goes); min_mapping_quality 38: 104
  assert packing.ReadTable.from_bam(path, 'chr20', 9999999, 10000100).n_reads == 105
  assert packing.ReadTable.from_bam(path, 'chr20', 9999999, 10000100, min_mapping_quality=38).n_reads == 104
  t = packing.ReadTable.from_bam(path, 'chr20', 999999, 100000000, **everything)
  assert t.n_reads == 105                                         # aligned reads only, whatever the flags
  hist = {int(q): int((t.read_mapq == q).sum()) for q in np.unique(t.read_mapq)}
  assert hist == {37: 1, 60: 104}                                 # "samtools view | cut -f 5 | sort | uniq -c"
  assert packing.ReadTable.from_bam(path, 'chr20', 999999, 2000000).n_reads == 0
  assert packing.ReadTable.from_bam(path, 'chr10', 9999999, 10000000).n_reads == 0
  # ThatRangeIsExactlyCorrect: this read spans [9999911, 10000010)
  name = 'HSQ1004:134:C0D8DACXX:4:1304:21341:94622'
  i = [k.rsplit('/', 1)[0] for k in t.keys].index(name)
  assert (int(t.read_pos[i]), int(t.read_end[i])) == (9999911, 10000010)
  present = lambda a, b: name in [k.rsplit('/', 1)[0] for k in
                                  packing.ReadTable.from_bam(path, 'chr20', a, b, **everything).keys]
  assert present(9999911, 10000010) and present(9999912, 10000009)
  assert present(9999901, 9999912) and not present(9999901, 9999911)
  assert present(10000009, 10000020) and not present(10000010, 10000020)


def test_query_reads_matches_the_table_query():
  """dv_query_reads (InMemoryReader::Query + ReadOverlapsRegion, make_examples_native.cc:802-810,
  nucleus/util/utils.cc:172-188) against ReadTable.query and a direct restatement, on reads
  with every reference-consuming / non-consuming CIGAR op."""
  import ctypes as C
  from deepvariant_amd import _lib
  rng = np.random.default_rng(4)
  n = 500
  pos = rng.integers(0, 3000, size=n).astype(np.int32)
  cig, off = [], [0]
  for _ in range(n):
    ops = [(int(rng.integers(1, 60)) << 4) | int(op) for op in rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9],
                                                                         size=int(rng.integers(1, 6)))]
    cig += ops
    off.append(len(cig))
  cig, off = np.array(cig, np.uint32), np.array(off, np.uint32)
  ref_len = np.array([sum(int(w) >> 4 for w in cig[off[i]:off[i + 1]] if (int(w) & 15) in (1, 3, 4, 8, 9))
                      for i in range(n)], np.int64)
  q0 = rng.integers(-50, 3100, size=64).astype(np.int64)
  q1 = q0 + rng.integers(0, 400, size=64)
  lib = _lib.lib()
  lib.dv_query_reads.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
  list_off = np.zeros(65, np.uint32)
  _lib.check(lib.dv_query_reads(n, pos.ctypes.data, off.ctypes.data, cig.ctypes.data, 64, q0.ctypes.data,
                                q1.ctypes.data, list_off.ctypes.data, None))
  reads = np.zeros(max(int(list_off[-1]), 1), np.uint32)
  _lib.check(lib.dv_query_reads(n, pos.ctypes.data, off.ctypes.data, cig.ctypes.data, 64, q0.ctypes.data,
                                q1.ctypes.data, list_off.ctypes.data, reads.ctypes.data))
  assert list_off[-1] > 500
  for k in range(64):
    want = [r for r in range(n) if q1[k] > pos[r] and q0[k] < pos[r] + ref_len[r]]
    assert reads[list_off[k]:list_off[k + 1]].tolist() == want, k


def test_long_cigar_in_the_cg_tag(tmp_path):
  """A record with more than 65535 CIGAR operations stores <l_seq>S<span>N and keeps the real
  operations in CG:B:I (SAMv1 4.2.2); htslib -- hence nucleus' SamReader -- hands out the real
  CIGAR (sam.c bam_tag2cigar).  Native reader == Python restatement == the operations written;
  a CG tag on a record whose CIGAR is NOT the placeholder is ignored, as htslib does."""
  rng = np.random.default_rng(3)
  n_ops = 70001
  ops = []
  for k in range(n_ops):
    ops.append((1 + int(rng.integers(0, 3)), 'MID'[k % 3] if 0 < k < n_ops - 1 else 'M'))
  l_seq = sum(n for n, op in ops if op in 'MI')
  span = sum(n for n, op in ops if op in 'MD')
  cg = b'CGBI' + struct.pack('<i', n_ops) + b''.join(
      struct.pack('<I', (n << 4) | BAM_OPS.index(op)) for n, op in ops)
  refs = [('chrA', 400000)]
  hdr = b'BAM\x01' + struct.pack('<i', 0) + struct.pack('<i', 1)
  hdr += struct.pack('<i', 5) + b'chrA\0' + struct.pack('<i', 400000)
  long_rec = _record(rng, 0, 1000, 'ont_read', 0, 60, [(l_seq, 'S'), (span, 'N')], -1, 0,
                     aux=b'HPC\x02' + cg + b'XAZtail\0')
  # same tag on an ordinary record: not a placeholder, the tag is just an aux field
  plain = _record(rng, 0, 2000, 'plain', 0, 60, [(30, 'M'), (2, 'D'), (20, 'M')], -1, 0,
                  aux=b'CGBI' + struct.pack('<iI', 1, (50 << 4) | 0))
  path = str(tmp_path / 'long.bam')
  with open(path, 'wb') as f:
    f.write(_bgzf(hdr + long_rec + plain, block=30000))
  nat = packing.ReadTable.from_bam(path, 'chrA', 0, 1 << 40)
  py = _python_table(path, 'chrA', 0, 1 << 40)
  _assert_same(nat, py)
  assert nat.n_reads == 2
  c0, c1 = int(nat.read_cigar_off[0]), int(nat.read_cigar_off[1])
  assert c1 - c0 == n_ops
  want = np.array([(n << 4) | (BAM_OPS.index(op) + 1) for n, op in ops], np.uint32)
  np.testing.assert_array_equal(np.asarray(nat.cigar[c0:c1]), want)
  assert int(nat.read_end[0]) == 1000 + span and int(nat.read_hp[0]) == 2
  assert int(nat.read_cigar_off[2]) - c1 == 3
  # the region test uses the real span: a window past the placeholder-free end finds nothing
  assert packing.ReadTable.from_bam(path, 'chrA', 1000 + span + 5, 1000 + span + 50).n_reads == 0


def _bai_to_csi(bai: bytes, depth: int) -> bytes:
  """The same index as a CSIv1 file (hts-specs CSIv1.tex): min_shift 14, `depth` levels (5 = the
  .bai's own bins; 6 = one more level on top, so every bin moves one level down with the same
  offset inside its level), loffset of a bin = the linear-index entry of its first 16 kb window."""
  assert bai[:4] == b'BAI\x01' and depth in (5, 6)
  n_ref = struct.unpack_from('<i', bai, 4)[0]
  p = 8
  out = bytearray(b'CSI\x01' + struct.pack('<iii', 14, depth, 0) + struct.pack('<i', n_ref))
  level_first = [((1 << (3 * l)) - 1) // 7 for l in range(8)]
  for _ in range(n_ref):
    n_bin = struct.unpack_from('<i', bai, p)[0]
    p += 4
    bins = []
    for _b in range(n_bin):
      b, n_chunk = struct.unpack_from('<Ii', bai, p)
      p += 8
      chunks = bai[p:p + 16 * n_chunk]
      p += 16 * n_chunk
      bins.append((b, n_chunk, chunks))
    n_intv = struct.unpack_from('<i', bai, p)[0]
    p += 4
    linear = struct.unpack_from('<%dQ' % n_intv, bai, p)
    p += 8 * n_intv
    out += struct.pack('<i', n_bin)
    for b, n_chunk, chunks in bins:
      if b == 37450:                       # the metadata pseudo-bin: its number depends on the depth
        new_bin, loffset = ((1 << (3 * (depth + 1))) - 1) // 7 + 1, 0
      else:
        level = max(l for l in range(6) if level_first[l] <= b)
        k = b - level_first[level]
        new_bin = level_first[level + depth - 5] + k
        window = (k << (14 + 3 * (5 - level))) >> 14
        loffset = linear[window] if window < n_intv else 0
      out += struct.pack('<IQi', new_bin, loffset, n_chunk) + chunks
  return _bgzf(bytes(out), block=60000)


@pytest.mark.parametrize('depth', [5, 6])
def test_csi_indexed_read_equals_bai_and_full_scan(tmp_path, depth, monkeypatch):
  """A BAM indexed with .csi (what `samtools index -c` writes, needed for contigs beyond 2^29 bases;
  htslib reads it wherever it reads a .bai) is queried through the same chunk logic: bins of every
  level + the deepest existing bin's loffset.  Same reads as with the .bai and as the full scan."""
  monkeypatch.delenv('DV_BAM_NO_INDEX', raising=False)
  with np.load(os.path.join(os.path.dirname(__file__), 'golden', 'na12878_100kb.npz')) as z:
    bam, bai = z['bam'].tobytes(), z['bai'].tobytes()
  with_bai = str(tmp_path / 'a.bam')
  with_csi = str(tmp_path / 'c.bam')
  for path in (with_bai, with_csi):
    with open(path, 'wb') as f:
      f.write(bam)
  with open(with_bai + '.bai', 'wb') as f:
    f.write(bai)
  with open(with_csi + '.csi', 'wb') as f:
    f.write(_bai_to_csi(bai, depth))
  for start, end in ((10_000_000, 10_001_000), (10_049_000, 10_066_500), (9_990_000, 10_000_050), (10_099_000, 10_200_000)):
    a = packing.ReadTable.from_bam(with_bai, 'chr20', start, end, min_mapping_quality=5)
    c = packing.ReadTable.from_bam(with_csi, 'chr20', start, end, min_mapping_quality=5)
    _assert_same(a, c)
    assert a.n_reads > 0
  monkeypatch.setenv('DV_BAM_NO_INDEX', '1')
  full = packing.ReadTable.from_bam(with_csi, 'chr20', 10_049_000, 10_066_500, min_mapping_quality=5)
  monkeypatch.delenv('DV_BAM_NO_INDEX')
  _assert_same(full, packing.ReadTable.from_bam(with_csi, 'chr20', 10_049_000, 10_066_500, min_mapping_quality=5))
  # the index was really used: a .csi that points nowhere yields nothing
  with open(with_csi + '.csi', 'wb') as f:
    f.write(_bgzf(b"CSI\x01" + struct.pack("<iii", 14, depth, 0) + struct.pack("<i", 25) + struct.pack("<i", 0) * 25, block=60000))
  assert packing.ReadTable.from_bam(with_csi, 'chr20', 10_049_000, 10_066_500).n_reads == 0
