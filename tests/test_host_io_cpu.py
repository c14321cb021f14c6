"""Host-side formats and drivers that need no GPU."""
import os

import numpy as np
import pytest

from deepvariant_amd import call_variants as cv
from deepvariant_amd import dv_types as T
from deepvariant_amd import make_examples_native as men
from deepvariant_amd import protowire as pw
from deepvariant_amd import tfrecord


@pytest.mark.parametrize('precision,expected', [
    (None, [0.2102311329, 0.099768768, 0.6899999991]),
    (1, [0.2, 0.1, 0.7]),
    (2, [0.21, 0.10, 0.69]),
])
def test_round_gls_known_answers(precision, expected):
  # deepvariant/call_variants_test.py:335-357
  assert cv.round_gls([0.2102311329, 0.099768768, 0.6899999991],
                      precision) == expected


def test_round_gls_rejects_unnormalised():
  with pytest.raises(ValueError, match='do not sum to one'):
    cv.round_gls([0.5, 0.5, 0.5], 2)


def test_tfrecord_roundtrip_with_crc(tmp_path):
  path = str(tmp_path / 'x.tfrecord.gz')
  recs = [b'', b'abc', os.urandom(100000)]
  with tfrecord.Writer(path) as w:
    for r in recs:
      w.write(r)
  assert list(tfrecord.read_tfrecords(path, verify_crc=True)) == recs
  # masked crc of the TFRecord spec: crc32c('123456789') = 0xe3069283
  assert tfrecord.crc32c(b'123456789') == 0xE3069283


def test_example_and_cvo_wire_roundtrip():
  v = T.Variant('chr20', 99, 100, 'A', ['C', 'G'],
                [T.VariantCall('s', [-1, -1])])
  vb = pw.encode_variant(v)
  assert pw.decode_variant(vb).alternate_bases == ['C', 'G']
  ex = pw.encode_example({
      'locus': [b'chr20:100-100'], 'variant/encoded': [vb],
      'variant_type': [1], 'alt_allele_indices/encoded': [b'\n\x01\x00'],
      'image/encoded': [bytes(range(256))], 'image/shape': [4, 8, 8],
      'sequencing_type': [0]})
  d = pw.decode_example(ex)
  assert d['image/shape'] == [4, 8, 8] and d['image/encoded'][0] == bytes(range(256))
  cvo = cv.create_cvo(vb, [0.1, 0.2, 0.7], b'\n\x01\x00')
  variant, alt, probs = pw.decode_call_variants_output(cvo)
  assert alt == [0] and probs == [0.1, 0.2, 0.7]
  assert variant.reference_bases == 'A' and variant.start == 99
  # MID tag landed in calls[0].info (call_variants.py:397-398)
  assert b'MID' in cvo and b'deepvariant' in cvo


def test_alt_allele_indices_encoding_matches_reference_bytes():
  # make_examples_native.cc:350-374: [0] -> 0a 01 00 ; [0, 1] -> 0a 02 00 01
  assert pw.encode_alt_allele_indices([0]) == b'\n\x01\x00'
  assert pw.encode_alt_allele_indices([0, 1]) == b'\n\x02\x00\x01'


def test_alt_allele_combinations():
  # deepvariant/make_examples_native_test.cc:489-556
  def cand(alts, idx=None):
    c = T.DeepVariantCall(variant=T.Variant('chr1', 5, 6, 'A', alts))
    for i in idx or []:
      c.make_examples_alt_allele_indices.append(T.AltAlleleIndices(i))
    return c
  add, no = T.MultiAllelicMode.ADD_HET_ALT_IMAGES, T.MultiAllelicMode.NO_HET_ALT_IMAGES
  assert men.alt_allele_combinations(cand(['C']), add) == [['C']]
  assert men.alt_allele_combinations(cand(['C', 'G']), add) == [['C'], ['G'], ['C', 'G']]
  assert men.alt_allele_combinations(cand(['C', 'G', 'T']), add) == [
      ['C'], ['G'], ['T'], ['C', 'G'], ['C', 'T'], ['G', 'T']]
  assert men.alt_allele_combinations(cand(['C', 'G']), no) == [['C'], ['G']]
  assert men.alt_allele_combinations(cand(['C', 'G'], [[1], [0, 1]]), add) == [
      ['G'], ['C', 'G']]
  assert men.alt_allele_combinations(cand(['C', 'G'], [[1], [0, 1]]), no) == [['G']]


def test_reference_window_is_n_padded():
  # deepvariant/make_examples_native_test.cc:792-836
  class Ref:
    def n_bases(self, _):
      return 10
    def get_bases(self, _, s, e):
      return 'ACGTACGTAC'[s:e]
  v = T.Variant('chr1', 1, 2, 'C', ['T'])
  assert men.get_reference_bases_for_pileup(Ref(), v, 7) == 'NNACGTA'
  v = T.Variant('chr1', 8, 9, 'A', ['T'])
  assert men.get_reference_bases_for_pileup(Ref(), v, 7) == 'CGTACNN'


def test_encoded_variant_type():
  t = lambda r, a: men.encoded_variant_type(T.Variant('c', 1, 2, r, a))
  assert t('A', ['C']) == men.K_SNP and t('A', ['C', 'G']) == men.K_SNP
  assert t('AT', ['A']) == men.K_INDEL and t('A', ['AT']) == men.K_INDEL
  assert t('A', []) == men.K_UNKNOWN


def test_golden_examples_decode_with_our_reader():
  """The reader consumes the reference's own golden TFRecord bytes -- checked
  offline when the fixture was built; here on a re-encoded sample."""
  from tests import golden_io
  from tests.test_oracle_golden import FIXTURE
  _, examples, _ = golden_io.load(FIXTURE)
  img = examples[0]['image']
  rec = pw.encode_example({'image/encoded': [img.tobytes()],
                           'image/shape': list(img.shape),
                           'variant/encoded': [pw.encode_variant(examples[0]['call'].variant)],
                           'alt_allele_indices/encoded': [b'\n\x01\x00']})
  d = pw.decode_example(rec)
  assert np.array_equal(np.frombuffer(d['image/encoded'][0], np.uint8).reshape(img.shape), img)


def test_round_gls_batch_equals_the_scalar_path_on_float32():
  """The batched rounding used by call_variants == the per-candidate function applied to
  np.float32 scalars (what the reference's loop sees: float32 predictions through Python's
  round), bit for bit, on softmax-like rows incl. ties, tiny minima and exact halves."""
  from deepvariant_amd import call_variants as cv
  rng = np.random.default_rng(7)
  logits = rng.normal(0, 4, size=(5000, 3)).astype(np.float32)
  e = np.exp(logits - logits.max(1, keepdims=True))
  p = (e / e.sum(1, keepdims=True)).astype(np.float32)
  p[:3] = np.array([[0.25, 0.25, 0.5], [1.0, 0.0, 0.0], [0.33333334, 0.33333334, 0.3333333]],
                   np.float32)
  got = cv.round_gls_batch(p, 10)
  assert got.dtype == np.float64
  for row, want_in in zip(got, p):
    want = cv.round_gls(list(want_in), 10)          # np.float32 scalars
    assert [float(v) for v in want] == row.tolist()
  # doubles are float32-granular, like the reference's CVO bytes
  assert (got == got.astype(np.float32).astype(np.float64)).all()
  with pytest.raises(ValueError, match='do not sum to one'):
    cv.round_gls_batch(np.array([[0.5, 0.5, 0.5]], np.float32))


def test_sharded_output_name_detection():
  from deepvariant_amd import call_variants as cv
  assert cv.is_sharded_filename('/x/cvo-00000-of-00001.tfrecord.gz')
  assert cv.is_sharded_filename('cvo-3-of-16')
  assert not cv.is_sharded_filename('/x/cvo.tfrecord.gz')
  assert not cv.is_sharded_filename('/x/cvo-00000-of-00000.tfrecord.gz')


def test_fasta_reader_indexed_and_whole_file_forms_agree(tmp_path):
  """FastaReader on a plain file with a .fai (mapped, read on demand), without one and on gzip
  (parsed once): same contig order, lengths and bases for queries that cross line ends, run past
  the contig end or are empty; lower case is upper-cased, header descriptions and CR LF are ignored."""
  import os
  from deepvariant_amd import genomics_io as G
  rng = np.random.default_rng(1)
  c1 = 'N' * 1000 + ''.join('acgtN'[i] for i in rng.integers(0, 5, size=12345))
  c2 = ''.join('ACGT'[i] for i in rng.integers(0, 4, size=120))
  for suffix in ('', '.gz'):
    for index in (False, True):
      path = os.path.join(str(tmp_path), 'r%d.fa%s' % (index, suffix))
      G.write_fasta(path, [('chrA', c1), ('chrB', c2)], index=index)
      r = G.FastaReader(path)
      assert (r._map is not None) == (index and not suffix)          # pylint: disable=protected-access
      assert r.contig_names() == ['chrA', 'chrB']
      assert r.n_bases('chrA') == len(c1) and r.n_bases('chrB') == 120
      for a, b in ((0, 10), (990, 1100), (59, 61), (60, 120), (0, len(c1)), (13000, 14000), (len(c1) - 1, len(c1) + 5),
                   (500, 500)):
        assert r.get_bases('chrA', a, b) == c1.upper()[a:b], (suffix, index, a, b)
      assert r.get_bases('chrB', 0, 120) == c2 and r.get_bases('chrB', 100, 500) == c2[100:]
      with pytest.raises(KeyError):
        r.n_bases('chrZ')
  path = os.path.join(str(tmp_path), 'x.fa')
  with open(path, 'wb') as f:
    f.write(b'>c1 some description\r\nacgt\r\nNN\r\n>c2\nTT\n')
  r = G.FastaReader(path)
  assert r.contig_names() == ['c1', 'c2'] and r.get_bases('c1', 0, 10) == 'ACGTNN' and r.get_bases('c2', 0, 5) == 'TT'
