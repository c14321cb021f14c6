"""Vectors of deepvariant/allelecounter_test.cc (AlleleCounterTest), as data: used against the
oracle (tests/test_allelecounter_oracle_cpu.py) and against the HIP product
(tests/test_hip_allelecounter.py).  Reference genome = third_party/nucleus/testdata/test.fasta
(chr1 in full, the head of chr2), reads as nucleus::MakeRead builds them (qualities 30,
mapping quality 90), min_base_quality 21 as the fixture sets it."""
from deepvariant_amd import dv_types as T

CHR1 = 'ACCACCATCCTCCGTGAAATCAATATCCCGCACAAGAGTGCTACTCTCCTAAATCCCTTCTCGTCCCCATGGATGA'
CHR2_HEAD = 'CGCTNCGGGCCCATAACACTTGGGGGTAGCTAAAGTGAAC'
MIN_BQ = 21
REF, SUB, INS, DEL, SOFT = 1, 2, 3, 4, 5
START, END = 10, 15       # "TCCGT"
_OPS = {'M': 1, 'I': 2, 'D': 3, 'N': 4, 'S': 5, 'H': 6, 'P': 7, '=': 8, 'X': 9}


class TestRef:
  """GenomeReference stand-in over the two contigs (chr2 padded with its real length)."""
  contigs = {'chr1': CHR1, 'chr2': CHR2_HEAD + 'A' * (121 - len(CHR2_HEAD))}

  def n_bases(self, contig):
    return len(self.contigs[contig])

  def get_bases(self, contig, start, end):
    return self.contigs[contig][start:end]


_counter = [0]


def make_read(chrom, start, bases, cigar, quals=None, mapq=90, name=None, read_number=0):
  _counter[0] += 1
  return T.Read(
      fragment_name=name or 'read_%d' % _counter[0], read_number=read_number, number_reads=2,
      proper_placement=True, aligned_sequence=bases,
      aligned_quality=bytes(quals if quals is not None else [30] * len(bases)),
      alignment=T.LinearAlignment(position=T.Position(chrom, start, False), mapping_quality=mapq,
                                  cigar=[T.CigarUnit(_OPS[c[-1]], int(c[:-1])) for c in cigar]))


def _refs(bases):
  return [[(b, REF, 1)] for b in bases]


def _q(n, **bad):
  q = [MIN_BQ + 1] * n
  for k, v in bad.items():
    q[int(k[1:])] = v
  return q


def cases():
  """-> [(name, contig, start, end, reads, expected per position [(bases, type, count)])]"""
  out = []

  def add(name, reads, expected, contig='chr1', start=START, end=END):
    out.append((name, contig, start, end, reads if isinstance(reads, list) else [reads], expected))

  for op in 'MX=':
    add('simple_' + op, make_read('chr1', START, 'TCCGT', ['5' + op]), _refs('TCCGT'))
  add('spanning_beyond', make_read('chr1', START - 2, 'AATCCGTAA', ['9M']), _refs('TCCGT'))
  seq = 'TCCGT'
  for s in range(5):
    for e in range(5, s, -1):
      add('sub_read_%d_%d' % (s, e), make_read('chr1', START + s, seq[s:e], ['%dM' % (e - s)]),
          [[(seq[i], REF, 1)] if s <= i < e else [] for i in range(5)])
  for subi in range(5):
    bases = seq[:subi] + 'A' + seq[subi + 1:]
    add('substitution_%d' % subi, make_read('chr1', START, bases, ['5M']),
        [[(bases[i], SUB if i == subi else REF, 1)] for i in range(5)])
  add('ins1', make_read('chr1', START, 'TCAAACGT', ['2M', '3I', '3M']),
      [[('T', REF, 1)], [('CAAA', INS, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('ins2', make_read('chr1', START, 'TAAACCGT', ['1M', '3I', '4M']),
      [[('TAAA', INS, 1)], [('C', REF, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('ins3', make_read('chr1', START, 'TCCGTAAA', ['5M', '3I']),
      [[('T', REF, 1)], [('C', REF, 1)], [('C', REF, 1)], [('G', REF, 1)], [('TAAA', INS, 1)]])
  for size in range(1, 10):
    add('ins_size_%d' % size, make_read('chr1', START, 'TC' + 'A' * size + 'CGT', ['2M', '%dI' % size, '3M']),
        [[('T', REF, 1)], [('C' + 'A' * size, INS, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('start_ins_dropped', make_read('chr1', START, 'AAATCCGT', ['3I', '5M']), _refs('TCCGT'))
  add('start_ins_kept', make_read('chr1', START + 1, 'AAACCGT', ['3I', '4M']),
      [[('TAAA', INS, 1)], [('C', REF, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('del1', make_read('chr1', START, 'TCGT', ['2M', '1D', '2M']),
      [[('T', REF, 1)], [('CC', DEL, 1)], [], [('G', REF, 1)], [('T', REF, 1)]])
  add('del2', make_read('chr1', START, 'TCGT', ['1M', '1D', '3M']),
      [[('TC', DEL, 1)], [], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('del3', make_read('chr1', START, 'TCCT', ['3M', '1D', '1M']),
      [[('T', REF, 1)], [('C', REF, 1)], [('CG', DEL, 1)], [], [('T', REF, 1)]])
  add('del_size2', make_read('chr1', START, 'TGT', ['1M', '2D', '2M']),
      [[('TCC', DEL, 1)], [], [], [('G', REF, 1)], [('T', REF, 1)]])
  add('del_size3', make_read('chr1', START, 'TT', ['1M', '3D', '1M']),
      [[('TCCG', DEL, 1)], [], [], [], [('T', REF, 1)]])
  add('del_size4', make_read('chr1', START, 'T', ['1M', '4D']), [[('TCCGT', DEL, 1)], [], [], [], []])
  add('starting_del_lost', make_read('chr1', START, 'CCGT', ['1D', '4M']),
      [[], [('C', REF, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('starting_del_kept', make_read('chr1', START + 1, 'CGT', ['1D', '3M']),
      [[('TC', DEL, 1)], [], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('del_to_end', make_read('chr1', START, 'TCCG', ['4M', '1D']),
      [[('T', REF, 1)], [('C', REF, 1)], [('C', REF, 1)], [('GT', DEL, 1)], []])
  add('del_off_interval', make_read('chr1', START, 'TCCG', ['4M', '3D']),
      [[('T', REF, 1)], [('C', REF, 1)], [('C', REF, 1)], [('GTGA', DEL, 1)], []])
  add('multiple_reads', [
      make_read('chr1', START, 'TCCGT', ['5M']), make_read('chr1', START, 'TCGT', ['2M', '1D', '2M']),
      make_read('chr1', START + 2, 'CGT', ['3M']), make_read('chr1', START, 'TCCAGT', ['3M', '1I', '2M']),
      make_read('chr1', START + 2, 'CG', ['2M'])],
      [[('T', REF, 3)], [('C', REF, 2), ('CC', DEL, 1)], [('C', REF, 3), ('CA', INS, 1)], [('G', REF, 5)],
       [('T', REF, 4)]])
  add('soft1', make_read('chr1', START + 2, 'AACGT', ['2S', '3M']),
      [[], [('CAA', SOFT, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('soft2', make_read('chr1', START + 1, 'ACCGT', ['1S', '4M']),
      [[('TA', SOFT, 1)], [('C', REF, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('soft3', make_read('chr1', START, 'AATCCGT', ['2S', '5M']), _refs('TCCGT'))
  add('soft4', make_read('chr1', START, 'TCCGTAA', ['5M', '2S']),
      [[('T', REF, 1)], [('C', REF, 1)], [('C', REF, 1)], [('G', REF, 1)], [('TAA', SOFT, 1)]])
  for op in ('2S', '2I'):
    add('at_chr_start_' + op, make_read('chr1', 0, 'AAAC', [op, '2M']), [[('A', REF, 1)], [('C', REF, 1)]],
        start=0, end=2)
  n = len(CHR1)
  for op, t in (('2S', SOFT), ('2I', INS)):
    add('at_chr_end_' + op, make_read('chr1', n - 2, 'GAAA', ['2M', op]), [[('G', REF, 1)], [('AAA', t, 1)]],
        start=n - 2, end=n)
  add('del_off_chr_end', make_read('chr1', n - 2, 'GA', ['2M', '2D']), [[('G', REF, 1)], [('A', REF, 1)]],
      start=n - 2, end=n)
  add('match_off_chr_end', make_read('chr1', n - 2, 'GAAAAAAA', ['8M']), [[('G', REF, 1)], [('A', REF, 1)]],
      start=n - 2, end=n)
  add('del_at_chr_start', make_read('chr1', 0, 'CA', ['2D', '2M']), [[], [], [('C', REF, 1)], [('A', REF, 1)]],
      start=0, end=4)
  for bad in range(5):
    exp = _refs('TCCGT')
    exp[bad] = []
    add('min_bq_snp_%d' % bad, make_read('chr1', START, 'TCCGT', ['5M'],
                                         quals=[30 if i != bad else MIN_BQ - 1 for i in range(5)]), exp)
  for bad in (1, 2, 3):
    add('min_bq_ins_%d' % bad, make_read('chr1', START, 'TAAAC', ['1M', '3I', '1M'],
                                         quals=_q(5, **{'i%d' % bad: MIN_BQ - 3})),
        [[], [('C', REF, 1)], [], [], []])
  good = [[('T', REF, 1)], [('CAAA', INS, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]]
  lost = [[('T', REF, 1)], [], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]]
  add('indel_bad_initial_good', make_read('chr1', START, 'TCAAACGT', ['2M', '3I', '3M']), good)
  add('indel_bad_insertion_base', make_read('chr1', START, 'TCAAACGT', ['2M', '3I', '3M'],
                                            quals=_q(8, i3=MIN_BQ - 4)), lost)
  add('indel_bad_anchor_and_insertion', make_read('chr1', START, 'TCAAACGT', ['2M', '3I', '3M'],
                                                  quals=_q(8, i3=MIN_BQ - 4, i1=MIN_BQ - 1)), lost)
  add('indel_bad_anchor_only', make_read('chr1', START, 'TCAAACGT', ['2M', '3I', '3M'],
                                         quals=_q(8, i1=MIN_BQ - 1)), good)
  add('snp_indel', make_read('chr1', START, 'TAAAACGT', ['2M', '3I', '3M']),
      [[('T', REF, 1)], [('AAAA', INS, 1)], [('C', REF, 1)], [('G', REF, 1)], [('T', REF, 1)]])
  add('paired_reads', [make_read('chr1', START, 'TCCAT', ['5M'], name='fragment', read_number=0),
                       make_read('chr1', START, 'TCAAT', ['5M'], name='fragment', read_number=1)],
      [[('T', REF, 2)], [('C', REF, 2)], [('C', REF, 1), ('A', SUB, 1)], [('A', SUB, 2)], [('T', REF, 2)]])
  add('n_base_matches', make_read('chr1', START, 'TCNGT', ['5M']),
      [[('T', REF, 1)], [('C', REF, 1)], [], [('G', REF, 1)], [('T', REF, 1)]])
  add('n_anchors_del', make_read('chr1', START, 'TNGT', ['2M', '1D', '2M']),
      [[('T', REF, 1)], [], [], [('G', REF, 1)], [('T', REF, 1)]])
  add('n_anchors_ins', make_read('chr1', START, 'TCNAGT', ['3M', '1I', '2M']),
      [[('T', REF, 1)], [('C', REF, 1)], [], [('G', REF, 1)], [('T', REF, 1)]])
  add('n_in_insertion', make_read('chr1', START, 'TCCNGT', ['3M', '1I', '2M']), _refs('TCCGT'))
  add('ref_n_substitution', make_read('chr2', 2, 'CTACG', ['5M']),
      [[('C', REF, 1)], [('T', REF, 1)], [('A', SUB, 1)], [('C', REF, 1)], [('G', REF, 1)]],
      contig='chr2', start=2, end=7)
  add('ref_n_deleted', make_read('chr2', 2, 'CTCG', ['2M', '1D', '2M']),
      [[('C', REF, 1)], [('T', REF, 1)], [], [('C', REF, 1)], [('G', REF, 1)]], contig='chr2', start=2, end=7)
  reads = []
  for pos, k, base in ((1, 1, 'C'), (1, 2, 'T'), (2, 3, 'C'), (2, 4, 'T'), (3, 5, 'A'), (3, 6, 'T')):
    reads += [make_read('chr1', pos, base, ['1M']) for _ in range(k)]
  add('count_summaries', reads, [[('C', REF, 1), ('T', SUB, 2)], [('C', REF, 3), ('T', SUB, 4)],
                                 [('A', REF, 5), ('T', SUB, 6)]], start=1, end=4)
  return out
