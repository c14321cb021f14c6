"""AlleleCounter::NormalizeCigar restated on the host (deepvariant_amd/allelecounter.py) against
the reference's vectors: deepvariant/allelecounter_test.cc NormalizeCigar* (:1224-1582)."""
import pytest

from deepvariant_amd import allelecounter as A
from deepvariant_amd import dv_types as T

_OPS = {'M': 1, 'I': 2, 'D': 3, 'S': 5}
LONG = ('GTCAAAGGGTGTTGCATCTGCTTAAACTCACACATCTCGAAGGTTGCTGTGAAGGTAAACAG'
        'AAAGCAACGTAAGGCACGGATGTTGATTCGTGTGTCGTGTGTGTGTGTGTGTGTGTGTGTGT'
        'GCGAAATTTGTACAGCAGTACCTGCAT')
TTCC = 'ATGTTCCTTCCTTCCTTCCTTCCTTCCTTCCACT'


def _cigar(ops):
  return [T.CigarUnit(_OPS[o[-1]], int(o[:-1])) for o in ops]


@pytest.mark.parametrize('name,ref,offset,read,cigar,want,shift', [
    ('del', LONG, 82, 'TGTTGATTCGTGTGTCGTGTGTGTGTGTGTGCGAAATTTGTACAGCAGTACCTGCAT', ['31M', '12D', '26M'],
     ['16M', '12D', '41M'], 0),
    ('ins', LONG, 82, 'TGTTGATTCGTGTGTGTCGTGTGTGTGTGTGTGTGTGTGTGTGTGCGAAATTTGTACAGCAGTACCTGCAT',
     ['13M', '2I', '56M'], ['9M', '2I', '60M'], 0),
    ('ins_del', 'AGTGGGGGGGGGATGGGGG', 0, 'AGTGGGGGGGGGGATGGGG', ['7M', '1I', '10M', '1D', '1M'],
     ['3M', '1I', '11M', '1D', '4M'], 0),
    ('insert_at_the_end', 'AGTGGGGGGGGGATGGGGG', 0, 'AGTGGGGGGGGGGG', ['12M', '2I'], ['3M', '2I', '9M'], 0),
    ('two_dels_merged', 'ATAGACAGATAGATAGATCGATAGAT'[:22], 5, 'CAGATAGA', ['4M', '9D', '1M', '3D', '3M'],
     ['2M', '12D', '6M'], 0),
    ('del_ins_merged', TTCC, 4, 'TCCTTCCTTCCTCCTTCCTTCCTTCCTTCCTTCCA', ['11M', '1D', '4M', '8I', '12M'],
     ['4M', '7I', '24M'], 0),
    ('ins_shifted_to_edge', TTCC, 8, 'TCCTTCCTTCCTTCCTTCCTTCCTTCCACT', ['4M', '4I', '22M'], ['30M'], -4),
    ('ins_shifted_to_soft_clip', TTCC, 8, 'GGGTCCTTCCTTCCTTCCTTCCTTCCTTCCACT', ['3S', '4M', '4I', '22M'],
     ['3S', '30M'], -4),
    ('del_ins_merged_no_shift', TTCC, 4, 'TCCTTCCTTCCTCCTTCCTTCCTTCCTTCCTTCCA', ['11M', '1D', '8I', '16M'],
     ['4M', '7I', '24M'], 0),
])
def test_normalize_cigar(name, ref, offset, read, cigar, want, shift):
  modified, got, read_shift = A.normalize_cigar(read, offset, _cigar(cigar), ref)
  want_ops = [(c.operation, c.operation_length) for c in _cigar(want)]
  assert [(c.operation, c.operation_length) for c in got] == want_ops, name
  assert read_shift == shift and modified
