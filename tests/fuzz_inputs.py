"""Seeded random proto-shaped pileup inputs exercising every CIGAR op, window
overlap, HP tags, strands, support lists, deep pile-ups (shuffle path)."""
import numpy as np

from deepvariant_amd import dv_types as T

_OPS = {'M': 1, 'I': 2, 'D': 3, 'N': 4, 'S': 5, 'H': 6, 'P': 7, '=': 8, 'X': 9}


def random_cigar(rng, min_ops=1, max_ops=7):
  ops = []
  n = int(rng.integers(min_ops, max_ops + 1))
  for k in range(n):
    choices = 'MMM=XIDNSHP' if 0 < k < n - 1 else 'MM=XSHIDN'
    op = choices[int(rng.integers(0, len(choices)))]
    ln = int(rng.integers(1, 40 if op in 'M=X' else 8))
    ops.append(T.CigarUnit(_OPS[op], ln))
  return ops


def query_len(cigar):
  return sum(c.operation_length for c in cigar if c.operation in (1, 2, 5, 8, 9))


def make_case(rng, width, n_reads, variant_start=1000, with_hp=False,
              with_mods=False, n_alts=2, fuzzy=False, with_ultima=False, with_ref_support=False):
  hw = (width - 1) // 2
  ref_window = ''.join('ACGTN'[int(i)] for i in rng.choice(5, size=width,
                                                           p=[.24, .24, .24, .24, .04]))
  alts = (['AC', 'ACC', 'ACCCC'] if fuzzy else ['C', 'G', 'T'])[:n_alts]
  reads = []
  for i in range(n_reads):
    cigar = random_cigar(rng)
    qlen = max(query_len(cigar), 1)
    if query_len(cigar) == 0:
      cigar.append(T.CigarUnit(1, 1))
    start = variant_start - int(rng.integers(-10, hw + 40))
    seq = ''.join('ACGT'[int(j)] for j in rng.integers(0, 4, size=qlen))
    quals = rng.integers(0, 60, size=qlen).astype(np.uint8)
    r = T.Read(
        fragment_name='frag%d' % int(rng.integers(0, max(n_reads // 2, 1))),
        read_number=int(rng.integers(0, 2)), number_reads=2,
        fragment_length=int(rng.integers(-1500, 1500)),
        aligned_sequence=seq, aligned_quality=bytes(quals),
        supplementary_alignment=bool(rng.integers(0, 2)),
        alignment=T.LinearAlignment(
            position=T.Position('chr1', start, bool(rng.integers(0, 2))),
            mapping_quality=int(rng.integers(0, 70)), cigar=cigar))
    if with_hp and rng.random() < 0.7:
      r.info['HP'] = T.ListValue(values=[T.Value(int_value=int(rng.integers(0, 4)))])
    if with_mods and rng.random() < 0.5:
      r.base_modifications[T.K5MC] = bytes(rng.integers(0, 256, size=qlen).astype(np.uint8))
    if with_mods and rng.random() < 0.5:
      r.base_modifications[T.K6MA] = bytes(rng.integers(0, 256, size=qlen).astype(np.uint8))
    if with_ultima:
      # Ultima flow-space tags: tp (per base, the direction / size of the homopolymer error QUAL prices) and t0
      # (per base, phred + 33).  Qualities stay >= 10 so that a homopolymer's summed error stays below 1 (above it the
      # reference casts a negative float to uint8); some reads carry no tag, a short one or a long one.
      runs = rng.integers(1, 7, size=qlen)
      seq = ''.join('ACGT'[int(b)] * int(k) for b, k in zip(rng.integers(0, 4, size=qlen), runs))[:qlen]
      r.aligned_sequence = seq
      r.aligned_quality = bytes(rng.integers(10, 60, size=qlen).astype(np.uint8))
      kind = rng.random()
      n_tag = qlen if kind < 0.7 else max(qlen - 3, 0) if kind < 0.8 else qlen + 2 if kind < 0.9 else -1
      if n_tag >= 0:
        tp = rng.choice([0, 0, 0, 1, -1, 2, -2], size=n_tag)
        r.info['tp'] = T.ListValue(values=[T.Value(int_value=int(v)) for v in tp])
      if n_tag >= 0 or rng.random() < 0.5:
        r.info['t0'] = T.ListValue(values=[T.Value(string_value=''.join(
            chr(33 + int(v)) for v in rng.integers(0, 61, size=max(n_tag, 0) or qlen)))])
    reads.append(r)
  keys = ['%s/%d' % (r.fragment_name, r.read_number) for r in reads]
  support = {}
  for a in alts:
    k = int(rng.integers(0, max(n_reads // 2, 1) + 1))
    if n_reads:
      support[a] = T.SupportingReads(
          [keys[int(j)] for j in rng.integers(0, n_reads, size=k)])
  call = T.DeepVariantCall(
      variant=T.Variant('chr1', variant_start, variant_start + 1, 'A', alts),
      allele_support=support)
  if fuzzy and n_reads:
    # what read_supports_variant_fuzzy reads besides allele_support: per-allele phases, the
    # rejected alleles with their reads, the reference-supporting reads
    pick = lambda k: [keys[int(j)] for j in rng.integers(0, n_reads, size=k)]
    call.variant.info['ALT_PS'] = T.ListValue(
        values=[T.Value(int_value=int(v)) for v in rng.integers(0, 3, size=n_alts + 1)])
    call.variant.alternate_bases_rejected = ['ACCC', 'ACCCCCCC']
    call.rejected_allele_support = {'ACCC': T.SupportingReads(pick(3)),
                                    'ACCCCCCC': T.SupportingReads(pick(2))}
    call.ref_support = pick(4)
  if with_ref_support and n_reads:
    call.ref_support = [keys[int(j)] for j in rng.integers(0, n_reads, size=int(rng.integers(0, 9)))]
  combo = [alts[int(j)] for j in sorted(set(rng.integers(0, n_alts, size=int(rng.integers(1, 3))).tolist()))]
  return call, ref_window, reads, variant_start - hw, combo


from tests import known_answers as KA  # noqa: E402


def options(channels, width, height, **kw):
  rr = T.ReadRequirements(min_mapping_quality=kw.pop('min_mapq', 5),
                          min_base_quality=kw.pop('min_bq', 10))
  o = KA.default_options(channels, read_requirements=rr, **kw)
  o.width, o.height = width, height
  return o


CONFIGS = [
    # (name, channels, width, height, options, case kwargs)
    ('wgs7', T.PILEUP_CHANNELS_WITH_INSERT_SIZE, 221, 100, {}, {}),
    ('narrow_even_band2', T.PILEUP_DEFAULT_CHANNELS, 31, 20,
     dict(reference_band_height=2), {}),
    ('pacbio_like', T.PILEUP_DEFAULT_CHANNELS + ['haplotype', 'base_methylation',
                                                 'base_6ma', 'supplementary_alignment'],
     147, 60, dict(sort_by_haplotypes=True, min_mapq=1), dict(with_hp=True, with_mods=True)),
    ('polishing_hp', T.PILEUP_DEFAULT_CHANNELS + ['haplotype'], 75, 40,
     dict(sort_by_haplotypes=True, hp_tag_for_assembly_polishing=2), dict(with_hp=True)),
    ('aux_channels', ['read_base', 'read_mapping_percent', 'avg_base_quality', 'identity',
                      'gap_compressed_identity', 'blank', 'mean_coverage', 'insert_size'],
     51, 30, {}, {}),
    ('seq_context', T.PILEUP_DEFAULT_CHANNELS + ['gc_content', 'is_homopolymer',
                                                 'homopolymer_weighted'], 45, 26, {}, {}),
    ('sort_by_support', T.PILEUP_DEFAULT_CHANNELS, 41, 24,
     dict(sort_by_alt_allele_support=True), {}),
    ('fuzzy_support', T.PILEUP_DEFAULT_CHANNELS + ['haplotype', 'read_supports_variant_fuzzy'],
     61, 40, dict(sort_by_haplotypes=True, other_allele_supporting_read_alpha=0.3),
     dict(with_hp=True, fuzzy=True)),
]

# Channel sets added in round 4 (the three Ultima flow-space channels, channels/homopolymer_*_quality_channel.cc and
# channels/inter_homopolymer_insertion_quality_channel.cc); kept apart from CONFIGS so that the GPU tests of the two
# lists can be told apart.
ULTIMA_CONFIGS = [
    ('ultima', T.PILEUP_DEFAULT_CHANNELS + ['homopolymer_insertion_quality', 'homopolymer_deletion_quality',
                                            'inter_homopolymer_insertion_quality'], 71, 36, {}, dict(with_ultima=True)),
    ('ultima_with_sequence_context', ['read_base', 'is_homopolymer', 'homopolymer_deletion_quality', 'base_quality',
                                      'inter_homopolymer_insertion_quality'], 33, 20, {}, dict(with_ultima=True)),
    ('ultima_one_channel', ['homopolymer_insertion_quality', 'read_base'], 21, 12, {}, dict(with_ultima=True)),
]

# allele_sample_probability (channels/allele_sample_probability_channel.cc): drawn with the alleles in KEY order --
# what the reference's code does over an ordered map (oracle/_ref); over protobuf's hash map its own output is not a
# function of its input.  Kept apart like the list above.
SAMPLE_PROBABILITY_CONFIGS = [
    ('allele_sample_probability', T.PILEUP_DEFAULT_CHANNELS + ['allele_sample_probability'], 51, 28, {}, dict(n_alts=3, with_ref_support=True)),
]

