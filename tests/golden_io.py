"""(De)serialises proto-shaped fixtures (reads, candidates, expected images)
into a single compressed .npz so the golden tests run without /root/reference."""
import io

import numpy as np

from deepvariant_amd import dv_types as T


def pack_reads(reads):
  names = '\n'.join(r.fragment_name for r in reads)
  seq = ''.join(r.aligned_sequence for r in reads)
  qual = b''.join(bytes(bytearray(r.aligned_quality)) for r in reads)
  seq_off = np.cumsum([0] + [len(r.aligned_sequence) for r in reads])
  cig_off = np.cumsum([0] + [len(r.alignment.cigar) for r in reads])
  cig_op = [c.operation for r in reads for c in r.alignment.cigar]
  cig_len = [c.operation_length for r in reads for c in r.alignment.cigar]
  hp = [r.info['HP'].values[0].int_value if 'HP' in r.info else -(1 << 31)
        for r in reads]
  return dict(
      r_names=np.frombuffer(names.encode(), np.uint8),
      r_contig=np.frombuffer(
          (reads[0].alignment.position.reference_name if reads else '').encode(),
          np.uint8),
      r_num=np.array([r.read_number for r in reads], np.int32),
      r_pos=np.array([r.alignment.position.position for r in reads], np.int64),
      r_mapq=np.array([r.alignment.mapping_quality for r in reads], np.int32),
      r_rev=np.array([r.alignment.position.reverse_strand for r in reads], np.uint8),
      r_supp=np.array([r.supplementary_alignment for r in reads], np.uint8),
      r_frag=np.array([r.fragment_length for r in reads], np.int32),
      r_hp=np.array(hp, np.int64),
      r_seq=np.frombuffer(seq.encode(), np.uint8),
      r_qual=np.frombuffer(qual, np.uint8),
      r_seq_off=seq_off.astype(np.int64),
      r_cig_off=cig_off.astype(np.int64),
      r_cig_op=np.array(cig_op, np.int32),
      r_cig_len=np.array(cig_len, np.int64))


def unpack_reads(z):
  names = bytes(z['r_names']).decode().split('\n') if len(z['r_pos']) else []
  contig = bytes(z['r_contig']).decode()
  seq = bytes(z['r_seq']).decode()
  qual = bytes(z['r_qual'])
  reads = []
  for i in range(len(z['r_pos'])):
    s0, s1 = int(z['r_seq_off'][i]), int(z['r_seq_off'][i + 1])
    c0, c1 = int(z['r_cig_off'][i]), int(z['r_cig_off'][i + 1])
    info = {}
    if int(z['r_hp'][i]) != -(1 << 31):
      info['HP'] = T.ListValue(values=[T.Value(int_value=int(z['r_hp'][i]))])
    reads.append(T.Read(
        fragment_name=names[i], read_number=int(z['r_num'][i]), number_reads=2,
        supplementary_alignment=bool(z['r_supp'][i]),
        fragment_length=int(z['r_frag'][i]),
        aligned_sequence=seq[s0:s1], aligned_quality=qual[s0:s1],
        alignment=T.LinearAlignment(
            position=T.Position(contig, int(z['r_pos'][i]), bool(z['r_rev'][i])),
            mapping_quality=int(z['r_mapq'][i]),
            cigar=[T.CigarUnit(int(z['r_cig_op'][k]), int(z['r_cig_len'][k]))
                   for k in range(c0, c1)]),
        info=info))
  return reads


def pack_examples(examples):
  """examples: list of dict(call, alt_alleles, ref_window, read_idx, image)."""
  lines = []
  read_idx, read_off = [], [0]
  rows, row_off, row_ids = [], [0], []
  shapes = []
  for ex in examples:
    call = ex['call']
    v = call.variant
    sup = ';'.join('%s=%s' % (a, ','.join(s.read_names))
                   for a, s in call.allele_support.items())
    lines.append('\t'.join([
        v.reference_name, str(v.start), str(v.end), v.reference_bases,
        ','.join(v.alternate_bases), ','.join(ex['alt_alleles']),
        ex['ref_window'], sup]))
    read_idx.extend(ex['read_idx'])
    read_off.append(len(read_idx))
    img = ex['image']
    shapes.append(img.shape)
    nz = [r for r in range(img.shape[0]) if img[r].any()]
    row_ids.extend(nz)
    rows.extend(img[r].reshape(-1) for r in nz)
    row_off.append(len(row_ids))
  return dict(
      e_meta=np.frombuffer('\n'.join(lines).encode(), np.uint8),
      e_read_idx=np.array(read_idx, np.int32),
      e_read_off=np.array(read_off, np.int64),
      e_row_ids=np.array(row_ids, np.int32),
      e_row_off=np.array(row_off, np.int64),
      e_rows=(np.stack(rows) if rows else np.zeros((0, 0), np.uint8)),
      e_shapes=np.array(shapes, np.int32))


def unpack_examples(z):
  lines = bytes(z['e_meta']).decode().split('\n')
  out = []
  for i, line in enumerate(lines):
    (contig, start, end, ref, alts, combo, window, sup) = line.split('\t')
    support = {}
    if sup:
      for part in sup.split(';'):
        a, names = part.split('=')
        support[a] = T.SupportingReads(names.split(',') if names else [])
    call = T.DeepVariantCall(
        variant=T.Variant(contig, int(start), int(end), ref,
                          alts.split(',') if alts else []),
        allele_support=support)
    shape = tuple(int(x) for x in z['e_shapes'][i])
    img = np.zeros(shape, np.uint8)
    r0, r1 = int(z['e_row_off'][i]), int(z['e_row_off'][i + 1])
    for k in range(r0, r1):
      img[int(z['e_row_ids'][k])] = z['e_rows'][k].reshape(shape[1:])
    a0, a1 = int(z['e_read_off'][i]), int(z['e_read_off'][i + 1])
    out.append(dict(call=call, alt_alleles=combo.split(',') if combo else [],
                    ref_window=window,
                    read_idx=[int(x) for x in z['e_read_idx'][a0:a1]],
                    image=img))
  return out


def save(path, reads, examples, **extra):
  d = {}
  d.update(pack_reads(reads))
  d.update(pack_examples(examples))
  d.update(extra)
  np.savez_compressed(path, **d)


def load(path):
  with np.load(path) as f:
    z = {k: f[k] for k in f.files}
  return unpack_reads(z), unpack_examples(z), z
