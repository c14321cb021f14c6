/*
 * dvhip.h -- C ABI of libdvhip.so, the MI355X (gfx950) implementation of
 * DeepVariant's make_examples -> call_variants hot path.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain
 * pointers and sizes (no torch / protobuf / STL types), and replaces one piece
 * of the reference's native hot path; the reference interface each one
 * replaces is cited as file:line relative to google/deepvariant v1.10.0.
 * INTEGRATION.md shows the pybind/ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - Return value: DV_OK (0) or a negative dv_status; dv_last_error() gives a
 *     thread-local message.  Conditions the reference CHECK-fails on
 *     (LOG(FATAL)) are reported as errors, never silently patched.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     device work is enqueued on it; calls that write HOST memory synchronise
 *     the stream before returning, calls that only touch device memory do not.
 *   - Pointers inside dv_batch are host or device pointers according to
 *     dv_batch.memory.  Device pointers must be 4-byte aligned.
 *   - There is NO CPU fallback: without a HIP device every compute entry
 *     point returns DV_ERR_NO_DEVICE.
 */
#ifndef DVHIP_H_
#define DVHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DV_ABI_VERSION 8
#define DV_MAX_CHANNELS 16
#define DV_READ_AUX_STRIDE 8

typedef enum dv_status {
  DV_OK = 0,
  DV_ERR_INVALID_ARGUMENT = -1,
  DV_ERR_UNSUPPORTED = -2, /* e.g. a channel the device encoder does not draw */
  DV_ERR_NO_DEVICE = -3,
  DV_ERR_HIP = -4,
  DV_ERR_OUT_OF_MEMORY = -5,
  DV_ERR_BAD_INPUT = -6, /* what the reference would CHECK-fail / LOG(FATAL) on */
} dv_status;

typedef enum dv_memory { DV_MEM_HOST = 0, DV_MEM_DEVICE = 1 } dv_memory;

/* DeepVariantChannelEnum values, deepvariant/protos/deepvariant.proto:1287-1342. */
enum {
  DV_CH_READ_BASE = 1,
  DV_CH_BASE_QUALITY = 2,
  DV_CH_MAPPING_QUALITY = 3,
  DV_CH_STRAND = 4,
  DV_CH_READ_SUPPORTS_VARIANT = 5,
  DV_CH_BASE_DIFFERS_FROM_REF = 6,
  DV_CH_HAPLOTYPE_TAG = 7,
  DV_CH_ALLELE_FREQUENCY = 8,          /* pixel supplied in list_aux */
  DV_CH_READ_MAPPING_PERCENT = 11,     /* pixel supplied in read_aux[0] */
  DV_CH_AVG_BASE_QUALITY = 12,         /* read_aux[1] */
  DV_CH_IDENTITY = 13,                 /* read_aux[2] */
  DV_CH_GAP_COMPRESSED_IDENTITY = 14,  /* read_aux[3] */
  DV_CH_GC_CONTENT = 15,               /* read: read_aux[4]; reference row: ref_aux2 */
  DV_CH_IS_HOMOPOLYMER = 16,           /* per base: base_aux0; reference row: ref_aux0 */
  DV_CH_HOMOPOLYMER_WEIGHTED = 17,     /* per base: base_aux1; reference row: ref_aux1 */
  DV_CH_BLANK = 18,
  DV_CH_INSERT_SIZE = 19,
  DV_CH_MEAN_COVERAGE = 22,
  DV_CH_BASE_METHYLATION = 23,
  DV_CH_BASE_6MA = 24,
  DV_CH_READ_SUPPORTS_VARIANT_FUZZY = 25, /* pixel supplied in list_aux */
  DV_CH_SUPPLEMENTARY_ALIGNMENT = 26,
  DV_CH_ALLELE_SAMPLE_PROBABILITY = 27, /* pixel supplied in list_aux */
  /* Ultima flow-space channels (channels/homopolymer_{insertion,deletion}_quality_channel.cc,
   * channels/inter_homopolymer_insertion_quality_channel.cc): per-base pixels the host computes from the tp / t0
   * aux tags (dv_flow_channel_pixels) and passes in the base_aux planes, see dv_batch */
  DV_CH_HOMOPOLYMER_INSERTION_QUALITY = 28,
  DV_CH_HOMOPOLYMER_DELETION_QUALITY = 29,
  DV_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY = 30,
};

/* CIGAR op codes = nucleus CigarUnit::Operation
 * (third_party/nucleus/protos/cigar.proto:38-82); cigar[] = (len << 4) | op. */
enum {
  DV_CIGAR_ALIGNMENT_MATCH = 1,
  DV_CIGAR_INSERT = 2,
  DV_CIGAR_DELETE = 3,
  DV_CIGAR_SKIP = 4,
  DV_CIGAR_CLIP_SOFT = 5,
  DV_CIGAR_CLIP_HARD = 6,
  DV_CIGAR_PAD = 7,
  DV_CIGAR_SEQUENCE_MATCH = 8,
  DV_CIGAR_SEQUENCE_MISMATCH = 9,
};

/* read_flags bits */
enum {
  DV_READ_REVERSE = 1,       /* alignment.position.reverse_strand */
  DV_READ_SUPPLEMENTARY = 2, /* supplementary_alignment */
  DV_READ_HAS_5MC = 4,       /* base_modifications has k5mC */
  DV_READ_HAS_6MA = 8,       /* base_modifications has k6mA */
};

#define DV_HP_NONE INT32_MIN /* read has no (single, integer) HP tag */

/* POD image of the PileupImageOptions fields the encoder reads
 * (deepvariant/protos/deepvariant.proto:500-638; defaults in
 * deepvariant/pileup_image.py:36-74).  `channels` is what
 * PileupImageEncoderNative::AllChannelsEnum("") returns
 * (deepvariant/pileup_image_native.cc:125-151). */
typedef struct dv_encoder_options {
  int32_t width;  /* = ref_bases.size(); the odd-width CHECK of pileup_image_native.cc:114 is the host mirror's */
  int32_t height;
  int32_t reference_band_height;
  int32_t n_channels;
  int32_t channels[DV_MAX_CHANNELS];
  int32_t base_color_offset_a_and_g;
  int32_t base_color_offset_t_and_c;
  int32_t base_color_stride;
  float allele_supporting_read_alpha;
  float allele_unsupporting_read_alpha;
  float other_allele_supporting_read_alpha;
  float reference_matching_read_alpha;
  float reference_mismatching_read_alpha;
  int32_t indel_anchoring_base_char;
  int32_t reference_base_quality;
  int32_t positive_strand_color;
  int32_t negative_strand_color;
  int32_t base_quality_cap;
  int32_t mapping_quality_cap;
  int32_t min_base_quality;    /* read_requirements.min_base_quality */
  int32_t min_mapping_quality; /* read_requirements.min_mapping_quality */
  uint32_t random_seed;
  int32_t sort_by_haplotypes;
  int32_t hp_tag_for_assembly_polishing;
  int32_t sort_by_alt_allele_support;
  float min_non_zero_allele_frequency;
} dv_encoder_options;

/*
 * A packed batch of pileup "items".  One item = one call of
 * PileupImageEncoderNative::BuildPileupForOneSample
 * (deepvariant/pileup_image_native.cc:297-447) followed by the kNone branch of
 * FillPileupArray (deepvariant/pileup_image_native.h:214-275): one candidate x
 * one alt-allele combination x one sample.  Multi-sample stacking and the
 * `rows` / `single_row` alt-aligned layouts are several items whose
 * item_out_off place them one below the other in the same example.
 *
 * Reads are a structure of arrays shared by all items of the batch (a read
 * overlaps many candidates).  Strings never reach the device: the host
 * resolves read names into
 *   read_name_rank  dense rank of (fragment_name, read_number) under the
 *                   reference's tie-break order (pileup_image_native.cc:97-101)
 *   list_code       ReadSupportsVariantChannel::ReadSupportsAlt: 0/1/2
 *                   (deepvariant/channels/read_supports_variant_channel.cc:75-104)
 *   list_group      allele-support sort group (pileup_image_native.cc:346-393)
 * per (item, read).
 */
typedef struct dv_batch {
  int32_t memory; /* dv_memory: where every pointer below lives */

  /* ---- reads (n_reads) ---- */
  int32_t n_reads;
  const int32_t* read_pos;        /* alignment.position.position */
  const int32_t* read_sort_pos;   /* original (pre-trim) position; NULL = read_pos */
  const uint32_t* read_seq_off;   /* [n_reads+1] into bases / quals / mods */
  const uint32_t* read_cigar_off; /* [n_reads+1] into cigar */
  const uint8_t* read_mapq;       /* alignment.mapping_quality (<= 255) */
  const uint8_t* read_flags;      /* DV_READ_* */
  const int32_t* read_frag_len;   /* fragment_length */
  const int32_t* read_hp;         /* HP tag value or DV_HP_NONE */
  const uint32_t* read_name_rank;
  const uint8_t* read_aux;        /* [n_reads][DV_READ_AUX_STRIDE] or NULL */
  const uint8_t* bases;           /* aligned_sequence, ASCII */
  const uint8_t* quals;           /* aligned_quality, raw phred */
  const uint8_t* mod_5mc;         /* parallel to bases, or NULL */
  const uint8_t* mod_6ma;
  const uint32_t* cigar;          /* (operation_length << 4) | operation */
  uint32_t n_bases;               /* = read_seq_off[n_reads] */
  uint32_t n_cigar;               /* = read_cigar_off[n_reads] */

  /* ---- items (n_items) ---- */
  int32_t n_items;
  const int32_t* item_variant_start; /* dv_call.variant.start */
  const int32_t* item_image_start;   /* image_start_pos (may be negative) */
  const uint32_t* item_ref_idx;      /* row of ref_windows */
  const uint32_t* item_list_off;     /* [n_items+1] into list_* */
  const uint16_t* item_height;       /* sample pileup_height (rows of this item) */
  const uint64_t* item_out_off;      /* byte offset of the item's row 0 in `out` */
  const uint32_t* item_blank_mask;   /* bit c = channel index c blanked for reads; NULL = 0 */
  const float* item_mean_coverage;   /* NULL = 0.0 */
  const uint8_t* ref_windows;        /* [n_ref_windows][width] ASCII, N-padded */
  uint32_t n_ref_windows;

  /* ---- per (item, read) lists, in InMemoryReader::Query order
   * (deepvariant/make_examples_native.cc:802-810); the encoder applies
   * DownsampleReadIndices itself ---- */
  const uint32_t* list_read;  /* read index */
  const uint8_t* list_code;   /* 0 / 1 / 2 */
  const uint8_t* list_group;  /* NULL = 0 */
  const uint8_t* list_aux;    /* NULL; host-computed pixel for DV_CH_ALLELE_* */
  uint32_t n_list;            /* = item_list_off[n_items] */
  uint32_t max_list_len;      /* upper bound on any item's list length */

  /* ---- sequence-context channels (channels/{is_homopolymer,homopolymer_weighted,
   * gc_content}_channel.cc): host-computed PIXELS, all optional (NULL unless the channel
   * list asks for them) ---- */
  const uint8_t* base_aux0; /* parallel to bases: is_homopolymer pixel of every read base */
  const uint8_t* base_aux1; /* parallel to bases: homopolymer_weighted pixel */
  const uint8_t* ref_aux0;  /* [n_ref_windows][width]: is_homopolymer pixel of the window */
  const uint8_t* ref_aux1;  /* [n_ref_windows][width]: homopolymer_weighted pixel */
  const uint8_t* ref_aux2;  /* [n_ref_windows][width]: gc_content pixel of the window, repeated */
  /* ABI v6.  A third per-base plane, and the rule that assigns planes: is_homopolymer always reads base_aux0 and
   * homopolymer_weighted base_aux1; each flow-space channel (DV_CH_HOMOPOLYMER_INSERTION_QUALITY, _DELETION_QUALITY,
   * DV_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY), in the order of the channel list, takes the first plane of
   * base_aux2, base_aux1, base_aux0 that no other channel of the list reads (dv_base_aux_plane answers for a given
   * list).  Their reference-row pixel is 0.  More than three per-base channels in one list: DV_ERR_UNSUPPORTED. */
  const uint8_t* base_aux2;
  /* ABI v7.  Optional hints (0 = unknown) that size the encoder's per-item LDS tables: an upper bound on the
   * CIGAR operations of any read and on any item_height.  With them the row pipeline keeps up to 64 operations
   * per kept read in LDS (8 without: reads with more take the row-at-a-time path -- every ONT read).  A host
   * batch is measured by dv_encode_batch itself; a DV_MEM_DEVICE batch relies on the hints, and a hint that
   * is too small is a caller error (items are clamped to it, never read out of bounds). */
  uint32_t max_cigar_ops;
  uint32_t max_item_height;
} dv_batch;

typedef struct dv_encoder dv_encoder;
typedef struct dv_model dv_model;

/* Which base_aux plane (0, 1, 2) the channel at index `channel_index` of `channels[n_channels]` (DV_CH_* values)
 * reads; DV_BASE_AUX_NONE if it reads none; a dv_status (< 0) otherwise: DV_ERR_UNSUPPORTED if the list needs more
 * planes than there are. */
#define DV_BASE_AUX_NONE 3
int dv_base_aux_plane(const int32_t* channels, int32_t n_channels, int32_t channel_index);

/* SampleOptions.use_non_uniform_downsampling (deepvariant/pileup_image_native.cc:242-294,326-341;
 * deepvariant/sampling_util.h:57-155): the reads of one pile-up that survive when every allele keeps at least
 * `min_per_partition` of its supporters.  part_off[n_parts + 1] / part_idx: for every allele of
 * DeepVariantCall.allele_support, in the order the map yields them, the indices (into the pile-up's read list,
 * 0 .. n_reads - 1) of the reads it lists, and as the LAST element the reads no allele lists (GetReadIndicesAllelePartition
 * finds reads by their "fragment_name/read_number" key: of several reads with one key only the last takes part -- the
 * others appear in no element and are never drawn).  A read listed twice belongs to the first element that lists it.
 * out[<= n_reads] receives the sample in ascending order, *n_out its size -- or -1 where the reference's
 * sampler fails (the thresholds alone exceed max_reads) and BuildPileupForOneSample falls back to the uniform
 * shuffle: the caller then passes the whole list to dv_encode_batch as usual.  With a sample in hand the caller passes
 * only those reads (the list is then no longer than max_reads and the device draws them all, in that order).
 * Draws: absl::Uniform over std::mt19937_64(random_seed), restated (csrc/sampling.cpp: parity UNPINNED for the bit
 * stream alone).  forced_draws (tests; NULL otherwise) replaces the draws, one per Uniform(0, index) call. */
int dv_downsample_with_partition_mins(int32_t n_reads, const int32_t* part_off, const int32_t* part_idx,
                                      int32_t n_parts, int32_t max_reads, int32_t min_per_partition,
                                      uint32_t random_seed, const uint64_t* forced_draws, int64_t n_forced,
                                      int32_t* out, int32_t* n_out);

/* Per-base pixels of a flow-space channel for every read of a table, to be passed as the base_aux plane the channel
 * reads (channels/homopolymer_indel_quality_channel.cc:68-183, channels/inter_homopolymer_insertion_quality_channel.cc:
 * 76-125, channels/channel_utils.cc:41-44).  `tags` is parallel to bases:
 *   DV_CH_HOMOPOLYMER_INSERTION_QUALITY / _DELETION_QUALITY   the tp tag's values (0 where a read has none or fewer)
 *   DV_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY                 the t0 tag's characters - 33 (0 where a read has none)
 * Host arithmetic as in the reference (double pow, float sum, float log10); no device work. */
int dv_flow_channel_pixels(int channel, const uint8_t* bases, const uint8_t* quals, const int8_t* tags,
                           const uint32_t* read_seq_off, int32_t n_reads, uint8_t* out);

const char* dv_last_error(void);
int dv_abi_version(void);
/* Number of visible HIP devices (0 without a GPU). */
int dv_device_count(void);

/* ---------------------------------------------------------------- encoder */

/* PileupImageEncoderNative::PileupImageEncoderNative
 * (deepvariant/pileup_image_native.cc:111-123).  Builds the pixel LUTs with the
 * reference's fp32 arithmetic on the host and uploads them. */
int dv_encoder_create(const dv_encoder_options* options, int device,
                      dv_encoder** out);
void dv_encoder_destroy(dv_encoder* enc);

/* BuildPileupForOneSample + FillPileupArray for every item of the batch
 * (deepvariant/pileup_image_native.cc:297-447,
 *  deepvariant/pileup_channel_lib.cc:91-261, deepvariant/channels/ *.cc,
 *  deepvariant/pileup_image_native.h:214-275), one workgroup per item.
 *   out           uint8, item i row r at out + item_out_off[i] + r*width*out_channels,
 *                 pixels HWC with `out_channels` >= n_channels bytes each
 *                 (extra trailing channels are zero-filled)
 *   out_rows      int32[n_items] read rows kept per item, may be NULL
 *   out_memory    dv_memory of out / out_rows */
int dv_encode_batch(dv_encoder* enc, const dv_batch* batch, int out_channels,
                    uint8_t* out, int32_t* out_rows, int out_memory,
                    void* stream);

/* The checks dv_encode_batch applies to a HOST batch before staging it: what the
 * reference LOG(FATAL)s / CHECKs on (unknown CIGAR op, pileup_channel_lib.cc:252; a CIGAR
 * that consumes more bases than aligned_sequence holds) and every index range the kernel
 * trusts.  A caller that uploads its own DV_MEM_DEVICE batch (device batches are not
 * readable from the host) runs this on the host image first.  Host only, no device work. */
int dv_validate_batch(const dv_batch* batch, int32_t reference_band_height);

/* DownsampleReadIndices (deepvariant/pileup_image_native.cc:153-165): iota,
 * std::shuffle'd with std::mt19937_64(seed) iff n > max_reads.  Host only. */
int dv_downsample_indices(int n, int max_reads, uint32_t seed, int32_t* out);

/* InMemoryReader::Query + nucleus::ReadOverlapsRegion for every item
 * (deepvariant/make_examples_native.cc:802-810,
 *  third_party/nucleus/util/utils.cc:172-240) over packed HOST reads:
 * item i gets the reads r (ascending index = caller order) with
 *   query_end[i] > read_pos[r] && query_start[i] < read_end(r).
 * Two-pass: call with list_read == NULL to get counts in list_off[n_items+1],
 * then again with a buffer of list_off[n_items] entries. */
int dv_query_reads(int32_t n_reads, const int32_t* read_pos,
                   const uint32_t* read_cigar_off, const uint32_t* cigar,
                   int32_t n_items, const int64_t* query_start,
                   const int64_t* query_end, uint32_t* list_off,
                   uint32_t* list_read);

/* ---- BAM -> packed read table (host only) --------------
 * Replaces SamReader::Query + ConvertToPb + InMemoryReader construction
 * (third_party/nucleus/io/sam_reader.cc:734-840, deepvariant/make_examples_core.py:
 * region_reads_norealign) for the fields the encoder consumes.  dv_read_requirements
 * mirrors nucleus.genomics.v1.ReadRequirements
 * (third_party/nucleus/protos/reads.proto:421-445; sam_reader.cc:217-247,
 *  third_party/nucleus/util/utils.cc:261-266); make_examples' defaults are all zero
 * except min_mapping_quality (deepvariant/make_examples_options.py:954-990). */
typedef struct dv_read_requirements {
  int32_t keep_duplicates;
  int32_t keep_failed_vendor_quality_checks;
  int32_t keep_secondary_alignments;
  int32_t keep_supplementary_alignments;
  int32_t keep_improperly_placed;
  int32_t min_mapping_quality;
  /* SamReaderOptions.use_original_base_quality_scores (third_party/nucleus/io/sam_reader.cc:722-740):
   * qualities come from the OQ:Z aux tag (char - 33) instead of QUAL.  nucleus leaves
   * aligned_quality EMPTY for a read without the tag -- which the encoder cannot draw -- so
   * here such a read is an error (DV_ERR_BAD_INPUT). */
  int32_t use_original_base_quality_scores;
  /* ABI v6: per-base planes from aux tags, what make_examples asks SamReader to parse when the channel list needs
   * them (deepvariant/make_examples_core.py:288-373 resolve_sam_aux_fields).
   * parse_base_modifications: MM / ML / MN -> 5mC and 6mA planes = Read.base_modifications
   * (third_party/nucleus/io/sam_reader.cc:521-719 ParseBaseModifications, :855-862); dv_read_table_fill_batch then
   * sets dv_batch::mod_5mc / mod_6ma, and read_flags carry DV_READ_HAS_5MC / DV_READ_HAS_6MA per read.
   * parse_flow_tags: the Ultima tp (B array) and t0 (Z) tags -> dv_read_table_aux_planes' tp / t0 planes, the
   * `tags` input of dv_flow_channel_pixels. */
  int32_t parse_base_modifications;
  int32_t parse_flow_tags;
} dv_read_requirements;

typedef struct dv_read_table dv_read_table; /* owns host arrays in dv_batch's read layout */

/* Reads `path` (BGZF BAM).  With `<path>.bai` present and a contig given, only the BGZF
 * members the index points at are read; otherwise the whole file is inflated on
 * `n_threads` threads.  Keeps the mapped reads of `contig` (NULL = all) that overlap
 * [start, end) (nucleus::ReadOverlapsRegion) and satisfy `req` (NULL = keep nothing
 * optional, mapq >= 0), in file order. */
int dv_bam_read_region(const char* path, const char* contig, int64_t start, int64_t end,
                       const dv_read_requirements* req, int n_threads, dv_read_table** out);
/* ---- CRAM 3.0 -> the same packed read table (host only) ------------------------
 * Replaces the CRAM side of SamReader (third_party/nucleus/io/sam_reader.cc:560-640: hts_open +
 * hts_set_opt(CRAM_OPT_REFERENCE) for SamReaderOptions.ref_path / --use_ref_for_cram, :1022-1035
 * the query iterator, :734-840 ConvertToPb).  `fetch` supplies the bases of the FASTA the file
 * was written against: it is asked for [start, end) of `contig` (0-based, half open), writes up to
 * end - start bases into `out` (fewer at a contig's end: *n_out), returns 0 on success; it may be
 * entered from a worker thread of the call, one thread at a time.  NULL = decode from the slices'
 * embedded references only (--nouse_ref_for_cram); a slice that needs the external reference is
 * then DV_ERR_BAD_INPUT, as is a FASTA whose MD5 differs from the slice header's.  With
 * `<path>.crai` present and a contig given, only the containers the index lists are decoded
 * (without one the container headers are walked); slices decode on `n_threads` threads.  Same
 * rows, in the same order, dv_bam_read_region returns for the BAM of the same alignments. */
typedef int (*dv_ref_fetch_fn)(void* ctx, const char* contig, int64_t start, int64_t end,
                               char* out, int64_t* n_out);
int dv_cram_read_region(const char* path, const char* contig, int64_t start, int64_t end,
                        const dv_read_requirements* req, dv_ref_fetch_fn fetch, void* fetch_ctx,
                        int n_threads, dv_read_table** out);
/* The optional per-base planes of a table, parallel to its bases (NULL where the requirements did not ask for them):
 * 5mC / 6mA modification probabilities; tp values (0 where a read has no tag or a shorter one); t0 characters - 33;
 * flow_present[read]: bit 0 = the read has a tp tag, bit 1 = a t0 tag.  Any output pointer may be NULL. */
int dv_read_table_aux_planes(const dv_read_table* table, const uint8_t** mod_5mc, const uint8_t** mod_6ma,
                             const int8_t** tp, const uint8_t** t0, const uint8_t** flow_present);
/* The SAM header text of a CRAM (its first container): `needed` = its length; up to
 * `capacity` bytes are copied to `text` (may be NULL). */
int dv_cram_header(const char* path, char* text, uint64_t capacity, uint64_t* needed);
/* Points the read-table fields of `b` (n_reads, read_*, bases, quals, cigar, n_bases,
 * n_cigar; memory = DV_MEM_HOST) at the table's arrays; item / list fields are left
 * alone.  The table must outlive the batch. */
int dv_read_table_fill_batch(const dv_read_table* t, dv_batch* b);
/* fragment_name (NUL-terminated) and read_number of read i: the key
 * "<fragment_name>/<read_number>" DeepVariantCall.allele_support refers to. */
const char* dv_read_table_name(const dv_read_table* t, int32_t i, int32_t* read_number);
/* All names at once: NUL-terminated strings in `blob` (blob_bytes long), name i at
 * blob + offsets[i], read_numbers[i] in {0, 1}. */
int dv_read_table_names(const dv_read_table* t, const char** blob, const uint32_t** offsets,
                        const uint8_t** read_numbers, uint64_t* blob_bytes);
/* exclusive reference end of every read (alignment start + reference span) */
const int64_t* dv_read_table_ends(const dv_read_table* t);
void dv_read_table_free(dv_read_table* t);

/* ---- candidates + region reads -> item / list arrays (host only) -------------
 * Replaces the per-candidate decisions of ExamplesGenerator::
 * CreateAndWriteExamplesForCandidate that precede the pixels
 * (deepvariant/make_examples_native.cc:632-736): InMemoryReader::Query per
 * candidate (:645-648,802-810), one item per alt-allele combination (:191-267),
 * and per (item, read) the ReadSupportsAlt code
 * (deepvariant/channels/read_supports_variant_channel.cc:54-116) and allele group
 * (deepvariant/pileup_image_native.cc:346-393) from DeepVariantCall.allele_support.
 * Single sample; multi-sample stacking stays with the host mirror. */
typedef struct dv_pack_reads {
  int32_t n_reads;
  const int32_t* read_pos;     /* alignment start */
  const int64_t* read_end;     /* exclusive reference end (dv_read_table_ends) */
  const char* names;           /* NUL-terminated fragment names, concatenated */
  const uint32_t* name_off;    /* [n_reads] offset of each name in `names` */
  const uint8_t* read_number;  /* [n_reads] 0 / 1: the key is "<name>/<read_number>" */
} dv_pack_reads;

typedef struct dv_pack_options {
  int32_t width;                   /* pic_options.width: image_start = variant.start - (width-1)/2 */
  int32_t read_overlap_buffer_bp;  /* pic_options.read_overlap_buffer_bp (5) */
  int32_t pileup_height;           /* sample_options.pileup_height */
  int32_t n_threads;               /* host threads over the candidates; 0 / 1 = the calling thread only */
  uint64_t example_bytes;          /* bytes of one example: item k is written at k * example_bytes */
} dv_pack_options;

typedef struct dv_pack_candidate {
  int64_t start, end;       /* variant.start, variant.end */
  int32_t n_alts;           /* variant.alternate_bases_size (<= 32) */
  int32_t ref_idx;          /* index of its reference window in the batch's ref_windows; < 0 = none
                               (contig edge): the candidate is skipped like the reference does */
  uint32_t first_combo, n_combos;      /* slice of combo_masks: bit i = alternate_bases[i] is in the combination */
  uint32_t first_support, n_support;   /* slice of the support arrays */
} dv_pack_candidate;

typedef struct dv_packed_region dv_packed_region; /* owns the item / list arrays */

/* support_keys: NUL-terminated "<fragment_name>/<read_number>" strings (allele_support
 * read_names), support_key_off[j] the offset of entry j, support_alt[j] the index in
 * alternate_bases of the allele that lists it. */
int dv_pack_region(const dv_pack_reads* reads, const dv_pack_options* opt, int32_t n_candidates,
                   const dv_pack_candidate* cands, const uint32_t* combo_masks,
                   const char* support_keys, const uint32_t* support_key_off,
                   const uint8_t* support_alt, dv_packed_region** out);
/* Points the item / list fields of `b` (n_items, n_list, max_list_len, item_*, list_read,
 * list_code, list_group iff use_groups) at the packed arrays; read-table, reference-window
 * and per-item option fields (blank mask, mean coverage) are the caller's. */
int dv_packed_region_fill_batch(const dv_packed_region* p, int use_groups, dv_batch* b);
/* Which (candidate, combination mask) each item is; returns the item count. */
int dv_packed_region_items(const dv_packed_region* p, const int32_t** item_candidate,
                           const uint32_t** item_combo);
void dv_packed_region_free(dv_packed_region* p);

/* ---- read realigner (host only) ----------------------------------------------
 * Replaces deepvariant/realigner/fast_pass_aligner.{h,cc} (class FastPassAligner) and the
 * local aligner it links (libssw v1.2.5 through deepvariant/realigner/ssw.{h,cc}); used by
 * RealignReadsToHaplotype for alt-aligned pileups (deepvariant/alt_aligned_pileup_lib.cc:
 * 278-313).  Options follow AlignerOptions (deepvariant/protos/realigner.proto:178-228):
 * 0 keeps the class default. */
typedef struct dv_aligner dv_aligner;

typedef struct dv_aligner_options {
  int32_t match, mismatch, gap_open, gap_extend;          /* defaults 4, 6, 8, 1 */
  int32_t kmer_size, read_size, max_num_of_mismatches;    /* defaults 32, 100, 2 */
  double realignment_similarity_threshold;                /* default 0.85 */
  int32_t force_alignment;   /* never return the original alignment (alt-aligned pileups) */
  int32_t normalize_reads;   /* --normalize_reads: keep alignments that could shift left */
  int32_t ref_prefix_len, ref_suffix_len;   /* reference padding around the haplotype */
} dv_aligner_options;

typedef struct dv_realigned_read {
  int32_t status;      /* 0 = original alignment kept, 1 = new alignment, 2 = read dropped
                          (force_alignment and no alignment found: the reference returns an empty Read) */
  int32_t n_cigar;
  int64_t position;    /* new alignment start (reference coordinates), status 1 only */
  uint32_t cigar_off;  /* first word of its CIGAR in the array dv_aligner_align_reads returns */
  uint32_t reserved;
} dv_realigned_read;

typedef struct dv_read_alignment {   /* ReadAlignment (fast_pass_aligner.h:104-128) */
  int32_t position;    /* offset in the haplotype; -1 = kNotAligned */
  int32_t score;
  char cigar[120];     /* text, as the reference keeps it ("15=", "3S3=2I13=") */
} dv_read_alignment;

typedef struct dv_local_alignment {  /* StripedSmithWaterman::Alignment, the fields used */
  int32_t score, ref_begin, ref_end, query_begin, query_end, mismatches;
  char cigar[512];
} dv_local_alignment;

int dv_aligner_create(const dv_aligner_options* options, dv_aligner** out);   /* set_options */
void dv_aligner_destroy(dv_aligner* a);
int dv_aligner_set_reference(dv_aligner* a, const char* reference, int64_t ref_start); /* set_reference + set_ref_start */
int dv_aligner_set_haplotypes(dv_aligner* a, int32_t n, const char* const* haplotypes);
int dv_aligner_set_reads(dv_aligner* a, int32_t n, const char* const* reads);
/* FastPassAligner::AlignReads.  `cigar` receives (length << 4 | op) words -- nucleus op
 * codes, dv_batch's CIGAR encoding -- owned by the aligner until its next call. */
int dv_aligner_align_reads(dv_aligner* a, int32_t n, const char* const* sequences,
                           dv_realigned_read* out, const uint32_t** cigar);
/* Individual stages, for tests that follow fast_pass_aligner_test.cc. */
enum {
  DV_ALIGNER_BUILD_INDEX = 0,        /* BuildIndex */
  DV_ALIGNER_INIT_LOCAL_ALIGNER = 1, /* InitSswLib */
  DV_ALIGNER_ALIGN_HAPLOTYPES = 2,   /* AlignHaplotypesToReference */
  DV_ALIGNER_POSITION_MAPS = 3,      /* CalculatePositionMaps */
  DV_ALIGNER_LOCAL_ALIGN_READS = 4,  /* SswAlignReadsToHaplotypes(arg = score threshold) */
  DV_ALIGNER_SCORE_THRESHOLD = 5     /* CalculateSswAlignmentScoreThreshold */
};
int dv_aligner_stage(dv_aligner* a, int32_t stage, int32_t arg);
int dv_aligner_fast_align(dv_aligner* a, const char* haplotype, int32_t* haplotype_score,
                          dv_read_alignment* out /* [n reads] */);   /* FastAlignReadsToHaplotype */
int dv_aligner_haplotype_info(const dv_aligner* a, int32_t k, int32_t* haplotype_index,
                              int32_t* haplotype_score, int64_t* ref_pos, int32_t* is_reference,
                              char* cigar, int32_t cigar_cap);
int dv_aligner_read_alignment(const dv_aligner* a, int32_t k, int32_t read, dv_read_alignment* out);
/* CalculateReadToRefAlignment; the merged CIGAR as text with M / I / D / S. */
int dv_aligner_merge_alignment(const dv_aligner* a, int32_t read, int32_t position,
                               const char* read_cigar, const char* haplotype_cigar, char* out, int32_t cap);
int dv_aligner_is_normalized(const dv_aligner* a, const char* cigar, int32_t ref_offset,
                             const char* read);                          /* 1 / 0 */
int dv_aligner_score_threshold(const dv_aligner* a);
/* Occurrences of a k-mer in the read index (count returned); "" returns the number of k-mers. */
int dv_aligner_kmer_occurrences(const dv_aligner* a, const char* kmer, int32_t cap, int32_t* reads,
                                int32_t* offsets);
int dv_positions_map(const char* cigar, int32_t haplotype_size, int32_t* out);        /* SetPositionsMap */
int dv_merge_cigar_op(char* cigar, int32_t cap, char op, int32_t length, int32_t read_len); /* MergeCigarOp */
/* One local alignment (Aligner::SetReferenceSequence + Align). */
int dv_local_align(const char* reference, const char* query, int32_t match, int32_t mismatch,
                   int32_t gap_open, int32_t gap_extend, dv_local_alignment* out);
/* The same for n queries against one reference, 16 alignments per SIMD batch (the path the
 * realigner uses for haplotypes -> reference and reads -> haplotypes); identical results.
 * out[k].score = -1 where dv_local_align would fail (empty query). */
int dv_local_align_many(const char* reference, int32_t n, const char* const* queries, int32_t match,
                        int32_t mismatch, int32_t gap_open, int32_t gap_extend, dv_local_alignment* out);

/* ---- local assembly for the window realigner (host only) -----------------------
 * Replaces deepvariant/realigner/debruijn_graph.{h,cc} (DeBruijnGraph::Build,
 * CandidateHaplotypes, GraphViz; python binding deepvariant/realigner/python/
 * debruijn_graph_pybind.cc).  Options are DeBruijnGraphOptions (deepvariant/protos/
 * realigner.proto).  Reads come as the region's packed read table (dv_batch's bases / quals /
 * read_seq_off / read_mapq arrays) plus the indices of the reads that overlap the window, in
 * the order the reference would add them. */
typedef struct dv_debruijn_graph dv_debruijn_graph;

typedef struct dv_debruijn_options {
  int32_t min_k, max_k, step_k;
  int32_t min_mapq, min_base_quality, min_edge_weight, max_num_paths;
  int32_t disable_graph_pruning;
} dv_debruijn_options;

/* *out = NULL (and DV_OK) when no k gives an acyclic graph: debruijn_graph.build() -> None. */
int dv_debruijn_build(const char* ref, int64_t ref_len, const uint8_t* bases, const uint8_t* quals, int64_t n_bases,
                      const uint32_t* read_seq_off, const uint8_t* read_mapq, int32_t n_table_reads,
                      const int32_t* reads, int32_t n_reads, const dv_debruijn_options* options,
                      dv_debruijn_graph** out);
void dv_debruijn_destroy(dv_debruijn_graph* g);
int dv_debruijn_kmer_size(const dv_debruijn_graph* g);                                 /* kmer_size */
/* candidate_haplotypes(): sorted; strings owned by the graph until its next call. */
int dv_debruijn_haplotypes(dv_debruijn_graph* g, int32_t* n, const char* const** haplotypes);
int dv_debruijn_graphviz(dv_debruijn_graph* g, const char** text);                     /* graphviz */

/* ---- the window realigner over many regions in one call (host only, threaded) -----
 * Replaces the body of Realigner.realign_reads (deepvariant/realigner/realigner.py:795-855) for a
 * BATCH of calling regions whose candidate windows are already selected: per window the reads
 * that overlap it are assembled (call_debruijn_graph, :703-738; windows whose only haplotype is
 * the reference are dropped), every read joins the assembled window it shares most bases with
 * (assign_reads_to_assembled_regions, :596-619; the first such window on ties), and each
 * window's reads are realigned against its haplotypes padded with the reference out to the
 * reads' span + ref_align_margin (call_fast_pass_aligner, :740-793; FastPassAligner::AlignReads,
 * deepvariant/realigner/fast_pass_aligner.cc:131-177).  The reference runs this window by
 * window from Python, one make_examples process per core; here the (region, window) tasks of
 * the whole batch are spread over `n_threads` host threads inside one call, and nothing but
 * arrays crosses the boundary.  Results are those of the per-window entry points above. */
typedef struct dv_realign_region {
  /* the region's reads: dv_batch's arrays (one row per read, in the caller's order) */
  const uint8_t* bases;
  const uint8_t* quals;
  int64_t n_bases;
  const uint32_t* read_seq_off;    /* [n_reads + 1] */
  const uint8_t* read_mapq;        /* [n_reads] */
  const int64_t* read_start;       /* [n_reads] alignment start */
  const int64_t* read_end;         /* [n_reads] alignment end, exclusive (ReadRange) */
  int32_t n_reads;
  int32_t n_windows;               /* candidate windows, sorted; each inside the contig and */
  const int64_t* window_start;     /*   no longer than ws_config.max_window_size (the caller's */
  const int64_t* window_end;       /*   filter, realigner.py:717-722) */
  const char* ref;                 /* reference bases [ref_start, ref_start + ref_len): must cover */
  int64_t ref_start;               /*   every read and window +- ref_align_margin, clipped to */
  int64_t ref_len;                 /*   [0, contig_len) */
  int64_t contig_len;
} dv_realign_region;

typedef struct dv_realign_options {
  dv_debruijn_options dbg;
  dv_aligner_options aln;          /* read_size, ref_prefix_len, ref_suffix_len are set per window */
  int32_t ref_align_margin;        /* _REF_ALIGN_MARGIN, realigner.py:263 (20) */
  int32_t n_threads;               /* <= 0: one per hardware thread, at most 16 */
} dv_realign_options;

typedef struct dv_realign_output {   /* views into the result; valid until dv_realign_result_free */
  const int64_t* region_row_off;     /* [n_regions + 1]: region g's rows are [off[g], off[g + 1]) below */
  const int32_t* order;              /* rows of the region in the order realign_reads returns its reads:
                                        first the ones no window claimed, then window by window */
  const int32_t* status;             /* per row (input order): 0 alignment kept, 1 new alignment */
  const int64_t* position;           /* new alignment start, status 1 */
  const int64_t* cigar_off;          /* [rows + 1] into `cigar`; empty unless status 1 */
  const uint32_t* cigar;             /* (length << 4 | op) words */
  const int32_t* region_assembled_off;   /* [n_regions + 1] into the assembled-window arrays */
  const int32_t* assembled_window;   /* index in the region's window list */
  const int32_t* assembled_hap_off;  /* [n_assembled + 1] into the haplotype list */
  const int64_t* hap_text_off;       /* [n_haplotypes + 1] into hap_text */
  const char* hap_text;              /* CandidateHaplotypes.haplotypes, sorted per window */
} dv_realign_output;

typedef struct dv_realign_result dv_realign_result;
int dv_realign_regions(const dv_realign_region* regions, int32_t n_regions, const dv_realign_options* options,
                       dv_realign_result** out, dv_realign_output* arrays);
void dv_realign_result_free(dv_realign_result* r);

/* ---- read phasing for the long-read path (host only) ---------------------------
 * Replaces deepvariant/direct_phasing.{h,cc} (DirectPhasing::PhaseReads / GetPhasedVariants;
 * python binding deepvariant/python/direct_phasing_pybind.cc).  A candidate is a
 * DeepVariantCall reduced to what phasing reads: variant.start / end and, per called allele
 * (allele_support_ext key, UNCALLED_ALLELE left out) plus one is_ref entry for
 * ref_support_ext, the supporting reads as indices into the reads being phased (-1: a read
 * that is not among them) with their is_low_quality flags. */
typedef struct dv_phasing_allele {
  int64_t bases_off;        /* into `bases` */
  int32_t bases_len;
  int32_t is_ref;
  int64_t support_off;      /* into support_reads / support_low_quality */
  int32_t n_support;
  int32_t reserved;
} dv_phasing_allele;

typedef struct dv_phasing_candidate {
  int64_t start, end;
  int32_t allele_off, n_alleles;   /* into `alleles` */
} dv_phasing_candidate;

/* read_phases[n_reads] receives 0 / 1 / 2 (PhaseReads' return value).  Optional outputs, per
 * allele-table entry: allele_phases (-1 = the allele is not a vertex of the graph) and
 * allele_flags (1 = first site of its phase block) -- what GetPhasedVariants reads -- and
 * the graph as text (GraphViz()).  Candidates must be strictly ordered by start. */
int dv_phase_reads(const dv_phasing_candidate* candidates, int32_t n_candidates, const dv_phasing_allele* alleles,
                   int32_t n_alleles, const char* bases, int64_t n_bases, const int32_t* support_reads,
                   const uint8_t* support_low_quality, int64_t n_support, int32_t n_reads,
                   int32_t min_alleles_to_phase, int32_t* read_phases, int32_t* allele_phases,
                   uint8_t* allele_flags, char* graphviz, int32_t graphviz_cap);

/* ---- alt-aligned channel merge (device) --------------------------------------
 * FillPileupArray's diff_channels / base_channels modes (deepvariant/pileup_image_native.h:
 * 246-271) on images that stay in HBM: `images` holds the examples followed, from
 * `scratch_offset`, by scratch images of `scratch_image_bytes` each (the alt images, items of
 * the same dv_encode_batch launch); per entry the two channels first_alt_channel, +1 of rows
 * [first_row, first_row + rows) of one example receive channel `source_channel` (5 or 0) of
 * scratch image scratch_alt1 and scratch_alt2 (alt 1 again when scratch_alt2 < 0). */
typedef struct dv_alt_merge_entry {
  int64_t example;
  int32_t first_row, rows;
  int64_t scratch_alt1, scratch_alt2;
} dv_alt_merge_entry;

int dv_merge_alt_channels(uint8_t* images, uint64_t scratch_offset, uint64_t example_bytes,
                          uint64_t scratch_image_bytes, int32_t width, int32_t channels,
                          int32_t first_alt_channel, int32_t source_channel,
                          const dv_alt_merge_entry* entries, int32_t n_entries, void* stream);

/* ---- allele counting (device) -------------------------------------------------
 * Replaces the per-read loop around AlleleCounter::Add for one region
 * (deepvariant/allelecounter.cc:873-979 with MakeIndelReadAllele :402-469, GetPrevBase
 * :386-400, CanBasesBeUsed :206-229, AddReadAlleles :471-543; constructed as in :349-369).
 * The reads come as the read-table fields of a dv_batch (host or device memory); the result
 * is AlleleCount's content per interval position: ref_supporting_read_count plus one event
 * per non-reference read allele (the host builds read_alleles / allele sums from events:
 * deepvariant_amd/allelecounter.py).  Not produced: methylation fields, REFERENCE read
 * alleles of track_ref_reads, sample_alleles. */
typedef struct dv_allele_counter_options {
  int64_t interval_start, interval_end;             /* AlleleCounter's range (counts are reported here) */
  int64_t reads_interval_start, reads_interval_end; /* full_range (:349-369); pass the interval again if unused */
  const char* ref_bases;                            /* reference of [ref_start, ref_start + n_ref_bases): must cover the
                                                       reads interval; indels that reach outside it are an error */
  int64_t ref_start, n_ref_bases;
  int64_t contig_n_bases;                           /* RefBases validity (:371-384); 0 = end of the window */
  int32_t min_mapping_quality, min_base_quality;    /* AlleleCounterOptions.read_requirements */
  int32_t keep_legacy_behavior;                     /* AlleleCounterOptions.keep_legacy_behavior */
  /* AlleleCounterOptions.track_ref_reads + the constructor's candidate_positions (absolute, any order):
   * at those positions reference-supporting reads are reported too, as events of type 1 (REFERENCE),
   * so that the caller can name them (ref_support_ext, read phasing); allelecounter.cc:504-512 */
  int32_t track_ref_reads;
  const int64_t* candidate_positions;
  int32_t n_candidate_positions;
} dv_allele_counter_options;

typedef struct dv_allele_event {   /* one ReadAllele that AddReadAlleles stores in read_alleles */
  int32_t position;      /* offset in the interval */
  uint32_t read;         /* index in the read table */
  uint32_t read_offset;  /* SUBSTITUTION: the base; INSERTION / SOFT_CLIP: first inserted base; DELETION: next read base */
  uint32_t length_type;  /* bits 0-27: operation length (1 for substitutions; a BAM CIGAR length has 28 bits,
                            so long HiFi / ONT soft clips and deletions are exact);
                            bits 28-30: AlleleType: 2 SUBSTITUTION, 3 INSERTION, 4 DELETION, 5 SOFT_CLIP;
                            1 REFERENCE (track_ref_reads, candidate positions only; read_offset = the base);
                            bit 31: Allele.is_low_quality */
} dv_allele_event;

typedef struct dv_allele_counts dv_allele_counts;  /* owns the host copies of the result */

int dv_count_alleles(const dv_batch* reads, const dv_allele_counter_options* options, dv_allele_counts** out,
                     void* stream);
/* The same for n regions (a region driver's batch of calling regions, each with its own read
 * table and options): every region's arrays travel in one pinned staging image, the kernels are
 * queued back to back and the stream is synchronised twice for the whole batch instead of twice
 * per region.  out[k] receives region k's result (each freed with dv_allele_counts_free); on an
 * error no result is left allocated. */
int dv_count_alleles_batch(int32_t n, const dv_batch* const* reads, const dv_allele_counter_options* const* options,
                           dv_allele_counts** out, void* stream);
/* Events are sorted by (position, read, read_offset) = the order AddReadAlleles stores them
 * for one position.  Returns the interval length. */
int dv_allele_counts_arrays(const dv_allele_counts* c, const int32_t** ref_supporting_read_count,
                            const dv_allele_event** events, uint32_t* n_events, int32_t* n_reads_counted);
void dv_allele_counts_free(dv_allele_counts* c);

/* CRC32C (Castagnoli) as used by TFRecord framing
 * (third_party/nucleus/io/example_writer.cc:88-104 via tensorflow::io::RecordWriter). */
uint32_t dv_crc32c(const uint8_t* data, size_t n);

/* ------------------------------------------------------------------ model */

/* Inception-v3 classifier as instantiated by
 * deepvariant/keras_modeling.py:246-336 (tf_keras InceptionV3,
 * include_top=False, pooling='avg') + Dense(3, softmax) head (:46-67), for
 * input [height, width, channels] uint8. */
typedef struct dv_model_desc {
  int32_t height;
  int32_t width;
  int32_t channels;
  int32_t num_classes; /* 3 */
  int32_t max_batch;   /* activations are sized for this many examples (<= 8192; ~6 MB of HBM each at 100x221) */
} dv_model_desc;

int dv_model_create(const dv_model_desc* desc, int device, dv_model** out);
void dv_model_destroy(dv_model* m);

/* Number of fp32 values dv_model_load_weights expects and, per layer i of
 * dv_model_num_layers(), its slice: conv kernel HWIO (kh*kw*cin*cout) then BN
 * beta, moving_mean, moving_variance (cout each); the last layer is the Dense
 * kernel [2048, num_classes] + bias.  Layer order = tf_keras CONSTRUCTION
 * order (applications/inception_v3.py).  That is NOT the
 * checkpoint's `layer_with_weights-N` numbering, which follows Keras'
 * depth-sorted `model.layers`: a TensorFlow checkpoint is mapped by variable
 * name and shape by the host importer (deepvariant_amd/keras_layout.py,
 * call_variants.import_keras_checkpoint), never by position. */
int64_t dv_model_num_params(const dv_model* m);
/* Multiply-accumulates of the 94 convolutions for ONE example of the model's input
 * shape (padding taps included, as in every FLOP count of this architecture): the
 * algorithmic work bench.py prices the conv kernels with. */
int64_t dv_model_conv_macs(const dv_model* m);
int dv_model_num_layers(const dv_model* m);
int dv_model_layer_info(const dv_model* m, int layer, int32_t* kh, int32_t* kw,
                        int32_t* cin, int32_t* cout, int64_t* param_offset);

/* Folds BatchNorm (scale=False, eps=1e-3) into the conv weights, converts to
 * fp16 in the MFMA fragment layout and uploads.  `weights` is host memory.
 * Replaces model.load_weights (deepvariant/call_variants.py:759-762). */
int dv_model_load_weights(dv_model* m, const float* weights, int64_t n);

/* Shift calibration (ABI v7; optional, after dv_model_load_weights with the SAME `weights`).
 * The classifier multiplies fp16 weights by fp16 activations where the reference computes in float32
 * (deepvariant/call_variants.py:913-918).  Both roundings have a per-channel MEAN (the weight
 * error is one fixed draw multiplying activations that are far from zero-mean; constant map regions
 * round identically everywhere); this call measures it on `n_images` example images -- two fp32
 * pipelines on the device, one exact, one with the MFMA kernels' roundings, walked layer by layer
 * (csrc/calib.h) -- and moves each layer's fp32 shift (and the Dense bias) by the difference of the
 * per-channel pre-activation means.  No run-time cost; deterministic for given weights and images;
 * calling it again replaces the previous correction.  Any batch drawn like the inputs the model
 * will see serves (a few hundred pile-ups; tests use OTHER images than the ones they check).
 *   weights  host, the array given to dv_model_load_weights
 *   images   device uint8 [n_images, height, width, channels]
 *   corrections  optional host array [capacity]: the shift corrections in layer order (cout values per
 *                convolution), then num_classes logit corrections; for tests and reports
 * The reference has no counterpart: it keeps float32 end to end. */
int dv_model_calibrate(dv_model* m, const float* weights, int64_t n_weights, const uint8_t* images,
                       int n_images, float* corrections, int64_t capacity);

/* Applies shift corrections measured elsewhere (ABI v7): `corrections` in the layout dv_model_calibrate
 * reports -- cout values per convolution in layer order, then num_classes logit corrections; `n` must be
 * exactly that many.  Replaces any previous correction (the loaded shifts minus these).  For host
 * processes that share one set of weights -- the ranks of `make_examples --ranks_per_gpu R`: one of them
 * calibrates, the others apply its result instead of repeating the measurement on the same GPU. */
int dv_model_apply_corrections(dv_model* m, const float* corrections, int64_t n);

/* Diagnostic (ABI v8): WHERE the fp16 error of the classifier enters.  Runs dv_model_calibrate's two fp32
 * pipelines (csrc/calib.h: R exact, E with the MFMA kernels' rounding points) on `n_images` device images with FIXED
 * shift corrections (`corrections` in dv_model_calibrate's layout, or NULL for none) and returns both pipelines'
 * logits (host, [n_images][num_classes]; logits_r may be NULL: R is skipped).  keep_f32[op] != 0 keeps the output
 * tensor of graph op `op` (0 .. dv_model_num_ops - 1, dv_model_op_label describes each) in float32 in E -- what
 * storing that tensor wider than fp16 would buy; NULL = the product's own rounding points.  flags bit 0: E multiplies
 * the float32 weights (isolates the activation roundings); bit 1: measure the corrections on these images under this
 * plan (the calibration proper; `corrections` must be NULL) and report them in `corrections_out` (optional, the
 * layout of dv_model_calibrate).  The model's state is not changed.  tools/r6_tensor_budget.py;
 * no reference counterpart (deepvariant/call_variants.py:913-918 computes in float32 end to end). */
int dv_model_num_ops(const dv_model* m);
int dv_model_op_label(const dv_model* m, int op_index, char* buf, int capacity);
int dv_model_probe_rounding(dv_model* m, const float* weights, int64_t n_weights, const uint8_t* images,
                            int n_images, const uint8_t* keep_f32, int flags, const float* corrections,
                            int64_t n_corrections, float* logits_r, float* logits_e, float* corrections_out);

/* preprocess_images ((x-128)/128, deepvariant/dv_utils.py:343-366) + model
 * forward + softmax (deepvariant/call_variants.py:904-932).
 *   images  device uint8 [n, height, width, channels]
 *   probs   device fp32  [n, num_classes] */
int dv_model_infer(dv_model* m, const uint8_t* images, int n, float* probs,
                   void* stream);

/* Blank-row skipping (ABI v8; on by default for inputs of <= 16 channels).  A pileup image is zero below its last
 * read row (deepvariant/pileup_image_native.cc:405-447 zero-pads every image to `height` rows; a 30x pile-up fills
 * ~40 of 100), and the model normalises zero to the constant -1 (deepvariant/dv_utils.py:343-366): every activation of
 * the stem whose receptive field lies inside those rows equals the all-blank image's response at the same position.
 * dv_model_infer finds each image's last nonzero row (one scan kernel) and the stem kernels copy such tiles from the
 * precomputed response instead of multiplying -- bit-identical to the dense path (tests/test_hip_blank_skip.py), so
 * probabilities are unchanged; only the executed work depends on the images' depth.  `enabled` = 0 runs the dense path
 * (bench.py reports both); the environment variable DV_BLANK_SKIP=0 does the same for a whole process.
 * dv_model_blank_thresholds: what the last forward's scan found for its first n examples, out[k * n + i] with
 * k = 0 first all-zero row, 1..4 the first blank-determined row of conv2 / stem_b / the 3x3 80->192 / its pooled
 * output (bench.py derives the executed FLOPs from them); DV_ERR_UNSUPPORTED when the model does not skip. */
int dv_model_set_blank_skip(dv_model* m, int enabled);

/* Precise mode (ABI v8).  The reference classifies in float32 (deepvariant/call_variants.py:913-918); with fp16 MFMA
 * operands the zero-mean rounding noise of the stored activations puts sigma(dp) of the long-read models' deeper pile-ups
 * at ~2.5e-4, and the largest of a few thousand candidates beyond north_star's 1e-3 on some weight seeds.  In precise
 * mode every fp16 tensor of the 17x17 and 8x8 stages (83 % of that variance; the per-tensor table is
 * profiles/r06_tensor_budget_*.txt) is stored as hi = fp16(x) and lo = fp16(x - hi), and its consumers run their K over
 * both with the same weights: 22-bit activations at twice the MFMA count there, +40 % on the forward.  dv_model_create
 * switches it on for inputs of more than 8 channels (PACBIO, ONT_R104) and off otherwise; the environment variable
 * DV_PRECISE=0 / 1 read at dv_model_create overrides.  Same ABI, same weights, same outputs to within the tolerance. */
int dv_model_is_precise(const dv_model* m);
int dv_model_blank_thresholds(dv_model* m, int n, int32_t* out);

/* dv_model_infer for a caller that KNOWS how many rows of each image are drawn (ABI v8): `rows_used` is a device
 * array int32[n], and the caller promises that in image i every byte of rows >= rows_used[i] + rows_add is zero --
 * what dv_encode_batch's `out_rows` (read rows kept) + the reference band height is for the images it has just drawn.
 * Blank-row skipping then takes its thresholds from the array instead of scanning the images (0.2 ms per 8 K
 * ILLUMINA30 pileups).  NULL = dv_model_infer.  A wrong promise gives wrong probabilities; callers that read images
 * from files (call_variants) use dv_model_infer.  The reference has no counterpart. */
int dv_model_infer_rows(dv_model* m, const uint8_t* images, int n, float* probs, const int32_t* rows_used,
                        int rows_add, void* stream);

/* On a non-default stream the forward is captured once per (n, stream) into a hipGraph and
 * replayed; the image / probability pointers are read from a device-side table, so they may
 * change from call to call without a new capture.  Testing hook: captures and replays so far. */
int dv_model_graph_stats(const dv_model* m, int64_t* captures, int64_t* replays);

/* Testing hook: copy activation buffer `index` (fp16, channel-blocked
 * [n][c/8][h][w][8], first n examples of the last dv_model_infer) to host
 * memory and report its (padded) shape.  index -1 = the last Inception block's
 * output (what GlobalAveragePooling reads), -2 = the stem's output. */
int dv_model_debug_tensor(dv_model* m, int index, int n, void* host_out,
                          int32_t* h, int32_t* w, int32_t* c);

/* Average time of kernels launched by the last dv_model_infer /
 * dv_encode_batch on `stream`, measured with HIP events around each launch
 * when profiling is on.  kind: 0 = encoder kernel, 1 = all conv kernels,
 * 2 = everything else.  Returns milliseconds summed over the call. */
int dv_set_profiling(int enabled);
double dv_profile_ms(int kind);
/* Number of launches summed by the last dv_profile_ms call. */
int dv_last_profile_count(void);

#ifdef __cplusplus
}
#endif
#endif /* DVHIP_H_ */
