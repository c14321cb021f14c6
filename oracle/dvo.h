/*
 * dvo.h -- C interface of the CPU ORACLE for the pileup-image encoder.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * The oracle is a from-scratch CPU restatement (C++17, no dependencies) of the
 * reference algorithm in google/deepvariant v1.10.0:
 *   deepvariant/pileup_image_native.cc   (BuildPileupForOneSample, EncodeRead,
 *                                         EncodeReference, DownsampleReadIndices,
 *                                         SortImageRows, GetHapIndex)
 *   deepvariant/pileup_channel_lib.cc    (CalculateChannels, CalculateBaseLevelData,
 *                                         CalculateRefRows)
 *   deepvariant/channels/ *.cc            (per-channel FillReadBase / FillRefBase)
 *   deepvariant/pileup_image_native.h    (FillPileupArray: CHW rows -> HWC bytes)
 * Parity is pinned by the reference's own known-answer vectors and golden
 * TFRecords (tests/test_oracle_*.py) AND by the reference itself: its encoder
 * sources compile here, unmodified, against generated structs and small abseil
 * stand-ins (oracle/ref_build/ -> oracle/_ref/libdvref.so, which exports this same
 * interface); tests/test_reference_encoder_cpu.py holds the two against each other.
 *
 * Inputs are proto-shaped (names as strings, allele_support as string lists):
 * the oracle does the reference's string matching itself, it does not consume
 * the product's packed/pre-resolved format (except through dvo_encode_packed,
 * which re-expands a packed batch into proto-shaped inputs first).
 */
#ifndef DVO_H_
#define DVO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVO_MAX_CHANNELS 32

/* DeepVariantChannelEnum, deepvariant/protos/deepvariant.proto:1287-1342 */
enum {
  DVO_CH_UNSPECIFIED = 0,
  DVO_CH_READ_BASE = 1,
  DVO_CH_BASE_QUALITY = 2,
  DVO_CH_MAPPING_QUALITY = 3,
  DVO_CH_STRAND = 4,
  DVO_CH_READ_SUPPORTS_VARIANT = 5,
  DVO_CH_BASE_DIFFERS_FROM_REF = 6,
  DVO_CH_HAPLOTYPE_TAG = 7,
  DVO_CH_ALLELE_FREQUENCY = 8,
  DVO_CH_DIFF_CHANNELS_ALTERNATE_ALLELE_1 = 9,
  DVO_CH_DIFF_CHANNELS_ALTERNATE_ALLELE_2 = 10,
  DVO_CH_READ_MAPPING_PERCENT = 11,
  DVO_CH_AVG_BASE_QUALITY = 12,
  DVO_CH_IDENTITY = 13,
  DVO_CH_GAP_COMPRESSED_IDENTITY = 14,
  DVO_CH_GC_CONTENT = 15,
  DVO_CH_IS_HOMOPOLYMER = 16,
  DVO_CH_HOMOPOLYMER_WEIGHTED = 17,
  DVO_CH_BLANK = 18,
  DVO_CH_INSERT_SIZE = 19,
  DVO_CH_BASE_CHANNELS_ALTERNATE_ALLELE_1 = 20,
  DVO_CH_BASE_CHANNELS_ALTERNATE_ALLELE_2 = 21,
  DVO_CH_MEAN_COVERAGE = 22,
  DVO_CH_BASE_METHYLATION = 23,
  DVO_CH_BASE_6MA = 24,
  DVO_CH_READ_SUPPORTS_VARIANT_FUZZY = 25,
  DVO_CH_SUPPLEMENTARY_ALIGNMENT = 26,
  DVO_CH_ALLELE_SAMPLE_PROBABILITY = 27,
  DVO_CH_HOMOPOLYMER_INSERTION_QUALITY = 28,       /* Ultima: tp tag */
  DVO_CH_HOMOPOLYMER_DELETION_QUALITY = 29,        /* Ultima: tp tag */
  DVO_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY = 30, /* Ultima: t0 tag */
};

/* CigarUnit::Operation, third_party/nucleus/protos/cigar.proto:38-82 */
enum {
  DVO_CIGAR_ALIGNMENT_MATCH = 1,
  DVO_CIGAR_INSERT = 2,
  DVO_CIGAR_DELETE = 3,
  DVO_CIGAR_SKIP = 4,
  DVO_CIGAR_CLIP_SOFT = 5,
  DVO_CIGAR_CLIP_HARD = 6,
  DVO_CIGAR_PAD = 7,
  DVO_CIGAR_SEQUENCE_MATCH = 8,
  DVO_CIGAR_SEQUENCE_MISMATCH = 9,
};

/* The fields of PileupImageOptions (deepvariant.proto:500-638) that the
 * encoder reads.  `channels` is AllChannelsEnum("") (pileup_image_native.cc:
 * 125-151): pic_options.channels mapped through ChannelStrToEnum with the
 * CH_UNSPECIFIED (alt-aligned) names dropped. */
typedef struct dvo_options {
  int32_t width;
  int32_t height;
  int32_t reference_band_height;
  int32_t n_channels;
  int32_t channels[DVO_MAX_CHANNELS];
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
  int32_t min_base_quality;     /* read_requirements.min_base_quality */
  int32_t min_mapping_quality;  /* read_requirements.min_mapping_quality */
  uint32_t random_seed;
  int32_t sort_by_haplotypes;
  int32_t hp_tag_for_assembly_polishing;
  int32_t sort_by_alt_allele_support;
  float min_non_zero_allele_frequency;
  /* SampleOptions.use_non_uniform_downsampling / non_uniform_downsampling_threshold (deepvariant.proto:715-720) */
  int32_t use_non_uniform_downsampling;
  int32_t non_uniform_downsampling_threshold;
} dvo_options;

/* nucleus.genomics.v1.Read, the fields the encoder touches. */
typedef struct dvo_read {
  const char* fragment_name; /* NUL-terminated */
  int32_t read_number;
  int64_t position; /* alignment.position.position */
  int32_t mapping_quality;
  int32_t reverse_strand;
  int32_t supplementary;
  int32_t fragment_length;
  const char* seq; /* aligned_sequence */
  int32_t seq_len;
  const uint8_t* qual; /* aligned_quality */
  int32_t qual_len;
  const int32_t* cigar_ops;
  const int64_t* cigar_lens;
  int32_t n_cigar;
  int32_t hp_present;  /* read.info contains "HP" */
  int32_t hp_n_values; /* read.info["HP"].values_size() */
  int32_t hp_is_int;   /* values(0).kind_case() == kIntValue */
  int32_t hp_value;    /* values(0).int_value() */
  const uint8_t* mod_5mc; /* base_modifications[k5mC], NULL if absent */
  int32_t mod_5mc_len;
  const uint8_t* mod_6ma;
  int32_t mod_6ma_len;
  const int8_t* tp; /* read.info["tp"].values(i).int_value(), NULL if the tag is absent */
  int32_t tp_len;
  const char* t0;   /* read.info["t0"].values(0).string_value(), NULL if the tag is absent */
  int32_t t0_len;
} dvo_read;

/* DeepVariantCall (deepvariant.proto:262-317), the fields the encoder touches. */
typedef struct dvo_call {
  int64_t variant_start;
  int32_t n_alts;
  const char* const* alts; /* variant.alternate_bases */
  int32_t n_support;       /* allele_support entries */
  const char* const* support_alleles;
  const int32_t* support_offsets; /* [n_support+1] into support_names */
  const char* const* support_names;
  int32_t n_af;
  const char* const* af_alleles;
  const float* af_values;
  int32_t n_ref_support;
  /* read_supports_variant_fuzzy only (channels/read_supports_variant_fuzzy_channel.cc) */
  const char* reference_bases;          /* variant.reference_bases, NULL -> "" */
  const char* const* ref_support_names; /* ref_support[n_ref_support], may be NULL */
  int32_t alt_ps_present;               /* variant.info contains "ALT_PS" */
  int32_t n_alt_ps;                     /* info["ALT_PS"].values_size() */
  const int32_t* alt_ps;                /* values(i).int_value() */
  int32_t n_rejected_alts;              /* variant.alternate_bases_rejected */
  const char* const* rejected_alts;
  int32_t n_rejected_support;           /* rejected_allele_support entries */
  const char* const* rejected_support_alleles;
  const int32_t* rejected_support_offsets; /* [n_rejected_support+1] */
  const char* const* rejected_support_names;
} dvo_call;

const char* dvo_last_error(void);

/* Channels::ChannelStrToEnum (pileup_channel_lib.cc:421-512). -1 if unknown. */
int dvo_channel_str_to_enum(const char* name);

/* EncodeReference -> one row, HWC bytes out[w * n_channels]. 0 on success. */
int dvo_encode_reference(const dvo_options* opt, const char* ref_bases, int w,
                         uint8_t* out_hwc);

/* EncodeRead -> 1 row written, 0 = nullptr (read rejected), <0 = error. */
int dvo_encode_read(const dvo_options* opt, const dvo_call* call,
                    const char* ref_bases, int w, const dvo_read* read,
                    int32_t image_start_pos, const char* const* alt_alleles,
                    int n_alt_alleles, const int32_t* channels_to_blank,
                    int n_blank, uint8_t* out_hwc);

/* BuildPileupForOneSample followed by FillPileupArray(kNone): out_hwc is
 * [pileup_height][w][n_channels].  pileup_height==0 -> opt->height.
 * out_row_read[h] = index into reads[] drawn on row h, or -1 (reference band /
 * blank).  Returns number of read rows kept, <0 on error. */
int dvo_build_pileup(const dvo_options* opt, const dvo_call* call,
                     const char* ref_bases, int w, const dvo_read* reads,
                     int n_reads, int32_t image_start_pos,
                     const char* const* alt_alleles, int n_alt_alleles,
                     int pileup_height, float mean_coverage,
                     const int64_t* alignment_positions,
                     const int32_t* channels_to_blank, int n_blank,
                     uint8_t* out_hwc, int32_t* out_row_read);

/* ReadSupportsVariantFuzzyChannel::ReadSupportsAlt: 0 / 1 / 2 / 10 / 9. */
int dvo_fuzzy_read_supports_alt(const dvo_call* call, const dvo_read* read,
                                const char* const* alt_alleles, int n_alt_alleles);

/* DownsampleReadIndices (pileup_image_native.cc:153-165): iota, shuffled with
 * std::mt19937_64(seed) iff n > max_reads.  out[n]. */
int dvo_downsample_indices(int n, int max_reads, uint32_t seed, int32_t* out);

/* ReadOverlapsRegion minus the contig test (nucleus/util/utils.cc:172-240). */
int dvo_read_overlaps(const dvo_read* read, int64_t start, int64_t end);

/* ---- packed-batch adapter (for full-size parity runs and the CPU baseline).
 * Mirrors the product's packed layout (include/dvhip.h, dv_batch) field for
 * field, re-expands every item into proto-shaped inputs (names synthesised
 * from the name ranks, allele_support rebuilt from the support codes) and runs
 * dvo_build_pileup on it.  n_threads>1 splits items over std::threads. */
typedef struct dvo_packed_batch {
  int32_t n_reads;
  const int32_t* read_pos;
  const int32_t* read_sort_pos; /* NULL -> read_pos */
  const uint32_t* read_seq_off;   /* [n_reads+1] */
  const uint32_t* read_cigar_off; /* [n_reads+1] */
  const uint8_t* read_mapq;
  const uint8_t* read_flags; /* bit0 reverse, bit1 supplementary, bit2 has 5mC, bit3 has 6mA */
  const int32_t* read_frag_len;
  const int32_t* read_hp; /* INT32_MIN = no HP tag */
  const uint32_t* read_name_rank;
  const uint8_t* bases;
  const uint8_t* quals;
  const uint8_t* mod_5mc; /* parallel to bases, may be NULL */
  const uint8_t* mod_6ma;
  const uint32_t* cigar; /* (len << 4) | op */
  int32_t n_items;
  const int32_t* item_variant_start;
  const int32_t* item_image_start;
  const uint32_t* item_ref_idx; /* index into ref_windows[][width] */
  const uint32_t* item_list_off; /* [n_items+1] */
  const uint16_t* item_height;   /* pileup height of this item */
  const uint64_t* item_out_off;  /* byte offset of the item's first row in out */
  const uint8_t* ref_windows;
  const uint32_t* list_read; /* read index, in Query() order (NOT shuffled) */
  const uint8_t* list_code;  /* ReadSupportsAlt: 0 / 1 / 2 */
  const uint8_t* list_group; /* allele support group (sort_by_alt_allele_support), may be NULL */
  const uint32_t* item_blank_mask;  /* bit i = blank channel i of opt->channels; may be NULL */
  const float* item_mean_coverage;  /* may be NULL (0) */
} dvo_packed_batch;

int dvo_encode_packed(const dvo_options* opt, const dvo_packed_batch* b,
                      int out_channels, uint8_t* out, int32_t* out_rows,
                      int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* DVO_H_ */
