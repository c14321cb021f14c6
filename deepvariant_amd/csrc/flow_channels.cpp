// flow_channels.cpp -- host side of the three Ultima flow-space channels: the per-base PIXELS the device encoder
// takes as dv_batch::base_aux planes (include/dvhip.h).  The reference computes them once per read from the read's
// `tp` / `t0` aux tags and its qualities:
//   homopolymer_insertion_quality / homopolymer_deletion_quality
//       deepvariant/channels/homopolymer_indel_quality_channel.cc:68-183 (GetTPValues, HomoPolymerWeighted,
//       HomoPolymerInDelQuality), drawn by homopolymer_{insertion,deletion}_quality_channel.cc:46-61
//   inter_homopolymer_insertion_quality
//       deepvariant/channels/inter_homopolymer_insertion_quality_channel.cc:76-125
// Both scale with deepvariant/channels/channel_utils.{h,cc}: pixel = uint8(255.0f * q / 93.0f) -- 255, not the 254 of
// every other channel.  The arithmetic is the reference's, operation for operation (double pow, float sum, float
// log10), so the bytes are the ones it draws.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>

#include "dv_internal.h"
#include "dvhip.h"

namespace {

inline uint8_t base_quality_color(int q) { return static_cast<uint8_t>(255.0f * q / 93.0f); }

constexpr int kMaxQScore = 93;

// One read: n bases / qualities / tp values -> n pixels.
void hmer_indel_quality(const uint8_t* bases, const uint8_t* quals, const int8_t* tp, size_t n, bool is_deletion,
                        uint8_t* out) {
  const uint8_t top = base_quality_color(kMaxQScore);
  size_t i = 0;
  while (i < n) {
    size_t j = i + 1;
    while (j < n && bases[j] == bases[i]) ++j;
    // The reference keeps run lengths in bytes (capped at 255) and walks the read in steps of the capped length; a
    // longer run is priced piece by piece, the last piece with the cap again -- past the end of the read there.  Here
    // the last piece stops at the end of the run.
    for (size_t p = i; p < j;) {
      const size_t len = std::min<size_t>(std::min<size_t>(j - i, 255), j - p);
      float err = 0;
      for (size_t k = p; k < p + len; ++k) {
        if (tp[k] == 0) continue;
        if ((tp[k] < 0) == is_deletion) {
          const float e = std::pow(10, (quals[k] / -10.0));
          err += e;
        }
      }
      int hq = err == 0 ? kMaxQScore : static_cast<int>(-10 * std::log10(err));
      if (hq > kMaxQScore) hq = kMaxQScore;
      const uint8_t px = err == 0 ? top : base_quality_color(hq);
      for (size_t k = p; k < p + len; ++k) out[k] = px;
      p += len;
    }
    i = j;
  }
}

}  // namespace

extern "C" int dv_flow_channel_pixels(int channel, const uint8_t* bases, const uint8_t* quals, const int8_t* tags,
                                      const uint32_t* read_seq_off, int32_t n_reads, uint8_t* out) {
  if (n_reads < 0 || (n_reads > 0 && (!read_seq_off || !out))) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_flow_channel_pixels: null argument");
  }
  const size_t total = n_reads ? read_seq_off[n_reads] : 0;
  if (total && !tags) return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_flow_channel_pixels: tags is NULL");
  switch (channel) {
    case DV_CH_HOMOPOLYMER_INSERTION_QUALITY:
    case DV_CH_HOMOPOLYMER_DELETION_QUALITY:
      if (total && (!bases || !quals)) {
        return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_flow_channel_pixels: bases / quals are NULL");
      }
      for (int32_t r = 0; r < n_reads; ++r) {
        const size_t s0 = read_seq_off[r], s1 = read_seq_off[r + 1];
        hmer_indel_quality(bases + s0, quals + s0, tags + s0, s1 - s0,
                           channel == DV_CH_HOMOPOLYMER_DELETION_QUALITY, out + s0);
      }
      return DV_OK;
    case DV_CH_INTER_HOMOPOLYMER_INSERTION_QUALITY:
      // tags = t0 character - 33 per base (0 where the read has no t0), as GetT0Values leaves them
      for (size_t i = 0; i < total; ++i) out[i] = base_quality_color(static_cast<uint8_t>(tags[i]));
      return DV_OK;
    default:
      return dv::fail(DV_ERR_INVALID_ARGUMENT,
                      "dv_flow_channel_pixels: channel " + std::to_string(channel) + " is not a flow-space channel");
  }
}
