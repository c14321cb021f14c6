// alt_merge.hip -- FillPileupArray's channel modes on device-resident images.
//
// For alt_aligned_pileup = diff_channels / base_channels (deepvariant/pileup_image_native.h:
// 246-271) the two trailing channels of every reference-image row block are one channel
// (5 = base_differs_from_ref, or 0 = read_base) of the image drawn against the haplotype of
// alt 1 and of alt 2; alt 1's again when there is no alt 2, zero when there is no alt 1
// (the encoder leaves them zero).  The alt images are items of the same dv_encode_batch
// launch, drawn into scratch rows behind the examples (make_examples_native.py); this kernel
// copies the one channel over, so the fused path never brings images back to the host.
#include <vector>

#include "dv_internal.h"

namespace {

struct MergeArgs {
  uint8_t* images;
  const dv_alt_merge_entry* entries;
  int32_t n_entries;
  uint64_t scratch_off, example_bytes, scratch_bytes;
  int32_t width, channels, first_alt_channel, source_channel;
};

__global__ __launch_bounds__(256) void merge_alt_channels_kernel(MergeArgs a) {
  const dv_alt_merge_entry e = a.entries[blockIdx.y];
  const int64_t pixels = static_cast<int64_t>(e.rows) * a.width;
  const int64_t row_bytes = static_cast<int64_t>(a.width) * a.channels;
  uint8_t* dst = a.images + e.example * a.example_bytes + static_cast<int64_t>(e.first_row) * row_bytes;
  const uint8_t* alt1 = a.images + a.scratch_off + e.scratch_alt1 * a.scratch_bytes;
  const uint8_t* alt2 = e.scratch_alt2 >= 0 ? a.images + a.scratch_off + e.scratch_alt2 * a.scratch_bytes : alt1;
  for (int64_t p = blockIdx.x * 256 + threadIdx.x; p < pixels; p += static_cast<int64_t>(gridDim.x) * 256) {
    dst[p * a.channels + a.first_alt_channel] = alt1[p * a.channels + a.source_channel];
    dst[p * a.channels + a.first_alt_channel + 1] = alt2[p * a.channels + a.source_channel];
  }
}

}  // namespace

extern "C" int dv_merge_alt_channels(uint8_t* images, uint64_t scratch_offset, uint64_t example_bytes,
                                     uint64_t scratch_image_bytes, int32_t width, int32_t channels,
                                     int32_t first_alt_channel, int32_t source_channel,
                                     const dv_alt_merge_entry* entries, int32_t n_entries, void* stream_v) {
  if (!images || n_entries < 0 || (n_entries && !entries) || width <= 0 || channels <= 0 || first_alt_channel < 0 ||
      first_alt_channel + 2 > channels || source_channel < 0 || source_channel >= channels) {
    return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_merge_alt_channels: bad argument");
  }
  if (n_entries == 0) return DV_OK;
  for (int32_t i = 0; i < n_entries; ++i) {
    if (entries[i].scratch_alt1 < 0 || entries[i].rows <= 0 || entries[i].first_row < 0 ||
        static_cast<uint64_t>(entries[i].first_row + entries[i].rows) * width * channels > example_bytes ||
        static_cast<uint64_t>(entries[i].rows) * width * channels > scratch_image_bytes) {
      return dv::fail(DV_ERR_INVALID_ARGUMENT, "dv_merge_alt_channels: entry out of range");
    }
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  dv::DeviceBuffer d;
  if (int rc = d.reserve(sizeof(dv_alt_merge_entry) * n_entries)) return rc;
  hipError_t err = hipMemcpyAsync(d.ptr, entries, sizeof(dv_alt_merge_entry) * n_entries, hipMemcpyHostToDevice, stream);
  if (err == hipSuccess) {
    MergeArgs a{images, static_cast<const dv_alt_merge_entry*>(d.ptr), n_entries, scratch_offset, example_bytes,
                scratch_image_bytes, width, channels, first_alt_channel, source_channel};
    hipLaunchKernelGGL(merge_alt_channels_kernel, dim3(8, n_entries), dim3(256), 0, stream, a);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipStreamSynchronize(stream);   // the entry table is freed below
  d.release();
  if (err != hipSuccess) return dv::fail(DV_ERR_HIP, std::string("dv_merge_alt_channels: ") + hipGetErrorString(err));
  return DV_OK;
}
