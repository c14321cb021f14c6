// Interface between model.hip (graph, weights) and stem.hip (the fused stem kernels).
#ifndef DV_STEM_FUSED_H_
#define DV_STEM_FUSED_H_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace dv {

// Geometry of an fp16 activation tensor in the channel-blocked, zero-haloed
// layout [N][C/8][h + 2 halo][w + 2 halo][8] (model.hip).
struct C8Geom {
  int h, w, halo, hp, wp, groups;
};

// ---- stem A: conv 3x3/2 (uint8 pileup, C <= 12 channels -> 32) + conv 3x3 32 -> 32 -------
// tf_keras InceptionV3 stem layers 1-2 (deepvariant/keras_modeling.py:268-274 builds the
// backbone); both 'valid'.  One workgroup produces a TH x TW tile of the SECOND conv: the
// (TH+2) x (TW+2) tile of the first conv's output lives in LDS only.
constexpr int kStemA_TH = 7, kStemA_TW = 54;
constexpr int kStemA_W1Halfs = 5 * 2 * 32 * 8;    // conv1: 5 chunks (2 taps x 8 channels each)
constexpr int kStemA_W2Halfs = 18 * 2 * 32 * 8;   // conv2: 9 taps x 2 channel chunks
constexpr int kStemA_W1WideHalfs = 9 * 2 * 32 * 8;  // conv1 at 9..12 input channels: 9 chunks (1 tap x 16 channels each)
constexpr int kStemA_MaxChannels = 12;            // a pixel's bytes must lie inside one aligned 16-byte load

struct StemAArgs {
  const uint8_t* in;      // [N][H][W][C]; or, when in_ind is set, *in_ind + in_off (the caller's
                          // pointer read from device memory, so that a captured graph does not bake it in)
  const uint8_t* const* in_ind;
  size_t in_off;
  const _Float16* w1;     // pack_stem_a_w1
  const _Float16* w2;     // pack_stem_a_w2
  const float* shift1;    // [32]
  const float* shift2;    // [32]
  _Float16* out;          // conv2 output, C8 layout, 4 groups
  C8Geom og;
  int N, H, W, C;
  int OH1, OW1, OH2, OW2;
  int tiles_y, tiles_x;
  int total_tiles;        // N * tiles_y * tiles_x
  unsigned in_bytes;      // N*H*W*C (< 2^31)
  // Blank-row skipping (model.hip): blank_thr[n] = first conv2 output row of example n whose receptive
  // field holds only the zero rows below the pile-up; a tile that starts at or below it equals the all-blank image's
  // response at the same position (blank_src: ONE example in `og`'s geometry, produced by this kernel) and is copied
  // instead of loaded and multiplied -- bit-identical.  NULL = off.
  const int* blank_thr;
  const _Float16* blank_src;
  // blank_need[n] = conv2 rows of example n that stem_b's computed tiles read: blank tiles below that row are not
  // even copied -- nothing reads them (the tensor keeps whatever an earlier forward left there)
  const int* blank_need;
};

// ---- stem B: conv 3x3 'same' 32 -> 64, max-pool 3x3/2, conv 1x1 64 -> 80 -----------------
// One workgroup produces a PH x PW tile of the 1x1's output; the conv3 tile
// ((2PH+1) x (2PW+1) x 64) and its pooled image live in LDS only.
// Round 4: tiles of 6 x 9 pooled pixels on FOUR waves, two workgroups per CU (80 KB of LDS each): the
// eight waves of the round-2/3 kernel moved in lockstep, so every phase's tail (DMA wait, two
// barriers, the MFMA-free pool) was exposed -- 46 % MFMA-busy; two independent workgroups cover
// each other's.  -DDV_STEM_B_WIDE builds the old 12 x 9 / eight-wave shape for A/B runs.
#ifdef DV_STEM_B_WIDE
constexpr int kStemB_PH = 12, kStemB_PW = 9, kStemB_Waves = 8;
#else
constexpr int kStemB_PH = 6, kStemB_PW = 9, kStemB_Waves = 4;
#endif
constexpr int kStemB_W3Halfs = 2 * 18 * 2 * 32 * 8;  // [cout half][9 taps x 2 chunks]
constexpr int kStemB_W4Halfs = 3 * 4 * 2 * 32 * 8;   // [cout subtile][4 chunks]

struct StemBArgs {
  const _Float16* in;     // conv2 output (C8, 4 groups, halo >= 1)
  const _Float16* w3;
  const _Float16* w4;
  const float* shift3;    // [64]
  const float* shift4;    // [80] (+ padding to 96 readable)
  _Float16* out;          // 1x1 output, C8 layout, 10 groups
  C8Geom ig, og;
  int N;
  int OH3, OW3;           // conv3 output size (= its input size)
  int PH, PW;             // pooled size
  int Cout4;              // 80
  int tiles_y, tiles_x;
  int total_tiles;
  size_t in_bytes;
  unsigned in_img_bytes;
  // tuning aid (DV_STEM_PROF): per-phase shader-clock sums [block][8], or NULL
  unsigned long long* prof;
  // Blank-row skipping (see StemAArgs): blank_thr[n] = first POOLED row of example n that is blank-determined
  // (pooled row py reads conv2 rows 2py-1 .. 2py+3), blank_src = the all-blank image's 1x1 output (one example, `og`)
  const int* blank_thr;
  const _Float16* blank_src;
  const int* blank_need;  // rows of example n the 3x3 80->192's walk reads (see StemAArgs::blank_need)
};

void launch_stem_a(const StemAArgs& a, int blocks, hipStream_t stream);
void launch_stem_b(const StemBArgs& a, int blocks, hipStream_t stream);
int stem_a_blocks(int device);
int stem_b_blocks(int device);

// Host-side weight packing.  `w` is the layer's HWIO kernel, `inv` the folded BatchNorm
// scale 1/sqrt(var + eps) per output channel.
void pack_stem_a_w1(const float* w, const float* inv, int cin, _Float16* dst);
void pack_stem_a_w1_wide(const float* w, const float* inv, int cin, _Float16* dst);   // cin in 9..kStemA_MaxChannels
void pack_stem_a_w2(const float* w, const float* inv, _Float16* dst);
void pack_stem_b_w3(const float* w, const float* inv, _Float16* dst);
void pack_stem_b_w4(const float* w, const float* inv, int cout, _Float16* dst);

}  // namespace dv

#endif  // DV_STEM_FUSED_H_
