// chain.hip's interface: a CHAIN of stride-1, 'same'-padded one-dimensional convolutions
// (1 x k and k x 1, k <= 7: the factorised 7x7 branches of Inception-v3's 17x17 blocks,
// deepvariant/keras_modeling.py:268-274 / SURVEY.md App. B mixed4..mixed8) as ONE persistent
// launch whose intermediate tensors never leave the CU.
#ifndef DV_CHAIN_H_
#define DV_CHAIN_H_

#include "conv_common.h"

namespace dv {

constexpr int kChainMaxLayers = 4;
constexpr int kChainTilePx = 192;    // small maps (<= 96 pixels): G whole maps in 6 MFMA fragments, 1-D filters
constexpr int kChainTilePxBig = 256; // maps of up to 256 pixels (the 35x35 stage): 8 fragments, 3x3 / 5x5 filters
constexpr int kChainMaxTaps = 7;     // of a one-dimensional filter

struct ChainLayer {
  const _Float16* w;      // packed [chunk][tap][2 k-groups][cout_pad][8]
  const float* shift;     // folded BatchNorm shift, readable up to cout_pad
  int n_chunks;           // Cin / 16
  int cout, cout_pad;     // cout_pad: multiple of 32
  int kh, kw;             // filter (odd sizes, stride 1, 'same' padding): 1 x k / k x 1 (k <= 7), 3 x 3, 5 x 5
  unsigned slab_bytes;    // one chunk of weights: kh * kw * 2 * cout_pad * 16
};

struct ChainArgs {
  const _Float16* in;     // C8 tensor the first layer reads
  convk::TensorGeom ig;
  unsigned in_img_bytes;  // bytes of one example of `in`
  int N, G, h, w;         // G whole images of h x w pixels per tile (G*h*w <= tpx)
  int tpx;                // kChainTilePx or kChainTilePxBig
  int n_tiles;            // ceil(N / G); the last tile is shifted back to end at image N
  int n_layers;
  ChainLayer L[kChainMaxLayers];
  _Float16* out;          // the last layer's destination (a concat buffer)
  convk::TensorGeom og;
  int out_goff;           // first destination channel group
  unsigned act_bytes;     // LDS: activation tile [channel group][tpx][8]
  unsigned slot_bytes;    // LDS: one weight-slab slot (two of them follow the activations)
  // tuning aid (DV_CHAIN_PROF, eager launches): shader-clock sums [block][computing wave][8] =
  // wait at the chunk barriers, MFMA steps, wait at the layer barrier, LDS epilogue, HBM epilogue, set-up
  unsigned long long* prof;
};

size_t chain_lds_bytes(const ChainArgs& a);
void launch_chain(const ChainArgs& a, int blocks, hipStream_t stream);

}  // namespace dv

#endif  // DV_CHAIN_H_
