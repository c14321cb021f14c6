// imgconv.hip's interface: stride-1 convolutions over tiles of whole feature maps
// (both MFMA operands staged in LDS by DMA, persistent workgroups).
#ifndef DV_IMGCONV_H_
#define DV_IMGCONV_H_

#include "conv_common.h"

namespace dv {

struct ImgConvArgs {
  convk::ConvArgs c;      // input, branches, geometry; c.w = weights packed by pack_imgconv
  int G;                  // images per tile
  int P;                  // output pixels per image (OH * OW)
  int RP, CP;             // rows / columns of one image's input patch (OH + KH - 1, OW + KW - 1)
  int n_img_tiles;        // ceil(N / G); the last tile is shifted back to end at image N
  int n_cout_tiles;       // cout tiles of NB*32
  int n_steps;            // ceil(Cin / 16 / KC)
  int act_pieces;         // 16-byte pieces of one activation slab = KC * 2 * G * RP * CP
  int act_slab_bytes;     // act_pieces * 16 rounded up to the DMA granule (1 KB per wave)
  int plane_pieces;       // G * RP * CP: pieces of one 8-channel plane of the slab
};

// Kernel shapes that exist (template instances): filter taps x NB.
bool imgconv_supported(int kh, int kw, int nb);
int imgconv_kc(int kh, int kw);                    // channel chunks per pipeline step
size_t imgconv_wslab_halfs(int kh, int kw, int nb);  // halfs of one (cout tile, step) weight slab
size_t imgconv_lds_bytes(const ImgConvArgs& a, int nb);
int imgconv_threads();
void launch_imgconv(const ImgConvArgs& a, int nb, int blocks, hipStream_t stream);

}  // namespace dv

#endif  // DV_IMGCONV_H_
