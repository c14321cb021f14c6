// calib.hip -- shift calibration of the fp16 classifier (dv_model_calibrate, include/dvhip.h).
//
// The CNN multiplies fp16 weights by fp16 activations (fp32 accumulate).  Both roundings are
// unbiased per element but not per CHANNEL: W16 - W is one fixed draw that multiplies activations
// with a large positive mean (post-ReLU maps, the constant rows below a pile-up), and the rounding
// of a stored activation is the same number wherever the map is constant.  The per-channel MEAN of
// that error is removed at no run-time cost by moving each layer's fp32 shift (the folded BatchNorm
// beta): this file measures it on a small calibration batch by running two fp32 pipelines side by
// side on the GPU, layer by layer --
//   R: the layer exactly as given (fp32 weights, fp32 activations), and
//   E: the arithmetic of the MFMA kernels (BN-folded weights rounded to fp16 -- or W_hi + W_lo for
//      split layers --, every stored activation rounded to fp16, fp32 sums)
// -- and setting corr[c] = mean_E(z[c]) - mean_R(z[c]) over all calibration pixels of the layer's
// pre-activation z (conv + shift, before ReLU).  E applies corr of layer L before it feeds layer
// L + 1, so the corrections compose (sequential mean matching).  Plain direct convolutions on NHWC
// fp32: ~0.5 TFLOP for 128 pile-ups, a fraction of a second, once per set of weights.
#ifndef DV_CALIB_H_
#define DV_CALIB_H_

#include <cstdint>
#include <vector>

namespace dv {

struct CalibOp {
  int type;                   // 0 conv, 1 max-pool 3x3/2 'valid', 2 avg-pool 3x3/1 'same' (divisor = cells inside)
  int in_buf, out_buf, out_coff;
  int pool_in;                // conv: its input is max-pooled (3x3/2) first
  int pool_out;               // conv: its output is max-pooled (3x3/2) before it is stored
  int kh, kw, stride, pad_h, pad_w;
  int cin, cout;              // cin = the kernel's input channels (HWIO)
  int64_t w_off;              // conv: floats into the flat weights (kernel, beta, moving mean, moving variance)
  int raw;                    // conv: no shift, no ReLU (the avg-pool behind it applies them)
  int shift_relu;             // avg-pool: + shift, ReLU after averaging
  int64_t shift_off;          // where this op's shifts (and corrections) live in the model's shift array; -1 none
  int split;                  // conv: the product multiplies W_hi + W_lo (two fp16 numbers) instead of W_hi
  int keep_f32;               // E keeps this op's output in fp32: the product stores it wider than fp16 (an fp32 tensor --
                              // pooled projections' raw outputs, the last block's outputs that feed the global pool -- or
                              // hi + lo fp16 pieces) or the probe below asks what that would buy
};

struct CalibBuf {
  int h, w, c;                // as stored (after a producer's pool_out), full concat width
};

struct CalibPlan {
  std::vector<CalibOp> ops;       // topological order
  std::vector<CalibBuf> bufs;     // bufs[0] = the input image (c = real channels)
  int feat_buf = -1;              // input of the head: global average pool -> Dense -> softmax
  int num_classes = 3;
  int64_t dense_off = 0;          // floats into the flat weights: Dense kernel [feat_c][classes], then bias
};

// Optional probe (dv_model_probe_rounding, a diagnostic): fixed corrections instead of measured ones, the logits of both
// pipelines, R skipped when only E is wanted.
struct CalibProbe {
  const float* corr_in = nullptr;        // same indexing as `shift`; nullptr = measure (the calibration proper)
  const float* dense_corr_in = nullptr;  // [num_classes], with corr_in
  float* logits_r = nullptr;             // HOST [n][num_classes], optional
  float* logits_e = nullptr;             // HOST [n][num_classes], optional (corrections applied, Dense bias included)
  bool skip_r = false;                   // needs corr_in
  bool weights_f32 = false;              // E multiplies the fp32 weights (isolates the activation roundings)
};

// images: DEVICE pointer, uint8 [n][h][w][c] of bufs[0].  shift: the model's host shift array (as uploaded).
// On success corr (same indexing as shift; zero where no correction applies) and dense_corr[num_classes]
// hold mean_E - mean_R; the caller subtracts them from the shifts / the Dense bias.
int run_calibration(const CalibPlan& plan, int device, const float* weights, int64_t n_weights,
                    const std::vector<float>& shift, const uint8_t* images, int n,
                    std::vector<float>* corr, std::vector<float>* dense_corr, const CalibProbe* probe = nullptr);

}  // namespace dv

#endif  // DV_CALIB_H_
