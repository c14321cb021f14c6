// calib.hip -- see calib.h.  Two fp32 pipelines (R: as given, E: the MFMA kernels' roundings) walk
// the classifier's op list on NHWC fp32 tensors; per layer the difference of their per-channel
// pre-activation means becomes that layer's shift correction.  Model-preparation code: it runs once
// per set of weights on a few hundred images, so the kernels are plain direct loops -- every
// reduction in a fixed order, so that the same weights and images always give the same corrections.
#include "calib.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>

#include "dv_internal.h"

namespace dv {
namespace {

constexpr int kPix = 8;          // output pixels per conv thread (their inputs are wave-uniform scalar loads)
constexpr int kZeroFloats = 4096;

__device__ __forceinline__ float round_f16(float v) { return static_cast<float>(static_cast<_Float16>(v)); }

__global__ void preprocess_u8(const uint8_t* in, float* out, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (static_cast<float>(in[i]) - 128.0f) / 128.0f;   // dv_utils.preprocess_images
}

// wq[k][co] = the weight the pipeline multiplies by: w * inv (R), fp16(w * inv) (E), or W_hi + W_lo (E, split).
__global__ void fold_weights(const float* w, const float* inv, float* wq, size_t k_total, int cout, int mode_e,
                             int split) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= k_total * cout) return;
  const float v = w[i] * inv[i % cout];
  if (!mode_e) {
    wq[i] = v;
    return;
  }
  const float hi = round_f16(v);
  wq[i] = split ? hi + round_f16(v - hi) : hi;
}

struct ConvGeom {
  int n, h, w, c;          // input tensor (NHWC, c = stored channels)
  int oh, ow, cout;
  int kh, kw, stride, pad_h, pad_w, cin;
};

// z[pix][co] = sum_{kh,kw,ci} x[n, ih, iw, ci] * wq[(kh*KW + kw)*cin + ci][co] (+ shift[co])
__global__ __launch_bounds__(64) void conv_direct(const float* __restrict__ in, const float* __restrict__ wq,
                                                  const float* __restrict__ shift, const float* __restrict__ zeros,
                                                  float* __restrict__ z, ConvGeom g, int m_total) {
  const int co = blockIdx.y * 64 + threadIdx.x;
  const int m0 = blockIdx.x * kPix;
  float acc[kPix];
  int pn[kPix], py[kPix], px[kPix];
#pragma unroll
  for (int p = 0; p < kPix; ++p) {
    acc[p] = 0.f;
    const int m = min(m0 + p, m_total - 1);
    pn[p] = m / (g.oh * g.ow);
    const int r = m - pn[p] * (g.oh * g.ow);
    py[p] = r / g.ow;
    px[p] = r - py[p] * g.ow;
  }
  const bool live = co < g.cout;
  const float* wcol = wq + (live ? co : 0);
  for (int kh = 0; kh < g.kh; ++kh) {
    for (int kw = 0; kw < g.kw; ++kw) {
      const float* xp[kPix];
#pragma unroll
      for (int p = 0; p < kPix; ++p) {
        const int iy = py[p] * g.stride - g.pad_h + kh, ix = px[p] * g.stride - g.pad_w + kw;
        const bool ok = iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
        xp[p] = ok ? in + ((static_cast<size_t>(pn[p]) * g.h + iy) * g.w + ix) * g.c : zeros;
      }
      const float* wk = wcol + static_cast<size_t>(kh * g.kw + kw) * g.cin * g.cout;
      for (int ci = 0; ci < g.cin; ++ci) {
        const float wv = wk[static_cast<size_t>(ci) * g.cout];
#pragma unroll
        for (int p = 0; p < kPix; ++p) acc[p] = fmaf(xp[p][ci], wv, acc[p]);
      }
    }
  }
  if (!live) return;
  const float sh = shift != nullptr ? shift[co] : 0.f;
#pragma unroll
  for (int p = 0; p < kPix; ++p) {
    if (m0 + p < m_total) z[static_cast<size_t>(m0 + p) * g.cout + co] = acc[p] + sh;
  }
}

// 3x3 average over the cells inside the map (Keras AveragePooling2D 'same'), + shift: z[pix][c]
__global__ void avgpool_direct(const float* __restrict__ in, const float* __restrict__ shift, float* __restrict__ z,
                               int n, int h, int w, int c_total, int c) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<size_t>(n) * h * w * c) return;
  const int ch = static_cast<int>(i % c);
  const size_t pix = i / c;
  const int x = static_cast<int>(pix % w), y = static_cast<int>((pix / w) % h);
  const size_t img = pix / (static_cast<size_t>(w) * h);
  float s = 0.f;
  int cnt = 0;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
      s += in[((img * h + yy) * w + xx) * c_total + ch];
      ++cnt;
    }
  z[i] = s / static_cast<float>(cnt) + (shift != nullptr ? shift[ch] : 0.f);
}

// 3x3 / stride 2 'valid' max-pool of channels [0, c) of `in` into channels [coff, coff + c) of `out`
__global__ void maxpool_direct(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w,
                               int c_in_total, int c, int oh, int ow, int c_out_total, int coff) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<size_t>(n) * oh * ow * c) return;
  const int ch = static_cast<int>(i % c);
  const size_t pix = i / c;
  const int x = static_cast<int>(pix % ow), y = static_cast<int>((pix / ow) % oh);
  const size_t img = pix / (static_cast<size_t>(ow) * oh);
  float best = -INFINITY;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx)
      best = fmaxf(best, in[((img * h + 2 * y + dy) * w + 2 * x + dx) * c_in_total + ch]);
  out[pix * c_out_total + coff + ch] = best;
}

// Per-channel sums of z[m][c] in a fixed order: block (channel tile of 32, segment s) -> partial[s][c].
__global__ __launch_bounds__(256) void channel_partials(const float* __restrict__ z, double* __restrict__ partial,
                                                        int m_total, int c_total, int seg_len) {
  __shared__ double sm[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), lane = threadIdx.x >> 5;
  const int m_lo = blockIdx.y * seg_len, m_hi = min(m_total, m_lo + seg_len);
  double s = 0.0;
  if (c < c_total) {
    for (int m = m_lo + lane; m < m_hi; m += 8) s += static_cast<double>(z[static_cast<size_t>(m) * c_total + c]);
  }
  sm[lane][threadIdx.x & 31] = s;
  __syncthreads();
  if (lane == 0 && c < c_total) {
    double t = 0.0;
    for (int j = 0; j < 8; ++j) t += sm[j][threadIdx.x & 31];
    partial[static_cast<size_t>(blockIdx.y) * c_total + c] = t;
  }
}

__global__ void channel_means(const double* __restrict__ partial, double* __restrict__ mean, int segs, int c_total,
                              int m_total) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= c_total) return;
  double t = 0.0;
  for (int s = 0; s < segs; ++s) t += partial[static_cast<size_t>(s) * c_total + c];
  mean[c] = t / static_cast<double>(m_total);
}

__global__ void mean_difference(const double* __restrict__ mean_e, const double* __restrict__ mean_r,
                                float* __restrict__ corr, int c_total) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < c_total) corr[c] = static_cast<float>(mean_e[c] - mean_r[c]);
}

// out[pix][coff + c] = f(z[pix][c] - corr[c]); f = ReLU (optional), then the fp16 rounding of a stored activation (E)
__global__ void finalize(const float* __restrict__ z, const float* __restrict__ corr, float* __restrict__ out,
                         size_t m_total, int c, int c_out_total, int coff, int relu, int round16) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m_total * c) return;
  const int ch = static_cast<int>(i % c);
  float v = z[i] - (corr != nullptr ? corr[ch] : 0.f);
  if (round16) v = round_f16(v);   // the kernels round first and clamp the fp16 value; same result
  if (relu) v = fmaxf(v, 0.f);
  out[(i / c) * c_out_total + coff + ch] = v;
}

// global average pool -> Dense: logits[n][k]
__global__ __launch_bounds__(256) void head_logits(const float* __restrict__ feat, const float* __restrict__ dw,
                                                   const float* __restrict__ db, float* __restrict__ logits,
                                                   int pixels, int c_total, int classes) {
  __shared__ float sm[256];
  const int n = blockIdx.x;
  for (int k = 0; k < classes; ++k) {
    float s = 0.f;
    for (int c = threadIdx.x; c < c_total; c += 256) {
      float f = 0.f;
      for (int p = 0; p < pixels; ++p) f += feat[(static_cast<size_t>(n) * pixels + p) * c_total + c];
      s += f / static_cast<float>(pixels) * dw[static_cast<size_t>(c) * classes + k];
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
      if (threadIdx.x < d) sm[threadIdx.x] += sm[threadIdx.x + d];
      __syncthreads();
    }
    if (threadIdx.x == 0) logits[static_cast<size_t>(n) * classes + k] = sm[0] + db[k];
    __syncthreads();
  }
}

inline unsigned blocks_for(size_t n, int threads) { return static_cast<unsigned>((n + threads - 1) / threads); }

struct Scratch {
  DeviceBuffer weights, inv, shift, corr, zeros, wq, z, tmp_in, tmp_out, partial, mean[2], logits[2];
  std::vector<DeviceBuffer> act[2];
  ~Scratch() {
    (void)hipDeviceSynchronize();   // an early error return must not free buffers that queued kernels still use
    for (DeviceBuffer* b : {&weights, &inv, &shift, &corr, &zeros, &wq, &z, &tmp_in, &tmp_out, &partial, &mean[0],
                            &mean[1], &logits[0], &logits[1]}) {
      b->release();
    }
    for (auto& v : act)
      for (DeviceBuffer& b : v) b.release();
  }
};

}  // namespace

int run_calibration(const CalibPlan& plan, int device, const float* weights, int64_t n_weights,
                    const std::vector<float>& shift, const uint8_t* images, int n, std::vector<float>* corr,
                    std::vector<float>* dense_corr, const CalibProbe* probe) {
  const bool fixed_corr = probe != nullptr && probe->corr_in != nullptr;
  const bool skip_r = fixed_corr && probe->skip_r;
  const int first_pipe = skip_r ? 1 : 0;
  if (n < 1 || plan.ops.empty() || plan.bufs.empty() || plan.feat_buf < 0) {
    return fail(DV_ERR_INVALID_ARGUMENT, "calibration: empty plan or batch");
  }
  DV_HIP_CHECK(hipSetDevice(device));
  Scratch s;
  // per-channel fold factors 1 / sqrt(var + eps), computed on the host exactly as dv_model_load_weights does
  std::vector<float> inv(shift.size(), 0.f);
  size_t max_w = 0, max_z = 0, max_tmp = 0;
  int max_c = plan.num_classes;
  for (const CalibOp& op : plan.ops) {
    if (op.type == 0) {
      const size_t wn = static_cast<size_t>(op.kh) * op.kw * op.cin * op.cout;
      const float* var = weights + op.w_off + wn + 2 * static_cast<size_t>(op.cout);
      if (op.w_off < 0 || op.w_off + static_cast<int64_t>(wn) + 3 * op.cout > n_weights || op.shift_off < 0 ||
          op.shift_off + op.cout > static_cast<int64_t>(shift.size())) {
        return fail(DV_ERR_INVALID_ARGUMENT, "calibration: op outside the weight / shift arrays");
      }
      for (int co = 0; co < op.cout; ++co) inv[op.shift_off + co] = 1.0f / std::sqrt(var[co] + 1e-3f);
      max_w = std::max(max_w, wn);
    }
    const CalibBuf& ib = plan.bufs[op.in_buf];
    const CalibBuf& ob = plan.bufs[op.out_buf];
    int ih = ib.h, iw = ib.w;
    if (op.type == 0 && op.pool_in) {
      ih = (ih - 3) / 2 + 1;
      iw = (iw - 3) / 2 + 1;
      max_tmp = std::max(max_tmp, static_cast<size_t>(ih) * iw * ib.c);
    }
    int oh = ob.h, ow = ob.w;
    if (op.type == 0) {
      oh = (ih + 2 * op.pad_h - op.kh) / op.stride + 1;
      ow = (iw + 2 * op.pad_w - op.kw) / op.stride + 1;
      if (op.pool_out) max_tmp = std::max(max_tmp, static_cast<size_t>(oh) * ow * op.cout);
    }
    max_z = std::max(max_z, static_cast<size_t>(oh) * ow * op.cout);
    max_c = std::max(max_c, op.cout);
  }
  int rc = DV_OK;
  auto need = [&](DeviceBuffer& b, size_t bytes) {
    if (rc == DV_OK) rc = b.reserve(std::max<size_t>(bytes, 16));
  };
  need(s.weights, static_cast<size_t>(n_weights) * 4);
  need(s.inv, shift.size() * 4);
  need(s.shift, shift.size() * 4);
  need(s.corr, shift.size() * 4);
  need(s.zeros, kZeroFloats * 4);
  need(s.wq, max_w * 4);
  need(s.z, max_z * n * 4);
  need(s.tmp_in, max_tmp * n * 4);
  need(s.tmp_out, max_tmp * n * 4);
  constexpr int kMaxSegs = 256;
  need(s.partial, static_cast<size_t>(kMaxSegs) * max_c * 8);
  for (int p = 0; p < 2; ++p) {
    need(s.mean[p], static_cast<size_t>(max_c) * 8);
    need(s.logits[p], static_cast<size_t>(n) * plan.num_classes * 4);
    s.act[p].resize(plan.bufs.size());
    for (size_t b = 0; b < plan.bufs.size(); ++b) {
      const CalibBuf& d = plan.bufs[b];
      if (d.c > kZeroFloats) return fail(DV_ERR_INVALID_ARGUMENT, "calibration: tensor wider than the zero row");
      // the two pipelines read the same input image
      if (b == 0 && p == 1) continue;
      if (b != 0 && p == 0 && skip_r) continue;
      need(s.act[p][b], static_cast<size_t>(n) * d.h * d.w * d.c * 4);
    }
  }
  if (rc != DV_OK) return rc;
  DV_HIP_CHECK(hipMemcpy(s.weights.ptr, weights, static_cast<size_t>(n_weights) * 4, hipMemcpyHostToDevice));
  DV_HIP_CHECK(hipMemcpy(s.inv.ptr, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
  DV_HIP_CHECK(hipMemcpy(s.shift.ptr, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
  if (fixed_corr) {
    DV_HIP_CHECK(hipMemcpy(s.corr.ptr, probe->corr_in, shift.size() * 4, hipMemcpyHostToDevice));
  } else {
    DV_HIP_CHECK(hipMemset(s.corr.ptr, 0, shift.size() * 4));
  }
  DV_HIP_CHECK(hipMemset(s.zeros.ptr, 0, kZeroFloats * 4));
  hipStream_t st = nullptr;
  auto F = [](DeviceBuffer& b) { return static_cast<float*>(b.ptr); };
  auto act = [&](int pipe, int buf) { return F(s.act[buf == 0 ? 0 : pipe][buf]); };
  {
    const CalibBuf& d = plan.bufs[0];
    const size_t cnt = static_cast<size_t>(n) * d.h * d.w * d.c;
    hipLaunchKernelGGL(preprocess_u8, dim3(blocks_for(cnt, 256)), dim3(256), 0, st, images, act(0, 0), cnt);
  }
  // per-channel means of z[m_total][c] into s.mean[pipe]
  auto channel_mean = [&](int pipe, const float* z, int m_total, int c) {
    const int segs = std::max(1, std::min(kMaxSegs, (m_total + 63) / 64));
    const int seg_len = (m_total + segs - 1) / segs;
    hipLaunchKernelGGL(channel_partials, dim3((c + 31) / 32, segs), dim3(256), 0, st, z,
                       static_cast<double*>(s.partial.ptr), m_total, c, seg_len);
    hipLaunchKernelGGL(channel_means, dim3((c + 63) / 64), dim3(64), 0, st,
                       static_cast<const double*>(s.partial.ptr), static_cast<double*>(s.mean[pipe].ptr), segs, c,
                       m_total);
  };
  auto take_corr = [&](int64_t shift_off, int c) {
    if (fixed_corr) return;   // the probe's corrections are already in s.corr
    hipLaunchKernelGGL(mean_difference, dim3((c + 63) / 64), dim3(64), 0, st,
                       static_cast<const double*>(s.mean[1].ptr), static_cast<const double*>(s.mean[0].ptr),
                       F(s.corr) + shift_off, c);
  };
  for (const CalibOp& op : plan.ops) {
    const CalibBuf& ib = plan.bufs[op.in_buf];
    const CalibBuf& ob = plan.bufs[op.out_buf];
    for (int pipe = first_pipe; pipe < 2; ++pipe) {
      const float* in = act(pipe, op.in_buf);
      const int round16 = pipe == 1 && !op.keep_f32;
      float* out = act(pipe, op.out_buf);
      if (op.type == 1) {
        const int oh = (ib.h - 3) / 2 + 1, ow = (ib.w - 3) / 2 + 1;
        const size_t cnt = static_cast<size_t>(n) * oh * ow * ib.c;
        hipLaunchKernelGGL(maxpool_direct, dim3(blocks_for(cnt, 256)), dim3(256), 0, st, in, out, n, ib.h, ib.w, ib.c,
                           ib.c, oh, ow, ob.c, op.out_coff);
        continue;
      }
      if (op.type == 2) {
        const int m_total = n * ib.h * ib.w;
        const size_t cnt = static_cast<size_t>(m_total) * ib.c;
        const float* sh = op.shift_relu ? F(s.shift) + op.shift_off : nullptr;
        hipLaunchKernelGGL(avgpool_direct, dim3(blocks_for(cnt, 256)), dim3(256), 0, st, in, sh, F(s.z), n, ib.h, ib.w,
                           ib.c, ib.c);
        const float* cr = nullptr;
        if (op.shift_relu) {
          if (!fixed_corr) channel_mean(pipe, F(s.z), m_total, ib.c);
          if (pipe == 1) {
            take_corr(op.shift_off, ib.c);
            cr = F(s.corr) + op.shift_off;
          }
        }
        hipLaunchKernelGGL(finalize, dim3(blocks_for(cnt, 256)), dim3(256), 0, st, F(s.z), cr, out,
                           static_cast<size_t>(m_total), ib.c, ob.c, op.out_coff, op.shift_relu, round16);
        continue;
      }
      // convolution
      ConvGeom g{};
      g.n = n;
      g.h = ib.h;
      g.w = ib.w;
      g.c = ib.c;
      if (op.pool_in) {
        const int ph = (ib.h - 3) / 2 + 1, pw = (ib.w - 3) / 2 + 1;
        const size_t cnt = static_cast<size_t>(n) * ph * pw * ib.c;
        hipLaunchKernelGGL(maxpool_direct, dim3(blocks_for(cnt, 256)), dim3(256), 0, st, in, F(s.tmp_in), n, ib.h, ib.w,
                           ib.c, ib.c, ph, pw, ib.c, 0);
        in = F(s.tmp_in);
        g.h = ph;
        g.w = pw;
      }
      g.kh = op.kh;
      g.kw = op.kw;
      g.stride = op.stride;
      g.pad_h = op.pad_h;
      g.pad_w = op.pad_w;
      g.cin = op.cin;
      g.cout = op.cout;
      g.oh = (g.h + 2 * op.pad_h - op.kh) / op.stride + 1;
      g.ow = (g.w + 2 * op.pad_w - op.kw) / op.stride + 1;
      if (g.cin > g.c) return fail(DV_ERR_INVALID_ARGUMENT, "calibration: kernel wider than its input tensor");
      const size_t k_total = static_cast<size_t>(op.kh) * op.kw * op.cin;
      hipLaunchKernelGGL(fold_weights, dim3(blocks_for(k_total * op.cout, 256)), dim3(256), 0, st,
                         F(s.weights) + op.w_off, F(s.inv) + op.shift_off, F(s.wq), k_total, op.cout,
                         pipe == 1 && !(probe != nullptr && probe->weights_f32) ? 1 : 0,
                         op.split);
      const int m_total = n * g.oh * g.ow;
      hipLaunchKernelGGL(conv_direct, dim3((m_total + kPix - 1) / kPix, (op.cout + 63) / 64), dim3(64), 0, st, in,
                         F(s.wq), op.raw ? nullptr : F(s.shift) + op.shift_off, F(s.zeros), F(s.z), g, m_total);
      const size_t cnt = static_cast<size_t>(m_total) * op.cout;
      const float* cr = nullptr;
      if (!op.raw) {
        if (!fixed_corr) channel_mean(pipe, F(s.z), m_total, op.cout);
        if (pipe == 1) {
          take_corr(op.shift_off, op.cout);
          cr = F(s.corr) + op.shift_off;
        }
      }
      if (op.pool_out) {
        hipLaunchKernelGGL(finalize, dim3(blocks_for(cnt, 256)), dim3(256), 0, st, F(s.z), cr, F(s.tmp_out),
                           static_cast<size_t>(m_total), op.cout, op.cout, 0, !op.raw, round16);
        const int ph = (g.oh - 3) / 2 + 1, pw = (g.ow - 3) / 2 + 1;
        const size_t pc = static_cast<size_t>(n) * ph * pw * op.cout;
        hipLaunchKernelGGL(maxpool_direct, dim3(blocks_for(pc, 256)), dim3(256), 0, st, F(s.tmp_out), out, n, g.oh,
                           g.ow, op.cout, op.cout, ph, pw, ob.c, op.out_coff);
      } else {
        hipLaunchKernelGGL(finalize, dim3(blocks_for(cnt, 256)), dim3(256), 0, st, F(s.z), cr, out,
                           static_cast<size_t>(m_total), op.cout, ob.c, op.out_coff, !op.raw, round16);
      }
    }
    DV_HIP_CHECK(hipGetLastError());
  }
  // head: the mean logit difference goes to the Dense bias
  const CalibBuf& fb = plan.bufs[plan.feat_buf];
  const float* dw = F(s.weights) + plan.dense_off;
  for (int pipe = first_pipe; pipe < 2; ++pipe) {
    hipLaunchKernelGGL(head_logits, dim3(n), dim3(256), 0, st, act(pipe, plan.feat_buf), dw,
                       dw + static_cast<size_t>(fb.c) * plan.num_classes, F(s.logits[pipe]), fb.h * fb.w, fb.c,
                       plan.num_classes);
    channel_mean(pipe, F(s.logits[pipe]), n, plan.num_classes);
  }
  DV_HIP_CHECK(hipGetLastError());
  DV_HIP_CHECK(hipDeviceSynchronize());
  corr->assign(shift.size(), 0.f);
  DV_HIP_CHECK(hipMemcpy(corr->data(), s.corr.ptr, shift.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> me(plan.num_classes), mr(plan.num_classes);
  DV_HIP_CHECK(hipMemcpy(me.data(), s.mean[1].ptr, me.size() * 8, hipMemcpyDeviceToHost));
  DV_HIP_CHECK(hipMemcpy(mr.data(), s.mean[0].ptr, mr.size() * 8, hipMemcpyDeviceToHost));
  dense_corr->resize(plan.num_classes);
  for (int k = 0; k < plan.num_classes; ++k) {
    (*dense_corr)[k] = fixed_corr ? (probe->dense_corr_in ? probe->dense_corr_in[k] : 0.f) : static_cast<float>(me[k] - mr[k]);
  }
  if (probe != nullptr) {
    const size_t cnt = static_cast<size_t>(n) * plan.num_classes;
    if (probe->logits_r != nullptr && !skip_r) {
      DV_HIP_CHECK(hipMemcpy(probe->logits_r, s.logits[0].ptr, cnt * 4, hipMemcpyDeviceToHost));
    }
    if (probe->logits_e != nullptr) {
      DV_HIP_CHECK(hipMemcpy(probe->logits_e, s.logits[1].ptr, cnt * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < cnt; ++i) probe->logits_e[i] -= (*dense_corr)[i % plan.num_classes];
    }
  }
  for (float v : *corr) {
    if (!std::isfinite(v)) return fail(DV_ERR_BAD_INPUT, "calibration produced a non-finite correction");
  }
  return DV_OK;
}

}  // namespace dv
