// misc_kernels.h -- GetCostVolume and DisparityRegression for gfx950.
// Reference: libs/GANet/modules/GANet.py:114-148 (a Python loop of 2*D slice
// copies, resp. arange upload + repeat + mul + sum).  Here: one streaming kernel
// each for forward and backward; pure HBM-bandwidth work, lanes along W.
#pragma once
#include "ga_common.h"

namespace ga {

// cost[n,c,i,h,w] = x[n,c,h,w] (c<C) | y[n,c-C,h,w-i] (c>=C) for w>=i, else 0
__global__ void __launch_bounds__(256)
cost_volume_fwd(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ cost,
                int N, int C, int Dn, int H, int W)
{
  const i64 total = (i64)N * 2 * C * Dn * H * W;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int w = (int)(o % W);
    i64 r = o / W;
    const int h = (int)(r % H); r /= H;
    const int i = (int)(r % Dn); r /= Dn;
    const int c = (int)(r % (2 * C));
    const int n = (int)(r / (2 * C));
    float v = 0.f;
    if (w >= i) {
      if (c < C) v = x[(((i64)n * C + c) * H + h) * W + w];
      else v = y[(((i64)n * C + (c - C)) * H + h) * W + (w - i)];
    }
    cost[o] = v;
  }
}

// the same with four consecutive columns per lane (W % 4 == 0, 16-byte aligned buffers): one index decomposition per
// 16-byte store instead of per element (0.23 -> 0.06 ms at [1,64,65,80,208]); the shifted right-image samples are four
// scalar loads (they hit L1/L2: the 2*C source planes are read Dn times)
static __global__ void __launch_bounds__(256)
cost_volume_fwd4(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ cost,
                 int N, int C, int Dn, int H, int W)
{
  const int W4 = W >> 2;
  const i64 total = (i64)N * 2 * C * Dn * H * W4;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
    const int w = (int)(q % W4) << 2;
    i64 r = q / W4;
    const int h = (int)(r % H); r /= H;
    const int i = (int)(r % Dn); r /= Dn;
    const int c = (int)(r % (2 * C));
    const int n = (int)(r / (2 * C));
    f4 v;
    if (c < C) {
      v = *reinterpret_cast<const f4 *>(x + (((i64)n * C + c) * H + h) * W + w);
      if (w < i) v.x = 0.f;
      if (w + 1 < i) v.y = 0.f;
      if (w + 2 < i) v.z = 0.f;
      if (w + 3 < i) v.w = 0.f;
    } else {
      const float *yr = y + (((i64)n * C + (c - C)) * H + h) * W;
      const int s0 = w - i;                        // source column of the first element (may be negative)
      const float a0 = yr[s0 > 0 ? s0 : 0], a1 = yr[s0 + 1 > 0 ? s0 + 1 : 0];
      const float a2 = yr[s0 + 2 > 0 ? s0 + 2 : 0], a3 = yr[s0 + 3 > 0 ? s0 + 3 : 0];
      v.x = s0 >= 0 ? a0 : 0.f;
      v.y = s0 + 1 >= 0 ? a1 : 0.f;
      v.z = s0 + 2 >= 0 ? a2 : 0.f;
      v.w = s0 + 3 >= 0 ? a3 : 0.f;
    }
    *reinterpret_cast<f4 *>(cost + (q << 2)) = v;
  }
}

// adjoint: gx[n,c,h,w] = sum_{i<=w} g[n,c,i,h,w];  gy[n,c,h,w] = sum_{i: w+i<W} g[n,C+c,i,h,w+i]
__global__ void __launch_bounds__(256)
cost_volume_bwd(const float *__restrict__ gcost, float *__restrict__ gx, float *__restrict__ gy,
                int N, int C, int Dn, int H, int W)
{
  const i64 total = (i64)N * C * H * W;
  const i64 HW = (i64)H * W;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int w = (int)(o % W);
    i64 r = o / W;
    const int h = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const float *gl = gcost + (((i64)n * 2 * C + c) * Dn) * HW + (i64)h * W;
    const float *gr = gcost + (((i64)n * 2 * C + C + c) * Dn) * HW + (i64)h * W;
    // unconditional loads (column clamped into the row, value masked) in groups of 8: a load under a condition is
    // followed by s_waitcnt vmcnt(0), one memory round trip per disparity
    float sx = 0.f, sy = 0.f;
    for (int i0 = 0; i0 < Dn; i0 += 8) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u < Dn ? i0 + u : Dn - 1;
        const int wr = w + i < W ? w + i : W - 1;
        a[u] = gl[(i64)i * HW + w];
        b[u] = gr[(i64)i * HW + wr];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u;
        if (i < Dn && i <= w) sx += a[u];
        if (i < Dn && w + i < W) sy += b[u];
      }
    }
    gx[o] = sx;
    gy[o] = sy;
  }
}

// out[n,h,w] = sum_d d * x[n,d,h,w]
__global__ void __launch_bounds__(256)
disp_regression_fwd(const float *__restrict__ x, float *__restrict__ out, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / HW, pix = o - n * HW;
    const float *xp = x + n * Dn * HW + pix;
    float acc = 0.f;
    for (int d0 = 0; d0 < Dn; d0 += 8) {       // 8 loads in flight per lane (same summation order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = xp[(i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) acc = fmaf(v[u], (float)(d0 + u), acc);
    }
    out[o] = acc;
  }
}

// gx[n,d,h,w] = d * gout[n,h,w]
__global__ void __launch_bounds__(256)
disp_regression_bwd(const float *__restrict__ gout, float *__restrict__ gx, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * Dn * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 pix = o % HW;
    const i64 r = o / HW;
    const int d = (int)(r % Dn);
    const i64 n = r / Dn;
    gx[o] = gout[n * HW + pix] * (float)d;
  }
}

// the same with four pixels per lane marching over d (HW % 4 == 0, 16-byte aligned buffers): gout is read once,
// every store is 16 bytes and no index is divided per element (0.10 -> 0.03 ms at [1,193,240,624])
static __global__ void __launch_bounds__(256)
disp_regression_bwd4(const float *__restrict__ gout, float *__restrict__ gx, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * (HW >> 2);
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
    const i64 n = q / (HW >> 2), pix = (q - n * (HW >> 2)) << 2;
    const f4 g = *reinterpret_cast<const f4 *>(gout + n * HW + pix);
    float *gp = gx + n * Dn * HW + pix;
    for (int d = 0; d < Dn; d++) {
      const float df = (float)d;
      f4 r;
      r.x = g.x * df; r.y = g.y * df; r.z = g.z * df; r.w = g.w * df;
      *reinterpret_cast<f4 *>(gp + (i64)d * HW) = r;
    }
  }
}

// =====================================================================================
// Callers' pre-/post-ops folded into single kernels.  In the reference they
// are chains of stock PyTorch kernels around the guided-aggregation ops:
//   models/GANet_deep.py:263-268  split + view + F.normalize(p=1, dim=2) of the SGA guidance (x4)
//   models/GANet_deep.py:235      F.normalize(p=1, dim=1) of the LGA filters
//   models/GANet_deep.py:246-247  F.normalize(p=1, dim=1) of the aggregated volume + DisparityRegression
// F.normalize(x, p=1, dim) = x / max(sum_dim |x|, eps), eps = 1e-12 (torch.nn.functional.normalize).
// =====================================================================================
constexpr float GA_NORM_EPS = 1e-12f;

struct NormPtrs {
  float *y[4];
  const float *gy[4];
};

// x [N][G][C][K][HW] -> y_g [N][C][K][HW], g < G <= 4: L1-normalise over K (the SGABlock split: G = 4,
// K = 5; LGA filters: G = C = 1, K = 3(2r+1)^2).  One lane per (n,g,c,pixel); the K values of a lane
// stay in registers when K is a template constant, otherwise the lane walks its column twice.
template <int KT>
__global__ void __launch_bounds__(256)
l1norm_fwd(const float *__restrict__ x, NormPtrs p, int N, int G, int C, int Krt, i64 HW)
{
  const int K = KT > 0 ? KT : Krt;
  const i64 total = (i64)N * G * C * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const i64 o = idx / HW, pix = idx - o * HW;
    const int c = (int)(o % C);
    const int g = (int)((o / C) % G);
    const i64 n = o / ((i64)C * G);
    const float *xin = x + o * K * HW + pix;
    float *yb = g == 0 ? p.y[0] : g == 1 ? p.y[1] : g == 2 ? p.y[2] : p.y[3];
    float *yo = yb + (n * C + c) * K * HW + pix;
    if (KT > 0) {
      float v[KT > 0 ? KT : 1];
      float sa = 0.f;
#pragma unroll
      for (int t = 0; t < KT; t++) { v[t] = xin[(i64)t * HW]; sa += fabsf(v[t]); }
      const float sden = fmaxf(sa, GA_NORM_EPS);
#pragma unroll
      for (int t = 0; t < KT; t++) yo[(i64)t * HW] = v[t] / sden;
    } else {
      float sa = 0.f;
      for (int t = 0; t < K; t++) sa += fabsf(xin[(i64)t * HW]);
      const float sden = fmaxf(sa, GA_NORM_EPS);
      for (int t = 0; t < K; t++) yo[(i64)t * HW] = xin[(i64)t * HW] / sden;
    }
  }
}

// adjoint: gx_t = (gy_t - sgn(x_t) * sum_u gy_u y_u) / s  with s = sum|x| (> eps); gy_t / eps where the norm was clamped
template <int KT>
__global__ void __launch_bounds__(256)
l1norm_bwd(const float *__restrict__ x, NormPtrs p, float *__restrict__ gx, int N, int G, int C, int Krt, i64 HW)
{
  const int K = KT > 0 ? KT : Krt;
  const i64 total = (i64)N * G * C * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const i64 o = idx / HW, pix = idx - o * HW;
    const int c = (int)(o % C);
    const int g = (int)((o / C) % G);
    const i64 n = o / ((i64)C * G);
    const float *xin = x + o * K * HW + pix;
    float *gxo = gx + o * K * HW + pix;
    const float *gb = g == 0 ? p.gy[0] : g == 1 ? p.gy[1] : g == 2 ? p.gy[2] : p.gy[3];
    const float *gyi = gb + (n * C + c) * K * HW + pix;
    if (KT > 0) {
      float v[KT > 0 ? KT : 1], gv[KT > 0 ? KT : 1];
      float sa = 0.f, dot = 0.f;
#pragma unroll
      for (int t = 0; t < KT; t++) { v[t] = xin[(i64)t * HW]; gv[t] = gyi[(i64)t * HW]; sa += fabsf(v[t]); }
      const bool clamped = sa < GA_NORM_EPS;
      const float sden = fmaxf(sa, GA_NORM_EPS);
#pragma unroll
      for (int t = 0; t < KT; t++) dot = fmaf(gv[t], v[t] / sden, dot);
      if (clamped) dot = 0.f;
#pragma unroll
      for (int t = 0; t < KT; t++) {
        const float sg = v[t] > 0.f ? 1.f : (v[t] < 0.f ? -1.f : 0.f);
        gxo[(i64)t * HW] = (gv[t] - sg * dot) / sden;
      }
    } else {
      float sa = 0.f, dot = 0.f;
      for (int t = 0; t < K; t++) sa += fabsf(xin[(i64)t * HW]);
      const bool clamped = sa < GA_NORM_EPS;
      const float sden = fmaxf(sa, GA_NORM_EPS);
      for (int t = 0; t < K; t++) dot = fmaf(gyi[(i64)t * HW], xin[(i64)t * HW] / sden, dot);
      if (clamped) dot = 0.f;
      for (int t = 0; t < K; t++) {
        const float xv = xin[(i64)t * HW];
        const float sg = xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f);
        gxo[(i64)t * HW] = (gyi[(i64)t * HW] - sg * dot) / sden;
      }
    }
  }
}

// out[n,h,w] = sum_d d * x[n,d,h,w] / max(sum_d |x[n,d,h,w]|, eps);  snorm[n,h,w] = that denominator
// (F.normalize(x, p=1, dim=1) followed by DisparityRegression: one read of the volume instead of
// norm + clamp + div + arange-product + sum).
static __global__ void __launch_bounds__(256)
norm_disp_regression_fwd(const float *__restrict__ x, float *__restrict__ out, float *__restrict__ snorm,
                         int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / HW, pix = o - n * HW;
    const float *xp = x + n * Dn * HW + pix;
    float acc = 0.f, sa = 0.f;
    for (int d0 = 0; d0 < Dn; d0 += 8) {       // 8 loads in flight per lane
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = xp[(i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) { acc = fmaf(v[u], (float)(d0 + u), acc); sa += fabsf(v[u]); }
    }
    const float sden = fmaxf(sa, GA_NORM_EPS);
    out[o] = acc / sden;
    snorm[o] = sden;
  }
}

// gx[n,d,h,w] = gout * (d - out * sgn(x_d)) / s    (gout * d / eps where the norm was clamped)
static __global__ void __launch_bounds__(256)
norm_disp_regression_bwd(const float *__restrict__ x, const float *__restrict__ out,
                         const float *__restrict__ snorm, const float *__restrict__ gout,
                         float *__restrict__ gx, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * Dn * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 pix = o % HW;
    const i64 r = o / HW;
    const int d = (int)(r % Dn);
    const i64 n = r / Dn;
    const float xv = x[o];
    const float sden = snorm[n * HW + pix];
    const float ov = sden > GA_NORM_EPS ? out[n * HW + pix] : 0.f;
    const float sg = xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f);
    gx[o] = gout[n * HW + pix] * (((float)d - ov * sg) / sden);
  }
}

// the same, four pixels per lane marching over d (HW % 4 == 0, 16-byte aligned buffers), 4 planes in flight
static __global__ void __launch_bounds__(256)
norm_disp_regression_bwd4(const float *__restrict__ x, const float *__restrict__ out,
                          const float *__restrict__ snorm, const float *__restrict__ gout,
                          float *__restrict__ gx, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * (HW >> 2);
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
    const i64 n = q / (HW >> 2), pix = (q - n * (HW >> 2)) << 2;
    const f4 so = *reinterpret_cast<const f4 *>(snorm + n * HW + pix);
    const f4 oo = *reinterpret_cast<const f4 *>(out + n * HW + pix);
    const f4 go = *reinterpret_cast<const f4 *>(gout + n * HW + pix);
    const float sd[4] = {so.x, so.y, so.z, so.w}, gg[4] = {go.x, go.y, go.z, go.w};
    float ov[4] = {oo.x, oo.y, oo.z, oo.w};
    float gs[4];                                  // gout / s: one division per pixel, not per element
#pragma unroll
    for (int j = 0; j < 4; j++) { ov[j] = sd[j] > GA_NORM_EPS ? ov[j] : 0.f; gs[j] = gg[j] / sd[j]; }
    const float *xp = x + n * Dn * HW + pix;
    float *gp = gx + n * Dn * HW + pix;
    for (int d0 = 0; d0 < Dn; d0 += 4) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const f4 *>(xp + (i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (d0 + u < Dn) {
          const float df = (float)(d0 + u);
          const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float sg = xv[j] > 0.f ? 1.f : (xv[j] < 0.f ? -1.f : 0.f);
            r[j] = gs[j] * (df - ov[j] * sg);
          }
          f4 o_; o_.x = r[0]; o_.y = r[1]; o_.z = r[2]; o_.w = r[3];
          *reinterpret_cast<f4 *>(gp + (i64)(d0 + u) * HW) = o_;
        }
      }
    }
  }
}

// y[n,d,h,w] = softmax_d(-x)  (nn.Softmin(dim=1), models/GANet_deep.py:244): one lane per pixel, two walks over its
// column (running max + rescaled sum, then the normalised exponentials), 8 loads in flight.  Stock PyTorch runs a
// negation kernel plus a strided softmax (0.24 ms at [1,193,240,624]).
static __global__ void __launch_bounds__(256)
softmin_fwd(const float *__restrict__ x, float *__restrict__ y, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / HW, pix = o - n * HW;
    const float *xp = x + n * Dn * HW + pix;
    float *yp = y + n * Dn * HW + pix;
    float m = -INFINITY, ssum = 0.f;
    for (int d0 = 0; d0 < Dn; d0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = -xp[(i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW];
      float mc = m;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) mc = fmaxf(mc, v[u]);
      ssum *= expf(m - mc);                   // (exp(-inf) = 0 on the first chunk)
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) ssum += expf(v[u] - mc);
      m = mc;
    }
    const float inv = 1.f / ssum;
    for (int d0 = 0; d0 < Dn; d0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = -xp[(i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) yp[(i64)(d0 + u) * HW] = expf(v[u] - m) * inv;
    }
  }
}

// gx = -y * (gy - sum_d gy*y)
static __global__ void __launch_bounds__(256)
softmin_bwd(const float *__restrict__ y, const float *__restrict__ gy, float *__restrict__ gx, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / HW, pix = o - n * HW;
    const float *yp = y + n * Dn * HW + pix, *gp = gy + n * Dn * HW + pix;
    float *gxp = gx + n * Dn * HW + pix;
    float dot = 0.f;
    for (int d0 = 0; d0 < Dn; d0 += 8) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const i64 off = (i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW;
        a[u] = yp[off]; b[u] = gp[off];
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) dot = fmaf(a[u], b[u], dot);
    }
    for (int d0 = 0; d0 < Dn; d0 += 8) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const i64 off = (i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW;
        a[u] = yp[off]; b[u] = gp[off];
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) gxp[(i64)(d0 + u) * HW] = -a[u] * (b[u] - dot);
    }
  }
}

// Four consecutive pixels per lane (16-byte requests, HW % 4 == 0, 16-byte aligned volumes): the same two walks.  The
// one-pixel forms above issue 4-byte requests, which reach ~2 TB/s on this chip against ~4.5 TB/s for 16-byte ones
// (scripts/ubench/stream_patterns.hip) -- and these kernels are pure streaming.
static __global__ void __launch_bounds__(256)
softmin_fwd4(const float *__restrict__ x, float *__restrict__ y, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * (HW >> 2);
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / (HW >> 2), pix = (o - n * (HW >> 2)) << 2;
    const float *xp = x + n * Dn * HW + pix;
    float *yp = y + n * Dn * HW + pix;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, ssum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < Dn; d0 += 4) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const f4 *>(xp + (i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float mc = m[j];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (d0 + u < Dn) mc = fmaxf(mc, -f4_get(v[u], j));
        ssum[j] *= expf(m[j] - mc);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (d0 + u < Dn) ssum[j] += expf(-f4_get(v[u], j) - mc);
        m[j] = mc;
      }
    }
    float inv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) inv[j] = 1.f / ssum[j];
    for (int d0 = 0; d0 < Dn; d0 += 4) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const f4 *>(xp + (i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW);
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (d0 + u < Dn) {
          f4 r;
          r.x = expf(-v[u].x - m[0]) * inv[0]; r.y = expf(-v[u].y - m[1]) * inv[1];
          r.z = expf(-v[u].z - m[2]) * inv[2]; r.w = expf(-v[u].w - m[3]) * inv[3];
          *reinterpret_cast<f4 *>(yp + (i64)(d0 + u) * HW) = r;
        }
    }
  }
}

static __global__ void __launch_bounds__(256)
softmin_bwd4(const float *__restrict__ y, const float *__restrict__ gy, float *__restrict__ gx, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * (HW >> 2);
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / (HW >> 2), pix = (o - n * (HW >> 2)) << 2;
    const float *yp = y + n * Dn * HW + pix, *gp = gy + n * Dn * HW + pix;
    float *gxp = gx + n * Dn * HW + pix;
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < Dn; d0 += 4) {
      f4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const i64 off = (i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW;
        a[u] = *reinterpret_cast<const f4 *>(yp + off);
        b[u] = *reinterpret_cast<const f4 *>(gp + off);
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (d0 + u < Dn) {
          dot[0] = fmaf(a[u].x, b[u].x, dot[0]); dot[1] = fmaf(a[u].y, b[u].y, dot[1]);
          dot[2] = fmaf(a[u].z, b[u].z, dot[2]); dot[3] = fmaf(a[u].w, b[u].w, dot[3]);
        }
    }
    for (int d0 = 0; d0 < Dn; d0 += 4) {
      f4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const i64 off = (i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW;
        a[u] = *reinterpret_cast<const f4 *>(yp + off);
        b[u] = *reinterpret_cast<const f4 *>(gp + off);
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (d0 + u < Dn) {
          f4 r;
          r.x = -a[u].x * (b[u].x - dot[0]); r.y = -a[u].y * (b[u].y - dot[1]);
          r.z = -a[u].z * (b[u].z - dot[2]); r.w = -a[u].w * (b[u].w - dot[3]);
          *reinterpret_cast<f4 *>(gxp + (i64)(d0 + u) * HW) = r;
        }
    }
  }
}

// out[n,h,w] = sum_d d * softmin_d(x)  (Disp.forward, models/GANet_deep.py:217-219: Softmin(dim=1) + DisparityRegression)
// in ONE walk over the lane's column: running max m of -x with rescaled sums s = sum e^(-x-m), t = sum d e^(-x-m);
// the probabilities are never written.  mx / ssum ([N,H,W]) are kept for the backward, which recomputes them:
// gx_d = -gout * p_d * (d - out),  p_d = e^(-x_d - m) / s.
static __global__ void __launch_bounds__(256)
softmin_regression_fwd(const float *__restrict__ x, float *__restrict__ out, float *__restrict__ mx,
                       float *__restrict__ ssum, int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / HW, pix = o - n * HW;
    const float *xp = x + n * Dn * HW + pix;
    float m = -INFINITY, s_ = 0.f, t_ = 0.f;
    for (int d0 = 0; d0 < Dn; d0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = -xp[(i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW];
      float mc = m;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) mc = fmaxf(mc, v[u]);
      const float resc = expf(m - mc);
      s_ *= resc; t_ *= resc;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) { const float e = expf(v[u] - mc); s_ += e; t_ = fmaf(e, (float)(d0 + u), t_); }
      m = mc;
    }
    out[o] = t_ / s_;
    mx[o] = m;
    ssum[o] = s_;
  }
}

static __global__ void __launch_bounds__(256)
softmin_regression_bwd(const float *__restrict__ x, const float *__restrict__ out, const float *__restrict__ mx,
                       const float *__restrict__ ssum, const float *__restrict__ gout, float *__restrict__ gx,
                       int N, int Dn, i64 HW)
{
  const i64 total = (i64)N * HW;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const i64 n = o / HW, pix = o - n * HW;
    const float *xp = x + n * Dn * HW + pix;
    float *gp = gx + n * Dn * HW + pix;
    const float m = mx[o], ov = out[o];
    const float gs = -gout[o] / ssum[o];
    for (int d0 = 0; d0 < Dn; d0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = -xp[(i64)(d0 + u < Dn ? d0 + u : Dn - 1) * HW];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < Dn) gp[(i64)(d0 + u) * HW] = gs * expf(v[u] - m) * ((float)(d0 + u) - ov);
    }
  }
}

// ---- SGABlock's residual epilogue (models/GANet_deep.py:270-277) -----------------------------------------
// Behind `conv_refine` (Conv3d + BatchNorm3d, no ReLU) the block ends with `x += rem; relu(x)`: per element
//   y = max(scale[c] * t + shift[c] + rem, 0)      t = the convolution's output, (scale, shift) = the BatchNorm3d folded from
//                                                  its running statistics (eval mode);  scale == nullptr: y = max(t + rem, 0)
//                                                  on t = bn(conv) as PyTorch computed it (training mode).
// Stock PyTorch: batch_norm (r V, w V) + add_ (r 2V, w V) + relu_ (r V, w V) = 7 V; here 2 V in, 1 V out.  y may BE t (the
// reference adds in place as well), which is why neither carries __restrict__.  blockIdx.y = slice (n, c): no division per element.
GA_DEV float relu_keep_nan(float v) { return v <= 0.f ? 0.f : v; }      // ATen's relu / threshold_backward: a NaN passes through
template <bool VEC4>
__global__ void __launch_bounds__(256)
residual_relu_fwd(const float *t, const float *__restrict__ rem, const float *__restrict__ scale,
                  const float *__restrict__ shift, float *y, i64 S /* N*C */, int C, i64 slice /* D*H*W */)
{
  const i64 stride = (i64)gridDim.x * blockDim.x, first = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  for (i64 s = blockIdx.y; s < S; s += gridDim.y) {
    float sc = 1.f, sh = 0.f;
    if (scale) { const int ch = (int)(s % C); sc = scale[ch]; sh = shift[ch]; }
    const i64 base = s * slice;
    if (VEC4) {
      for (i64 i = first; i < (slice >> 2); i += stride) {
        const i64 e = base + (i << 2);
        const f4 a = *reinterpret_cast<const f4 *>(t + e), r = *reinterpret_cast<const f4 *>(rem + e);
        f4 o;
        o.x = relu_keep_nan(fmaf(a.x, sc, sh) + r.x); o.y = relu_keep_nan(fmaf(a.y, sc, sh) + r.y);
        o.z = relu_keep_nan(fmaf(a.z, sc, sh) + r.z); o.w = relu_keep_nan(fmaf(a.w, sc, sh) + r.w);
        *reinterpret_cast<f4 *>(y + e) = o;
      }
    } else {
      for (i64 i = first; i < slice; i += stride)
        y[base + i] = relu_keep_nan(fmaf(t[base + i], sc, sh) + rem[base + i]);
    }
  }
}

// adjoint: g = [y <= 0] ? 0 : gy (what relu_'s threshold_backward computes from the saved output);  g_rem = g;
// g_t = scale[c] * g (scale == nullptr: g_t = g; g_t == nullptr: not written -- the caller hands g_rem to both inputs)
template <bool VEC4>
__global__ void __launch_bounds__(256)
residual_relu_bwd(const float *__restrict__ y, const float *__restrict__ gy, const float *__restrict__ scale,
                  float *__restrict__ g_t, float *__restrict__ g_rem, i64 S, int C, i64 slice)
{
  const i64 stride = (i64)gridDim.x * blockDim.x, first = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  for (i64 s = blockIdx.y; s < S; s += gridDim.y) {
    const float sc = scale ? scale[(int)(s % C)] : 1.f;
    const i64 base = s * slice;
    if (VEC4) {
      for (i64 i = first; i < (slice >> 2); i += stride) {
        const i64 e = base + (i << 2);
        const f4 v = *reinterpret_cast<const f4 *>(y + e), g = *reinterpret_cast<const f4 *>(gy + e);
        f4 o;
        o.x = v.x <= 0.f ? 0.f : g.x; o.y = v.y <= 0.f ? 0.f : g.y;
        o.z = v.z <= 0.f ? 0.f : g.z; o.w = v.w <= 0.f ? 0.f : g.w;
        *reinterpret_cast<f4 *>(g_rem + e) = o;
        if (g_t) {
          o.x *= sc; o.y *= sc; o.z *= sc; o.w *= sc;
          *reinterpret_cast<f4 *>(g_t + e) = o;
        }
      }
    } else {
      for (i64 i = first; i < slice; i += stride) {
        const float g = y[base + i] <= 0.f ? 0.f : gy[base + i];
        g_rem[base + i] = g;
        if (g_t) g_t[base + i] = g * sc;
      }
    }
  }
}

// ---- trilinear up-sampling of a volume (Disp / DispAgg.forward, models/GANet_deep.py:212, 240) -----------------------------
// y = F.interpolate(x, size=[Do, Ho, Wo], mode='trilinear', align_corners=False) on [S, Di, Hi, Wi] -> [S, Do, Ho, Wo]
// (S = N * C slices).  Per axis, PyTorch's rule (area_pixel_compute_source_index): src = max(scale * (o + 0.5) - 0.5, 0)
// with scale = in / out as float, i0 = floor(src), i1 = min(i0 + 1, in - 1), weights (1 - l, l), l = src - i0.
// The forward is a plain 8-point gather.  The BACKWARD is a gather too: one lane per INPUT voxel sums the (up to ~6^3 at a
// 3x zoom) output gradients whose interpolation footprint contains it, weights recomputed on the fly -- where ATen's
// upsample_trilinear3d_backward issues eight atomicAdds per OUTPUT element (22.5 ms of a 114 ms GANet-deep training step
// for its three calls on [1,1,193,240,624], profiles/r2_model_train_stock.json).
struct UpAxis {
  int in, out;
  float scale;      // (float)in / out
};
GA_DEV void up_src(const UpAxis &ax, int o, int &i0, int &i1, float &l1)
{
  float src = ax.scale * ((float)o + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 < ax.in - 1 ? i0 : ax.in - 1;
  i1 = i0 + (i0 < ax.in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);     // (guard_index_and_lambda)
}
// weight with which output o reads input i (0 if it does not)
GA_DEV float up_weight(const UpAxis &ax, int o, int i)
{
  int i0, i1; float l1;
  up_src(ax, o, i0, i1, l1);
  return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}
// outputs that can read input i: [lo, hi] (a superset by at most one on each side; up_weight() decides)
GA_DEV void up_range(const UpAxis &ax, int i, int &lo, int &hi)
{
  const float inv = (float)ax.out / (float)ax.in;
  lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > ax.out - 1 ? ax.out - 1 : hi;
}

static __global__ void __launch_bounds__(256)
trilinear_up_fwd(const float *__restrict__ x, float *__restrict__ y, i64 S, UpAxis ad, UpAxis ah, UpAxis aw)
{
  // one lane per output element, w fastest (coalesced stores; the 8 reads hit a volume 27x smaller: cache resident)
  const i64 total = S * ad.out * ah.out * aw.out;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int ow = (int)(e % aw.out);
    i64 r = e / aw.out;
    const int oh = (int)(r % ah.out); r /= ah.out;
    const int od = (int)(r % ad.out);
    const i64 sl = r / ad.out;
    int d0, d1, h0, h1, w0, w1; float ld, lh, lw;
    up_src(ad, od, d0, d1, ld);
    up_src(ah, oh, h0, h1, lh);
    up_src(aw, ow, w0, w1, lw);
    const float *xs = x + sl * ad.in * ah.in * aw.in;
    const i64 p00 = ((i64)d0 * ah.in + h0) * aw.in, p01 = ((i64)d0 * ah.in + h1) * aw.in;
    const i64 p10 = ((i64)d1 * ah.in + h0) * aw.in, p11 = ((i64)d1 * ah.in + h1) * aw.in;
    // ATen's order: (1-ld) * ((1-lh) * ((1-lw) a + lw b) + lh * (...)) + ld * (...)
    const float a00 = (1.f - lw) * xs[p00 + w0] + lw * xs[p00 + w1];
    const float a01 = (1.f - lw) * xs[p01 + w0] + lw * xs[p01 + w1];
    const float a10 = (1.f - lw) * xs[p10 + w0] + lw * xs[p10 + w1];
    const float a11 = (1.f - lw) * xs[p11 + w0] + lw * xs[p11 + w1];
    y[e] = (1.f - ld) * ((1.f - lh) * a00 + lh * a01) + ld * ((1.f - lh) * a10 + lh * a11);
  }
}

constexpr int UP_MAXFP = 12;     // outputs per axis that can read one input (zoom <= ~5; larger zooms take the slow loop)
static __global__ void __launch_bounds__(256)
trilinear_up_bwd(const float *__restrict__ gy, float *__restrict__ gx, i64 S, UpAxis ad, UpAxis ah, UpAxis aw)
{
  // one lane per INPUT voxel, w fastest: neighbouring lanes read neighbouring (overlapping) runs of gy
  const i64 total = S * ad.in * ah.in * aw.in;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int iw = (int)(e % aw.in);
    i64 r = e / aw.in;
    const int ih = (int)(r % ah.in); r /= ah.in;
    const int id = (int)(r % ad.in);
    const i64 sl = r / ad.in;
    int dlo, dhi, hlo, hhi, wlo, whi;
    up_range(ad, id, dlo, dhi);
    up_range(ah, ih, hlo, hhi);
    up_range(aw, iw, wlo, whi);
    const float *gs = gy + sl * ad.out * ah.out * aw.out;
    float acc = 0.f;
    if (whi - wlo < UP_MAXFP) {
      float ww[UP_MAXFP];
#pragma unroll
      for (int k = 0; k < UP_MAXFP; k++) ww[k] = wlo + k <= whi ? up_weight(aw, wlo + k, iw) : 0.f;
      for (int od = dlo; od <= dhi; od++) {
        const float wd = up_weight(ad, od, id);
        if (wd == 0.f) continue;
        for (int oh = hlo; oh <= hhi; oh++) {
          const float wdh = wd * up_weight(ah, oh, ih);
          if (wdh == 0.f) continue;
          const float *row = gs + ((i64)od * ah.out + oh) * aw.out;
          float t = 0.f;
#pragma unroll
          for (int k = 0; k < UP_MAXFP; k++) {
            const int ow = wlo + k <= whi ? wlo + k : whi;       // clamped address, zero weight past the range
            t = fmaf(ww[k], row[ow], t);
          }
          acc = fmaf(wdh, t, acc);
        }
      }
    } else {
      for (int od = dlo; od <= dhi; od++) {
        const float wd = up_weight(ad, od, id);
        for (int oh = hlo; oh <= hhi; oh++) {
          const float wdh = wd * up_weight(ah, oh, ih);
          const float *row = gs + ((i64)od * ah.out + oh) * aw.out;
          float t = 0.f;
          for (int ow = wlo; ow <= whi; ow++) t = fmaf(up_weight(aw, ow, iw), row[ow], t);
          acc = fmaf(wdh, t, acc);
        }
      }
    }
    gx[e] = acc;
  }
}

}  // namespace ga
