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
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < Dn; i++) {
      if (i <= w) sx += gl[(i64)i * HW + w];
      if (w + i < W) sy += gr[(i64)i * HW + w + i];
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
    for (int d = 0; d < Dn; d++) acc = fmaf(xp[(i64)d * HW], (float)d, acc);
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

}  // namespace ga
