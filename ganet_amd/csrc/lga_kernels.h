// lga_kernels.h -- local guided aggregation (LGA) for gfx950.
//
// What it computes: SURVEY.md Appendix A.3, i.e. the reference's
// lga_filtering_forward / lga_filter_backward / lga_data_backward
// (libs/GANet/src/GANet_kernel.cu:1131-1269): a per-pixel 3 x (2r+1) x (2r+1)
// filter over (disparity, row, col) where an out-of-range neighbour (in ANY of the
// three axes) is replaced by the centre sample.
//
// Design (instead of one CUDA thread per output element doing 75 global RMWs):
//  * one lane per PIXEL, marching over disparity; the pixel's 3K filter taps live
//    in VGPRs for the whole march (they do not depend on d), so the filter volume
//    is read from HBM exactly once;
//  * the input plane tile (+halo, zero outside the image) is staged through LDS in
//    chunks of PB planes, double-buffered, one barrier per chunk; each LDS read
//    feeds three FMAs (the plane contributes to y[d-1], y[d], y[d+1]);
//  * the centre-replacement rule is folded into per-pixel constants: taps that are
//    spatially out of range get weight 0 and their sum multiplies the centre
//    sample; the d = 0 / d = D-1 planes add the in-range part of the missing
//    depth slab.  Border and interior pixels run the same straight-line code.
//  * data-backward is the SAME kernel with transposed weights gathered from the
//    neighbouring pixels' filters (flipped tap), so it inherits the tiling;
//  * filter-backward keeps the 3K partial sums of a pixel in VGPRs across the
//    whole disparity march (the reference does a global RMW per disparity).
//
// MFMA note (north_star asks for it "where it is a true dense contraction"): per
// pixel this is a [D x K] . [K x 3] product whose two operands are BOTH private
// to the pixel; there is no operand shared across pixels, so an MFMA tile would
// run at N = 3 of 16/32 columns (<= 19 % of the fp32 MFMA rate, which on gfx950
// equals the fp32 VALU rate).  The VALU formulation is the faster one; see
// DESIGN.md.
#pragma once
#include "ga_common.h"

namespace ga {

constexpr int LGA_TW = 32;   // tile width  (pixels, = lanes along W)
constexpr int LGA_TH = 8;    // tile height
constexpr int LGA_PB = 4;    // planes per LDS stage

template <int R> struct LgaCfg {
  static constexpr int WS = 2 * R + 1;
  static constexpr int K = WS * WS;
  static constexpr int TW2 = LGA_TW + 2 * R;
  static constexpr int TH2 = LGA_TH + 2 * R;
  static constexpr int PLANE = TW2 * TH2;
  static constexpr int STAGE = PLANE * LGA_PB;
  static constexpr int NLD = (STAGE + 255) / 256;   // staged elements per thread
};

struct LgaGeom {
  int D, H, W;
  i64 HW;
};

// cooperative stage load: planes [d0, d0+PB) of the tile (+halo) -> registers
template <int R>
GA_DEV void lga_stage_fetch(const float *__restrict__ xb, const LgaGeom &geo, int ty0, int tx0,
                            int d0, float (&regs)[LgaCfg<R>::NLD])
{
  typedef LgaCfg<R> C;
#pragma unroll
  for (int l = 0; l < C::NLD; l++) {
    const int e = l * 256 + (int)threadIdx.x;
    float v = 0.f;
    if (e < C::STAGE) {
      const int pl = e / C::PLANE, rem = e - pl * C::PLANE;
      const int r = rem / C::TW2, cc = rem - r * C::TW2;
      const int d = d0 + pl, i = ty0 + r - R, j = tx0 + cc - R;
      if (d < geo.D && i >= 0 && i < geo.H && j >= 0 && j < geo.W)
        v = xb[(i64)d * geo.HW + (i64)i * geo.W + j];
    }
    regs[l] = v;
  }
}
template <int R>
GA_DEV void lga_stage_commit(float *__restrict__ buf, const float (&regs)[LgaCfg<R>::NLD])
{
  typedef LgaCfg<R> C;
#pragma unroll
  for (int l = 0; l < C::NLD; l++) {
    const int e = l * 256 + (int)threadIdx.x;
    if (e < C::STAGE) buf[e] = regs[l];
  }
}

// ---- forward (TRANSPOSED = false) and data-backward (TRANSPOSED = true) ---------
// y[b,d,i,j] = sum_t w_t * xs(d+dd, i+a, j+b)  with centre replacement.
template <int R, bool TRANSPOSED>
__global__ void __launch_bounds__(256)
lga_apply(const float *__restrict__ x, const float *__restrict__ f, float *__restrict__ y,
          LgaGeom geo)
{
  typedef LgaCfg<R> C;
  __shared__ float tile[2][C::STAGE];
  const int tx = threadIdx.x % LGA_TW, ty = threadIdx.x / LGA_TW;
  const int tx0 = blockIdx.x * LGA_TW, ty0 = blockIdx.y * LGA_TH;
  const int b = blockIdx.z;
  const int i = ty0 + ty, j = tx0 + tx;
  const bool inb = i < geo.H && j < geo.W;
  const int ic = i < geo.H ? i : geo.H - 1, jc = j < geo.W ? j : geo.W - 1;
  const float *xb = x + (i64)b * geo.D * geo.HW;
  const float *fb = f + (i64)b * 3 * C::K * geo.HW;
  float *yb = y + (i64)b * geo.D * geo.HW;
  const i64 pix = (i64)ic * geo.W + jc;

  // per-pixel weights and centre coefficients
  float w[3][C::K];
  float cmid = 0.f, sin_m = 0.f, sin_p = 0.f;
#pragma unroll
  for (int dd = 0; dd < 3; dd++) {
#pragma unroll
    for (int a = -R; a <= R; a++) {
#pragma unroll
      for (int bb = -R; bb <= R; bb++) {
        const int t = dd * C::K + (a + R) * C::WS + (bb + R);
        const int i2 = ic + a, j2 = jc + bb;
        const bool ok = i2 >= 0 && i2 < geo.H && j2 >= 0 && j2 < geo.W;
        const float own = fb[(i64)t * geo.HW + pix];
        float wv = own;
        if (TRANSPOSED) {
          const int tf = (2 - dd) * C::K + (-a + R) * C::WS + (-bb + R);
          wv = ok ? fb[(i64)tf * geo.HW + (i64)i2 * geo.W + j2] : 0.f;
        }
        w[dd][(a + R) * C::WS + (bb + R)] = ok ? wv : 0.f;
        if (!ok) cmid += own;
        else if (dd == 0) sin_m += own;
        else if (dd == 2) sin_p += own;
      }
    }
  }

  const int nchunks = (geo.D + LGA_PB - 1) / LGA_PB;
  float regs[C::NLD];
  lga_stage_fetch<R>(xb, geo, ty0, tx0, 0, regs);
  lga_stage_commit<R>(tile[0], regs);
  __syncthreads();

  float acc_a = 0.f, acc_b = 0.f;   // partial y[d-1], y[d] while visiting plane d
  float xc_prev = 0.f;
  for (int c = 0; c < nchunks; c++) {
    const bool more = c + 1 < nchunks;
    if (more) lga_stage_fetch<R>(xb, geo, ty0, tx0, (c + 1) * LGA_PB, regs);
    const float *buf = tile[c & 1];
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      const int d = c * LGA_PB + pl;
      if (d < geo.D) {
        const float *pb = buf + pl * C::PLANE + ty * C::TW2 + tx;
        float zm = 0.f, z0 = 0.f, zp = 0.f;
#pragma unroll
        for (int a = 0; a < C::WS; a++) {
#pragma unroll
          for (int bb = 0; bb < C::WS; bb++) {
            const float v = pb[a * C::TW2 + bb];
            zm = fmaf(v, w[0][a * C::WS + bb], zm);   // dd = -1 -> y[d+1]
            z0 = fmaf(v, w[1][a * C::WS + bb], z0);   // dd =  0 -> y[d]
            zp = fmaf(v, w[2][a * C::WS + bb], zp);   // dd = +1 -> y[d-1]
          }
        }
        const float xc = pb[R * C::TW2 + R];
        if (d >= 1) {
          const int dy = d - 1;
          float cc = cmid;
          if (dy == 0) cc += sin_m;
          if (dy == geo.D - 1) cc += sin_p;   // unreachable here (dy <= D-2), kept for clarity
          const float r = fmaf(xc_prev, cc, acc_a + zp);
          if (inb) yb[(i64)dy * geo.HW + pix] = r;
        }
        acc_a = acc_b + z0;
        acc_b = zm;
        xc_prev = xc;
      }
    }
    if (more) lga_stage_commit<R>(tile[(c + 1) & 1], regs);
    __syncthreads();
  }
  {
    const int dy = geo.D - 1;
    float cc = cmid + sin_p;
    if (dy == 0) cc += sin_m;
    const float r = fmaf(xc_prev, cc, acc_a);
    if (inb) yb[(i64)dy * geo.HW + pix] = r;
  }
}

// ---- filter backward --------------------------------------------------------------
// gf[b,t,i,j] (+)= sum_d gy[b,d,i,j] * xs(d+dd, i+a, j+b)   (centre replacement)
template <int R>
__global__ void __launch_bounds__(256)
lga_filter_grad(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gf,
                LgaGeom geo, int accumulate)
{
  typedef LgaCfg<R> C;
  __shared__ float tile[2][C::STAGE];
  const int tx = threadIdx.x % LGA_TW, ty = threadIdx.x / LGA_TW;
  const int tx0 = blockIdx.x * LGA_TW, ty0 = blockIdx.y * LGA_TH;
  const int b = blockIdx.z;
  const int i = ty0 + ty, j = tx0 + tx;
  const bool inb = i < geo.H && j < geo.W;
  const int ic = i < geo.H ? i : geo.H - 1, jc = j < geo.W ? j : geo.W - 1;
  const float *xb = x + (i64)b * geo.D * geo.HW;
  const float *gyb = gy + (i64)b * geo.D * geo.HW;
  float *gfb = gf + (i64)b * 3 * C::K * geo.HW;
  const i64 pix = (i64)ic * geo.W + jc;

  float acc[3][C::K];
#pragma unroll
  for (int dd = 0; dd < 3; dd++)
#pragma unroll
    for (int t = 0; t < C::K; t++) acc[dd][t] = 0.f;
  float gc = 0.f;                 // sum_d gy[d] * x[d][centre]
  float e_lo = 0.f, e_hi = 0.f;   // gy[0]*x[0][c], gy[D-1]*x[D-1][c]

  const int nchunks = (geo.D + LGA_PB - 1) / LGA_PB;
  float regs[C::NLD];
  lga_stage_fetch<R>(xb, geo, ty0, tx0, 0, regs);
  lga_stage_commit<R>(tile[0], regs);
  __syncthreads();

  // gy at planes d-1, d, d+1 of the own pixel (rolling)
  float g_m = 0.f, g_0 = gyb[pix], g_p = 0.f;
  for (int c = 0; c < nchunks; c++) {
    const bool more = c + 1 < nchunks;
    if (more) lga_stage_fetch<R>(xb, geo, ty0, tx0, (c + 1) * LGA_PB, regs);
    const float *buf = tile[c & 1];
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      const int d = c * LGA_PB + pl;
      if (d < geo.D) {
        g_p = d + 1 < geo.D ? gyb[(i64)(d + 1) * geo.HW + pix] : 0.f;
        const float *pb = buf + pl * C::PLANE + ty * C::TW2 + tx;
        // plane d pairs with gy[d+1] for dd=-1, gy[d] for dd=0, gy[d-1] for dd=+1
#pragma unroll
        for (int a = 0; a < C::WS; a++) {
#pragma unroll
          for (int bb = 0; bb < C::WS; bb++) {
            const float v = pb[a * C::TW2 + bb];
            acc[0][a * C::WS + bb] = fmaf(g_p, v, acc[0][a * C::WS + bb]);
            acc[1][a * C::WS + bb] = fmaf(g_0, v, acc[1][a * C::WS + bb]);
            acc[2][a * C::WS + bb] = fmaf(g_m, v, acc[2][a * C::WS + bb]);
          }
        }
        const float xc = pb[R * C::TW2 + R];
        const float e = g_0 * xc;
        gc += e;
        if (d == 0) e_lo = e;
        if (d == geo.D - 1) e_hi = e;
        g_m = g_0;
        g_0 = g_p;
      }
    }
    if (more) lga_stage_commit<R>(tile[(c + 1) & 1], regs);
    __syncthreads();
  }

  if (inb) {
#pragma unroll
    for (int dd = 0; dd < 3; dd++) {
#pragma unroll
      for (int a = -R; a <= R; a++) {
#pragma unroll
        for (int bb = -R; bb <= R; bb++) {
          const int t = dd * C::K + (a + R) * C::WS + (bb + R);
          const int i2 = i + a, j2 = j + bb;
          const bool ok = i2 >= 0 && i2 < geo.H && j2 >= 0 && j2 < geo.W;
          float r = acc[dd][(a + R) * C::WS + (bb + R)];
          if (dd == 0) r += e_lo;
          if (dd == 2) r += e_hi;
          if (!ok) r = gc;
          float *dst = gfb + (i64)t * geo.HW + pix;
          *dst = accumulate ? *dst + r : r;
        }
      }
    }
  }
}

}  // namespace ga
