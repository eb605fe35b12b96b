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
#ifndef LGA_WAVES_PER_SIMD
#define LGA_WAVES_PER_SIMD 3   // register cap for R <= 2: the march is latency-bound, occupancy pays
                               // (R = 3 holds 147 taps per pixel: left uncapped so nothing spills)
#endif

// Aligned-window layout.  The tile's column halo is rounded up to an even width RE, so tile
// column 0 sits at an even LDS offset; a lane whose (2R+1)-wide window starts at an odd tile
// column (par = 1) reads from one column earlier.  Every lane therefore fetches the SAME shape
// -- NP = R+1 eight-byte-aligned ds_read_b64 per tile row (256 B/clk, twice the ds_read2_b32
// rate; neighbouring lanes of a pair read identical addresses, which broadcast) -- and the
// parity is folded into the WEIGHTS once per kernel: w6[k] = w[k - par], zero outside.  Each
// b64 result is a natural even-aligned register pair, which is exactly what v_pk_fma_f32 wants:
//   (slab -1, slab 0) accumulators += (v_k, v_k) * (w6a[k], w6b[k])       2 per pair
//   slab +1 accumulator pair       += (v_2q, v_2q+1) * (w6c[2q], w6c[2q+1])  1 per pair
// = 3*NP packed FMAs per row (45 per plane at R = 2, against 75 scalar FMAs + 25 LDS dwords in
// the first version, profiles/r1b_pmc_summary.txt).
template <int R> struct LgaCfg {
  static constexpr int WS = 2 * R + 1;
  static constexpr int K = WS * WS;
  static constexpr int RE = (R + 1) & ~1;            // even column halo
  static constexpr int NP = R + 1;                   // b64 pairs per window
  static constexpr int NK = 2 * NP;                  // window slots (one is a zero-weight dummy)
  static constexpr int TW2 = LGA_TW + 2 * RE;        // even
  static constexpr int TH2 = LGA_TH + 2 * R;
  static constexpr int PLANE = TW2 * TH2;
  static constexpr int STAGE = PLANE * LGA_PB;
  static constexpr int NLD = (STAGE + 255) / 256;    // staged elements per thread
};

struct LgaGeom {
  int D, H, W;
  i64 HW;
};

// Cooperative stage load: planes [d0, d0+PB) of the tile (+halo) -> registers -> LDS.
// Which tile cell a thread copies does not depend on the chunk, so the (plane-in-chunk,
// element offset, in-image) triple is computed ONCE per kernel.
template <int R> struct LgaStage {
  int off[LgaCfg<R>::NLD];     // element offset from the chunk's first plane, or -1 = write zero
  int pl[LgaCfg<R>::NLD];      // plane within the chunk
};
template <int R>
GA_DEV void lga_stage_init(LgaStage<R> &st, const LgaGeom &geo, int ty0, int tx0)
{
  typedef LgaCfg<R> C;
#pragma unroll
  for (int l = 0; l < C::NLD; l++) {
    const int e = l * 256 + (int)threadIdx.x;
    st.off[l] = -1;
    st.pl[l] = 0;
    if (e < C::STAGE) {
      const int pl = e / C::PLANE, rem = e - pl * C::PLANE;
      const int r = rem / C::TW2, cc = rem - r * C::TW2;
      const int i = ty0 + r - R, j = tx0 + cc - C::RE;
      st.pl[l] = pl;
      if (i >= 0 && i < geo.H && j >= 0 && j < geo.W) st.off[l] = (int)(pl * geo.HW + (i64)i * geo.W + j);
    }
  }
}
template <int R>
GA_DEV void lga_stage_fetch(const float *__restrict__ xb, const LgaGeom &geo, const LgaStage<R> &st,
                            int d0, float (&regs)[LgaCfg<R>::NLD])
{
  typedef LgaCfg<R> C;
  const float *base = xb + (i64)d0 * geo.HW;
#pragma unroll
  for (int l = 0; l < C::NLD; l++)
    regs[l] = (st.off[l] >= 0 && d0 + st.pl[l] < geo.D) ? base[st.off[l]] : 0.f;
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

// weights of one pixel in the aligned-window packing; CHECK = false when every tap is in the image
template <int R, bool TRANSPOSED, bool CHECK>
GA_DEV void lga_gather_weights(const float *__restrict__ fb, const LgaGeom &geo, int ic, int jc, int par,
                               f2 (&wab)[LgaCfg<R>::WS][LgaCfg<R>::NK], f2 (&wc)[LgaCfg<R>::WS][LgaCfg<R>::NP],
                               float &cmid, float &sin_m, float &sin_p)
{
  typedef LgaCfg<R> C;
  const float *fp = fb + (i64)ic * geo.W + jc;          // own pixel, tap plane 0
#pragma unroll
  for (int a = -R; a <= R; a++) {
    float wt[3][C::WS];
#pragma unroll
    for (int dd = 0; dd < 3; dd++) {
#pragma unroll
      for (int bb = -R; bb <= R; bb++) {
        const int t = dd * C::K + (a + R) * C::WS + (bb + R);
        bool ok = true;
        if (CHECK) {
          const int i2 = ic + a, j2 = jc + bb;
          ok = i2 >= 0 && i2 < geo.H && j2 >= 0 && j2 < geo.W;
        }
        float own = 0.f;
        if (CHECK || !TRANSPOSED || dd != 1) own = fp[(i64)t * geo.HW];   // interior gX needs own taps only for the d-edge sums
        float wv = own;
        if (TRANSPOSED) {
          const int tf = (2 - dd) * C::K + (-a + R) * C::WS + (-bb + R);
          wv = ok ? fp[(i64)tf * geo.HW + a * geo.W + bb] : 0.f;
        }
        wt[dd][bb + R] = ok ? wv : 0.f;
        if (!ok) cmid += own;
        else if (dd == 0) sin_m += own;
        else if (dd == 2) sin_p += own;
      }
    }
    float w6[3][C::NK];
#pragma unroll
    for (int dd = 0; dd < 3; dd++)
#pragma unroll
      for (int k = 0; k < C::NK; k++) {
        const float we = k < C::WS ? wt[dd][k < C::WS ? k : 0] : 0.f;            // par = 0: slot k = tap k
        const float wo = k >= 1 ? wt[dd][k >= 1 ? k - 1 : 0] : 0.f;              // par = 1: slot k = tap k-1
        w6[dd][k] = par ? wo : we;
      }
#pragma unroll
    for (int k = 0; k < C::NK; k++) wab[a + R][k] = mk2(w6[0][k], w6[1][k]);
#pragma unroll
    for (int q = 0; q < C::NP; q++) wc[a + R][q] = mk2(w6[2][2 * q], w6[2][2 * q + 1]);
  }
}

// ---- forward (TRANSPOSED = false) and data-backward (TRANSPOSED = true) ---------
// y[b,d,i,j] = sum_t w_t * xs(d+dd, i+a, j+b)  with centre replacement.
template <int R, bool TRANSPOSED>
__global__ void __launch_bounds__(256, (R <= 2 ? LGA_WAVES_PER_SIMD : 1))
lga_apply(const float *__restrict__ x, const float *__restrict__ f, float *__restrict__ y,
          LgaGeom geo)
{
  typedef LgaCfg<R> C;
  __shared__ __attribute__((aligned(16))) float tile[2][C::STAGE];
  const int tx = threadIdx.x % LGA_TW, ty = threadIdx.x / LGA_TW;
  // XCD-aware tile order: hardware puts consecutive block ids on different XCDs (id % 8), each
  // with its own L2; neighbouring tiles share halo lines, so give every XCD a contiguous band
  // of tiles (measured: 3.2x DRAM over-fetch without it, profiles/r1h_pmc_memory_side.txt)
  int bx, by, b;
  {
    const int nb = gridDim.x * gridDim.y * gridDim.z;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nb);
    bx = lid % gridDim.x;
    by = (lid / gridDim.x) % gridDim.y;
    b = lid / (gridDim.x * gridDim.y);
  }
  const int tx0 = bx * LGA_TW, ty0 = by * LGA_TH;
  const int i = ty0 + ty, j = tx0 + tx;
  const bool inb = i < geo.H && j < geo.W;
  const int ic = i < geo.H ? i : geo.H - 1, jc = j < geo.W ? j : geo.W - 1;
  const float *xb = x + (i64)b * geo.D * geo.HW;
  const float *fb = f + (i64)b * 3 * C::K * geo.HW;
  float *yb = y + (i64)b * geo.D * geo.HW;
  const i64 pix = (i64)ic * geo.W + jc;
  const int wcol = tx + C::RE - R;          // tile column where the window starts
  const int par = wcol & 1;                 // 1: the aligned read starts one column earlier
  const int rcol = wcol - par;              // even

  // per-pixel weights (parity-shifted, packed) and centre coefficients.  Tiles whose halo lies
  // entirely inside the image (84 % of them at 240x624) skip every bounds test: the weight
  // set-up was ~30 % of this kernel's VALU instructions (profiles/r1f_pmc_lga_apply_packed.txt
  // vs the 67 VALU per plane of the steady-state loop).
  f2 wab[C::WS][C::NK];                     // (slab -1, slab 0) of window slot k
  f2 wc[C::WS][C::NP];                      // slab +1 of slots (2q, 2q+1)
  float cmid = 0.f, sin_m = 0.f, sin_p = 0.f;
  const bool interior = ty0 >= R && ty0 + LGA_TH + R <= geo.H && tx0 >= R && tx0 + LGA_TW + R <= geo.W;
  if (interior)
    lga_gather_weights<R, TRANSPOSED, false>(fb, geo, ic, jc, par, wab, wc, cmid, sin_m, sin_p);
  else
    lga_gather_weights<R, TRANSPOSED, true>(fb, geo, ic, jc, par, wab, wc, cmid, sin_m, sin_p);

  const int nchunks = (geo.D + LGA_PB - 1) / LGA_PB;
  LgaStage<R> stg;
  lga_stage_init<R>(stg, geo, ty0, tx0);
  float regs[C::NLD];
  lga_stage_fetch<R>(xb, geo, stg, 0, regs);
  lga_stage_commit<R>(tile[0], regs);
  __syncthreads();

  float acc_a = 0.f, acc_b = 0.f;   // partial y[d-1], y[d] while visiting plane d
  float xc_prev = 0.f;
  float *yp = yb + pix;               // output cursor: plane d-1 of the own pixel
  for (int c = 0; c < nchunks; c++) {
    const bool more = c + 1 < nchunks;
    if (more) lga_stage_fetch<R>(xb, geo, stg, (c + 1) * LGA_PB, regs);
    const lds_cptr buf = GA_LDS_CPTR(&tile[0][0]) + (c & 1) * C::STAGE;
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      const int d = c * LGA_PB + pl;
      if (d < geo.D) {
        const lds_cptr pb = buf + pl * C::PLANE + ty * C::TW2 + rcol;
        // four independent accumulator chains (a single chain of dependent v_pk_fma_f32 was
        // slower than three scalar chains; 15 chains cost registers -> occupancy)
        f2 s_x = mk2(0.f, 0.f), s_y = mk2(0.f, 0.f), s_p0 = mk2(0.f, 0.f), s_p1 = mk2(0.f, 0.f);
        float xc = 0.f;
#pragma unroll
        for (int a = 0; a < C::WS; a++) {
#pragma unroll
          for (int q = 0; q < C::NP; q++) {
            const f2 vv = lds_read_b64(pb + a * C::TW2 + 2 * q);
            s_x = fma2(mk2(vv.x, vv.x), wab[a][2 * q], s_x);
            s_y = fma2(mk2(vv.y, vv.y), wab[a][2 * q + 1], s_y);
            if (a & 1) s_p1 = fma2(vv, wc[a][q], s_p1);
            else s_p0 = fma2(vv, wc[a][q], s_p0);
            if (a == R && 2 * q <= R && R <= 2 * q + 1) {
              // centre sample: slot R (par 0) or R+1 (par 1)
              const float c0 = (R & 1) ? vv.y : vv.x;
              xc = par ? xc : c0;
            }
            if (a == R && 2 * q <= R + 1 && R + 1 <= 2 * q + 1) {
              const float c1 = ((R + 1) & 1) ? vv.y : vv.x;
              xc = par ? c1 : xc;
            }
          }
        }
        const float zm = s_x.x + s_y.x;                       // depth slab -1 -> y[d+1]
        const float z0 = s_x.y + s_y.y;                       // depth slab  0 -> y[d]
        const float zp = (s_p0.x + s_p0.y) + (s_p1.x + s_p1.y);   // depth slab +1 -> y[d-1]
        if (d >= 1) {
          const int dy = d - 1;
          float cc = cmid;
          if (dy == 0) cc += sin_m;
          const float r = fmaf(xc_prev, cc, acc_a + zp);
          if (inb) *yp = r;
          yp += geo.HW;
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
    if (inb) *yp = r;
  }
}

// ---- filter backward --------------------------------------------------------------
// gf[b,t,i,j] (+)= sum_d gy[b,d,i,j] * xs(d+dd, i+a, j+b)   (centre replacement)
template <int R>
__global__ void __launch_bounds__(256, (R <= 2 ? LGA_WAVES_PER_SIMD : 1))
lga_filter_grad(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gf,
                LgaGeom geo, int accumulate)
{
  typedef LgaCfg<R> C;
  __shared__ __attribute__((aligned(16))) float tile[2][C::STAGE];
  const int tx = threadIdx.x % LGA_TW, ty = threadIdx.x / LGA_TW;
  // XCD-aware tile order: hardware puts consecutive block ids on different XCDs (id % 8), each
  // with its own L2; neighbouring tiles share halo lines, so give every XCD a contiguous band
  // of tiles (measured: 3.2x DRAM over-fetch without it, profiles/r1h_pmc_memory_side.txt)
  int bx, by, b;
  {
    const int nb = gridDim.x * gridDim.y * gridDim.z;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nb);
    bx = lid % gridDim.x;
    by = (lid / gridDim.x) % gridDim.y;
    b = lid / (gridDim.x * gridDim.y);
  }
  const int tx0 = bx * LGA_TW, ty0 = by * LGA_TH;
  const int i = ty0 + ty, j = tx0 + tx;
  const bool inb = i < geo.H && j < geo.W;
  const int ic = i < geo.H ? i : geo.H - 1, jc = j < geo.W ? j : geo.W - 1;
  const float *xb = x + (i64)b * geo.D * geo.HW;
  const float *gyb = gy + (i64)b * geo.D * geo.HW;
  float *gfb = gf + (i64)b * 3 * C::K * geo.HW;
  const i64 pix = (i64)ic * geo.W + jc;
  const int wcol = tx + C::RE - R;
  const int par = wcol & 1;
  const int rcol = wcol - par;

  // partial sums per window slot: sab[a][k] = (slab -1, slab 0), sc[a][q] = slab +1 of (2q, 2q+1)
  f2 sab[C::WS][C::NK], sc[C::WS][C::NP];
#pragma unroll
  for (int a = 0; a < C::WS; a++) {
#pragma unroll
    for (int k = 0; k < C::NK; k++) sab[a][k] = mk2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < C::NP; q++) sc[a][q] = mk2(0.f, 0.f);
  }
  float gc = 0.f;                 // sum_d gy[d] * x[d][centre]
  float e_lo = 0.f, e_hi = 0.f;   // gy[0]*x[0][c], gy[D-1]*x[D-1][c]

  const int nchunks = (geo.D + LGA_PB - 1) / LGA_PB;
  LgaStage<R> stg;
  lga_stage_init<R>(stg, geo, ty0, tx0);
  float regs[C::NLD];
  lga_stage_fetch<R>(xb, geo, stg, 0, regs);
  lga_stage_commit<R>(tile[0], regs);
  __syncthreads();

  // gy of the own pixel at planes d-1, d, d+1 (rolling); the next chunk's values are fetched
  // one chunk ahead so the march never waits on a dependent global load
  float g_m = 0.f, g_0 = gyb[pix];
  float gnext[LGA_PB];
#pragma unroll
  for (int pl = 0; pl < LGA_PB; pl++) gnext[pl] = pl + 1 < geo.D ? gyb[(i64)(pl + 1) * geo.HW + pix] : 0.f;
  for (int c = 0; c < nchunks; c++) {
    const bool more = c + 1 < nchunks;
    if (more) lga_stage_fetch<R>(xb, geo, stg, (c + 1) * LGA_PB, regs);
    float gcur[LGA_PB];
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      gcur[pl] = gnext[pl];
      const int dn = (c + 1) * LGA_PB + pl + 1;
      gnext[pl] = dn < geo.D ? gyb[(i64)dn * geo.HW + pix] : 0.f;
    }
    const lds_cptr buf = GA_LDS_CPTR(&tile[0][0]) + (c & 1) * C::STAGE;
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      const int d = c * LGA_PB + pl;
      if (d < geo.D) {
        const float g_p = gcur[pl];                       // gy[d+1] (0 past the end)
        const lds_cptr pb = buf + pl * C::PLANE + ty * C::TW2 + rcol;
        // plane d pairs with gy[d+1] for slab -1, gy[d] for slab 0, gy[d-1] for slab +1
        const f2 g01 = mk2(g_p, g_0), gmm = mk2(g_m, g_m);
        float xc = 0.f;
#pragma unroll
        for (int a = 0; a < C::WS; a++) {
#pragma unroll
          for (int q = 0; q < C::NP; q++) {
            const f2 vv = lds_read_b64(pb + a * C::TW2 + 2 * q);
            sab[a][2 * q] = fma2(mk2(vv.x, vv.x), g01, sab[a][2 * q]);
            sab[a][2 * q + 1] = fma2(mk2(vv.y, vv.y), g01, sab[a][2 * q + 1]);
            sc[a][q] = fma2(vv, gmm, sc[a][q]);
            if (a == R && 2 * q <= R && R <= 2 * q + 1) {
              const float c0 = (R & 1) ? vv.y : vv.x;
              xc = par ? xc : c0;
            }
            if (a == R && 2 * q <= R + 1 && R + 1 <= 2 * q + 1) {
              const float c1 = ((R + 1) & 1) ? vv.y : vv.x;
              xc = par ? c1 : xc;
            }
          }
        }
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
          // tap bb lives in window slot k = (bb + R) + par
          const int ke = bb + R, ko = bb + R + 1, ra = a + R;
          float re, ro;
          if (dd == 0) { re = sab[ra][ke].x; ro = sab[ra][ko].x; }
          else if (dd == 1) { re = sab[ra][ke].y; ro = sab[ra][ko].y; }
          else {
            re = (ke & 1) ? sc[ra][ke >> 1].y : sc[ra][ke >> 1].x;
            ro = (ko & 1) ? sc[ra][ko >> 1].y : sc[ra][ko >> 1].x;
          }
          float r = par ? ro : re;
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
