/*
 * ganet_oracle.c -- CPU restatement of GANet's guided-aggregation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ganet_amd/ (the product) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / baseline.
 *
 * Parity status: PINNED against the reference's own kernel arithmetic.  The
 * reference ships no tests or golden vectors, so the pin is
 * (a) oracle/_ref (the reference's __global__ bodies host-compiled through
 * oracle/ref_shim, see oracle/Makefile) compared with this file in
 * tests/test_oracle_vs_ref.py whenever /root/reference is present, and
 * (b) tests/golden/*.npz generated from oracle/_ref by
 * tests/golden/make_golden.py and committed.
 *
 * Every function cites the reference lines it restates (paths relative to
 * the reference root, file libs/GANet/src/GANet_kernel.cu unless noted).
 * This is a from-scratch restatement: one generic scan routine with a
 * direction->offset map instead of the reference's four hand-mirrored
 * copies, out-of-place scans instead of memcpy + in-place, explicit fmaf()
 * for every `temp += a * b` (what nvcc's default -fmad=true contraction
 * produces on the reference source), so the forward values -- and therefore
 * every discrete decision (in-scan argmax k, direction mask, MaxDepth idx)
 * -- are bit-identical to a CUDA build of the reference.
 *
 * Layouts (all contiguous fp32): volumes [N,C,D,H,W]; guidance [N,C,5,H,W];
 * LGA input [B,D,H,W], filters [B,3*(2r+1)^2,H,W].
 * Direction ids follow the reference mask values: 0 down, 1 up, 2 right,
 * 3 left (GANet_kernel.cu:964-994).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef long long i64;

int oracle_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- scanline geometry ------------------------------------------------ */
/* A scanline of direction `dir` is identified by q (column for down/up,
 * row for right/left); pos(p) is the spatial offset (row*W+col) of its p-th
 * visited pixel.  down :66-127 rows 0..H-1; up :285-346 rows H-1..0;
 * right :507-565 cols 0..W-1; left :720-778 cols W-1..0. */
static inline int scan_len(int dir, int H, int W) { return dir < 2 ? H : W; }
static inline int scan_lines(int dir, int H, int W) { return dir < 2 ? W : H; }
static inline i64 scan_pos(int dir, int H, int W, int q, int p)
{
  switch (dir) {
    case 0: return (i64)p * W + q;
    case 1: return (i64)(H - 1 - p) * W + q;
    case 2: return (i64)q * W + p;
    default: return (i64)q * W + (W - 1 - p);
  }
}

/* ---- SGA forward, one direction --------------------------------------- */
/* Restates sga_{down,up,right,left}_forward (:66-127, :285-346, :507-565,
 * :720-778).  A[p][d] = x*w0 + P1*w1 + P2*w2 + P3*w3 + P4*w4 accumulated
 * left to right with fma; any unavailable tap is replaced by x[p][d];
 * P4 = A[p-1][k], k = first argmax_d A[p-1][.] (strict '<', :122-123).
 * If idx != NULL it receives first-argmax_d A[p][.] per pixel (what
 * MaxDepth :50-64 computes from the finished scan). */
void oracle_sga_scan_forward(const float *x, const float *g, float *A, float *idx,
                             int NC, int D, int H, int W, int dir)
{
  const i64 HW = (i64)H * W;
  const int L = scan_len(dir, H, W), Q = scan_lines(dir, H, W);
#pragma omp parallel for collapse(2) schedule(static)
  for (int s = 0; s < NC; s++) {
    for (int q = 0; q < Q; q++) {
      const float *xs = x + (i64)s * D * HW;
      const float *gs = g + (i64)s * 5 * HW;
      float *As = A + (i64)s * D * HW;
      int k = 0;                 /* argmax of previous position */
      for (int p = 0; p < L; p++) {
        const i64 o = scan_pos(dir, H, W, q, p);
        const i64 op = p > 0 ? scan_pos(dir, H, W, q, p - 1) : 0;
        const float w0 = gs[o], w1 = gs[HW + o], w2 = gs[2 * HW + o],
                    w3 = gs[3 * HW + o], w4 = gs[4 * HW + o];
        int kp = 0;
        float best = 0.f;
        for (int d = 0; d < D; d++) {
          const float xv = xs[d * HW + o];
          float P1 = xv, P2 = xv, P3 = xv, P4 = xv;
          if (p > 0) {
            P1 = As[d * HW + op];
            if (d >= 1) P2 = As[(d - 1) * HW + op];
            if (d + 1 < D) P3 = As[(d + 1) * HW + op];
            P4 = As[(i64)k * HW + op];
          }
          float t = fmaf(xv, w0, 0.0f);
          t = fmaf(P1, w1, t);
          t = fmaf(P2, w2, t);
          t = fmaf(P3, w3, t);
          t = fmaf(P4, w4, t);
          As[d * HW + o] = t;
          if (d == 0) { best = t; kp = 0; }
          else if (best < t) { best = t; kp = d; }
        }
        if (idx) idx[(i64)s * HW + o] = (float)kp;
        k = kp;
      }
    }
  }
}

/* ---- SGA forward, full ------------------------------------------------- */
/* Restates sga_kernel_forward (:935-998) + Max (:23-36): out = A_down,
 * mask = 0; then for dir in (up=1, right=2, left=3): where out < A_dir,
 * out = A_dir and mask = dir.  temp_out ends holding A_left (F6).
 * `out`, `mask` need not be pre-zeroed.  A_all (optional, may be NULL) is
 * [4][NC*D*H*W] and receives the four directional volumes. */
void oracle_sga_forward(const float *x, const float *g0, const float *g1,
                        const float *g2, const float *g3, float *temp_out,
                        float *out, float *mask, float *A_all,
                        int N, int C, int D, int H, int W)
{
  const int NC = N * C;
  const i64 n = (i64)NC * D * H * W;
  const float *gs[4] = {g0, g1, g2, g3};
  for (int dir = 0; dir < 4; dir++) {
    oracle_sga_scan_forward(x, gs[dir], temp_out, NULL, NC, D, H, W, dir);
    if (A_all) memcpy(A_all + dir * n, temp_out, sizeof(float) * n);
    if (dir == 0) {
#pragma omp parallel for schedule(static)
      for (i64 i = 0; i < n; i++) { out[i] = temp_out[i]; mask[i] = 0.f; }
    } else {
#pragma omp parallel for schedule(static)
      for (i64 i = 0; i < n; i++)
        if (out[i] < temp_out[i]) { out[i] = temp_out[i]; mask[i] = (float)dir; }
    }
  }
}

/* ---- SGA backward, one direction --------------------------------------- */
/* Restates, for one direction, the block of sga_kernel_backward (:1040-1128):
 *   G <- 0; get_temp_grad (:38-48): G = gradOut where (int)mask == dir
 *   MaxDepth (:50-64): idx = first-argmax_d A
 *   sga_*_data_backward (:129-208 and mirrors): reverse-scan adjoint,
 *       gradX += ... (including the reference's inexact first-position
 *       terms, F4: only w0 (+w2 at d=0, +w3 at d=D-1) reach the input)
 *   sga_*_weight_backward (:210-281 and mirrors): gw += five sum_d products;
 *       w1..w4 get no gradient at the first scan position (F4).
 * A is the directional forward volume (what `top_temp` holds at that point).
 * G (scratch, n floats) and idx (scratch, NC*H*W floats) are overwritten.
 * gradX and gw are ACCUMULATED into, as in the reference. */
void oracle_sga_backward_dir(const float *x, const float *g, const float *A,
                             const float *mask, const float *gradOut,
                             float *G, float *idx, float *gradX, float *gw,
                             int NC, int D, int H, int W, int dir)
{
  const i64 HW = (i64)H * W;
  const i64 n = (i64)NC * D * HW;
  const int L = scan_len(dir, H, W), Q = scan_lines(dir, H, W);

#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < n; i++) G[i] = ((int)mask[i] == dir) ? gradOut[i] : 0.f;

#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < (i64)NC * HW; i++) {
    const i64 base = i / HW * HW * D + i % HW;
    int k = 0;
    for (int d = 1; d < D; d++)
      if (A[base + k * HW] < A[base + d * HW]) k = d;
    idx[i] = (float)k;
  }

  /* data backward: one sequential program per scanline */
#pragma omp parallel for collapse(2) schedule(static)
  for (int s = 0; s < NC; s++) {
    for (int q = 0; q < Q; q++) {
      const float *gs = g + (i64)s * 5 * HW;
      float *Gs = G + (i64)s * D * HW;
      float *gx = gradX + (i64)s * D * HW;
      const float *is = idx + (i64)s * HW;
      for (int p = L - 1; p >= 0; p--) {
        const i64 o = scan_pos(dir, H, W, q, p);
        const int has_next = p + 1 < L;
        const i64 on = has_next ? scan_pos(dir, H, W, q, p + 1) : 0;
        for (int d = 0; d < D; d++) {
          float t = Gs[d * HW + o];
          if (has_next) t = fmaf(Gs[d * HW + on], gs[HW + on], t);
          if (has_next && d + 1 < D) t = fmaf(Gs[(d + 1) * HW + on], gs[2 * HW + on], t);
          if (has_next && d - 1 >= 0) t = fmaf(Gs[(d - 1) * HW + on], gs[3 * HW + on], t);
          Gs[d * HW + o] = t;
          gx[d * HW + o] = fmaf(t, gs[o], gx[d * HW + o]);
        }
        if (has_next) {
          const int k = (int)is[o];
          float t = 0.f;
          for (int d = 0; d < D; d++) t = fmaf(Gs[d * HW + on], gs[4 * HW + on], t);
          Gs[(i64)k * HW + o] += t;
          gx[(i64)k * HW + o] = fmaf(t, gs[o], gx[(i64)k * HW + o]);
        }
      }
      for (int p = 0; p < L; p++) {
        const i64 o = scan_pos(dir, H, W, q, p);
        gx[o] = fmaf(Gs[o], gs[2 * HW + o], gx[o]);
        const i64 top = (i64)(D - 1) * HW + o;
        gx[top] = fmaf(Gs[top], gs[3 * HW + o], gx[top]);
      }
    }
  }

  /* weight backward: one program per guidance pixel */
#pragma omp parallel for collapse(2) schedule(static)
  for (int s = 0; s < NC; s++) {
    for (int q = 0; q < Q; q++) {
      const float *xs = x + (i64)s * D * HW;
      const float *As = A + (i64)s * D * HW;
      const float *Gs = G + (i64)s * D * HW;
      const float *is = idx + (i64)s * HW;
      float *gws = gw + (i64)s * 5 * HW;
      for (int p = 0; p < L; p++) {
        const i64 o = scan_pos(dir, H, W, q, p);
        float a0 = gws[o];
        for (int d = 0; d < D; d++) a0 = fmaf(Gs[d * HW + o], xs[d * HW + o], a0);
        gws[o] = a0;
        if (p >= 1) {
          const i64 op = scan_pos(dir, H, W, q, p - 1);
          float a1 = gws[HW + o];
          for (int d = 0; d < D; d++) a1 = fmaf(Gs[d * HW + o], As[d * HW + op], a1);
          gws[HW + o] = a1;
          float a2 = gws[2 * HW + o];
          a2 = fmaf(Gs[o], xs[o], a2);
          for (int d = 1; d < D; d++) a2 = fmaf(Gs[d * HW + o], As[(d - 1) * HW + op], a2);
          gws[2 * HW + o] = a2;
          float a3 = gws[3 * HW + o];
          a3 = fmaf(Gs[(i64)(D - 1) * HW + o], xs[(i64)(D - 1) * HW + o], a3);
          for (int d = 0; d < D - 1; d++) a3 = fmaf(Gs[d * HW + o], As[(d + 1) * HW + op], a3);
          gws[3 * HW + o] = a3;
          const int k = (int)is[op];
          float a4 = gws[4 * HW + o];
          for (int d = 0; d < D; d++) a4 = fmaf(Gs[d * HW + o], As[(i64)k * HW + op], a4);
          gws[4 * HW + o] = a4;
        }
      }
    }
  }
}

/* ---- SGA backward, full ------------------------------------------------ */
/* Restates sga_kernel_backward (:1000-1129): direction order left(3, using
 * the saved temp_out = A_left, F6), down(0), up(1), right(2); the other three
 * directional volumes are recomputed into temp_out.  gradInput and the four
 * weight grads must be zero-filled by the caller (functions/GANet.py:33-37);
 * temp_grad [n] and max_idx [NC*H*W] are scratch. */
void oracle_sga_backward(const float *x, const float *g0, const float *g1,
                         const float *g2, const float *g3, float *temp_out,
                         const float *mask, float *max_idx, const float *gradOut,
                         float *temp_grad, float *gradInput, float *grad0,
                         float *grad1, float *grad2, float *grad3,
                         int N, int C, int D, int H, int W)
{
  const int NC = N * C;
  const float *gs[4] = {g0, g1, g2, g3};
  float *gw[4] = {grad0, grad1, grad2, grad3};
  const int order[4] = {3, 0, 1, 2};
  for (int i = 0; i < 4; i++) {
    const int dir = order[i];
    if (dir != 3)
      oracle_sga_scan_forward(x, gs[dir], temp_out, NULL, NC, D, H, W, dir);
    oracle_sga_backward_dir(x, gs[dir], temp_out, mask, gradOut, temp_grad,
                            max_idx, gradInput, gw[dir], NC, D, H, W, dir);
  }
}

/* ---- LGA --------------------------------------------------------------- */
/* lga_filtering_forward (:1131-1175).  Taps t = (dd+1)*K + (a+r)*(2r+1) + (b+r),
 * K = (2r+1)^2.  A neighbour out of range in depth, row OR column is replaced
 * by the centre value (:1162-1165).  `y` is ACCUMULATED into (`+=`, :1168),
 * so the caller zero-fills it (functions/GANet.py:181-182). */
void oracle_lga_forward(const float *x, const float *f, float *y,
                        int B, int D, int H, int W, int r)
{
  const i64 HW = (i64)H * W;
  const int ws = 2 * r + 1, K = ws * ws;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++) {
    for (int d = 0; d < D; d++) {
      const float *xb = x + (i64)b * D * HW;
      const float *fb = f + (i64)b * 3 * K * HW;
      float *yb = y + (i64)b * D * HW;
      for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
          const i64 c = d * HW + (i64)i * W + j;
          float acc = yb[c];
          for (int dd = -1; dd <= 1; dd++)
            for (int a = -r; a <= r; a++)
              for (int bb = -r; bb <= r; bb++) {
                const int d2 = d + dd, i2 = i + a, j2 = j + bb;
                i64 src = c;
                if (d2 >= 0 && d2 < D && i2 >= 0 && i2 < H && j2 >= 0 && j2 < W)
                  src = d2 * HW + (i64)i2 * W + j2;
                const int t = (dd + 1) * K + (a + r) * ws + (bb + r);
                acc = fmaf(xb[src], fb[t * HW + (i64)i * W + j], acc);
              }
          yb[c] = acc;
        }
    }
  }
}

/* lga_backward (:1299-1322) = lga_filter_backward (:1177-1216; gradFilters is
 * ACCUMULATED into) then gradInput <- 0 and lga_data_backward (:1218-1269). */
void oracle_lga_backward(const float *x, const float *f, const float *gy,
                         float *gx, float *gf, int B, int D, int H, int W, int r)
{
  const i64 HW = (i64)H * W;
  const int ws = 2 * r + 1, K = ws * ws;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++) {
    for (int t = 0; t < 3 * K; t++) {
      const float *xb = x + (i64)b * D * HW;
      const float *gb = gy + (i64)b * D * HW;
      float *gfb = gf + ((i64)b * 3 * K + t) * HW;
      const int dd = t / K - 1, a = (t / ws) % ws - r, bb = t % ws - r;
      for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
          const int i2 = i + a, j2 = j + bb;
          const int sp_ok = i2 >= 0 && i2 < H && j2 >= 0 && j2 < W;
          const i64 pix = (i64)i * W + j;
          float acc = gfb[pix];
          for (int d = 0; d < D; d++) {
            const int d2 = d + dd;
            i64 src = d * HW + pix;
            if (sp_ok && d2 >= 0 && d2 < D) src = d2 * HW + (i64)i2 * W + j2;
            acc = fmaf(gb[d * HW + pix], xb[src], acc);
          }
          gfb[pix] = acc;
        }
    }
  }
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++) {
    for (int d = 0; d < D; d++) {
      const float *fb = f + (i64)b * 3 * K * HW;
      const float *gb = gy + (i64)b * D * HW;
      float *gxb = gx + (i64)b * D * HW;
      for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
          const i64 pix = (i64)i * W + j;
          const i64 c = d * HW + pix;
          float acc = 0.f;
          for (int dd = -1; dd <= 1; dd++)
            for (int a = -r; a <= r; a++)
              for (int bb = -r; bb <= r; bb++) {
                const int d2 = d + dd, i2 = i + a, j2 = j + bb;
                if (d2 >= 0 && d2 < D && i2 >= 0 && i2 < H && j2 >= 0 && j2 < W) {
                  const int tf = (-dd + 1) * K + (-a + r) * ws + (-bb + r);
                  acc = fmaf(gb[d2 * HW + (i64)i2 * W + j2],
                             fb[tf * HW + (i64)i2 * W + j2], acc);
                } else {
                  const int t = (dd + 1) * K + (a + r) * ws + (bb + r);
                  acc = fmaf(gb[c], fb[t * HW + pix], acc);
                }
              }
          gxb[c] = acc;
        }
    }
  }
}

/* ---- GetCostVolume / DisparityRegression ------------------------------- */
/* libs/GANet/modules/GANet.py:119-134: cost[n,c,i,h,w] = x[n,c,h,w],
 * cost[n,C+c,i,h,w] = y[n,c,h,w-i] for w >= i, zero for w < i; Dn = maxdisp+1. */
void oracle_cost_volume(const float *x, const float *y, float *cost,
                        int N, int C, int Dn, int H, int W)
{
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; n++)
    for (int c = 0; c < 2 * C; c++)
      for (int i = 0; i < Dn; i++)
        for (int h = 0; h < H; h++)
          for (int w = 0; w < W; w++) {
            const i64 o = ((((i64)n * 2 * C + c) * Dn + i) * H + h) * W + w;
            float v = 0.f;
            if (w >= i) {
              if (c < C) v = x[(((i64)n * C + c) * H + h) * W + w];
              else v = y[(((i64)n * C + (c - C)) * H + h) * W + (w - i)];
            }
            cost[o] = v;
          }
}

/* libs/GANet/modules/GANet.py:142-148: out[n,h,w] = sum_d d * x[n,d,h,w]. */
void oracle_disparity_regression(const float *x, float *out, int N, int Dn, int H, int W)
{
  const i64 HW = (i64)H * W;
#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < (i64)N * HW; i++) {
    const i64 n = i / HW, pix = i % HW;
    float acc = 0.f;
    for (int d = 0; d < Dn; d++) acc += x[(n * Dn + d) * HW + pix] * (float)d;
    out[i] = acc;
  }
}
