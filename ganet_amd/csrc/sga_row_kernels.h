// sga_row_kernels.h -- horizontal SGA scans (right / left), one wavefront per image row.
//
// Why a second kernel family: along W the scan direction IS the contiguous axis.  The
// lane-per-(row,disparity-group) float4 kernels of sga_kernels.h touch 64 different
// cache lines per load and come back to every line 8 times; on MI355X the per-wave
// working set (4 rows x 65 planes x 5 arrays x 128 B = 166 KB) lives in neither the
// 32 KB L1 nor comfortably in the XCD's 4 MB L2, so each 16-byte access re-fetches a
// whole line from L2 (measured 0.92 ms per backward direction = 0.8 TB/s algorithmic).
//
// Here ONE wavefront owns ONE row (n,c,h) and all its D disparities:
//   * lane g keeps disparities [g*DPL, g*DPL+DPL) of the recurrence state in VGPRs
//     (D = 65 -> 33 lanes x 2);
//   * the row is walked in batches of SBH positions; for every batch the wave
//     cooperatively copies the [D][SBH] tiles of each input array from HBM to LDS
//     with fully coalesced 16-byte pieces (PP = SBH/4 lanes per plane segment), so a
//     cache line is pulled from L2 once per batch instead of once per position group;
//   * compute lanes read their (d, 4 positions) cells back with ds_read_b128, results
//     go to an LDS tile and leave with the same coalesced pattern;
//   * d+-1 halos and the max / arg-max / sums over disparity span the whole wave:
//     16-lane DPP butterflies + two ds_bpermute exchanges (xor 16, xor 32).
//   * backward computes the guidance-weight sums of position p+1 when it visits p
//     (that is when A[p] is in registers), so no look-ahead tile is needed.
// Arithmetic order of the forward recurrence is identical to sga_kernels.h (bit-exact).
#pragma once
#include "ga_common.h"

namespace ga {

struct RowGeom {
  int D, H, W;
  i64 HW;
};

template <int SBH> struct RowCfg {
  static constexpr int RS = SBH + 4;    // tile row stride in floats (16 B aligned, de-phased banks)
  static constexpr int PP = SBH / 4;    // 16-byte pieces per plane segment
  static constexpr int PPI = 64 / PP;   // plane segments covered by one wave-wide access
};

template <int DPL>
GA_DEV void fwd_row_step(const float (&xs)[DPL], const float (&w)[5], float (&A)[DPL], float &m,
                         bool first, int lane, int d0, int D)
{
  float An[DPL];
  if (first) {
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      float t = fmaf(xs[i], w[0], 0.0f);
      t = fmaf(xs[i], w[1], t);
      t = fmaf(xs[i], w[2], t);
      t = fmaf(xs[i], w[3], t);
      An[i] = fmaf(xs[i], w[4], t);
    }
  } else {
    const float lo = wave_from_prev(xs[0], A[DPL - 1], lane);
    const float hi = wave_from_next(xs[DPL - 1], A[0], lane);
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      const float P2 = i > 0 ? A[i - 1] : lo;
      float P3 = i < DPL - 1 ? A[i + 1] : hi;
      if (d0 + i + 1 >= D) P3 = xs[i];
      float t = fmaf(xs[i], w[0], 0.0f);
      t = fmaf(A[i], w[1], t);
      t = fmaf(P2, w[2], t);
      t = fmaf(P3, w[3], t);
      An[i] = fmaf(m, w[4], t);
    }
  }
  float mm = -INFINITY;
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    A[i] = An[i];
    if (d0 + i < D) mm = fmaxf(mm, An[i]);
  }
  m = wave_allmax(mm, lane);
}

// grid.x = S*H rows, block = 64.  desc: visit w = W-1 .. 0 (direction `left`).
// dynamic LDS: (2*D*RS + 5*SBH) floats.
template <int DPL, int SBH>
__global__ void __launch_bounds__(64)
sga_row_fwd(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ A,
            RowGeom geo, int desc)
{
  typedef RowCfg<SBH> C;
  GA_DYN_SMEM(smem);
  const int D = geo.D, W = geo.W;
  float *xt = smem;
  float *at = xt + D * C::RS;
  float *wt = at + D * C::RS;
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  const int s = row / geo.H, h = row - s * geo.H;
  const i64 rowoff = (i64)h * W;
  const float *xb = x + (i64)s * D * geo.HW + rowoff;
  float *Ab = A + (i64)s * D * geo.HW + rowoff;
  const float *gb = g + (i64)s * 5 * geo.HW + rowoff;
  const int d0 = lane * DPL;
  const int piece = lane % C::PP, psub = lane / C::PP;
  const int nb = (W + SBH - 1) / SBH;
  float Ap[DPL], m = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Ap[i] = 0.f;

  for (int b = 0; b < nb; b++) {
    const int w_lo = desc ? W - (b + 1) * SBH : b * SBH;
    const int wq = w_lo + 4 * piece;
    const bool col_ok = wq >= 0 && wq < W;
    for (int p0 = 0; p0 < D; p0 += C::PPI) {
      const int pl = p0 + psub;
      if (pl < D && col_ok)
        *reinterpret_cast<f4 *>(xt + pl * C::RS + 4 * piece) =
            *reinterpret_cast<const f4 *>(xb + (i64)pl * geo.HW + wq);
    }
    if (psub < 5 && col_ok)
      *reinterpret_cast<f4 *>(wt + psub * SBH + 4 * piece) =
          *reinterpret_cast<const f4 *>(gb + (i64)psub * geo.HW + wq);
    __syncthreads();
#pragma unroll
    for (int kq = 0; kq < C::PP; kq++) {
      if (b * SBH + 4 * kq < W) {
        const int cq = desc ? C::PP - 1 - kq : kq;
        f4 xv[DPL], wv[5], ov[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          const int d = d0 + i < D ? d0 + i : D - 1;
          xv[i] = *reinterpret_cast<const f4 *>(xt + d * C::RS + 4 * cq);
        }
#pragma unroll
        for (int t = 0; t < 5; t++) wv[t] = *reinterpret_cast<const f4 *>(wt + t * SBH + 4 * cq);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int kk = desc ? 3 - k : k;
          float xs[DPL], w[5];
#pragma unroll
          for (int i = 0; i < DPL; i++) xs[i] = f4_get(xv[i], kk);
#pragma unroll
          for (int t = 0; t < 5; t++) w[t] = f4_get(wv[t], kk);
          fwd_row_step<DPL>(xs, w, Ap, m, b == 0 && kq == 0 && k == 0, lane, d0, D);
#pragma unroll
          for (int i = 0; i < DPL; i++) f4_set(ov[i], kk, Ap[i]);
        }
#pragma unroll
        for (int i = 0; i < DPL; i++)
          if (d0 + i < D) *reinterpret_cast<f4 *>(at + (d0 + i) * C::RS + 4 * cq) = ov[i];
      }
    }
    __syncthreads();
    for (int p0 = 0; p0 < D; p0 += C::PPI) {
      const int pl = p0 + psub;
      if (pl < D && col_ok)
        *reinterpret_cast<f4 *>(Ab + (i64)pl * geo.HW + wq) =
            *reinterpret_cast<const f4 *>(at + pl * C::RS + 4 * piece);
    }
    __syncthreads();
  }
}

// ---- backward -------------------------------------------------------------------------
struct RowCarry {
  float wn[5];   // guidance at the previously visited position (forward p+1)
  float sgn;     // sum_d G[p+1][d]
  float s0n;     // sum_d G[p+1][d] * x[p+1][d]
};

// Visit of forward position p (visit order = reverse scan).  Emits the guidance-weight sums
// of the PREVIOUS visit (position p+1) in gwprev when has_nx, because they need A[p].
template <int DPL>
GA_DEV void bwd_row_step(const float (&go)[DPL], const int (&mk)[DPL], const float (&xs)[DPL],
                         const float (&Ac)[DPL], const float (&w)[5], float (&Gn)[DPL],
                         float (&xp)[DPL], RowCarry &cy, float (&gxo)[DPL], float (&gwprev)[5],
                         bool has_nx, int lane, int d0, int D, int dir)
{
  float G[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) G[i] = (d0 + i < D && mk[i] == dir) ? go[i] : 0.f;
  if (has_nx) {
    // first arg-max over d of A[p] (routing target) and its value (5th tap of position p+1)
    float mv = -INFINITY;
    int cand = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < DPL; i++)
      if (d0 + i < D && Ac[i] > mv) { mv = Ac[i]; cand = d0 + i; }
    const float wm = wave_allmax(mv, lane);
    const int kp = wave_allmin_i(mv == wm ? cand : 0x7fffffff, lane);
    // guidance-weight sums of position p+1: G[p+1] (= Gn) against A[p] and x[p+1] (= xp)
    const float alo = wave_from_prev(0.f, Ac[DPL - 1], lane);
    const float ahi = wave_from_next(0.f, Ac[0], lane);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      const int d = d0 + i;
      const float a2 = d >= 1 ? (i > 0 ? Ac[i - 1] : alo) : xp[i];
      const float a3 = d + 1 < D ? (i < DPL - 1 ? Ac[i + 1] : ahi) : xp[i];
      s1 = fmaf(Gn[i], Ac[i], s1);
      s2 = fmaf(Gn[i], a2, s2);
      s3 = fmaf(Gn[i], a3, s3);
    }
    gwprev[0] = cy.s0n;
    gwprev[1] = wave_allsum(s1, lane);
    gwprev[2] = wave_allsum(s2, lane);
    gwprev[3] = wave_allsum(s3, lane);
    gwprev[4] = cy.sgn * wm;
    // reverse-scan adjoint
    const float lo = wave_from_prev(0.f, Gn[DPL - 1], lane);
    const float hi = wave_from_next(0.f, Gn[0], lane);
    const float t4 = cy.wn[4] * cy.sgn;
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      const float up = i < DPL - 1 ? Gn[i + 1] : hi;
      const float dn = i > 0 ? Gn[i - 1] : lo;
      float t = G[i];
      t = fmaf(Gn[i], cy.wn[1], t);
      t = fmaf(up, cy.wn[2], t);
      t = fmaf(dn, cy.wn[3], t);
      if (d0 + i == kp) t += t4;
      G[i] = (d0 + i < D) ? t : 0.f;
    }
  }
  float s0 = 0.f, sg = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    float r = G[i] * w[0];
    if (d0 + i == 0) r = fmaf(G[i], w[2], r);
    if (d0 + i == D - 1) r = fmaf(G[i], w[3], r);
    gxo[i] = r;
    s0 = fmaf(G[i], xs[i], s0);
    sg += G[i];
    Gn[i] = G[i];
    xp[i] = xs[i];
  }
#pragma unroll
  for (int t = 0; t < 5; t++) cy.wn[t] = w[t];
  cy.s0n = wave_allsum(s0, lane);
  cy.sgn = wave_allsum(sg, lane);
}

// grid.x = S*H rows, block = 64.  desc: VISIT order w = W-1..0 (backward of `right`).
// dynamic LDS floats: 4*D*RS (go, x, A, gx) + roundup4(D*(PP+1)) (mask words) + 2*5*SBH (w, gw).
template <int DPL, int SBH>
__global__ void __launch_bounds__(64)
sga_row_bwd(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ A,
            const uint8_t *__restrict__ mask, const float *__restrict__ gout,
            float *__restrict__ gradX, float *__restrict__ gw, RowGeom geo, int dir, int accumulate,
            int desc)
{
  typedef RowCfg<SBH> C;
  GA_DYN_SMEM(smem);
  const int D = geo.D, W = geo.W;
  constexpr int MS = C::PP + 1;
  float *got = smem;
  float *xt = got + D * C::RS;
  float *at = xt + D * C::RS;
  float *gxt = at + D * C::RS;
  uint32_t *mt = reinterpret_cast<uint32_t *>(gxt + D * C::RS);
  float *wt = reinterpret_cast<float *>(mt + ((D * MS + 3) & ~3));   // keep 16-byte alignment
  float *gwt = wt + 5 * SBH;
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  const int s = row / geo.H, h = row - s * geo.H;
  const i64 rowoff = (i64)h * W;
  const i64 vbase = (i64)s * D * geo.HW + rowoff;
  const float *xb = x + vbase, *Ab = A + vbase, *gob = gout + vbase;
  const uint8_t *mb = mask + vbase;
  float *gxb = gradX + vbase;
  const float *gb = g + (i64)s * 5 * geo.HW + rowoff;
  float *gwb = gw + (i64)s * 5 * geo.HW + rowoff;
  const int d0 = lane * DPL;
  const int piece = lane % C::PP, psub = lane / C::PP;
  const int nb = (W + SBH - 1) / SBH;
  float Gn[DPL], xp[DPL];
  RowCarry cy;
#pragma unroll
  for (int i = 0; i < DPL; i++) { Gn[i] = 0.f; xp[i] = 0.f; }
#pragma unroll
  for (int t = 0; t < 5; t++) cy.wn[t] = 0.f;
  cy.sgn = 0.f;
  cy.s0n = 0.f;

  for (int b = 0; b < nb; b++) {
    const int w_lo = desc ? W - (b + 1) * SBH : b * SBH;
    const int wq = w_lo + 4 * piece;
    const bool col_ok = wq >= 0 && wq < W;
    for (int p0 = 0; p0 < D; p0 += C::PPI) {
      const int pl = p0 + psub;
      if (pl < D && col_ok) {
        const i64 o = (i64)pl * geo.HW + wq;
        const int lo = pl * C::RS + 4 * piece;
        *reinterpret_cast<f4 *>(got + lo) = *reinterpret_cast<const f4 *>(gob + o);
        *reinterpret_cast<f4 *>(xt + lo) = *reinterpret_cast<const f4 *>(xb + o);
        *reinterpret_cast<f4 *>(at + lo) = *reinterpret_cast<const f4 *>(Ab + o);
        if (accumulate) *reinterpret_cast<f4 *>(gxt + lo) = *reinterpret_cast<const f4 *>(gxb + o);
        mt[pl * MS + piece] = *reinterpret_cast<const uint32_t *>(mb + o);
      }
    }
    if (psub < 5 && col_ok)
      *reinterpret_cast<f4 *>(wt + psub * SBH + 4 * piece) =
          *reinterpret_cast<const f4 *>(gb + (i64)psub * geo.HW + wq);
    __syncthreads();
#pragma unroll
    for (int kq = 0; kq < C::PP; kq++) {
      if (b * SBH + 4 * kq < W) {
        const int cq = desc ? C::PP - 1 - kq : kq;
        f4 gov[DPL], xv[DPL], av[DPL], gxv[DPL], wv[5];
        uint32_t mw[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          const int d = d0 + i < D ? d0 + i : D - 1;
          const int lo = d * C::RS + 4 * cq;
          gov[i] = *reinterpret_cast<const f4 *>(got + lo);
          xv[i] = *reinterpret_cast<const f4 *>(xt + lo);
          av[i] = *reinterpret_cast<const f4 *>(at + lo);
          if (accumulate) gxv[i] = *reinterpret_cast<const f4 *>(gxt + lo);
          else { gxv[i].x = 0.f; gxv[i].y = 0.f; gxv[i].z = 0.f; gxv[i].w = 0.f; }
          mw[i] = mt[d * MS + cq];
        }
#pragma unroll
        for (int t = 0; t < 5; t++) wv[t] = *reinterpret_cast<const f4 *>(wt + t * SBH + 4 * cq);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int kk = desc ? 3 - k : k;
          const int j = 4 * kq + k;                       // visit index inside the batch
          float go[DPL], xs[DPL], Ac[DPL], w[5], gxo[DPL], gwprev[5];
          int mk[DPL];
#pragma unroll
          for (int i = 0; i < DPL; i++) {
            go[i] = f4_get(gov[i], kk);
            xs[i] = f4_get(xv[i], kk);
            Ac[i] = f4_get(av[i], kk);
            mk[i] = (int)((mw[i] >> (8 * kk)) & 0xffu);
          }
#pragma unroll
          for (int t = 0; t < 5; t++) { w[t] = f4_get(wv[t], kk); gwprev[t] = 0.f; }
          const bool has_nx = !(b == 0 && j == 0);
          bwd_row_step<DPL>(go, mk, xs, Ac, w, Gn, xp, cy, gxo, gwprev, has_nx, lane, d0, D, dir);
#pragma unroll
          for (int i = 0; i < DPL; i++) f4_set(gxv[i], kk, f4_get(gxv[i], kk) + gxo[i]);
          if (lane < 5) {
            // lane t keeps tap t (all lanes hold identical sums)
            const float v = lane == 0 ? gwprev[0] : lane == 1 ? gwprev[1] : lane == 2 ? gwprev[2]
                            : lane == 3 ? gwprev[3] : gwprev[4];
            gwt[lane * SBH + j] = v;
          }
        }
#pragma unroll
        for (int i = 0; i < DPL; i++)
          if (d0 + i < D) *reinterpret_cast<f4 *>(gxt + (d0 + i) * C::RS + 4 * cq) = gxv[i];
      }
    }
    __syncthreads();
    for (int p0 = 0; p0 < D; p0 += C::PPI) {
      const int pl = p0 + psub;
      if (pl < D && col_ok)
        *reinterpret_cast<f4 *>(gxb + (i64)pl * geo.HW + wq) =
            *reinterpret_cast<const f4 *>(gxt + pl * C::RS + 4 * piece);
    }
    // guidance grads of visits [b*SBH - 1, b*SBH + SBH - 2]: slot j holds visit b*SBH + j - 1
    for (int e = lane; e < 5 * SBH; e += 64) {
      const int t = e / SBH, j = e - t * SBH;
      const int v = b * SBH + j, u = v - 1;
      if (v < W && u >= 0) {
        const int wu = desc ? W - 1 - u : u;
        gwb[(i64)t * geo.HW + wu] = gwt[t * SBH + j];
      }
    }
    __syncthreads();
  }
  // last visited position (forward position 0): only w0 receives a gradient
  if (lane < 5) {
    const int wu = desc ? 0 : W - 1;
    gwb[(i64)lane * geo.HW + wu] = lane == 0 ? cy.s0n : 0.f;
  }
}

}  // namespace ga
