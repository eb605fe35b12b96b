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
//   * the recurrence state lives in ONE 16-lane DPP row: lane g < 16 keeps disparities
//     [g*DPL, g*DPL+DPL) in VGPRs (D = 65 -> 13 lanes x 5), so every cross-lane step is a
//     row-local DPP op (first version spread D over all 64 lanes and paid ~18 ds_bpermute
//     round trips per position: 58 % of wave time in s_waitcnt, profiles/r1b_pmc_summary.txt);
//     lanes 16..63 mirror row 0 and only add bandwidth to the tile copies;
//   * the row is walked in batches of SBH positions; for every batch the wave
//     cooperatively copies the [D][SBH] tiles of each input array from HBM to LDS
//     with fully coalesced 16-byte pieces (PP = SBH/4 lanes per plane segment), so a
//     cache line is pulled from L2 once per batch instead of once per position group;
//   * compute lanes read their (d, 4 positions) cells back with ds_read_b128, results
//     go to an LDS tile and leave with the same coalesced pattern;
//   * d+-1 halos: DPP row_shr/row_shl; max / arg-max / sums over disparity: 4-step DPP
//     butterflies folded into the VALU op (v_max_f32_dpp, v_add_f32_dpp);
// Arithmetic order of the forward recurrence is identical to sga_kernels.h (bit-exact).
#pragma once
#include "ga_common.h"
#include "sga_kernels.h"

namespace ga {


#ifndef GA_ROW_ALIGN
#define GA_ROW_ALIGN 1   // 0: batches counted from the row start (A/B only)
#endif

struct RowGeom {
  int D, H, W;
  int total_rows;   // S * H
  i64 HW;
  // forward scans only -- what the copy-out does with the finished tile (inference path, ganet_sga_forward_infer):
  //   0  A = tile;   1  A = max(A, tile)  (running direction max, A is the output volume);
  //   2  A = relu(scale[c] * max(A, tile) + shift[c])  (last direction + eval-mode BatchNorm3d + ReLU)
  int out_mode;
  int C;                       // channels per sample (slice % C indexes scale / shift)
  const float *scale, *shift;
};

template <int SBH, int PAD> struct RowCfg {
  static constexpr int RS = SBH + PAD;  // tile row stride in floats (PAD = 4 de-phases banks, 0 saves LDS)
  static constexpr int PP = SBH / 4;    // 16-byte pieces per plane segment
  static constexpr int PPI = 64 / PP;   // plane segments covered by one wave-wide access
};

// grid.x = ceil(S*H / LN), block = 64: DPP row r of the wave owns image row blockIdx.x*LN + r
// (rows >= LN mirror row r % LN).  desc: visit w = W-1 .. 0 (direction `left`).
// dynamic LDS: LN * (D*RS + 5*SBH) floats (the A tile aliases the x tile).
// GD = 64, NPCT: see sga_row_bwdg.
template <int DPL, int SBH, int PAD, int LN, bool desc, bool FULL, int GD = 16, int NPCT = 0>
__global__ void __launch_bounds__(64, (GD == 64 && DPL == 1 ? 4 : DPL <= 5 ? 3 : 1))   // (<= 168 VGPRs incl. the prefetched tile where that fits)
sga_row_fwd(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ A,
            RowGeom geo)
{
  typedef RowCfg<SBH, PAD> C;
  GA_DYN_SMEM(smem);
  const int D = geo.D, W = geo.W;
  const int total_rows = geo.total_rows;
  float *xt = smem;                       // [LN][D][RS]
  float *at = xt;                         // A overwrites the x tile in place (a lane rewrites
                                          // exactly the cells it read; mirror rows do not write)
  float *wt = xt + LN * D * C::RS;        // [LN][5][SBH]
  static_assert(GD == 16 || (GD == 64 && LN == 1), "the depth axis lies in one DPP row (mirrored) or over the whole wavefront");
  const int lane = threadIdx.x;
  const int rl = GD == 64 ? lane : lane & 15;
  const int r = GD == 64 ? 0 : (lane >> 4) % LN;
  const bool owner = GD == 64 || (lane >> 4) < LN;    // this lane's DPP row carries row r (not a mirror)
  const int d0 = rl * DPL;
  LaneCtx c;
  c.lg = rl; c.d0 = d0; c.line_ok = true; c.s = 0; c.q = 0; c.cap = lane_cap(c.d0, D);
  const int piece = lane % C::PP, psub = lane / C::PP;
  i64 vb[LN], gbo[LN];
  float bn_sc[LN], bn_sh[LN];
  bool rok[LN];
#pragma unroll
  for (int q = 0; q < LN; q++) {
    int row = xcd_remap(blockIdx.x, gridDim.x) * LN + q;   // neighbouring rows share lines at their ends
    rok[q] = row < total_rows;
    if (!rok[q]) row = total_rows - 1;
    const int s = row / geo.H, h = row - s * geo.H;
    vb[q] = (i64)s * D * geo.HW + (i64)h * W;
    gbo[q] = (i64)s * 5 * geo.HW + (i64)h * W;
    bn_sc[q] = 1.f; bn_sh[q] = 0.f;
    if (geo.out_mode == 2) { bn_sc[q] = geo.scale[s % geo.C]; bn_sh[q] = geo.shift[s % geo.C]; }
  }
  // Line-aligned batches: a batch covers SBH elements that start on a multiple of SBH elements of the
  // ADDRESS (one or two whole 128-byte lines per plane), not of the row.  Rows of W = 208 floats start
  // 64 B into a line every other row; batches counted from the row start then straddle two lines per
  // piece and every line is pulled from the fabric twice, 18 us apart -- too long for the XCD's L2 to
  // keep it (profiles/r1m_pmc_memory_side.txt: 227 MB read per scan for 149 MB of input).  The first
  // and the last batch of a row may therefore be partial.
  static_assert(LN == 1 || !GA_ROW_ALIGN, "line-aligned batching assumes one image row per wavefront");
  const int a0 = GA_ROW_ALIGN ? (int)(((reinterpret_cast<uintptr_t>(x) >> 2) + (uintptr_t)vb[0]) & (uintptr_t)(SBH - 1)) : 0;
  const int nb = (W + a0 + SBH - 1) / SBH;
  float Ap[DPL], m = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Ap[i] = 0.f;
  float *xr = xt + r * D * C::RS, *ar = at + r * D * C::RS, *wr = wt + r * 5 * SBH;

  // Software pipeline: the next batch's tile is requested (into registers) BEFORE the current
  // batch is computed and committed to LDS after it, so one HBM round trip overlaps 32 positions
  // of recurrence instead of preceding them (written the obvious way -- load, ds_write, next piece
  // -- hipcc emits load / s_waitcnt vmcnt(0) / ds_write per piece: nine serial round trips per
  // batch; scripts/isa_lint.py counts such patterns).  Loads are unconditional (addresses clamped
  // into the row, the clamped copies are never committed): a conditional load makes the compiler
  // guard every later use of its registers with vmcnt(0).
  constexpr int NPC = NPCT ? NPCT : (GD * DPL + C::PPI - 1) / C::PPI;      // pieces per lane and image row
  float xpre[LN][NPC][4], wpre[LN][4];       // (plain floats: an f4 array carried around the loop stays in scratch)
  auto batch_col0 = [&](int b) { return (desc ? nb - 1 - b : b) * SBH - a0; };     // column of the batch's first element (may be < 0)
  auto batch_col = [&](int b) { return batch_col0(b) + 4 * piece; };
  auto prefetch = [&](int b) {
    int wq = batch_col(b < nb ? b : nb - 1);
    wq = wq < 0 ? 0 : (wq > W - 4 ? W - 4 : wq);
#pragma unroll
    for (int q = 0; q < LN; q++) {
      const float *xb = x + vb[q] + wq;
#pragma unroll
      for (int n = 0; n < NPC; n++) {
        int pl = n * C::PPI + psub;
        pl = pl < D ? pl : D - 1;
        const f4 t = stream_load<(GA_NT_LOADS & 4) != 0>(reinterpret_cast<const f4 *>(xb + (i64)pl * geo.HW));
        xpre[q][n][0] = t.x; xpre[q][n][1] = t.y; xpre[q][n][2] = t.z; xpre[q][n][3] = t.w;
      }
      const f4 t = *reinterpret_cast<const f4 *>(g + gbo[q] + (i64)(psub < 5 ? psub : 4) * geo.HW + wq);
      wpre[q][0] = t.x; wpre[q][1] = t.y; wpre[q][2] = t.z; wpre[q][3] = t.w;
    }
  };
  auto commit = [&](int b) {
    const int wq = batch_col(b);
    const bool col_ok = wq >= 0 && wq < W;
#pragma unroll
    for (int q = 0; q < LN; q++) {
#pragma unroll
      for (int n = 0; n < NPC; n++) {
        const int pl = n * C::PPI + psub;
        if (pl < D && col_ok)
          *reinterpret_cast<f4 *>(xt + (q * D + pl) * C::RS + 4 * piece) =
              f4{xpre[q][n][0], xpre[q][n][1], xpre[q][n][2], xpre[q][n][3]};
      }
      if (psub < 5 && col_ok)
        *reinterpret_cast<f4 *>(wt + (q * 5 + psub) * SBH + 4 * piece) = f4{wpre[q][0], wpre[q][1], wpre[q][2], wpre[q][3]};
    }
  };

  prefetch(0);
  bool started = false;                    // a position of this row has been visited (uniform)
  for (int b = 0; b < nb; b++) {
    const int wq = batch_col(b);
    const bool col_ok = wq >= 0 && wq < W;
    const int bc0 = batch_col0(b);
    commit(b);
    GA_WAVE_SYNC();   // the workgroup is one wavefront: its LDS queue is in order, nothing to drain
    prefetch(b + 1);
#pragma unroll
    for (int kq = 0; kq < C::PP; kq++) {
      const int cq = desc ? C::PP - 1 - kq : kq;
      const int gc = bc0 + 4 * cq;           // image column of this group of 4 positions (W % 4 == 0: in or out as a whole)
      if (gc >= 0 && gc < W) {
        const bool first_group = !started;
        started = true;
        f4 xv[DPL], wv[5], ov[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          const int d = d0 + i < D ? d0 + i : D - 1;
          xv[i] = *reinterpret_cast<const f4 *>(xr + d * C::RS + 4 * cq);
        }
#pragma unroll
        for (int t = 0; t < 5; t++) wv[t] = *reinterpret_cast<const f4 *>(wr + t * SBH + 4 * cq);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int kk = desc ? 3 - k : k;
          float xs[DPL], w[5];
#pragma unroll
          for (int i = 0; i < DPL; i++) xs[i] = f4_get(xv[i], kk);
#pragma unroll
          for (int t = 0; t < 5; t++) w[t] = f4_get(wv[t], kk);
          fwd_step<GD, DPL, FULL>(xs, w, Ap, m, k == 0 && first_group, c, D);
#pragma unroll
          for (int i = 0; i < DPL; i++) f4_set(ov[i], kk, Ap[i]);
        }
#pragma unroll
        for (int i = 0; i < DPL; i++)
          if (owner && d0 + i < D) *reinterpret_cast<f4 *>(ar + (d0 + i) * C::RS + 4 * cq) = ov[i];
      }
    }
    GA_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < LN; q++) {
      float *Ab = A + vb[q];
#pragma unroll
      for (int n = 0; n < NPC; n++) {
        const int pl = n * C::PPI + psub;
        if (pl < D && col_ok && rok[q]) {
          f4 v = *reinterpret_cast<const f4 *>(at + (q * D + pl) * C::RS + 4 * piece);
          if (geo.out_mode) {                      // (uniform) running max into the output volume
            const f4 o = *reinterpret_cast<const f4 *>(Ab + (i64)pl * geo.HW + wq);
            v.x = v.x < o.x ? o.x : v.x; v.y = v.y < o.y ? o.y : v.y;
            v.z = v.z < o.z ? o.z : v.z; v.w = v.w < o.w ? o.w : v.w;
            if (geo.out_mode == 2) {
              v.x = fmaxf(fmaf(v.x, bn_sc[q], bn_sh[q]), 0.f); v.y = fmaxf(fmaf(v.y, bn_sc[q], bn_sh[q]), 0.f);
              v.z = fmaxf(fmaf(v.z, bn_sc[q], bn_sh[q]), 0.f); v.w = fmaxf(fmaf(v.w, bn_sc[q], bn_sh[q]), 0.f);
            }
            *reinterpret_cast<f4 *>(Ab + (i64)pl * geo.HW + wq) = v;      // (read back by the next direction's scan: a plain store)
          } else {
            stream_store<(GA_NT_STORES & 2) != 0>(reinterpret_cast<f4 *>(Ab + (i64)pl * geo.HW + wq), v);
          }
        }
      }
    }
    GA_WAVE_SYNC();
  }
}

// ---- adjoint scan (backward step 1, see sga_kernels.h) with the same row staging --------------
// grid.x = ceil(S*H / LN), block = 64.  desc: VISIT order w = W-1..0 (adjoint of `right`).
// The G tile overwrites the gradOut tile in place (a lane rewrites exactly the cells it read).
// dynamic LDS per image row: (D+1)*RS ([mask == dir] * gradOut -> G; one row of zeros) + 5*SBH (w) + SBH/2 words (kp as uint16).
// The direction mask is applied when a tile is committed to LDS -- 64 lanes, 64 different pieces -- not in the recurrence,
// whose 16 lanes are mirrored four times: 2 instead of 15 instructions per position, and no mask tile.
// GD = 64: the depth axis over all 64 lanes (DPL disparities each, D <= 64 * DPL) instead of over one 16-lane DPP row that is
// mirrored in the other three -- 2.5x fewer elements per lane at D = 65 for three more cross-lane steps per position; NPCT: the
// planes a lane stages per batch, ceil(D_max / PPI), when that is less than GD * DPL / PPI.
template <int DPL, int SBH, int PAD, int LN, bool desc, bool FULL, int GD = 16, int NPCT = 0>     // FULL: D % DPL == 0 (bwdg_step)
__global__ void __launch_bounds__(64, (GD == 64 ? 4 : DPL <= 5 ? 3 : 1))
sga_row_bwdg(const float *__restrict__ g, const uint8_t *__restrict__ mask,
             const uint16_t *__restrict__ kp, const float *__restrict__ gout,
             float *__restrict__ G, RowGeom geo, int dir)
{
  typedef RowCfg<SBH, PAD> C;
  GA_DYN_SMEM(smem);
  const int D = geo.D, W = geo.W;
  const int total_rows = geo.total_rows;
  const int TS = (D + 1) * C::RS;
  float *gt = smem;                                            // [LN][D + 1][RS]: the MASKED gradient ([mask == dir] * gradOut,
                                                               // applied by commit()), G in place; row D stays zero: what the
                                                               // elements outside [0, D) of the last lanes read
  float *wt = gt + LN * TS;                                    // [LN][5][SBH]
  uint32_t *kt = reinterpret_cast<uint32_t *>(wt + LN * 5 * SBH);   // [LN][SBH/2]
  static_assert(GD == 16 || (GD == 64 && LN == 1), "the depth axis lies in one DPP row (mirrored) or over the whole wavefront");
  const int lane = threadIdx.x;
  const int rl = GD == 64 ? lane : lane & 15;
  const int r = GD == 64 ? 0 : (lane >> 4) % LN;
  const bool owner = GD == 64 || (lane >> 4) < LN;
  LaneCtx c;
  c.lg = rl; c.d0 = rl * DPL; c.line_ok = true; c.s = 0; c.q = 0; c.cap = lane_cap(c.d0, D);
  const int piece = lane % C::PP, psub = lane / C::PP;
  i64 vb[LN], gbo[LN], kbo[LN];
  bool rok[LN];
#pragma unroll
  for (int q = 0; q < LN; q++) {
    int row = xcd_remap(blockIdx.x, gridDim.x) * LN + q;   // neighbouring rows share lines at their ends
    rok[q] = row < total_rows;
    if (!rok[q]) row = total_rows - 1;
    const int s = row / geo.H, h = row - s * geo.H;
    vb[q] = (i64)s * D * geo.HW + (i64)h * W;
    gbo[q] = (i64)s * 5 * geo.HW + (i64)h * W;
    kbo[q] = (i64)s * geo.HW + (i64)h * W;
  }
  // line-aligned batches, see sga_row_fwd (alignment taken from the gradOut volume, the widest stream)
  static_assert(LN == 1 || !GA_ROW_ALIGN, "line-aligned batching assumes one image row per wavefront");
  const int a0 = GA_ROW_ALIGN ? (int)(((reinterpret_cast<uintptr_t>(gout) >> 2) + (uintptr_t)vb[0]) & (uintptr_t)(SBH - 1)) : 0;
  const int nb = (W + a0 + SBH - 1) / SBH;
  float Gn[DPL], wn[5], sgn = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Gn[i] = 0.f;
#pragma unroll
  for (int t = 0; t < 5; t++) wn[t] = 0.f;
  float *gr = gt + r * TS;
  const uint32_t *kr = kt + r * (SBH / 2);
  const float *wr = wt + r * 5 * SBH;
  if (lane < C::RS) {
#pragma unroll
    for (int q = 0; q < LN; q++) gt[q * TS + D * C::RS + lane] = 0.f;
  }

  // software pipeline as in sga_row_fwd: the next batch's gradOut / mask / guidance / arg-max pieces
  // are requested before the current batch is computed and committed to LDS after it
  constexpr int NPC = NPCT ? NPCT : (GD * DPL + C::PPI - 1) / C::PPI;
  float gpre[LN][NPC][4], wpre[LN][4];
  uint32_t mpre[LN][NPC], kpre[LN][2];
  auto batch_col0 = [&](int b) { return (desc ? nb - 1 - b : b) * SBH - a0; };
  auto batch_col = [&](int b) { return batch_col0(b) + 4 * piece; };
  auto prefetch = [&](int b) {
    int wq = batch_col(b < nb ? b : nb - 1);
    wq = wq < 0 ? 0 : (wq > W - 4 ? W - 4 : wq);
#pragma unroll
    for (int q = 0; q < LN; q++) {
#pragma unroll
      for (int n = 0; n < NPC; n++) {
        int pl = n * C::PPI + psub;
        pl = pl < D ? pl : D - 1;
        const i64 o = vb[q] + (i64)pl * geo.HW + wq;
        const f4 t = stream_load<(GA_NT_LOADS & 4) != 0>(reinterpret_cast<const f4 *>(gout + o));
        gpre[q][n][0] = t.x; gpre[q][n][1] = t.y; gpre[q][n][2] = t.z; gpre[q][n][3] = t.w;
        mpre[q][n] = *reinterpret_cast<const uint32_t *>(mask + o);
      }
      const f4 t = *reinterpret_cast<const f4 *>(g + gbo[q] + (i64)(psub < 5 ? psub : 4) * geo.HW + wq);
      wpre[q][0] = t.x; wpre[q][1] = t.y; wpre[q][2] = t.z; wpre[q][3] = t.w;
      const uint2 kk2 = *reinterpret_cast<const uint2 *>(kp + kbo[q] + wq);   // 4 x uint16
      kpre[q][0] = kk2.x; kpre[q][1] = kk2.y;
    }
  };
  auto commit = [&](int b) {
    const int wq = batch_col(b);
    const bool col_ok = wq >= 0 && wq < W;
#pragma unroll
    for (int q = 0; q < LN; q++) {
#pragma unroll
      for (int n = 0; n < NPC; n++) {
        const int pl = n * C::PPI + psub;
        if (pl < D && col_ok) {
          const uint32_t m = mpre[q][n];       // the winning direction of the piece's four elements, one byte each
          *reinterpret_cast<f4 *>(gt + q * TS + pl * C::RS + 4 * piece) =
              f4{(int)(m & 0xffu) == dir ? gpre[q][n][0] : 0.f, (int)((m >> 8) & 0xffu) == dir ? gpre[q][n][1] : 0.f,
                 (int)((m >> 16) & 0xffu) == dir ? gpre[q][n][2] : 0.f, (int)(m >> 24) == dir ? gpre[q][n][3] : 0.f};
        }
      }
      if (psub < 5 && col_ok)
        *reinterpret_cast<f4 *>(wt + (q * 5 + psub) * SBH + 4 * piece) = f4{wpre[q][0], wpre[q][1], wpre[q][2], wpre[q][3]};
      if (psub == 5 && col_ok) {
        kt[q * (SBH / 2) + 2 * piece] = kpre[q][0];
        kt[q * (SBH / 2) + 2 * piece + 1] = kpre[q][1];
      }
    }
  };

  prefetch(0);
  bool started = false;
  for (int b = 0; b < nb; b++) {
    const int wq = batch_col(b);
    const bool col_ok = wq >= 0 && wq < W;
    const int bc0 = batch_col0(b);
    commit(b);
    GA_WAVE_SYNC();   // the workgroup is one wavefront: its LDS queue is in order, nothing to drain
    prefetch(b + 1);
#pragma unroll
    for (int kq = 0; kq < C::PP; kq++) {
      const int cq = desc ? C::PP - 1 - kq : kq;
      const int gc = bc0 + 4 * cq;
      if (gc >= 0 && gc < W) {
        const bool first_group = !started;
        started = true;
        f4 gov[DPL], wv[5], ov[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          const int d = c.d0 + i < D ? c.d0 + i : D;         // (row D: zeros)
          gov[i] = *reinterpret_cast<const f4 *>(gr + d * C::RS + 4 * cq);
        }
#pragma unroll
        for (int t = 0; t < 5; t++) wv[t] = *reinterpret_cast<const f4 *>(wr + t * SBH + 4 * cq);
        const uint32_t k01 = kr[2 * cq], k23 = kr[2 * cq + 1];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int kk = desc ? 3 - k : k;
          float go[DPL], w[5];
          const uint8_t mk[DPL] = {};          // (not read: the tile holds the masked gradient)
#pragma unroll
          for (int i = 0; i < DPL; i++) go[i] = f4_get(gov[i], kk);
#pragma unroll
          for (int t = 0; t < 5; t++) w[t] = f4_get(wv[t], kk);
          const int kpv = (int)(((kk < 2 ? k01 : k23) >> (16 * (kk & 1))) & 0xffffu);
          bwdg_step<GD, DPL, uint8_t, FULL, true>(go, mk, Gn, wn, sgn, w, kpv, !(k == 0 && first_group), c, D, dir);
#pragma unroll
          for (int i = 0; i < DPL; i++) f4_set(ov[i], kk, Gn[i]);
        }
#pragma unroll
        for (int i = 0; i < DPL; i++)
          if (owner && c.d0 + i < D) *reinterpret_cast<f4 *>(gr + (c.d0 + i) * C::RS + 4 * cq) = ov[i];
      }
    }
    GA_WAVE_SYNC();   // the workgroup is one wavefront: its LDS queue is in order, nothing to drain
#pragma unroll
    for (int q = 0; q < LN; q++) {
      for (int p0 = 0; p0 < D; p0 += C::PPI) {
        const int pl = p0 + psub;
        if (pl < D && col_ok && rok[q])
          stream_store<(GA_NT_STORES & 2) != 0>(reinterpret_cast<f4 *>(G + vb[q] + (i64)pl * geo.HW + wq),
                       *reinterpret_cast<const f4 *>(gt + q * TS + pl * C::RS + 4 * piece));
      }
    }
    GA_WAVE_SYNC();   // the workgroup is one wavefront: its LDS queue is in order, nothing to drain
  }
}

}  // namespace ga
