// sga_kernels.h -- semi-global guided aggregation (SGA) for gfx950, wave64.
//
// What it computes: SURVEY.md Appendix A.1/A.2, i.e. the reference's
// sga_{down,up,right,left}_{forward,data_backward,weight_backward}, Max,
// get_temp_grad and MaxDepth (libs/GANet/src/GANet_kernel.cu:23-933) -- but
// re-designed for CDNA4 instead of one CUDA thread per scanline:
//
//  * A scanline (one column for down/up, one row for right/left, of one (n,c)
//    slice) is owned by a SEGMENT of GD consecutive lanes; each lane keeps DPL
//    consecutive disparities of the previous position in VGPRs.  The d+-1 taps
//    of the recurrence are register neighbours; only the two segment-internal
//    halos move across lanes (DPP row_shr/row_shl) and the max over disparity
//    is an in-register scan + a log2(GD) DPP butterfly.  No LDS, no barrier,
//    no global round trip per step (the reference scans in place in HBM).
//  * The "best previous disparity" tap only needs max_d A[p-1][d] -- the value
//    at the first arg-max IS the max -- so the forward never materialises k.
//  * Traversal is software-pipelined: batch b+1 (SB positions) is in flight in
//    a second VGPR set while batch b is consumed.  Vertical scans read 4-byte
//    elements (lanes of GD-lane segments side by side along W -> coalesced);
//    horizontal scans read float4 along W (4 positions per load) so every 16-byte
//    piece of a cache line is fetched by exactly one lane.
//  * Backward = four light adjoint scans (G only) + ONE per-pixel kernel that produces
//    gradX and all guidance-weight gradients for the four directions together (see the
//    "Backward" banner below).  The reference uses, per direction, memset + get_temp_grad
//    + MaxDepth + data_backward + weight_backward (global RMW per disparity) and three
//    recomputed forward scans.
//
// Numerics: forward uses the reference's exact fma order
//   ((((x*w0) + P1*w1) + P2*w2) + P3*w3) + P4*w4
// so directional volumes, the direction mask and arg-max indices are bit-exact.
// Backward sums over disparity in a different (tree) order: values agree to
// fp32 rounding (tests bound it at 1e-4 abs, north_star).
#pragma once
#include "ga_common.h"

namespace ga {

// One traversal over [S slices][D][H][W] (+ guidance [S][5][H][W]), in VISIT order
// (forward pass: p = 0..L-1; backward pass: p = L-1..0).
struct ScanGeom {
  int D, H, W;
  int L;            // positions per scanline
  int Q;            // scanlines per slice
  int total_lines;  // S * Q
  i64 HW;
  i64 line_stride;  // spatial offset between neighbouring scanlines (1 | W)
  i64 step_stride;  // signed spatial offset between consecutive visits
  i64 start;        // spatial offset of visit 0 on scanline 0
};

struct LaneCtx {
  int lg;        // lane within segment
  int d0;        // first disparity owned
  bool line_ok;  // scanline exists (else the lane shadows the last line, no stores)
  int s, q;      // slice, scanline within slice
  float cap;     // +inf if the lane's first disparity is inside [0, D), else -inf (see lane_cap)
};

// The running max of a lane that lies outside the disparity range is neutralised with
// fminf(max, cap).  The value is made opaque on purpose: written as a select on (d0 < D), hipcc
// turns it back into select(max, -inf) and then quiets every operand of the cross-lane
// reduction that follows (v_max_f32 x, x: 8 extra instructions per scan position).
GA_DEV float lane_cap(int d0, int D)
{
  float cap = d0 < D ? INFINITY : -INFINITY;
#if !defined(GA_HIPSIM)
  asm volatile("" : "+v"(cap));
#endif
  return cap;
}

template <int GD, int DPL> GA_DEV LaneCtx make_ctx(const ScanGeom &geo)
{
  LaneCtx c;
  const int tid = threadIdx.x;
  c.lg = tid % GD;
  c.d0 = c.lg * DPL;
  // XCD-aware order: blocks that share cache lines (neighbouring column groups) stay on one XCD
  int line = xcd_remap(blockIdx.x, gridDim.x) * (blockDim.x / GD) + tid / GD;
  c.line_ok = line < geo.total_lines;
  if (!c.line_ok) line = geo.total_lines - 1;
  c.s = line / geo.Q;
  c.q = line - c.s * geo.Q;
  c.cap = lane_cap(c.d0, geo.D);
  return c;
}

// ---- forward recurrence, one position ------------------------------------------
// Aprev/mprev: directional volume and its max over d at the previous position.
// FULL: the caller guarantees D % DPL == 0 (e.g. 65 = 13 x 5), so every lane lies wholly inside or
// wholly outside [0, D) and the per-element range tests collapse to two per-lane selects.  A wave
// issues about one instruction per 4-8 clk whatever it is, so a position costs what its instruction
// count says (74 per position before this; static mix of the current kernels: profiles/r1q_instruction_mix.txt).
template <int GD, int DPL, bool FULL = false>
GA_DEV void fwd_step(const float (&xs)[DPL], const float (&w)[5], float (&A)[DPL], float &m,
                     bool first, const LaneCtx &c, int D)
{
  float An[DPL];
  float mm;
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
    const float lo = seg_from_prev<GD>(xs[0], A[DPL - 1], c.lg);      // A[p-1][d0-1] | x
    float hi = seg_from_next<GD>(xs[DPL - 1], A[0], c.lg);            // A[p-1][d0+DPL] | x
    if (FULL) hi = c.d0 + DPL == D ? xs[DPL - 1] : hi;                // the lane that owns d = D-1
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      const float P2 = i > 0 ? A[i - 1] : lo;
      float P3 = i < DPL - 1 ? A[i + 1] : hi;
      if (!FULL && c.d0 + i + 1 >= D) P3 = xs[i];
      float t = fmaf(xs[i], w[0], 0.0f);
      t = fmaf(A[i], w[1], t);
      t = fmaf(P2, w[2], t);
      t = fmaf(P3, w[3], t);
      An[i] = fmaf(m, w[4], t);
    }
  }
  // (max taken straight on the FMA results: hipcc knows those are not signalling NaNs and emits bare
  // v_max_f32 / v_max3_f32.  Lanes outside [0, D) are capped with fminf, not a select: behind a select
  // every operand of the cross-lane reduction gets a quieting v_max_f32 x, x of its own, 8 per position)
  if (FULL) {
    mm = An[0];
#pragma unroll
    for (int i = 1; i < DPL; i++) mm = fmaxf(mm, An[i]);
    mm = fminf(mm, c.cap);
  } else {
    mm = -INFINITY;
#pragma unroll
    for (int i = 0; i < DPL; i++)
      if (c.d0 + i < D) mm = fmaxf(mm, An[i]);
  }
#pragma unroll
  for (int i = 0; i < DPL; i++) A[i] = An[i];
  m = seg_allmax<GD>(mm);
}

// ---- forward scan, element-strided traversal (any direction) -------------------
template <int GD, int DPL, int SB>
__global__ void __launch_bounds__(256)
sga_fwd_strided(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ A,
                ScanGeom geo)
{
  const LaneCtx c = make_ctx<GD, DPL>(geo);
  const i64 lineoff = geo.start + (i64)c.q * geo.line_stride;
  const float *xb = x + (i64)c.s * geo.D * geo.HW + lineoff;
  float *Ab = A + (i64)c.s * geo.D * geo.HW + lineoff;
  const float *gb = g + (i64)c.s * 5 * geo.HW + lineoff;
  i64 eoff[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int d = c.d0 + i;
    eoff[i] = (i64)(d < geo.D ? d : geo.D - 1) * geo.HW;
  }
  const int L = geo.L;
  const int nb = (L + SB - 1) / SB;
  float xbuf[2][SB][DPL], wbuf[2][SB][5];
  float Ap[DPL], m = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Ap[i] = 0.f;

#define GA_LOAD_BATCH(B, BUF)                                                   \
  _Pragma("unroll") for (int j = 0; j < SB; j++) {                              \
    int p = (B) * SB + j;                                                       \
    p = p < L ? p : L - 1;                                                      \
    const i64 po = (i64)p * geo.step_stride;                                    \
    _Pragma("unroll") for (int i = 0; i < DPL; i++) xbuf[BUF][j][i] = xb[eoff[i] + po]; \
    _Pragma("unroll") for (int t = 0; t < 5; t++) wbuf[BUF][j][t] = gb[t * geo.HW + po]; \
  }
#define GA_COMPUTE_BATCH(B, BUF)                                                \
  _Pragma("unroll") for (int j = 0; j < SB; j++) {                              \
    const int p = (B) * SB + j;                                                 \
    if (p < L) {                                                                \
      fwd_step<GD, DPL>(xbuf[BUF][j], wbuf[BUF][j], Ap, m, p == 0, c, geo.D);   \
      const i64 po = (i64)p * geo.step_stride;                                  \
      _Pragma("unroll") for (int i = 0; i < DPL; i++)                           \
        if (c.line_ok && c.d0 + i < geo.D) Ab[eoff[i] + po] = Ap[i];            \
    }                                                                           \
  }

  GA_LOAD_BATCH(0, 0)
  for (int b = 0; b < nb; b += 2) {
    GA_LOAD_BATCH(b + 1, 1)
    GA_COMPUTE_BATCH(b, 0)
    if (b + 1 >= nb) break;
    GA_LOAD_BATCH(b + 2, 0)
    GA_COMPUTE_BATCH(b + 1, 1)
  }
#undef GA_LOAD_BATCH
#undef GA_COMPUTE_BATCH
}

// ---- forward scan along W with float4 traffic (right / left) ---------------------
// Requires W % 4 == 0 and 16-byte aligned bases.  desc: visit w = W-1 .. 0.
// A batch is SB = 4*NV positions = NV float4 per owned disparity.
template <int GD, int DPL, int NV>
__global__ void __launch_bounds__(256)
sga_fwd_rowvec(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ A,
               ScanGeom geo, int desc)
{
  const LaneCtx c = make_ctx<GD, DPL>(geo);
  const i64 rowoff = (i64)c.q * geo.W;
  const float *xb = x + (i64)c.s * geo.D * geo.HW + rowoff;
  float *Ab = A + (i64)c.s * geo.D * geo.HW + rowoff;
  const float *gb = g + (i64)c.s * 5 * geo.HW + rowoff;
  i64 eoff[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int d = c.d0 + i;
    eoff[i] = (i64)(d < geo.D ? d : geo.D - 1) * geo.HW;
  }
  const int NF = geo.W >> 2;                    // float4 per row
  const int nb = (NF + NV - 1) / NV;
  f4 xbuf[2][NV][DPL], wbuf[2][NV][5];
  float Ap[DPL], m = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Ap[i] = 0.f;

#define GA_LOAD_BATCH(B, BUF)                                                   \
  _Pragma("unroll") for (int j = 0; j < NV; j++) {                              \
    int f = (B) * NV + j;                                                       \
    f = f < NF ? f : NF - 1;                                                    \
    const int fo = (desc ? NF - 1 - f : f) << 2;                                \
    _Pragma("unroll") for (int i = 0; i < DPL; i++)                             \
      xbuf[BUF][j][i] = *reinterpret_cast<const f4 *>(xb + eoff[i] + fo);       \
    _Pragma("unroll") for (int t = 0; t < 5; t++)                               \
      wbuf[BUF][j][t] = *reinterpret_cast<const f4 *>(gb + t * geo.HW + fo);    \
  }
#define GA_COMPUTE_BATCH(B, BUF)                                                \
  _Pragma("unroll") for (int j = 0; j < NV; j++) {                              \
    const int f = (B) * NV + j;                                                 \
    if (f < NF) {                                                               \
      f4 ov[DPL];                                                               \
      _Pragma("unroll") for (int k = 0; k < 4; k++) {                           \
        const int kk = desc ? 3 - k : k;                                        \
        float xs[DPL], w[5];                                                    \
        _Pragma("unroll") for (int i = 0; i < DPL; i++) xs[i] = f4_get(xbuf[BUF][j][i], kk); \
        _Pragma("unroll") for (int t = 0; t < 5; t++) w[t] = f4_get(wbuf[BUF][j][t], kk);    \
        fwd_step<GD, DPL>(xs, w, Ap, m, f == 0 && k == 0, c, geo.D);            \
        _Pragma("unroll") for (int i = 0; i < DPL; i++) f4_set(ov[i], kk, Ap[i]); \
      }                                                                         \
      const int fo = (desc ? NF - 1 - f : f) << 2;                              \
      _Pragma("unroll") for (int i = 0; i < DPL; i++)                           \
        if (c.line_ok && c.d0 + i < geo.D)                                      \
          *reinterpret_cast<f4 *>(Ab + eoff[i] + fo) = ov[i];                   \
    }                                                                           \
  }

  GA_LOAD_BATCH(0, 0)
  for (int b = 0; b < nb; b += 2) {
    GA_LOAD_BATCH(b + 1, 1)
    GA_COMPUTE_BATCH(b, 0)
    if (b + 1 >= nb) break;
    GA_LOAD_BATCH(b + 2, 0)
    GA_COMPUTE_BATCH(b + 1, 1)
  }
#undef GA_LOAD_BATCH
#undef GA_COMPUTE_BATCH
}

// =====================================================================================
// Backward.  Split in two (measured: the single-sweep version spent ~200 VALU/position on
// work that is NOT part of the recurrence and ran at 0.26-0.52 ms per direction):
//
//   1. sga_bwdg_*: the reverse-scan adjoint ONLY.  G[p][d] = [mask==dir]*gradOut
//        + G[p+1][d]*w1[p+1] + G[p+1][d+1]*w2[p+1] + G[p+1][d-1]*w3[p+1]
//        + [d == k_p] * w4[p+1] * sum_d' G[p+1][d'],   k_p = first-argmax_d A[p][.]
//      (GANet_kernel.cu:144-181 and mirrors).  k_p comes from the forward merge kernel
//      (uint16 per pixel and direction), so the scan touches gradOut, mask, G only.
//   2. sga_bwd_point: everything else is independent per pixel -- one lane per pixel loops
//      over d for ALL directions at once: input gradient (read x once, write gradX once,
//      no read-modify-write per direction) and the five guidance-weight sums
//      (GANet_kernel.cu:164-207, 226-272).  No cross-lane traffic at all.
// HBM traffic: 4 x (1.25 V in + 1 V out) + (1 + 4 + 4) V in + 1 V out = 19 V, the same as
// four fused sweeps, with the serial path ~5x shorter.
// =====================================================================================

// ---- reverse-scan adjoint, one visited position -------------------------------------
// Gn: G at the previously visited position (forward p+1); wn its guidance; sgn = sum_d Gn.
// FULL (the caller guarantees D % DPL == 0, so a lane lies wholly inside or wholly outside [0, D)): the range test of every
// element collapses to two per-lane values -- `dl`, the direction an inside lane compares its mask bytes with (-1 outside: no
// byte matches), and a zero for the one term an outside lane could receive from an inside one (lo of the first outside lane);
// outside lanes then stay exactly 0 by themselves.
// PRE: `go` is the MASKED gradient already -- [mask == dir] * gradOut for elements inside [0, D), 0 outside (the row kernel
// applies the mask when it stages a tile, where all 64 lanes hold different elements, instead of here, where a 16-lane
// recurrence is mirrored four times: 2 instead of 10-15 instructions per position for it); mk is not read.
template <int GD, int DPL, typename MaskT, bool FULL = false, bool PRE = false>
GA_DEV void bwdg_step(const float (&go)[DPL], const MaskT (&mk)[DPL], float (&Gn)[DPL],
                      float (&wn)[5], float &sgn, const float (&w)[5], int kp, bool has_nx,
                      const LaneCtx &c, int D, int dir)
{
  float G[DPL];
  if (FULL) {
    const bool in = c.cap > 0.f;                       // (lane_cap: +inf inside)
    if (PRE) {
#pragma unroll
      for (int i = 0; i < DPL; i++) G[i] = go[i];
    } else {
      int dl = in ? dir : -1;
#if !defined(GA_HIPSIM)
      asm volatile("" : "+v"(dl));                     // (left visible, hipcc turns it back into `in && mask == dir`: the s_and again)
#endif
#pragma unroll
      for (int i = 0; i < DPL; i++) G[i] = (int)mk[i] == dl ? go[i] : 0.f;
    }
    if (has_nx) {
      const float lo = seg_from_prev0<GD>(Gn[DPL - 1], c.lg);
      const float hi = seg_from_next0<GD>(Gn[0], c.lg);
      const float w3 = in ? wn[3] : 0.f;               // the first outside lane receives nothing from the last inside one
      const float t4 = wn[4] * sgn;
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        const float up = i < DPL - 1 ? Gn[i + 1] : hi;
        // (an outside lane stays 0 because its neighbour term is MULTIPLIED by w3 = 0, not selected away -- one instruction less
        // per position in kernels that are bound by their instruction count.  Finite gradients only: 0 * Inf would put a NaN into
        // the outside lane, from where it returns through `hi`; the reference propagates a non-finite gradient through the whole
        // scanline as well (every term of its recurrence is a product with it), so nothing finite is lost.  ADVICE r3.)
        const float dn = i > 0 ? Gn[i - 1] : lo;
        float t = G[i];
        t = fmaf(Gn[i], wn[1], t);
        t = fmaf(up, wn[2], t);
        t = fmaf(dn, w3, t);
        if (c.d0 + i == kp) t += t4;
        G[i] = t;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < DPL; i++) G[i] = PRE ? go[i] : ((c.d0 + i < D && (int)mk[i] == dir) ? go[i] : 0.f);
    if (has_nx) {
      const float lo = seg_from_prev0<GD>(Gn[DPL - 1], c.lg);
      const float hi = seg_from_next0<GD>(Gn[0], c.lg);
      const float t4 = wn[4] * sgn;
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        const float up = i < DPL - 1 ? Gn[i + 1] : hi;
        const float dn = i > 0 ? Gn[i - 1] : lo;
        float t = G[i];
        t = fmaf(Gn[i], wn[1], t);
        t = fmaf(up, wn[2], t);
        t = fmaf(dn, wn[3], t);
        if (c.d0 + i == kp) t += t4;
        G[i] = (c.d0 + i < D) ? t : 0.f;
      }
    }
  }
  float sg = G[0];
  Gn[0] = G[0];
#pragma unroll
  for (int i = 1; i < DPL; i++) { Gn[i] = G[i]; sg += G[i]; }
#pragma unroll
  for (int t = 0; t < 5; t++) wn[t] = w[t];
  sgn = seg_allsum<GD>(sg);
}

// ---- adjoint scan, element-strided traversal (geo in VISIT order = reverse of forward) ----
template <int GD, int DPL, int SB, typename MaskT>
__global__ void __launch_bounds__(256)
sga_bwdg_strided(const float *__restrict__ g, const MaskT *__restrict__ mask,
                 const uint16_t *__restrict__ kp, const float *__restrict__ gout,
                 float *__restrict__ G, ScanGeom geo, int dir)
{
  const LaneCtx c = make_ctx<GD, DPL>(geo);
  const i64 lineoff = geo.start + (i64)c.q * geo.line_stride;
  const i64 vbase = (i64)c.s * geo.D * geo.HW + lineoff;
  const float *gob = gout + vbase;
  const MaskT *mb = mask + vbase;
  float *Gb = G + vbase;
  const float *gb = g + (i64)c.s * 5 * geo.HW + lineoff;
  const uint16_t *kb = kp + (i64)c.s * geo.HW + lineoff;
  i64 eoff[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int d = c.d0 + i;
    eoff[i] = (i64)(d < geo.D ? d : geo.D - 1) * geo.HW;
  }
  const int L = geo.L;
  const int nb = (L + SB - 1) / SB;
  float gobuf[2][SB][DPL], wbuf[2][SB][5];
  MaskT mbuf[2][SB][DPL];
  int kbuf[2][SB];
  float Gn[DPL], wn[5], sgn = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Gn[i] = 0.f;
#pragma unroll
  for (int t = 0; t < 5; t++) wn[t] = 0.f;

#define GA_LOAD_BATCH(B, BUF)                                                   \
  _Pragma("unroll") for (int j = 0; j < SB; j++) {                              \
    int v = (B) * SB + j;                                                       \
    v = v < L ? v : L - 1;                                                      \
    const i64 po = (i64)v * geo.step_stride;                                    \
    _Pragma("unroll") for (int i = 0; i < DPL; i++) {                           \
      gobuf[BUF][j][i] = gob[eoff[i] + po];                                     \
      mbuf[BUF][j][i] = mb[eoff[i] + po];                                       \
    }                                                                           \
    _Pragma("unroll") for (int t = 0; t < 5; t++) wbuf[BUF][j][t] = gb[t * geo.HW + po]; \
    kbuf[BUF][j] = (int)kb[po];                                                 \
  }
#define GA_COMPUTE_BATCH(B, BUF)                                                \
  _Pragma("unroll") for (int j = 0; j < SB; j++) {                              \
    const int v = (B) * SB + j;                                                 \
    if (v < L) {                                                                \
      bwdg_step<GD, DPL, MaskT>(gobuf[BUF][j], mbuf[BUF][j], Gn, wn, sgn, wbuf[BUF][j], kbuf[BUF][j], \
                         v > 0, c, geo.D, dir);                                 \
      const i64 po = (i64)v * geo.step_stride;                                  \
      _Pragma("unroll") for (int i = 0; i < DPL; i++)                           \
        if (c.line_ok && c.d0 + i < geo.D) Gb[eoff[i] + po] = Gn[i];            \
    }                                                                           \
  }

  GA_LOAD_BATCH(0, 0)
  for (int b = 0; b < nb; b += 2) {
    GA_LOAD_BATCH(b + 1, 1)
    GA_COMPUTE_BATCH(b, 0)
    if (b + 1 >= nb) break;
    GA_LOAD_BATCH(b + 2, 0)
    GA_COMPUTE_BATCH(b + 1, 1)
  }
#undef GA_LOAD_BATCH
#undef GA_COMPUTE_BATCH
}

// ---- adjoint scan along W with float4 traffic.  desc: visit w = W-1 .. 0 ----------------
template <int GD, int DPL, int NV>
__global__ void __launch_bounds__(256)
sga_bwdg_rowvec(const float *__restrict__ g, const uint8_t *__restrict__ mask,
                const uint16_t *__restrict__ kp, const float *__restrict__ gout,
                float *__restrict__ G, ScanGeom geo, int dir, int desc)
{
  const LaneCtx c = make_ctx<GD, DPL>(geo);
  const i64 rowoff = (i64)c.q * geo.W;
  const i64 vbase = (i64)c.s * geo.D * geo.HW + rowoff;
  const float *gob = gout + vbase;
  const uint8_t *mb = mask + vbase;
  float *Gb = G + vbase;
  const float *gb = g + (i64)c.s * 5 * geo.HW + rowoff;
  const uint16_t *kb = kp + (i64)c.s * geo.HW + rowoff;
  i64 eoff[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int d = c.d0 + i;
    eoff[i] = (i64)(d < geo.D ? d : geo.D - 1) * geo.HW;
  }
  const int NF = geo.W >> 2;
  const int nb = (NF + NV - 1) / NV;
  f4 gobuf[2][NV][DPL], wbuf[2][NV][5];
  uint32_t mbuf[2][NV][DPL];
  uint2 kbuf[2][NV];
  float Gn[DPL], wn[5], sgn = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Gn[i] = 0.f;
#pragma unroll
  for (int t = 0; t < 5; t++) wn[t] = 0.f;

#define GA_LOAD_BATCH(B, BUF)                                                   \
  _Pragma("unroll") for (int j = 0; j < NV; j++) {                              \
    int f = (B) * NV + j;                                                       \
    f = f < NF ? f : NF - 1;                                                    \
    const int fo = (desc ? NF - 1 - f : f) << 2;                                \
    _Pragma("unroll") for (int i = 0; i < DPL; i++) {                           \
      gobuf[BUF][j][i] = *reinterpret_cast<const f4 *>(gob + eoff[i] + fo);     \
      mbuf[BUF][j][i] = *reinterpret_cast<const uint32_t *>(mb + eoff[i] + fo); \
    }                                                                           \
    _Pragma("unroll") for (int t = 0; t < 5; t++)                               \
      wbuf[BUF][j][t] = *reinterpret_cast<const f4 *>(gb + t * geo.HW + fo);    \
    kbuf[BUF][j] = *reinterpret_cast<const uint2 *>(kb + fo);                   \
  }
#define GA_COMPUTE_BATCH(B, BUF)                                                \
  _Pragma("unroll") for (int j = 0; j < NV; j++) {                              \
    const int f = (B) * NV + j;                                                 \
    if (f < NF) {                                                               \
      f4 ov[DPL];                                                               \
      _Pragma("unroll") for (int k = 0; k < 4; k++) {                           \
        const int kk = desc ? 3 - k : k;                                        \
        float go[DPL], w[5];                                                    \
        uint8_t mk[DPL];                                                        \
        _Pragma("unroll") for (int i = 0; i < DPL; i++) {                       \
          go[i] = f4_get(gobuf[BUF][j][i], kk);                                 \
          mk[i] = (uint8_t)(mbuf[BUF][j][i] >> (8 * kk));                       \
        }                                                                       \
        _Pragma("unroll") for (int t = 0; t < 5; t++) w[t] = f4_get(wbuf[BUF][j][t], kk); \
        const uint32_t kw = kk < 2 ? kbuf[BUF][j].x : kbuf[BUF][j].y;           \
        const int kpv = (int)((kw >> (16 * (kk & 1))) & 0xffffu);               \
        bwdg_step<GD, DPL, uint8_t>(go, mk, Gn, wn, sgn, w, kpv, !(f == 0 && k == 0), c, geo.D, dir); \
        _Pragma("unroll") for (int i = 0; i < DPL; i++) f4_set(ov[i], kk, Gn[i]); \
      }                                                                         \
      const int fo = (desc ? NF - 1 - f : f) << 2;                              \
      _Pragma("unroll") for (int i = 0; i < DPL; i++)                           \
        if (c.line_ok && c.d0 + i < geo.D)                                      \
          *reinterpret_cast<f4 *>(Gb + eoff[i] + fo) = ov[i];                   \
    }                                                                           \
  }

  GA_LOAD_BATCH(0, 0)
  for (int b = 0; b < nb; b += 2) {
    GA_LOAD_BATCH(b + 1, 1)
    GA_COMPUTE_BATCH(b, 0)
    if (b + 1 >= nb) break;
    GA_LOAD_BATCH(b + 2, 0)
    GA_COMPUTE_BATCH(b + 1, 1)
  }
#undef GA_LOAD_BATCH
#undef GA_COMPUTE_BATCH
}

// ---- per-pixel gradients for NDIR directions at once ---------------------------------------
// One lane per pixel (n,c,h,w), marching over d.  For every direction q with adjoint volume
// G_q and forward volume A_q, pp = the position visited just before the pixel in q's forward
// scan (pixel offset prev_off[q]; has_prev false on the scan's first row / column):
//   gradX[d] (+)= sum_q G_q[d]*w0_q (+ G_q[0]*w2_q at d = 0, + G_q[D-1]*w3_q at d = D-1)
//   gw0_q = sum_d G_q[d]*x[d]
//   gw1_q = sum_d G_q[d]*A_q[pp][d]
//   gw2_q = G_q[0]*x[0]     + sum_{d>=1}  G_q[d]*A_q[pp][d-1]
//   gw3_q = G_q[D-1]*x[D-1] + sum_{d<D-1} G_q[d]*A_q[pp][d+1]
//   gw4_q = (sum_d G_q[d]) * max_d A_q[pp][d]          (gw1..4 = 0 without pp)
struct PointArgs {
  const float *G[4];
  const float *A[4];
  const float *g[4];
  float *gw[4];
  int dir[4];
};

#ifndef GA_POINT_DU
#define GA_POINT_DU 2
#endif
#ifndef GA_POINT_WAVES
#define GA_POINT_WAVES 5
#endif
// TG: the ADJOINT volumes of the two vertical directions (G_down, G_up) have the private tiled layout of sga_col_kernels.h:
// plane stride 64 instead of H W, the pixel's offset computed once; 64-byte runs per plane, which this kernel's reads tolerate
// (+10 us of 290; the column adjoint scans gain 31 us from writing them as bursts, profiles/r7e_ab_sga_stages.txt).
template <int NDIR, bool ACC, bool TG = false>
__global__ void __launch_bounds__(256, GA_POINT_WAVES)
sga_bwd_point(const float *__restrict__ x, float *__restrict__ gradX, PointArgs pa,
              int D, int H, int W, i64 npix)
{
  // Register diet: this kernel is pure load->use latency (91 % of wave time in s_waitcnt), so
  // waves per SIMD is what counts.  Measured at cfg2: 127 VGPR / 4 waves 0.50 ms, 96 VGPR /
  // 5 waves 0.39 ms; spilling to reach 6 waves loses again (0.43-0.62 ms), and so does fetching
  // w2 / w3 only on the first / last plane.  Hence: 32-bit neighbour offsets, a bit mask for
  // "has previous position", two planes in flight per step, capped at 5 waves per SIMD.
  const i64 HW = (i64)H * W;
  const i64 stride = (i64)gridDim.x * blockDim.x;
#ifndef GA_POINT_XCD
// XCD-aware block order: neighbouring blocks on one XCD's L2.  Round 3 (API-layout volumes only): the kernel alone 0.332 -> 0.304
// ms, the step unchanged -- left off.  Round 4, with the vertical adjoint volumes tiled: a 128-byte line of those holds the same
// 16 columns of TWO rows, i.e. the 64-byte runs of two waves a row apart (0.8 blocks); round-robin over the XCDs each of them
// fetched the line into an L2 of its own: 1.57 GB read per launch, 1.30 GB with this order (profiles/r7r_pmc_point_block_order.txt)
// at the same time (+-0.1 % on the step, profiles/r7l_*: the kernel is bound by load latency, not by the fabric) -- on for the
// traffic.
#define GA_POINT_XCD 1
#endif
  const int bid = GA_POINT_XCD ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  for (i64 pidx = (i64)bid * blockDim.x + threadIdx.x; pidx < npix; pidx += stride) {
    const i64 s = pidx / HW, pix = pidx - s * HW;
    const int h = (int)(pix / W), w = (int)(pix - (i64)h * W);
    const i64 vb = s * D * HW + pix;
    const i64 gbo = s * 5 * HW + pix;
    // tiled volumes: offset of (s, plane 0, h, w)
    i64 tb = 0;
    if (TG) tb = (((s * (W >> 4) + (w >> 4)) * (H >> 2) + (h >> 2)) * D) * 64 + (h & 3) * 16 + (w & 15);
    float w0[NDIR], w2[NDIR], w3[NDIR];
    int poff[NDIR];          // previous position in forward order (0 = none, see hpm)
    unsigned hpm = 0;        // bit q: direction q has a previous position at this pixel
    float s0[NDIR], s1[NDIR], s2[NDIR], s3[NDIR], sg[NDIR], mx[NDIR];
    float a_m[NDIR], a_0[NDIR];        // A[pp][d-1], A[pp][d]
#pragma unroll
    for (int q = 0; q < NDIR; q++) {
      const int dir = pa.dir[q];
      w0[q] = pa.g[q][gbo];
      w2[q] = pa.g[q][gbo + 2 * HW];
      w3[q] = pa.g[q][gbo + 3 * HW];
      // down: h-1, up: h+1, right: w-1, left: w+1
      const bool hp = dir == 0 ? h > 0 : dir == 1 ? h + 1 < H : dir == 2 ? w > 0 : w + 1 < W;
      poff[q] = hp ? (dir == 0 ? -W : dir == 1 ? W : dir == 2 ? -1 : 1) : 0;
      static_assert(!TG || NDIR == 4, "tiled adjoint volumes: the four-direction launch only (direction q in slot q)");
      hpm |= hp ? (1u << q) : 0u;
      s0[q] = s1[q] = s2[q] = s3[q] = sg[q] = 0.f;
      mx[q] = -INFINITY;
      a_m[q] = 0.f;
      a_0[q] = pa.A[q][vb + poff[q]];
    }
    constexpr int DU = GA_POINT_DU;
    for (int dc = 0; dc < D; dc += DU) {
      float xv[DU], gxv[DU], Gv[DU][NDIR], Av[DU][NDIR];
      // every load is unconditional (plane indices clamped into the volume, values masked afterwards): a load
      // under a condition sits in a branch of its own and hipcc guards the next use of its register with
      // s_waitcnt vmcnt(0) right behind it -- eight serialised memory round trips per step instead of one
#pragma unroll
      for (int u = 0; u < DU; u++) {
        const int d = dc + u;
        const i64 o = vb + (i64)(d < D ? d : D - 1) * HW;
        const i64 on = vb + (i64)(d + 1 < D ? d + 1 : D - 1) * HW;
        const i64 ot = tb + (i64)(d < D ? d : D - 1) * 64;
        xv[u] = stream_load<(GA_NT_LOADS & 8) != 0>(x + o);
        gxv[u] = ACC ? gradX[o] : 0.f;
#pragma unroll
        for (int q = 0; q < NDIR; q++) {
          const bool vert = NDIR == 4 && q < 2;      // (the four-direction launch passes direction q in slot q)
          Gv[u][q] = stream_load<(GA_NT_LOADS & 2) != 0>(pa.G[q] + ((TG && vert) ? ot : o));
          Av[u][q] = stream_load<(GA_NT_LOADS & 2) != 0>(pa.A[q] + on + poff[q]);                          // A[pp][d+1] (used only where d + 1 < D)
        }
      }
#pragma unroll
      for (int u = 0; u < DU; u++) {
        const int d = dc + u;
        if (d < D) {
          float gacc = gxv[u];
#pragma unroll
          for (int q = 0; q < NDIR; q++) {
            const float G_ = Gv[u][q], a_p = Av[u][q];
            float r = G_ * w0[q];
            if (d == 0) r = fmaf(G_, w2[q], r);
            if (d == D - 1) r = fmaf(G_, w3[q], r);
            gacc += r;
            s0[q] = fmaf(G_, xv[u], s0[q]);
            sg[q] += G_;
            s1[q] = fmaf(G_, a_0[q], s1[q]);
            s2[q] = fmaf(G_, d >= 1 ? a_m[q] : xv[u], s2[q]);
            s3[q] = fmaf(G_, d + 1 < D ? a_p : xv[u], s3[q]);
            mx[q] = fmaxf(mx[q], a_0[q]);
            a_m[q] = a_0[q];
            a_0[q] = a_p;
          }
          stream_store<(GA_NT_STORES & 8) != 0>(gradX + vb + (i64)d * HW, gacc);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NDIR; q++) {
      float *gw = pa.gw[q] + gbo;
      const bool hp = (hpm >> q) & 1u;
      gw[0] = s0[q];
      gw[HW] = hp ? s1[q] : 0.f;
      gw[2 * HW] = hp ? s2[q] : 0.f;
      gw[3 * HW] = hp ? s3[q] : 0.f;
      gw[4 * HW] = hp ? sg[q] * mx[q] : 0.f;
    }
  }
}

// ---- direction merge + arg-max, one lane per pixel --------------------------------------------
// out = A0; mask = 0; for dir 1..3: if (out < A_dir) { out = A_dir; mask = dir; }
// (Max, GANet_kernel.cu:23-36, fused over the 4 volumes) and, in the same sweep,
// kp[dir][pixel] = first-argmax_d A_dir[.][pixel] (MaxDepth, :50-64) for the backward scans.
template <typename MaskT>
__global__ void __launch_bounds__(256)
sga_merge_px(const float *__restrict__ A0, const float *__restrict__ A1, const float *__restrict__ A2,
             const float *__restrict__ A3, float *__restrict__ out, MaskT *__restrict__ mask,
             uint16_t *__restrict__ kp, int D, i64 HW, i64 npix)
{
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 pidx = (i64)blockIdx.x * blockDim.x + threadIdx.x; pidx < npix; pidx += stride) {
    const i64 s = pidx / HW, pix = pidx - s * HW;
    const i64 vb = s * D * HW + pix;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    for (int d = 0; d < D; d++) {
      const i64 o = vb + (i64)d * HW;
      const float a0 = A0[o], a1 = A1[o], a2 = A2[o], a3 = A3[o];
      float ov = a0;
      int mk = 0;
      if (ov < a1) { ov = a1; mk = 1; }
      if (ov < a2) { ov = a2; mk = 2; }
      if (ov < a3) { ov = a3; mk = 3; }
      out[o] = ov;
      mask[o] = (MaskT)mk;
      if (d == 0) { m0 = a0; m1 = a1; m2 = a2; m3 = a3; }
      else {
        if (m0 < a0) { m0 = a0; k0 = d; }
        if (m1 < a1) { m1 = a1; k1 = d; }
        if (m2 < a2) { m2 = a2; k2 = d; }
        if (m3 < a3) { m3 = a3; k3 = d; }
      }
    }
    kp[pidx] = (uint16_t)k0;
    kp[npix + pidx] = (uint16_t)k1;
    kp[2 * npix + pidx] = (uint16_t)k2;
    kp[3 * npix + pidx] = (uint16_t)k3;
  }
}

// Same merge, FOUR consecutive pixels per lane: every access is a 16-byte load / store (the mask
// leaves as one packed dword, kp as one 8-byte store), DU planes of loads in flight per lane.  The
// one-pixel-per-lane form above issues 4-byte requests and tops out at ~4.4 TB/s; a streaming kernel
// of 16-byte requests reaches ~6.3 TB/s on this chip (scripts/ubench/mall_probe.py).
// Needs HW % 4 == 0 and 16-byte aligned volumes (launcher checks).
#ifndef GA_MERGE_DU
#define GA_MERGE_DU 2      // planes of loads in flight per lane: 1 was best with plain loads (round 1); with the non-temporal loads 2 is (whole step -0.9 ... -1.2 % on two boxes, profiles/r4a_*)
#endif
static __global__ void __launch_bounds__(64)
sga_merge_px4(const float *__restrict__ A0, const float *__restrict__ A1, const float *__restrict__ A2,
              const float *__restrict__ A3, float *__restrict__ out, uint8_t *__restrict__ mask,
              uint16_t *__restrict__ kp, int D, i64 HW, i64 npix)
{
  constexpr int DU = GA_MERGE_DU;
  const i64 nq = npix >> 2;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 qidx = (i64)blockIdx.x * blockDim.x + threadIdx.x; qidx < nq; qidx += stride) {
    const i64 pidx = qidx << 2;
    const i64 s = pidx / HW, pix = pidx - s * HW;
    const i64 vb = s * D * HW + pix;
    float m[4][4];
    int k[4][4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int j = 0; j < 4; j++) { m[q][j] = 0.f; k[q][j] = 0; }
    for (int dc = 0; dc < D; dc += DU) {
      f4 a[DU][4];
#pragma unroll
      for (int u = 0; u < DU; u++) {
        const int d = dc + u < D ? dc + u : D - 1;
        const i64 o = vb + (i64)d * HW;
        a[u][0] = stream_load<(GA_NT_LOADS & 1) != 0>(reinterpret_cast<const f4 *>(A0 + o));
        a[u][1] = stream_load<(GA_NT_LOADS & 1) != 0>(reinterpret_cast<const f4 *>(A1 + o));
        a[u][2] = stream_load<(GA_NT_LOADS & 1) != 0>(reinterpret_cast<const f4 *>(A2 + o));
        a[u][3] = stream_load<(GA_NT_LOADS & 1) != 0>(reinterpret_cast<const f4 *>(A3 + o));
      }
#pragma unroll
      for (int u = 0; u < DU; u++) {
        const int d = dc + u;
        if (d < D) {
          float v[4][4];
#pragma unroll
          for (int q = 0; q < 4; q++) { v[q][0] = a[u][q].x; v[q][1] = a[u][q].y; v[q][2] = a[u][q].z; v[q][3] = a[u][q].w; }
          float ov[4];
          unsigned mk = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            float o_ = v[0][j];
            unsigned mj = 0;
            if (o_ < v[1][j]) { o_ = v[1][j]; mj = 1; }
            if (o_ < v[2][j]) { o_ = v[2][j]; mj = 2; }
            if (o_ < v[3][j]) { o_ = v[3][j]; mj = 3; }
            ov[j] = o_;
            mk |= mj << (8 * j);
          }
          const i64 o = vb + (i64)d * HW;
          f4 r;
          r.x = ov[0]; r.y = ov[1]; r.z = ov[2]; r.w = ov[3];
          stream_store<(GA_NT_STORES & 4) != 0>(reinterpret_cast<f4 *>(out + o), r);
          stream_store<(GA_NT_STORES & 128) != 0>(reinterpret_cast<unsigned *>(mask + o), mk);
#pragma unroll
          for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (d == 0) m[q][j] = v[q][j];
              else if (m[q][j] < v[q][j]) { m[q][j] = v[q][j]; k[q][j] = d; }
            }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint2 pk;
      pk.x = (unsigned)k[q][0] | ((unsigned)k[q][1] << 16);
      pk.y = (unsigned)k[q][2] | ((unsigned)k[q][3] << 16);
      *reinterpret_cast<uint2 *>(kp + (i64)q * npix + pidx) = pk;
    }
  }
}

// Inference-only merge: out = relu(scale[c] * max_dir A_dir + shift[c]) -- the direction max with the
// eval-mode BatchNorm3d + ReLU that follow SGA in SGABlock.forward (models/GANet_deep.py:269-271) folded in;
// no mask, no arg-max (nothing is kept for a backward).  scale == nullptr: plain max.  Elementwise, linear
// order, 16-byte requests when n % 4 == 0 and the slice size is a multiple of 4.
template <bool VEC4>
__global__ void __launch_bounds__(256)
sga_merge_infer(const float *__restrict__ A0, const float *__restrict__ A1, const float *__restrict__ A2,
                const float *__restrict__ A3, float *__restrict__ out, const float *__restrict__ scale,
                const float *__restrict__ shift, int C, i64 slice /* D*H*W */, i64 n)
{
  const i64 stride = (i64)gridDim.x * blockDim.x;
  if (VEC4) {
    const i64 n4 = n >> 2;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const i64 e = i << 2;
      const f4 a = *reinterpret_cast<const f4 *>(A0 + e), b = *reinterpret_cast<const f4 *>(A1 + e);
      const f4 c_ = *reinterpret_cast<const f4 *>(A2 + e), d = *reinterpret_cast<const f4 *>(A3 + e);
      float v[4] = {a.x, a.y, a.z, a.w};
      const float w1[4] = {b.x, b.y, b.z, b.w}, w2[4] = {c_.x, c_.y, c_.z, c_.w}, w3[4] = {d.x, d.y, d.z, d.w};
      float sc = 1.f, sh = 0.f;
      if (scale) { const int ch = (int)((e / slice) % C); sc = scale[ch]; sh = shift[ch]; }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (v[j] < w1[j]) v[j] = w1[j];
        if (v[j] < w2[j]) v[j] = w2[j];
        if (v[j] < w3[j]) v[j] = w3[j];
        if (scale) v[j] = fmaxf(fmaf(v[j], sc, sh), 0.f);
      }
      f4 r; r.x = v[0]; r.y = v[1]; r.z = v[2]; r.w = v[3];
      *reinterpret_cast<f4 *>(out + e) = r;
    }
  } else {
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
      float v = A0[e];
      const float b = A1[e], c_ = A2[e], d = A3[e];
      if (v < b) v = b;
      if (v < c_) v = c_;
      if (v < d) v = d;
      if (scale) { const int ch = (int)((e / slice) % C); v = fmaxf(fmaf(v, scale[ch], shift[ch]), 0.f); }
      out[e] = v;
    }
  }
}

// first-argmax over d of one directional volume (reference-compatible path; MaxDepth :50-64)
static __global__ void __launch_bounds__(256)
sga_argmax_px(const float *__restrict__ A, uint16_t *__restrict__ kp, int D, i64 HW, i64 npix)
{
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 pidx = (i64)blockIdx.x * blockDim.x + threadIdx.x; pidx < npix; pidx += stride) {
    const i64 s = pidx / HW, pix = pidx - s * HW;
    const i64 vb = s * D * HW + pix;
    float m = A[vb];
    int k = 0;
    for (int d0 = 1; d0 < D; d0 += 8) {          // 8 loads in flight per lane
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = A[vb + (i64)(d0 + u < D ? d0 + u : D - 1) * HW];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (d0 + u < D && m < a[u]) { m = a[u]; k = d0 + u; }
    }
    kp[pidx] = (uint16_t)k;
  }
}

// running form used by the reference-compatible entry point: out/mask updated with
// one more direction (first == 1: out = tmp is already in place, mask <- (out<tmp)?dir:0)
template <typename MaskT>
__global__ void __launch_bounds__(256)
sga_merge_running(const float *__restrict__ tmp, float *__restrict__ out, MaskT *__restrict__ mask,
                  i64 n, int dir, int first)
{
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float t = tmp[i], o = out[i];
    if (o < t) { out[i] = t; mask[i] = (MaskT)dir; }
    else if (first) mask[i] = (MaskT)0;
  }
}

}  // namespace ga
