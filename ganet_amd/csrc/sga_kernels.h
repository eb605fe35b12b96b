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
//  * Backward is ONE pass per direction: the reverse-scan adjoint, the masked
//    gradOutput gather, the first-argmax routing and all five guidance-weight
//    reductions happen in the same sweep (the reference uses memset +
//    get_temp_grad + MaxDepth + data_backward + weight_backward = 5 launches and
//    re-reads everything; its weight kernel does a global RMW per disparity).
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
};

template <int GD, int DPL> GA_DEV LaneCtx make_ctx(const ScanGeom &geo)
{
  LaneCtx c;
  const int tid = threadIdx.x;
  c.lg = tid % GD;
  c.d0 = c.lg * DPL;
  int line = blockIdx.x * (blockDim.x / GD) + tid / GD;
  c.line_ok = line < geo.total_lines;
  if (!c.line_ok) line = geo.total_lines - 1;
  c.s = line / geo.Q;
  c.q = line - c.s * geo.Q;
  return c;
}

// ---- forward recurrence, one position ------------------------------------------
// Aprev/mprev: directional volume and its max over d at the previous position.
template <int GD, int DPL>
GA_DEV void fwd_step(const float (&xs)[DPL], const float (&w)[5], float (&A)[DPL], float &m,
                     bool first, const LaneCtx &c, int D)
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
    const float lo = seg_from_prev<GD>(xs[0], A[DPL - 1], c.lg);      // A[p-1][d0-1] | x
    const float hi = seg_from_next<GD>(xs[DPL - 1], A[0], c.lg);      // A[p-1][d0+DPL] | x
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      const float P2 = i > 0 ? A[i - 1] : lo;
      float P3 = i < DPL - 1 ? A[i + 1] : hi;
      if (c.d0 + i + 1 >= D) P3 = xs[i];
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
    if (c.d0 + i < D) mm = fmaxf(mm, An[i]);
  }
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

// ---- backward, one visited position ---------------------------------------------
// Visit order is the REVERSE of the forward scan.  "nx" = the position visited just
// before (forward position p+1), "pv" = the position visited next (forward p-1).
struct BwdCarry {
  float wn[5];   // guidance at p+1
  float SGn;     // sum_d G[p+1][d]
  int kp;        // first-argmax_d A[p][.]   (routing target at this visit)
};

template <int GD, int DPL>
GA_DEV void bwd_step(const float (&go)[DPL], const uint8_t (&mk)[DPL], const float (&xs)[DPL],
                     const float (&Am)[DPL], const float (&w)[5], float (&Gn)[DPL], BwdCarry &cy,
                     float (&gxo)[DPL], float (&gwo)[5], bool has_nx, bool has_pv,
                     const LaneCtx &c, int D, int dir)
{
  float G[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) G[i] = (c.d0 + i < D && (int)mk[i] == dir) ? go[i] : 0.f;
  if (has_nx) {
    const float lo = seg_from_prev<GD>(0.f, Gn[DPL - 1], c.lg);   // G[p+1][d0-1] | 0
    const float hi = seg_from_next<GD>(0.f, Gn[0], c.lg);         // G[p+1][d0+DPL] | 0
    const float t4 = cy.wn[4] * cy.SGn;
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      const float up = i < DPL - 1 ? Gn[i + 1] : hi;
      const float dn = i > 0 ? Gn[i - 1] : lo;
      float t = G[i];
      t = fmaf(Gn[i], cy.wn[1], t);
      t = fmaf(up, cy.wn[2], t);
      t = fmaf(dn, cy.wn[3], t);
      if (c.d0 + i == cy.kp) t += t4;
      G[i] = (c.d0 + i < D) ? t : 0.f;
    }
  }
  // input gradient contribution of this direction (A.2 incl. the d=0 / d=D-1 terms)
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    float r = G[i] * w[0];
    if (c.d0 + i == 0) r = fmaf(G[i], w[2], r);
    if (c.d0 + i == D - 1) r = fmaf(G[i], w[3], r);
    gxo[i] = r;
  }
  // guidance-weight reductions
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, sg = 0.f, mv = -INFINITY;
  int mk_idx = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < DPL; i++) { s0 = fmaf(G[i], xs[i], s0); sg += G[i]; }
  if (has_pv) {
    const float alo = seg_from_prev<GD>(0.f, Am[DPL - 1], c.lg);
    const float ahi = seg_from_next<GD>(0.f, Am[0], c.lg);
#pragma unroll
    for (int i = 0; i < DPL; i++) {
      const int d = c.d0 + i;
      const float a2 = d >= 1 ? (i > 0 ? Am[i - 1] : alo) : xs[i];
      const float a3 = d + 1 < D ? (i < DPL - 1 ? Am[i + 1] : ahi) : xs[i];
      s1 = fmaf(G[i], Am[i], s1);
      s2 = fmaf(G[i], a2, s2);
      s3 = fmaf(G[i], a3, s3);
      if (d < D && (Am[i] > mv)) { mv = Am[i]; mk_idx = d; }
    }
    seg_argmax<GD>(mv, mk_idx);
  }
  s0 = seg_allsum<GD>(s0);
  sg = seg_allsum<GD>(sg);
  if (has_pv) {
    s1 = seg_allsum<GD>(s1);
    s2 = seg_allsum<GD>(s2);
    s3 = seg_allsum<GD>(s3);
  }
  gwo[0] = s0;
  gwo[1] = has_pv ? s1 : 0.f;
  gwo[2] = has_pv ? s2 : 0.f;
  gwo[3] = has_pv ? s3 : 0.f;
  gwo[4] = has_pv ? sg * mv : 0.f;
  // carry to the next visit (forward position p-1)
#pragma unroll
  for (int i = 0; i < DPL; i++) Gn[i] = G[i];
#pragma unroll
  for (int t = 0; t < 5; t++) cy.wn[t] = w[t];
  cy.SGn = sg;
  cy.kp = mk_idx;
}

// ---- backward scan, element-strided traversal ------------------------------------
// geo is in VISIT order (start = forward position L-1, step_stride = -forward step).
// A is this direction's forward volume.  gradX: accumulate ? += : =.   gw: plain store
// (every guidance pixel is produced exactly once per direction).
template <int GD, int DPL, int SB>
__global__ void __launch_bounds__(256)
sga_bwd_strided(const float *__restrict__ x, const float *__restrict__ g,
                const float *__restrict__ A, const uint8_t *__restrict__ mask,
                const float *__restrict__ gout, float *__restrict__ gradX, float *__restrict__ gw,
                ScanGeom geo, int dir, int accumulate)
{
  const LaneCtx c = make_ctx<GD, DPL>(geo);
  const i64 lineoff = geo.start + (i64)c.q * geo.line_stride;
  const i64 vbase = (i64)c.s * geo.D * geo.HW + lineoff;
  const float *xb = x + vbase, *Ab = A + vbase, *gob = gout + vbase;
  const uint8_t *mb = mask + vbase;
  float *gxb = gradX + vbase;
  const float *gb = g + (i64)c.s * 5 * geo.HW + lineoff;
  float *gwb = gw + (i64)c.s * 5 * geo.HW + lineoff;
  i64 eoff[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int d = c.d0 + i;
    eoff[i] = (i64)(d < geo.D ? d : geo.D - 1) * geo.HW;
  }
  const int L = geo.L;
  const int nb = (L + SB - 1) / SB;
  float gobuf[2][SB][DPL], xbuf[2][SB][DPL], abuf[2][SB][DPL], wbuf[2][SB][5], gxbuf[2][SB][DPL];
  uint8_t mbuf[2][SB][DPL];
  float Gn[DPL];
  BwdCarry cy;
#pragma unroll
  for (int i = 0; i < DPL; i++) Gn[i] = 0.f;
#pragma unroll
  for (int t = 0; t < 5; t++) cy.wn[t] = 0.f;
  cy.SGn = 0.f;
  cy.kp = -1;

#define GA_LOAD_BATCH(B, BUF)                                                   \
  _Pragma("unroll") for (int j = 0; j < SB; j++) {                              \
    int v = (B) * SB + j;                                                       \
    v = v < L ? v : L - 1;                                                      \
    const i64 po = (i64)v * geo.step_stride;                                    \
    const i64 pa = (i64)(v + 1 < L ? v + 1 : v) * geo.step_stride;              \
    _Pragma("unroll") for (int i = 0; i < DPL; i++) {                           \
      gobuf[BUF][j][i] = gob[eoff[i] + po];                                     \
      mbuf[BUF][j][i] = mb[eoff[i] + po];                                       \
      xbuf[BUF][j][i] = xb[eoff[i] + po];                                       \
      abuf[BUF][j][i] = Ab[eoff[i] + pa];                                       \
      gxbuf[BUF][j][i] = accumulate ? gxb[eoff[i] + po] : 0.f;                  \
    }                                                                           \
    _Pragma("unroll") for (int t = 0; t < 5; t++) wbuf[BUF][j][t] = gb[t * geo.HW + po]; \
  }
#define GA_COMPUTE_BATCH(B, BUF)                                                \
  _Pragma("unroll") for (int j = 0; j < SB; j++) {                              \
    const int v = (B) * SB + j;                                                 \
    if (v < L) {                                                                \
      float gxo[DPL], gwo[5];                                                   \
      bwd_step<GD, DPL>(gobuf[BUF][j], mbuf[BUF][j], xbuf[BUF][j], abuf[BUF][j], wbuf[BUF][j], \
                        Gn, cy, gxo, gwo, v > 0, v + 1 < L, c, geo.D, dir);     \
      const i64 po = (i64)v * geo.step_stride;                                  \
      _Pragma("unroll") for (int i = 0; i < DPL; i++)                           \
        if (c.line_ok && c.d0 + i < geo.D) gxb[eoff[i] + po] = gxbuf[BUF][j][i] + gxo[i]; \
      if (c.line_ok && c.lg == 0) {                                             \
        _Pragma("unroll") for (int t = 0; t < 5; t++) gwb[t * geo.HW + po] = gwo[t]; \
      }                                                                         \
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

// ---- backward scan along W with float4 traffic (right / left) ----------------------
// desc: VISIT order runs w = W-1 .. 0 (i.e. the backward pass of `right`); otherwise
// w = 0 .. W-1 (backward pass of `left`).  Requires W % 4 == 0, 16-byte aligned bases.
// The forward volume at the NEXT visited position is the next component in visit
// order; for the last component of a float4 it is the first of the following float4,
// which is already resident in the other (prefetched) buffer.
template <int GD, int DPL, int NV>
__global__ void __launch_bounds__(256)
sga_bwd_rowvec(const float *__restrict__ x, const float *__restrict__ g,
               const float *__restrict__ A, const uint8_t *__restrict__ mask,
               const float *__restrict__ gout, float *__restrict__ gradX, float *__restrict__ gw,
               ScanGeom geo, int dir, int accumulate, int desc)
{
  const LaneCtx c = make_ctx<GD, DPL>(geo);
  const i64 rowoff = (i64)c.q * geo.W;
  const i64 vbase = (i64)c.s * geo.D * geo.HW + rowoff;
  const float *xb = x + vbase, *Ab = A + vbase, *gob = gout + vbase;
  const uint8_t *mb = mask + vbase;
  float *gxb = gradX + vbase;
  const float *gb = g + (i64)c.s * 5 * geo.HW + rowoff;
  float *gwb = gw + (i64)c.s * 5 * geo.HW + rowoff;
  i64 eoff[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int d = c.d0 + i;
    eoff[i] = (i64)(d < geo.D ? d : geo.D - 1) * geo.HW;
  }
  const int NF = geo.W >> 2;
  const int nb = (NF + NV - 1) / NV;
  f4 gobuf[2][NV][DPL], xbuf[2][NV][DPL], abuf[2][NV][DPL], wbuf[2][NV][5], gxbuf[2][NV][DPL];
  uint32_t mbuf[2][NV][DPL];
  float Gn[DPL];
  BwdCarry cy;
#pragma unroll
  for (int i = 0; i < DPL; i++) Gn[i] = 0.f;
#pragma unroll
  for (int t = 0; t < 5; t++) cy.wn[t] = 0.f;
  cy.SGn = 0.f;
  cy.kp = -1;

#define GA_LOAD_BATCH(B, BUF)                                                   \
  _Pragma("unroll") for (int j = 0; j < NV; j++) {                              \
    int f = (B) * NV + j;                                                       \
    f = f < NF ? f : NF - 1;                                                    \
    const int fo = (desc ? NF - 1 - f : f) << 2;                                \
    _Pragma("unroll") for (int i = 0; i < DPL; i++) {                           \
      gobuf[BUF][j][i] = *reinterpret_cast<const f4 *>(gob + eoff[i] + fo);     \
      mbuf[BUF][j][i] = *reinterpret_cast<const uint32_t *>(mb + eoff[i] + fo); \
      xbuf[BUF][j][i] = *reinterpret_cast<const f4 *>(xb + eoff[i] + fo);       \
      abuf[BUF][j][i] = *reinterpret_cast<const f4 *>(Ab + eoff[i] + fo);       \
      if (accumulate) gxbuf[BUF][j][i] = *reinterpret_cast<const f4 *>(gxb + eoff[i] + fo); \
      else { gxbuf[BUF][j][i].x = 0.f; gxbuf[BUF][j][i].y = 0.f; gxbuf[BUF][j][i].z = 0.f; gxbuf[BUF][j][i].w = 0.f; } \
    }                                                                           \
    _Pragma("unroll") for (int t = 0; t < 5; t++)                               \
      wbuf[BUF][j][t] = *reinterpret_cast<const f4 *>(gb + t * geo.HW + fo);    \
  }
#define GA_COMPUTE_BATCH(B, BUF)                                                \
  _Pragma("unroll") for (int j = 0; j < NV; j++) {                              \
    const int f = (B) * NV + j;                                                 \
    if (f < NF) {                                                               \
      f4 gxv[DPL], gwv[5];                                                      \
      _Pragma("unroll") for (int k = 0; k < 4; k++) {                           \
        const int kk = desc ? 3 - k : k;                                        \
        const int kn = desc ? 3 : 0;        /* first component in visit order */ \
        float go[DPL], xs[DPL], Am[DPL], w[5], gxo[DPL], gwo[5];                \
        uint8_t mk[DPL];                                                        \
        _Pragma("unroll") for (int i = 0; i < DPL; i++) {                       \
          go[i] = f4_get(gobuf[BUF][j][i], kk);                                 \
          xs[i] = f4_get(xbuf[BUF][j][i], kk);                                  \
          mk[i] = (uint8_t)(mbuf[BUF][j][i] >> (8 * kk));                       \
          if (k < 3) Am[i] = f4_get(abuf[BUF][j][i], desc ? kk - 1 : kk + 1);   \
          else if (j + 1 < NV) Am[i] = f4_get(abuf[BUF][j + 1 < NV ? j + 1 : j][i], kn); \
          else Am[i] = f4_get(abuf[1 - BUF][0][i], kn);                         \
        }                                                                       \
        _Pragma("unroll") for (int t = 0; t < 5; t++) w[t] = f4_get(wbuf[BUF][j][t], kk); \
        const bool has_nx = !(f == 0 && k == 0);                                \
        const bool has_pv = !(f == NF - 1 && k == 3);                           \
        bwd_step<GD, DPL>(go, mk, xs, Am, w, Gn, cy, gxo, gwo, has_nx, has_pv, c, geo.D, dir); \
        _Pragma("unroll") for (int i = 0; i < DPL; i++)                         \
          f4_set(gxv[i], kk, f4_get(gxbuf[BUF][j][i], kk) + gxo[i]);            \
        _Pragma("unroll") for (int t = 0; t < 5; t++) f4_set(gwv[t], kk, gwo[t]); \
      }                                                                         \
      const int fo = (desc ? NF - 1 - f : f) << 2;                              \
      _Pragma("unroll") for (int i = 0; i < DPL; i++)                           \
        if (c.line_ok && c.d0 + i < geo.D)                                      \
          *reinterpret_cast<f4 *>(gxb + eoff[i] + fo) = gxv[i];                 \
      if (c.line_ok && c.lg == 0) {                                             \
        _Pragma("unroll") for (int t = 0; t < 5; t++)                           \
          *reinterpret_cast<f4 *>(gwb + t * geo.HW + fo) = gwv[t];              \
      }                                                                         \
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

// ---- direction merge (Max, GANet_kernel.cu:23-36, fused over the 4 volumes) --------
// out = A0; mask = 0; for dir 1..3: if (out < A_dir) { out = A_dir; mask = dir; }
template <typename MaskT>
__global__ void __launch_bounds__(256)
sga_merge4(const float *__restrict__ A0, const float *__restrict__ A1, const float *__restrict__ A2,
           const float *__restrict__ A3, float *__restrict__ out, MaskT *__restrict__ mask, i64 n)
{
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float o = A0[i];
    int mk = 0;
    const float a1 = A1[i], a2 = A2[i], a3 = A3[i];
    if (o < a1) { o = a1; mk = 1; }
    if (o < a2) { o = a2; mk = 2; }
    if (o < a3) { o = a3; mk = 3; }
    out[i] = o;
    mask[i] = (MaskT)mk;
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

// float-valued mask (reference layout) -> uint8
__global__ void __launch_bounds__(256)
mask_f32_to_u8(const float *__restrict__ m, uint8_t *__restrict__ o, i64 n)
{
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) o[i] = (uint8_t)(int)m[i];
}

}  // namespace ga
