// sga_row_tu.hip -- the horizontal scan kernels (forward and adjoint) in a translation unit of their own.
// Reason: compiler flags.  With hipcc's SLP vectoriser on, neighbouring scalar FMAs of the recurrence
// are packed into v_pk_fma_f32 and paid for with v_mov shuffles (12 per scan position in the forward scan,
// 0.098 -> 0.082 ms without it; the adjoint with its lane-uniform range tests spills 24 registers with it);
// the vertical scans in ganet_capi.hip are a few per cent faster WITH it.  A row kernel is bound by its
// instruction count per position -- scalar instructions included (profiles/r5d_*) -- so that is what is
// kept small here.  build.py compiles this file with -fno-slp-vectorize and links both objects into
// libganet_hip.so.
#include "ga_launch.h"

#ifndef GA_ROW_GD64
#define GA_ROW_GD64 1        // forward: depth axis over the whole wavefront for the models' depths
#endif
#ifndef GA_ROW_BWDG_GD64
#define GA_ROW_BWDG_GD64 1   // the same for the adjoint
#endif

namespace ga {

void launch_row_fwd(const float *x, const float *g, float *A, int S, int D, int H, int W, int dir, hipStream_t st,
                    int out_mode, int C, const float *scale, const float *shift)
{
  RowGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W; geo.total_rows = S * H;
  geo.out_mode = out_mode; geo.C = C > 0 ? C : 1; geo.scale = scale; geo.shift = shift;
  const int dpl = row_dpl(D);
  const bool full = dpl > 0 && D % dpl == 0;     // lanes wholly inside / outside [0, D): leaner recurrence
  const size_t smem = row_smem_fwd(D);
  const dim3 grid((S * H + ROW_LN_F - 1) / ROW_LN_F), block(64);
#if GA_ROW_GD64
  // D <= 72 with the depth axis over the whole wavefront: one disparity per lane up to 64, two up to 72
#define L64(P, DESC, F, NP) GA_LAUNCH_SMEM((sga_row_fwd<P, ROW_SBH_F, ROW_PAD_F, 1, DESC, F, 64, NP>), grid, block, smem, st, x, g, A, geo)
  // (NP = staged pieces per lane = ceil(D_max / 8) of the range: the models' 33 / 48 / 65 get exactly what they need, the ranges
  //  in between share an instantiation)
#define L64_1(NP) { if (dir == 3) L64(1, true, true, NP); else L64(1, false, true, NP); return; }
#define L64_2(NP) { if (D % 2 == 0) { if (dir == 3) L64(2, true, true, NP); else L64(2, false, true, NP); }     \
                    else { if (dir == 3) L64(2, true, false, NP); else L64(2, false, false, NP); } return; }
  if (ROW_LN_F == 1) {
    if (D <= 40) L64_1(5)
    if (D <= 48) L64_1(6)
    if (D <= 64) L64_1(8)
    if (D <= 72) L64_2(9)      // (12 / 16 pieces for D <= 96 / 128 spill at these register caps: those depths keep the 16-lane kernels)
  }
#undef L64_1
#undef L64_2
#undef L64
#endif
#define X(P)                                                                                        \
  if (dpl == (P)) {                                                                                 \
    if (dir == 3 && full) GA_LAUNCH_SMEM((sga_row_fwd<P, ROW_SBH_F, ROW_PAD_F, ROW_LN_F, true, true>), grid, block, smem, st, x, g, A, geo);  \
    else if (dir == 3) GA_LAUNCH_SMEM((sga_row_fwd<P, ROW_SBH_F, ROW_PAD_F, ROW_LN_F, true, false>), grid, block, smem, st, x, g, A, geo);  \
    else if (full) GA_LAUNCH_SMEM((sga_row_fwd<P, ROW_SBH_F, ROW_PAD_F, ROW_LN_F, false, true>), grid, block, smem, st, x, g, A, geo);      \
    else GA_LAUNCH_SMEM((sga_row_fwd<P, ROW_SBH_F, ROW_PAD_F, ROW_LN_F, false, false>), grid, block, smem, st, x, g, A, geo);               \
  }
  GA_ROW_DPLS(X)
#undef X
}

void launch_row_bwdg(const float *g, const uint8_t *mask, const uint16_t *kp, const float *gout, float *G,
                     int S, int D, int H, int W, int dir, hipStream_t st)
{
  RowGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W; geo.total_rows = S * H;
  geo.out_mode = 0; geo.C = 1; geo.scale = nullptr; geo.shift = nullptr;
  const int dpl = row_dpl(D);
  const size_t smem = row_smem_bwdg(D);
  const dim3 grid((S * H + ROW_LN_B - 1) / ROW_LN_B), block(64);
  // the adjoint of `right` (2) walks w downwards, of `left` (3) upwards.  Lanes wholly inside / outside [0, D): the leaner
  // recurrence of bwdg_step<FULL>, instantiated for the depths the models use (33 and 48 at three, 65 at five per lane)
  const bool full = dpl > 0 && D % dpl == 0;
#define L(P, DESC, F) GA_LAUNCH_SMEM((sga_row_bwdg<P, ROW_SBH_B, ROW_PAD_B, ROW_LN_B, DESC, F>), grid, block, smem, st, g, mask, kp, gout, G, geo, dir)
#if GA_ROW_BWDG_GD64
  // D <= 72 with the depth axis over the whole wavefront: one disparity per lane up to 64, two up to 72
#define L64(P, DESC, F, NP) GA_LAUNCH_SMEM((sga_row_bwdg<P, ROW_SBH_B, ROW_PAD_B, 1, DESC, F, 64, NP>), grid, block, smem, st, g, mask, kp, gout, G, geo, dir)
#define L64_1(NP) { if (dir == 2) L64(1, true, true, NP); else L64(1, false, true, NP); return; }
#define L64_2(NP) { if (D % 2 == 0) { if (dir == 2) L64(2, true, true, NP); else L64(2, false, true, NP); }     \
                    else { if (dir == 2) L64(2, true, false, NP); else L64(2, false, false, NP); } return; }
  if (ROW_LN_B == 1) {
    if (D <= 40) L64_1(5)
    if (D <= 48) L64_1(6)
    if (D <= 64) L64_1(8)
    if (D <= 72) L64_2(9)      // (12 / 16 pieces for D <= 96 / 128 spill at these register caps: those depths keep the 16-lane kernels)
  }
#undef L64_1
#undef L64_2
#undef L64
#endif
#define X(P)                                                                                        \
  if (dpl == (P)) {                                                                                 \
    constexpr bool FL = (P) == 3 || (P) == 5;                                                       \
    if (full && FL) { if (dir == 2) L(P, true, FL); else L(P, false, FL); }                         \
    else { if (dir == 2) L(P, true, false); else L(P, false, false); }                              \
  }
  GA_ROW_DPLS(X)
#undef X
#undef L
}

}  // namespace ga
