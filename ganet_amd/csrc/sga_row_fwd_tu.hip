// sga_row_fwd_tu.hip -- the horizontal forward-scan kernels in a translation unit of their own.
// Reason: compiler flags.  With hipcc's SLP vectoriser on, neighbouring scalar FMAs of the recurrence
// are packed into v_pk_fma_f32 and paid for with v_mov shuffles (12 per scan position here, the scan
// 0.098 -> 0.082 ms without it); the vertical and adjoint scans in ganet_capi.hip are a few per cent
// faster WITH it (instruction mix of the current kernels: profiles/r1q_instruction_mix.txt).  build.py compiles this file with
// -fno-slp-vectorize and links both objects into libganet_hip.so.
#include "ga_launch.h"

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

}  // namespace ga
