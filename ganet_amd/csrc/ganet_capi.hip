// ganet_capi.hip -- host side of libganet_hip.so: the C ABI declared in
// include/ganet_hip.h, argument validation, kernel selection and launch.
// Compiled by hipcc for gfx950 (product) and by g++ -DGA_HIPSIM (CPU test emulator).
#include "ga_common.h"
#include "lga_kernels.h"
#include "misc_kernels.h"
#include "sga_kernels.h"
#include "sga_row_kernels.h"
#include "sga_col_kernels.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/ganet_hip.h"

#include "ga_launch.h"

namespace {

using namespace ga;

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char *what)
{
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GANET_E_RUNTIME, "%s: %s", what, hipGetErrorString(e));
  return GANET_OK;
}

#define GA_TRY(expr)                         \
  do {                                       \
    const int rc_ = (expr);                  \
    if (rc_ != GANET_OK) return rc_;         \
  } while (0)

#define GA_HIP(expr)                                                                   \
  do {                                                                                 \
    const hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) return fail(GANET_E_RUNTIME, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

// ---- options ------------------------------------------------------------------
// One default path per operator (measured on MI355X, profiles/) plus its general fallback; the knobs exist so that tests can
// reach the fallbacks and the forced modes.  Fields are atomics: ganet_set_option() may race with launches on other threads
// (each launcher reads a field once).
#ifndef GA_SGA_TILED_DEFAULT
#define GA_SGA_TILED_DEFAULT 1
#endif
struct Options {
  std::atomic<int> sga_tiled{GA_SGA_TILED_DEFAULT};  // SGA backward: the vertical directions' adjoint volumes G_down / G_up in the private tiled layout
                                    // of sga_col_kernels.h (1; where W % 16 == 0 and H % 4 == 0) or in the API layout (0)
  std::atomic<int> lga_wave{2};     // LGA kernel family (radius <= 2): 2 plane-pair kernels, the forward / data-backward with ONE x ring per 256-thread workgroup on 32 x 8 tiles and a barrier per plane pair (lga_apply_pp_w*; default: whole step -1.4 ... -3.5 % over five boxes against 1, profiles/r8*_ab_step*; where W % 4 != 0 on API-layout x: as 1), 1 plane-pair kernels with one ring per wave on 32 x 2 tiles (lga_apply_pp_* / lga_filter_grad_pp_*).  The filter gradient under 2: where x is pair-interleaved (the second pass of an LGA2) the taps of a 32 x 2 tile are split over a WAVE PAIR that shares the rings (lga_filter_grad_pair_xp: five waves per SIMD instead of three, the launch -7 %, round 6 profiles/r9k_*, r9l_*); with an API-layout x it runs the one-wave kernel in both settings (its wave-pair form lost +3 %, its four-wave ring form of round 5 was no gain: both removed).  0 the 256-thread tile kernels (any radius; the general fallback)
  std::atomic<int> lga_mix{1};      // plane-pair forward / data-backward: mixed item list (whole tiles + segments of the rest); 0 off, 1 on (measured: forward pass 0.103 -> 0.0955 ms, profiles/r3a_*), n > 1: n SIMDs assumed (tests)
  std::atomic<int> lga_segs{0};     // depth segments per tile for the plane-pair forward / data-backward (0 = automatic)
  std::atomic<int> wide_col{1};     // vertical scans: LDS-staged column blocks with one wavefront per column (1,024-thread blocks): 1 for inputs with few column blocks and D >= 96 (measured on [1,1,192,240,624]: forward 0.33 -> 0.21 ms, adjoint 0.52 -> 0.40, profiles/r3a_check_wide_col.txt), 0 never, 2 whenever the kernel applies (tests)
  std::atomic<int> wide_scan{1};    // SGA scans with the whole wavefront on one scanline: 1 for inputs with few scanlines (and D > 272), 0 never, 2 whenever D > 48 (tests)
  std::atomic<int> rowwave{1};      // horizontal scans: one wavefront per row, LDS-staged (sga_row_kernels.h); 0 = segment kernels (the fallback)
  std::atomic<int> colblock{1};     // vertical scans: 16-column blocks, LDS-staged (sga_col_kernels.h); 0 = segment kernels (the fallback)
};
Options g_opt;
std::once_flag g_opt_once;

// the one place an option value is normalised: ganet_set_option and the environment go through it (GANET_LGA_WAVE=-1
// from the environment used to be stored as is, truthy, and reported as -1)
bool store_option(const char *name, int value)
{
  auto tri = [](int v) { return v < 0 ? 0 : (v > 2 ? 2 : v); };
  if (!strcmp(name, "GANET_LGA_WAVE")) g_opt.lga_wave = tri(value);
  else if (!strcmp(name, "GANET_SGA_TILED")) g_opt.sga_tiled = value ? 1 : 0;
  else if (!strcmp(name, "GANET_LGA_MIX")) g_opt.lga_mix = value < 0 ? 0 : value;      // 1: S = SIMDs of the device; n > 1: S = n (tests)
  else if (!strcmp(name, "GANET_LGA_SEGS")) g_opt.lga_segs = value > 0 ? value : 0;
  else if (!strcmp(name, "GANET_SGA_WIDE_SCAN")) g_opt.wide_scan = tri(value);
  else if (!strcmp(name, "GANET_SGA_WIDE_COL")) g_opt.wide_col = tri(value);
  else if (!strcmp(name, "GANET_SGA_ROWWAVE")) g_opt.rowwave = value ? 1 : 0;
  else if (!strcmp(name, "GANET_SGA_COLBLOCK")) g_opt.colblock = value ? 1 : 0;
  else return false;
  return true;
}

void load_env_options()
{
  for (const char *name : {"GANET_LGA_WAVE", "GANET_SGA_TILED", "GANET_LGA_SEGS", "GANET_LGA_MIX", "GANET_SGA_WIDE_SCAN",
                           "GANET_SGA_WIDE_COL", "GANET_SGA_ROWWAVE", "GANET_SGA_COLBLOCK"}) {
    const char *v = getenv(name);
    if (v && *v) store_option(name, atoi(v));
  }
}
const Options &opts()
{
  std::call_once(g_opt_once, load_env_options);
  return g_opt;
}

// ---- SGA kernel selection -----------------------------------------------------------
// (lanes per scanline GD, disparities per lane DPL) pairs compiled in.
// GD = 64: the whole wavefront owns one scanline (D up to 1,088; 4x the waves of GD = 16 for inputs with few scanlines)
#define GA_SGA_PAIRS(X) \
  X(16, 1) X(16, 2) X(16, 3) X(16, 5) X(16, 9) X(16, 13) X(16, 17) \
  X(64, 3) X(64, 5) X(64, 9) X(64, 17)

constexpr int fwd_sb(int dpl) { return dpl <= 3 ? 8 : dpl <= 5 ? 4 : dpl <= 9 ? 2 : 1; }
constexpr int fwd_nv(int dpl) { return dpl <= 5 ? 2 : 1; }
constexpr int bwd_sb(int dpl) { return dpl <= 3 ? 8 : dpl <= 5 ? 4 : dpl <= 9 ? 2 : 1; }
constexpr int bwd_nv(int dpl) { return dpl <= 5 ? 2 : 1; }

bool pick_pair(int D, int want_gd, int *gd, int *dpl)
{
  int best_gd = 0, best_dpl = 0;
#define X(G, P)                                                                    \
  if ((G) == want_gd && (G) * (P) >= D && (best_gd == 0 || (P) < best_dpl)) {      \
    best_gd = (G);                                                                 \
    best_dpl = (P);                                                                \
  }
  GA_SGA_PAIRS(X)
#undef X
  if (best_gd == 0 && want_gd == 16) return pick_pair(D, 64, gd, dpl);      // D > 272: wave-wide scanlines
  if (best_gd == 0) return false;
  *gd = best_gd;
  *dpl = best_dpl;
  return true;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

ScanGeom make_geom(int S, int D, int H, int W, int dir, bool backward)
{
  ScanGeom g;
  g.D = D; g.H = H; g.W = W;
  g.HW = (i64)H * W;
  const bool vertical = dir < 2;
  g.L = vertical ? H : W;
  g.Q = vertical ? W : H;
  g.total_lines = S * g.Q;
  g.line_stride = vertical ? 1 : W;
  const i64 unit = vertical ? W : 1;
  // forward visit order: down/right ascending, up/left descending; backward reverses it
  const bool ascending = ((dir == 0 || dir == 2) != backward);
  g.step_stride = ascending ? unit : -unit;
  g.start = ascending ? 0 : (i64)(g.L - 1) * unit;
  return g;
}

template <int GD, int DPL>
int launch_scan_fwd(const float *x, const float *g, float *A, int S, int D, int H, int W, int dir,
                    hipStream_t st)
{
  ScanGeom geo = make_geom(S, D, H, W, dir, false);
  const bool rowvec = DPL <= 17 && dir >= 2 && (W % 4 == 0) && aligned16(x) && aligned16(g) && aligned16(A);
  const int block = dir < 2 ? 128 : 64;      // (measured defaults of round 1, profiles/r1_tune_sga.txt)
  const int lpb = block / GD;
  const int grid = (geo.total_lines + lpb - 1) / lpb;
  if (rowvec) {
    if constexpr (DPL <= 17)
      GA_LAUNCH((sga_fwd_rowvec<GD, DPL, fwd_nv(DPL)>), dim3(grid), dim3(block), st, x, g, A, geo,
                dir == 3 ? 1 : 0);
  } else {
    GA_LAUNCH((sga_fwd_strided<GD, DPL, fwd_sb(DPL)>), dim3(grid), dim3(block), st, x, g, A, geo);
  }
  return check_launch("sga scan forward");
}

template <int GD, int DPL>
int launch_scan_bwdg(const float *g, const uint8_t *mask, const uint16_t *kp, const float *gout,
                     float *G, int S, int D, int H, int W, int dir, hipStream_t st)
{
  ScanGeom geo = make_geom(S, D, H, W, dir, true);
  const bool rowvec = DPL <= 17 && dir >= 2 && (W % 4 == 0) && aligned16(g) && aligned16(gout) &&
                      aligned16(G) && (((uintptr_t)mask & 3) == 0) && (((uintptr_t)kp & 7) == 0);
  const int block = dir < 2 ? 128 : 64;      // (measured defaults of round 1, profiles/r1_tune_sga.txt)
  const int lpb = block / GD;
  const int grid = (geo.total_lines + lpb - 1) / lpb;
  if (rowvec) {
    // the adjoint of `right` (2) visits w descending, of `left` (3) ascending
    if constexpr (DPL <= 17)
      GA_LAUNCH((sga_bwdg_rowvec<GD, DPL, bwd_nv(DPL)>), dim3(grid), dim3(block), st, g, mask, kp, gout,
                G, geo, dir, dir == 2 ? 1 : 0);
  } else {
    GA_LAUNCH((sga_bwdg_strided<GD, DPL, bwd_sb(DPL), uint8_t>), dim3(grid), dim3(block), st, g, mask,
              kp, gout, G, geo, dir);
  }
  return check_launch("sga adjoint scan");
}

// float-valued mask (reference buffer contract): element-strided traversal, GD = 16 family (64 for D > 272)
template <int GD, int DPL>
int launch_scan_bwdg_f32mask(const float *g, const float *mask, const uint16_t *kp, const float *gout,
                             float *G, int S, int D, int H, int W, int dir, hipStream_t st)
{
  ScanGeom geo = make_geom(S, D, H, W, dir, true);
  const int block = 64, lpb = block / GD;
  const int grid = (geo.total_lines + lpb - 1) / lpb;
  GA_LAUNCH((sga_bwdg_strided<GD, DPL, bwd_sb(DPL), float>), dim3(grid), dim3(block), st, g, mask, kp,
            gout, G, geo, dir);
  return check_launch("sga adjoint scan (f32 mask)");
}

int check_dims5(const char *who, int N, int C, int D, int H, int W)
{
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "%s: non-positive size N=%d C=%d D=%d H=%d W=%d", who, N, C, D, H, W);
  if ((i64)D * H * W > 0x7fffffffLL / 4 || (i64)N * C * H > 0x7fffffffLL || (i64)N * C * W > 0x7fffffffLL)
    return fail(GANET_E_UNSUPPORTED, "%s: slice too large for 32-bit line indexing", who);
  return GANET_OK;
}

// ---- horizontal scans, one wavefront per row (sga_row_kernels.h) --------------------------

bool rowwave_ok(int D, int W, int dir, size_t smem)
{
  return opts().rowwave && dir >= 2 && W % 4 == 0 && D <= 16 * 13 && smem <= ROW_SMEM_MAX;
}

int row_fwd(const float *x, const float *g, float *A, int S, int D, int H, int W, int dir, hipStream_t st,
            int out_mode = 0, int C = 1, const float *scale = nullptr, const float *shift = nullptr)
{
  launch_row_fwd(x, g, A, S, D, H, W, dir, st, out_mode, C, scale, shift);   // sga_row_tu.hip (own translation unit, own flags)
  return check_launch("sga row forward");
}

// ---- vertical scans over LDS-staged 16-column blocks (sga_col_kernels.h) ----------------------------
size_t col_smem_fwd(int D) { return sizeof(float) * ((size_t)2 * COL_NC * D * COL_SBV + COL_NC * 5 * COL_SBV); }
size_t col_smem_bwdg(int D)
{
  return sizeof(float) * ((size_t)2 * COL_NC * D * COL_SBV + COL_NC * 5 * COL_SBV + COL_NC * COL_SBV) +
         (size_t)D * COL_SBV * 16;
}
bool colblock_ok(int D, int W, int dir, size_t smem)
{
  return opts().colblock && dir < 2 && W % 4 == 0 && D <= 16 * 13 && smem <= ROW_SMEM_MAX;
}

int col_fwd(const float *x, const float *g, float *A, int S, int D, int H, int W, int dir, hipStream_t st,
            int out_mode = 0)
{
  ColGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W; geo.out_mode = out_mode; geo.tiled = 0;
  const int dpl = row_dpl(D);
  const bool full = dpl > 0 && D % dpl == 0;
  const size_t smem = col_smem_fwd(D);
  const dim3 grid((W + COL_NC - 1) / COL_NC, S), block(256);
#define X(P)                                                                                        \
  if (dpl == (P)) {                                                                                 \
    if (dir == 0 && full) GA_LAUNCH_SMEM((sga_col_fwd<P, true, true>), grid, block, smem, st, x, g, A, geo);      \
    else if (dir == 0) GA_LAUNCH_SMEM((sga_col_fwd<P, true, false>), grid, block, smem, st, x, g, A, geo);        \
    else if (full) GA_LAUNCH_SMEM((sga_col_fwd<P, false, true>), grid, block, smem, st, x, g, A, geo);            \
    else GA_LAUNCH_SMEM((sga_col_fwd<P, false, false>), grid, block, smem, st, x, g, A, geo);                     \
  }
  GA_ROW_DPLS(X)
#undef X
  return check_launch("sga column-block forward");
}

// ---- the same column blocks with one WAVEFRONT per column (sga_col_fwd_wide / sga_col_bwdg_wide; GANET_SGA_WIDE_COL) -----
// For inputs with few column blocks (SURVEY 8d's stress shape [1,1,192,240,624]: 39 blocks) and many disparities: 16 waves
// per block instead of 4, 3-5 disparities of serial work per lane instead of 12-13, and -- unlike the register-only wide
// segment kernels -- the same 64-byte global pieces as the 16-lane column blocks.  D <= 192; up to 112 KB of LDS per block.
int device_cus();
constexpr size_t COL_WIDE_SMEM_MAX = 152 * 1024;
int col_wide_dpl(int D) { return D <= 64 * 3 ? 3 : 0; }      // (5 disparities per lane spill at the 128 registers a 1,024-thread block leaves)
bool col_wide_ok(int D, int W, int dir, size_t smem, int S)
{
  const int mode = opts().wide_col;
  if (!mode || dir >= 2 || W % 4 != 0 || col_wide_dpl(D) <= 0 || smem > COL_WIDE_SMEM_MAX || S > 65535) return false;
  if (mode >= 2) return true;
  // automatic: fewer 16-column blocks than half the compute units (the 16-lane blocks would leave most SIMDs idle, each with
  // 12-13 disparities of serial work per lane) -- the literal stress shape [1,1,192,240,624] has 39
  return D >= 96 && (i64)S * ((W + COL_NC - 1) / COL_NC) * 2 <= device_cus();
}

int col_fwd_wide(const float *x, const float *g, float *A, int S, int D, int H, int W, int dir, hipStream_t st)
{
  ColGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W; geo.out_mode = 0; geo.tiled = 0;
  const int dpl = col_wide_dpl(D);
  const bool full = D % dpl == 0;
  const size_t smem = col_smem_fwd(D);
  const dim3 grid((W + COL_NC - 1) / COL_NC, S), block(1024);
#define X(P)                                                                                        \
  if (dpl == (P)) {                                                                                 \
    if (dir == 0 && full) GA_LAUNCH_SMEM_BIG((sga_col_fwd_wide<P, true, true>), grid, block, smem, st, x, g, A, geo);   \
    else if (dir == 0) GA_LAUNCH_SMEM_BIG((sga_col_fwd_wide<P, true, false>), grid, block, smem, st, x, g, A, geo);     \
    else if (full) GA_LAUNCH_SMEM_BIG((sga_col_fwd_wide<P, false, true>), grid, block, smem, st, x, g, A, geo);         \
    else GA_LAUNCH_SMEM_BIG((sga_col_fwd_wide<P, false, false>), grid, block, smem, st, x, g, A, geo);                  \
  }
  X(3)
#undef X
  static const bool trace = getenv("GANET_TRACE_DISPATCH") != nullptr;
  if (trace) fprintf(stderr, "[ganet] sga_col_fwd_wide DPL=%d dir=%d D=%d smem=%zu\n", dpl, dir, D, smem);
  return check_launch("sga wide column-block forward");
}

int col_bwdg_wide(const float *g, const uint8_t *mask, const uint16_t *kp, const float *gout, float *G,
                  int S, int D, int H, int W, int dir, hipStream_t st)
{
  ColGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W; geo.out_mode = 0; geo.tiled = 0;
  const int dpl = col_wide_dpl(D);
  const size_t smem = col_smem_bwdg(D);
  const dim3 grid((W + COL_NC - 1) / COL_NC, S), block(1024);
  const bool m16 = W % 16 == 0 && aligned16(mask);
#define X(P)                                                                                        \
  if (dpl == (P)) {                                                                                 \
    if (dir == 1 && m16) GA_LAUNCH_SMEM_BIG((sga_col_bwdg_wide<P, true, true, false>), grid, block, smem, st, g, mask, kp, gout, G, geo, dir);  \
    else if (dir == 1) GA_LAUNCH_SMEM_BIG((sga_col_bwdg_wide<P, true, false, false>), grid, block, smem, st, g, mask, kp, gout, G, geo, dir);   \
    else if (m16) GA_LAUNCH_SMEM_BIG((sga_col_bwdg_wide<P, false, true, false>), grid, block, smem, st, g, mask, kp, gout, G, geo, dir);        \
    else GA_LAUNCH_SMEM_BIG((sga_col_bwdg_wide<P, false, false, false>), grid, block, smem, st, g, mask, kp, gout, G, geo, dir);                \
  }
  X(3)
#undef X
  static const bool trace = getenv("GANET_TRACE_DISPATCH") != nullptr;
  if (trace) fprintf(stderr, "[ganet] sga_col_bwdg_wide DPL=%d dir=%d D=%d smem=%zu\n", dpl, dir, D, smem);
  return check_launch("sga wide column-block adjoint scan");
}

int col_bwdg(const float *g, const uint8_t *mask, const uint16_t *kp, const float *gout, float *G,
             int S, int D, int H, int W, int dir, hipStream_t st, int tiled = 0)
{
  ColGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W; geo.out_mode = 0; geo.tiled = tiled;
  const int dpl = row_dpl(D);
  const size_t smem = col_smem_bwdg(D);
  const dim3 grid((W + COL_NC - 1) / COL_NC, S), block(256);
  // the adjoint of `down` (0) walks rows upwards (H-1..0), of `up` (1) downwards
  const bool m16 = W % 16 == 0 && aligned16(mask);
  // (lanes wholly inside / outside [0, D): the leaner recurrence of bwdg_step<FULL>, instantiated for the depths the models use
  //  -- 33 and 48 at three, 65 at five disparities per lane)
  const bool full = D % dpl == 0;
#define L(P, ASC, M, F) GA_LAUNCH_SMEM((sga_col_bwdg<P, ASC, M, F>), grid, block, smem, st, g, mask, kp, gout, G, geo, dir)
#define X(P)                                                                                        \
  if (dpl == (P)) {                                                                                 \
    constexpr bool FL = (P) == 3 || (P) == 5;                                                       \
    if (full && FL) {                                                                               \
      if (dir == 1 && m16) L(P, true, true, FL); else if (dir == 1) L(P, true, false, FL);          \
      else if (m16) L(P, false, true, FL); else L(P, false, false, FL);                             \
    } else {                                                                                        \
      if (dir == 1 && m16) L(P, true, true, false); else if (dir == 1) L(P, true, false, false);    \
      else if (m16) L(P, false, true, false); else L(P, false, false, false);                       \
    }                                                                                               \
  }
  GA_ROW_DPLS(X)
#undef X
#undef L
  return check_launch("sga column-block adjoint scan");
}

int row_bwdg(const float *g, const uint8_t *mask, const uint16_t *kp, const float *gout, float *G,
             int S, int D, int H, int W, int dir, hipStream_t st)
{
  launch_row_bwdg(g, mask, kp, gout, G, S, D, H, W, dir, st);       // sga_row_tu.hip
  return check_launch("sga row adjoint scan");
}

int device_cus();

// Inputs with few scanlines (a single slice at full resolution: SURVEY 8d's literal stress shape [1,1,192,240,624] has 624
// columns / 240 rows) leave most of the 1,024 SIMDs without a wave when a scanline is 16 lanes wide (156 / 240 waves): the
// segment kernels with the whole wavefront on one scanline give 4x the waves and a quarter of the serial work per position.
// Measured on that shape (profiles/r2n_sga_wide_scan_stress_shape.txt): horizontal scans 0.25 -> 0.10 ms forward and
// 0.31 -> 0.21 ms adjoint; VERTICAL scans do not gain (0.33 -> 0.32, adjoint 0.49 -> 0.54-0.64 ms): with one column per wave
// every lane of a load touches a different plane (64 cache lines for 256 bytes), which the LDS-staged column blocks avoid.
// So the automatic mode widens horizontal scans only; D > 272 takes the wide kernels in every direction (pick_pair).
bool few_lines(int N, int C, int D, int H, int W, int dir)
{
  if (opts().wide_scan == 0) return false;
  if (opts().wide_scan == 2) return D > 48;
  if (dir < 2) return false;
  const i64 lines = (i64)N * C * H;
  return D >= 96 && lines * 16 < (i64)4 * device_cus() * 64 * 2;      // fewer 16-lane segments than two waves per SIMD hold
}

// 1 if ganet_sga_backward keeps G_down / G_up in the tiled layout of sga_col_kernels.h for these dimensions: where the 16-lane
// column-block adjoint kernel runs (not the wide one), W % 16 == 0, H % 4 == 0.  Decided from the dimensions and GANET_SGA_TILED.
int sga_ws_tiled(int N, int C, int D, int H, int W)
{
  if (!opts().sga_tiled || W % 16 != 0 || H % 4 != 0 || N * C > 65535 || opts().wide_scan == 2) return 0;
  return (!col_wide_ok(D, W, 0, col_smem_bwdg(D), N * C) && colblock_ok(D, W, 0, col_smem_bwdg(D))) ? 1 : 0;
}

int scan_fwd(const float *x, const float *g, float *A, int N, int C, int D, int H, int W, int dir,
             hipStream_t st)
{
  if (col_wide_ok(D, W, dir, col_smem_fwd(D), N * C) && aligned16(x) && aligned16(g) && aligned16(A))
    return col_fwd_wide(x, g, A, N * C, D, H, W, dir, st);
  if (few_lines(N, C, D, H, W, dir)) {
    int gd, dpl;
    if (pick_pair(D, 64, &gd, &dpl)) {
#define X(G, P) \
      if ((G) == 64 && dpl == (P)) return launch_scan_fwd<G, P>(x, g, A, N * C, D, H, W, dir, st);
      GA_SGA_PAIRS(X)
#undef X
    }
  }
  if (rowwave_ok(D, W, dir, row_smem_fwd(D)) && aligned16(x) && aligned16(g) && aligned16(A))
    return row_fwd(x, g, A, N * C, D, H, W, dir, st);
  if (colblock_ok(D, W, dir, col_smem_fwd(D)) && aligned16(x) && aligned16(g) && aligned16(A) && N * C <= 65535)
    return col_fwd(x, g, A, N * C, D, H, W, dir, st);
  int gd, dpl;
  if (!pick_pair(D, 16, &gd, &dpl))
    return fail(GANET_E_UNSUPPORTED, "SGA: D=%d exceeds the compiled maximum (1088)", D);
#define X(G, P) \
  if (gd == (G) && dpl == (P)) return launch_scan_fwd<G, P>(x, g, A, N * C, D, H, W, dir, st);
  GA_SGA_PAIRS(X)
#undef X
  return fail(GANET_E_UNSUPPORTED, "SGA: no kernel for GD=%d DPL=%d", gd, dpl);
}

int scan_bwdg(const float *g, const uint8_t *mask, const uint16_t *kp, const float *gout, float *G,
              int N, int C, int D, int H, int W, int dir, hipStream_t st, int tiled = 0)
{
  if (tiled) {
    if (!(aligned16(g) && aligned16(gout) && aligned16(G) && (((uintptr_t)mask & 3) == 0)))
      return fail(GANET_E_UNSUPPORTED, "SGA: tiled workspace needs 16-byte aligned volumes");
    return col_bwdg(g, mask, kp, gout, G, N * C, D, H, W, dir, st, 1);
  }
  if (col_wide_ok(D, W, dir, col_smem_bwdg(D), N * C) && aligned16(g) && aligned16(gout) && aligned16(G) &&
      (((uintptr_t)mask & 3) == 0))
    return col_bwdg_wide(g, mask, kp, gout, G, N * C, D, H, W, dir, st);
  if (few_lines(N, C, D, H, W, dir)) {
    int gd, dpl;
    if (pick_pair(D, 64, &gd, &dpl)) {
#define X(G_, P) \
      if ((G_) == 64 && dpl == (P)) return launch_scan_bwdg<G_, P>(g, mask, kp, gout, G, N * C, D, H, W, dir, st);
      GA_SGA_PAIRS(X)
#undef X
    }
  }
  if (rowwave_ok(D, W, dir, row_smem_bwdg(D)) && aligned16(g) && aligned16(gout) && aligned16(G) &&
      (((uintptr_t)mask & 3) == 0) && (((uintptr_t)kp & 7) == 0))
    return row_bwdg(g, mask, kp, gout, G, N * C, D, H, W, dir, st);
  if (colblock_ok(D, W, dir, col_smem_bwdg(D)) && aligned16(g) && aligned16(gout) && aligned16(G) &&
      (((uintptr_t)mask & 3) == 0) && N * C <= 65535)
    return col_bwdg(g, mask, kp, gout, G, N * C, D, H, W, dir, st);
  int gd, dpl;
  if (!pick_pair(D, 16, &gd, &dpl))
    return fail(GANET_E_UNSUPPORTED, "SGA: D=%d exceeds the compiled maximum (1088)", D);
#define X(G_, P) \
  if (gd == (G_) && dpl == (P)) return launch_scan_bwdg<G_, P>(g, mask, kp, gout, G, N * C, D, H, W, dir, st);
  GA_SGA_PAIRS(X)
#undef X
  return fail(GANET_E_UNSUPPORTED, "SGA: no kernel for GD=%d DPL=%d", gd, dpl);
}

int scan_bwdg_f32mask(const float *g, const float *mask, const uint16_t *kp, const float *gout, float *G,
                      int N, int C, int D, int H, int W, int dir, hipStream_t st)
{
  int gd, dpl;
  if (!pick_pair(D, 16, &gd, &dpl))
    return fail(GANET_E_UNSUPPORTED, "SGA: D=%d exceeds the compiled maximum (1088)", D);
#define X(G_, P) \
  if (((G_) == 16 || (G_) == 64) && gd == (G_) && dpl == (P)) return launch_scan_bwdg_f32mask<G_, P>(g, mask, kp, gout, G, N * C, D, H, W, dir, st);
  GA_SGA_PAIRS(X)
#undef X
  return fail(GANET_E_UNSUPPORTED, "SGA: no kernel for DPL=%d", dpl);
}

int px_grid(i64 npix)
{
  i64 g = (npix + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return (int)g;
}

// per-pixel gradients for `ndir` (1 or 4) directions
int bwd_point(const float *x, float *gx, const PointArgs &pa, int ndir, int N, int C, int D, int H, int W,
              int accumulate, hipStream_t st, int tiled = 0)
{
  const i64 npix = (i64)N * C * H * W;
  const int pb = 256;      // (64 / 128 measured equal)
  i64 gsz = (npix + pb - 1) / pb;
  const i64 gmax = (i64)256 * 32 * (256 / pb);
  if (gsz > gmax) gsz = gmax;
  if (gsz < 1) gsz = 1;
  if (ndir == 4 && !accumulate && tiled) GA_LAUNCH((sga_bwd_point<4, false, true>), dim3((unsigned)gsz), dim3(pb), st, x, gx, pa, D, H, W, npix);
  else if (ndir == 4 && accumulate) GA_LAUNCH((sga_bwd_point<4, true>), dim3((unsigned)gsz), dim3(pb), st, x, gx, pa, D, H, W, npix);
  else if (ndir == 4) GA_LAUNCH((sga_bwd_point<4, false>), dim3((unsigned)gsz), dim3(pb), st, x, gx, pa, D, H, W, npix);
  else if (accumulate) GA_LAUNCH((sga_bwd_point<1, true>), dim3((unsigned)gsz), dim3(pb), st, x, gx, pa, D, H, W, npix);
  else GA_LAUNCH((sga_bwd_point<1, false>), dim3((unsigned)gsz), dim3(pb), st, x, gx, pa, D, H, W, npix);
  return check_launch("sga per-pixel gradients");
}

int ew_grid(i64 n)
{
  i64 g = (n + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// compute units of the current device (256 on MI355X); the emulator build reports 256
int device_cus()
{
#if defined(GA_HIPSIM)
  return 256;
#else
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
#endif
}

// ---- LGA dispatch -------------------------------------------------------------------
// Work items of the plane-pair forward / data-backward (LgaSegMix).  Every item re-gathers its pixels' filter taps (75 loads per
// lane; for the data-backward from 25 neighbouring pixels) and fills its ring, 0.038 ms of a 0.107 ms pass
// (profiles/r2f_lga_pp_ablation.txt), so whole tiles are best (profiles/r2e_ab_lga_plane_pairs_v2.txt) unless there are too few
// of them to fill the wave slots; with a whole number q < 3 of tiles per SIMD plus a remainder, the remainder is cut into
// segments, at most one per SIMD (the mixed list).  `whole_only`: the pass carries a per-pixel reduction over all of D.
// `wg`: items of the workgroup kernels (32 x 8 tiles, four waves each: the unit that takes an item is a CU, not a SIMD).
LgaSegMix lga_items(int W, int H, int B, int D, bool whole_only, i64 *items, bool wg = false)
{
  LgaSegMix mx;
  const int th = wg ? 8 : LGAW_TH, units = (wg ? 1 : 4) * device_cus();
  mx.tiles_x = (W + LGA_TW - 1) / LGA_TW;
  mx.tiles_y = (H + th - 1) / th;
  const i64 tiles = (i64)mx.tiles_x * mx.tiles_y * B;
  mx.n_whole = (int)(tiles < (1ll << 30) ? tiles : (1ll << 30));
  mx.nsub = 1;
  mx.sub_len = (D + 1) & ~1;
  *items = tiles;
  if (whole_only || tiles >= (1ll << 30)) return mx;
  const i64 slots = (i64)LGA_WAVES_PER_SIMD * units;
  int nseg = opts().lga_segs;
  if (nseg <= 0) {
    nseg = tiles * 2 > slots ? 1 : (int)((slots + tiles - 1) / tiles);
    if (nseg > D / 16) nseg = D / 16 > 1 ? D / 16 : 1;
  }
  if (nseg > D) nseg = D;
  if (nseg > 1) {                                   // every tile cut into equal segments
    mx.n_whole = 0;
    mx.sub_len = ((D + nseg - 1) / nseg + 1) & ~1;
    mx.nsub = (D + mx.sub_len - 1) / mx.sub_len;
    *items = tiles * mx.nsub;
    return mx;
  }
  const int mixo = opts().lga_mix;
  const i64 S = mixo > 1 ? (i64)mixo : (i64)units;
  if (mixo && opts().lga_segs <= 0 && tiles / S < LGA_WAVES_PER_SIMD && tiles % S != 0) {
    const i64 r = tiles % S;
    int nsub = (int)(S / r);
    if (nsub > D / 16) nsub = D / 16;
    if (nsub >= 2) {
      mx.n_whole = (int)(tiles - r);
      mx.sub_len = ((D + nsub - 1) / nsub + 1) & ~1;
      mx.nsub = (D + mx.sub_len - 1) / mx.sub_len;
      *items = mx.n_whole + r * mx.nsub;
    }
  }
  return mx;
}

#ifndef GA_LGA_PLANAR
#ifndef GA_FGP_DEFAULT
#define GA_FGP_DEFAULT 1  // (0: a build whose GANET_LGA_WAVE=2 keeps the one-wave filter gradient -- same-box A/B builds, scripts/build_variants.py)
#endif
#ifndef GA_FGP_WPS
#define GA_FGP_WPS 5      // waves per SIMD the wave-pair filter gradient is compiled for
#endif
#define GA_LGA_PLANAR 1     // API-layout operands of the plane-pair kernels staged planar by 16-byte copies where W % 4 == 0
#endif
template <int R>
int launch_lga_fwd(const float *x, const float *f, float *y, int B, int D, int H, int W,
                   bool transposed, hipStream_t st)
{
  LgaGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W;
  if constexpr (R <= 2) {
    const int family = opts().lga_wave;
    if (family && (i64)H * W < (1ll << 28)) {
      i64 items;
      bool planar = false;
      if constexpr (R == 2) planar = GA_LGA_PLANAR && W % 4 == 0 && aligned16(x);      // input staged by 16-byte copies
      if constexpr (R == 2) {
        if (planar && family == 2) {                      // one ring per 256-thread workgroup (32 x 8 tiles)
          const LgaSegMix wx = lga_items(W, H, B, D, false, &items, true);
          if (items < (1ll << 31)) {
            if (transposed) GA_LAUNCH((lga_apply_pp_wx<2, true>), dim3((unsigned)items), dim3(256), st, x, f, y, geo, wx);
            else GA_LAUNCH((lga_apply_pp_wx<2, false>), dim3((unsigned)items), dim3(256), st, x, f, y, geo, wx);
            return check_launch("lga apply (plane pairs, workgroup ring)");
          }
        }
      }
      const LgaSegMix mx = lga_items(W, H, B, D, false, &items);
      if (items < (1ll << 31)) {
        if constexpr (R == 2) {
          if (planar && transposed) GA_LAUNCH((lga_apply_pp_x<2, true>), dim3((unsigned)items), dim3(64), st, x, f, y, geo, mx);
          else if (planar) GA_LAUNCH((lga_apply_pp_x<2, false>), dim3((unsigned)items), dim3(64), st, x, f, y, geo, mx);
        }
        if (planar) {}
        else if (transposed) GA_LAUNCH((lga_apply_pp<R, true>), dim3((unsigned)items), dim3(64), st, x, f, y, geo, mx);
        else GA_LAUNCH((lga_apply_pp<R, false>), dim3((unsigned)items), dim3(64), st, x, f, y, geo, mx);
        static const bool trace = getenv("GANET_TRACE_DISPATCH") != nullptr;      // (development: which LGA kernel ran)
        if (trace) fprintf(stderr, "[ganet] lga_apply_pp R=%d T=%d items=%lld: %d whole tiles + segments of %d planes x %d\n", R, (int)transposed, (long long)items, mx.n_whole, mx.sub_len, mx.nsub);
        return check_launch("lga apply (plane pairs)");
      }
    }
  }
  const dim3 grid((W + LGA_TW - 1) / LGA_TW, (H + LGA_TH - 1) / LGA_TH, B);
  if (transposed) GA_LAUNCH((lga_apply<R, true>), grid, dim3(LGA_NT), st, x, f, y, geo);
  else GA_LAUNCH((lga_apply<R, false>), grid, dim3(LGA_NT), st, x, f, y, geo);
  return check_launch("lga apply");
}

// one LGA pass / data-backward with one side in the pair-interleaved layout (lga_apply_pp_pi / lga_apply_pp_po / lga_apply_pp_xo)
int launch_lga_paired(const float *x, const float *f, float *y, int B, int D, int H, int W, bool transposed, bool x_paired,
                      hipStream_t st, float *edge = nullptr)
{
  if ((i64)H * W >= (1ll << 28) || W % 2 != 0 || !aligned16(x) || !aligned16(y))
    return fail(GANET_E_UNSUPPORTED, "ganet_lga_apply_paired: needs W even, planes below 2^28 pixels and 16-byte aligned volumes");
  LgaGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W;
  i64 items;
  float *const none = nullptr;
  const bool wg_ring = opts().lga_wave == 2;
  if (x_paired && wg_ring) {                                       // one ring per 256-thread workgroup (32 x 8 tiles)
    const LgaSegMix wx = lga_items(W, H, B, D, false, &items, true);
    if (items < (1ll << 31)) {
      if (transposed) GA_LAUNCH((lga_apply_pp_wpi<2, true>), dim3((unsigned)items), dim3(256), st, x, f, y, geo, wx, none, none, edge);
      else GA_LAUNCH((lga_apply_pp_wpi<2, false>), dim3((unsigned)items), dim3(256), st, x, f, y, geo, wx, none, none, edge);
      return check_launch("lga apply (plane pairs, interleaved input, workgroup ring)");
    }
  }
  if (!x_paired && GA_LGA_PLANAR && W % 4 == 0 && wg_ring) {      // one ring per 256-thread workgroup (32 x 8 tiles)
    const LgaSegMix wx = lga_items(W, H, B, D, false, &items, true);
    if (items < (1ll << 31)) {
      if (transposed) GA_LAUNCH((lga_apply_pp_wxo<2, true>), dim3((unsigned)items), dim3(256), st, x, f, y, geo, wx, none, none, edge);
      else GA_LAUNCH((lga_apply_pp_wxo<2, false>), dim3((unsigned)items), dim3(256), st, x, f, y, geo, wx, none, none, edge);
      return check_launch("lga apply (plane pairs, interleaved volume, workgroup ring)");
    }
  }
  const LgaSegMix sg = lga_items(W, H, B, D, false, &items);
  if (items >= (1ll << 31)) return fail(GANET_E_UNSUPPORTED, "ganet_lga_apply_paired: too many tiles");
#define L(K, T) GA_LAUNCH((K<2, T>), dim3((unsigned)items), dim3(64), st, x, f, y, geo, sg, none, none, edge)
  if (x_paired) {
    if (transposed) L(lga_apply_pp_pi, true); else L(lga_apply_pp_pi, false);
  } else if (GA_LGA_PLANAR && W % 4 == 0) {      // API-layout input staged planar, two 16-byte copies per plane pair
    if (transposed) L(lga_apply_pp_xo, true); else L(lga_apply_pp_xo, false);
  } else {
    if (transposed) L(lga_apply_pp_po, true); else L(lga_apply_pp_po, false);
  }
#undef L
  return check_launch("lga apply (plane pairs, interleaved volume)");
}

int launch_lga_gf_paired(const float *x, const float *gy, float *gf, int B, int D, int H, int W, int acc, bool x_paired,
                         hipStream_t st)
{
  if ((i64)H * W >= (1ll << 28) || W % 2 != 0 || !aligned16(x) || !aligned16(gy))
    return fail(GANET_E_UNSUPPORTED, "ganet_lga_filter_grad_paired: needs W even, planes below 2^28 pixels and 16-byte aligned volumes");
  LgaGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W;
  LgaSeg sg;
  sg.tiles_x = (W + LGA_TW - 1) / LGA_TW;
  sg.tiles_y = (H + LGAW_TH - 1) / LGAW_TH;
  sg.nseg = 1; sg.seg_len = D;
  const i64 items = (i64)sg.tiles_x * sg.tiles_y * B;
  if (items >= (1ll << 31)) return fail(GANET_E_UNSUPPORTED, "ganet_lga_filter_grad_paired: too many tiles");
  const bool fg_pair = GA_FGP_DEFAULT && opts().lga_wave >= 2;      // the workgroup forms: here the taps of a tile split over a wave pair (lga_filter_grad_pair.inc)
  if (x_paired && fg_pair) GA_LAUNCH((lga_filter_grad_pair_xp<GA_FGP_WPS>), dim3((unsigned)items), dim3(128), st, x, gy, gf, geo, sg, acc);
  else if (x_paired) GA_LAUNCH((lga_filter_grad_pp_xp<2, 3, 0>), dim3((unsigned)items), dim3(64), st, x, gy, gf, geo, sg, acc);
  else if (GA_LGA_PLANAR && W % 4 == 0) GA_LAUNCH((lga_filter_grad_pp_gypx<2, 3, 0>), dim3((unsigned)items), dim3(64), st, x, gy, gf, geo, sg, acc);
  else GA_LAUNCH((lga_filter_grad_pp_gyp<2, 3, 0>), dim3((unsigned)items), dim3(64), st, x, gy, gf, geo, sg, acc);
  return check_launch("lga filter grad (plane pairs, interleaved volume)");
}

// one LGA pass whose output is also reduced over d per pixel (plane-pair kernel, one depth segment per tile)
template <int R>
int launch_lga_fwd_regress(const float *x, const float *f, float *y, float *snorm, float *sdy, int B, int D, int H, int W,
                           hipStream_t st)
{
  if constexpr (R <= 2) {
    if ((i64)H * W < (1ll << 28)) {
      LgaGeom geo;
      geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W;
      i64 items;
      const LgaSegMix sg = lga_items(W, H, B, D, true, &items);      // (the epilogue reduces over all of D: whole tiles only)
      if (items < (1ll << 31)) {
        bool planar = false;
        if constexpr (R == 2) {
          planar = GA_LGA_PLANAR && W % 4 == 0 && aligned16(x);
          if (planar) GA_LAUNCH((lga_apply_pp_x<2, false, true>), dim3((unsigned)items), dim3(64), st, x, f, y, geo, sg, snorm, sdy);
        }
        if (!planar) GA_LAUNCH((lga_apply_pp<R, false, true>), dim3((unsigned)items), dim3(64), st, x, f, y, geo, sg, snorm, sdy);
        return check_launch("lga apply + regression epilogue (plane pairs)");
      }
    }
  }
  return fail(GANET_E_UNSUPPORTED, "ganet_lga_forward_regress: no fused kernel for this shape / radius: run ganet_lga_forward "
                                   "and ganet_norm_disparity_regression_forward instead");
}

template <int R>
int launch_lga_gf(const float *x, const float *gy, float *gf, int B, int D, int H, int W, int acc,
                  hipStream_t st)
{
  LgaGeom geo;
  geo.D = D; geo.H = H; geo.W = W; geo.HW = (i64)H * W;
  if constexpr (R <= 2) {
    if (opts().lga_wave && (i64)H * W < (1ll << 28)) {
      LgaSeg sg;
      sg.tiles_x = (W + LGA_TW - 1) / LGA_TW;
      sg.tiles_y = (H + LGAW_TH - 1) / LGAW_TH;
      sg.nseg = 1; sg.seg_len = D;
      const i64 items = (i64)sg.tiles_x * sg.tiles_y * B;
      if (items < (1ll << 31)) {
        bool planar = false;
        if constexpr (R == 2) {
          planar = GA_LGA_PLANAR && W % 4 == 0 && aligned16(x);
          if (planar) GA_LAUNCH((lga_filter_grad_pp_x<2, 3, 0>), dim3((unsigned)items), dim3(64), st, x, gy, gf, geo, sg, acc);
        }
        if (!planar) GA_LAUNCH((lga_filter_grad_pp<R, 3, 0>), dim3((unsigned)items), dim3(64), st, x, gy, gf, geo, sg, acc);
        return check_launch("lga filter grad (plane pairs)");
      }
    }
  }
  const dim3 grid((W + LGA_TW - 1) / LGA_TW, (H + LGA_TH - 1) / LGA_TH, B);
  GA_LAUNCH((lga_filter_grad<R>), grid, dim3(LGA_NT), st, x, gy, gf, geo, acc);
  return check_launch("lga filter grad");
}

int check_lga(const char *who, int B, int D, int H, int W, int radius)
{
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "%s: non-positive size B=%d D=%d H=%d W=%d", who, B, D, H, W);
  if (radius < 1 || radius > 3)
    return fail(GANET_E_UNSUPPORTED, "%s: radius %d not in the compiled set {1,2,3}", who, radius);
  if (B > 65535) return fail(GANET_E_UNSUPPORTED, "%s: batch %d > 65535 (grid.z)", who, B);
  return GANET_OK;
}

__global__ void dpp_probe(int *out)
{
  const int lane = threadIdx.x;
  out[0 * 64 + lane] = dpp_i<DPP_QP_XOR1>(-1, lane);
  out[1 * 64 + lane] = dpp_i<DPP_QP_XOR2>(-1, lane);
  out[2 * 64 + lane] = dpp_i<DPP_ROW_SHL1>(-1, lane);
  out[3 * 64 + lane] = dpp_i<DPP_ROW_SHR1>(-1, lane);
  out[4 * 64 + lane] = dpp_i<DPP_ROW_MIRROR>(-1, lane);
  out[5 * 64 + lane] = dpp_i<DPP_ROW_HALF_MIRROR>(-1, lane);
  float v = (float)((lane * 37) % 64);
  int k = lane;
  seg_argmax<16>(v, k);
  out[6 * 64 + lane] = k;
  out[7 * 64 + lane] = (int)seg_allsum<16>((float)lane);
}

// whole-wavefront patterns of the 64-lanes-per-scanline scans
__global__ void dpp_probe_wave(int *out)
{
  const int lane = threadIdx.x;
  out[0 * 64 + lane] = dpp_i<DPP_WAVE_SHL1>(-1, lane);
  out[1 * 64 + lane] = dpp_i<DPP_WAVE_SHR1>(-1, lane);
  out[2 * 64 + lane] = (int)seg_allmax<64>((float)((lane * 37) % 64));
  out[3 * 64 + lane] = (int)seg_allsum<64>((float)lane);
}

}  // namespace

// =====================================================================================
GA_EXPORT int ganet_abi_version(void) { return GANET_ABI_VERSION; }
GA_EXPORT const char *ganet_last_error(void) { return g_err; }
GA_EXPORT int ganet_is_simulator(void)
{
#if defined(GA_HIPSIM)
  return 1;
#else
  return 0;
#endif
}

GA_EXPORT int ganet_get_option(const char *name)
{
  opts();
  if (!name) return fail(GANET_E_INVALID, "ganet_get_option: null name");
  if (!strcmp(name, "GANET_LGA_WAVE")) return g_opt.lga_wave;
  if (!strcmp(name, "GANET_SGA_TILED")) return g_opt.sga_tiled;
  if (!strcmp(name, "GANET_LGA_MIX")) return g_opt.lga_mix;
  if (!strcmp(name, "GANET_LGA_SEGS")) return g_opt.lga_segs;
  if (!strcmp(name, "GANET_SGA_WIDE_SCAN")) return g_opt.wide_scan;
  if (!strcmp(name, "GANET_SGA_WIDE_COL")) return g_opt.wide_col;
  if (!strcmp(name, "GANET_SGA_ROWWAVE")) return g_opt.rowwave;
  if (!strcmp(name, "GANET_SGA_COLBLOCK")) return g_opt.colblock;
  return fail(GANET_E_INVALID, "ganet_get_option: unknown option %s", name);
}

GA_EXPORT int ganet_set_option(const char *name, int value)
{
  opts();
  if (!name) return fail(GANET_E_INVALID, "ganet_set_option: null name");
  if (store_option(name, value)) {}      // a library option, normalised
#if defined(GA_HIPSIM)
  else if (!strcmp(name, "HIPSIM_LATE_DMA")) hipsim::S().late_dma = value != 0;   // emulator only: see tests/hipsim/hipsim.h
  else if (!strcmp(name, "HIPSIM_LANE_ORDER")) hipsim::S().lane_order = value ? 1 : 0;
  else if (!strcmp(name, "HIPSIM_WAVE_GREEDY")) hipsim::S().wave_greedy = value ? 1 : 0;   // one wavefront runs as far as its synchronisation lets it
  else if (!strcmp(name, "HIPSIM_LATE_LDS")) hipsim::S().late_lds = value != 0;
  else if (!strcmp(name, "HIPSIM_LGKM_SLACK")) hipsim::S().lgkm_slack = value;     // tests: every counted LDS wait loosened by `value`
  else if (!strcmp(name, "HIPSIM_VMCNT_SLACK")) hipsim::S().vm_slack = value;      // tests: every counted copy wait loosened by `value`
#endif
  else {
    // names retired in ABI 7 / 10 (their kernels were removed, became the only path, or the switch moved into another option:
    // GANET_LGA_WG -> GANET_LGA_WAVE = 2 | 1): accepted and ignored, so that a caller written against an older ABI keeps running;
    // one note per process
    static const char *const retired[] = {"GANET_LGA_BWD_STREAMS", "GANET_LGA_FG_WPS", "GANET_LGA_SPLIT", "GANET_LGA_VMCNT_SAFE",
                                          "GANET_SGA_BLOCK_H", "GANET_SGA_BLOCK_V", "GANET_SGA_GD", "GANET_SGA_GD_H", "GANET_SGA_GD_V",
                                          "GANET_SGA_INFER_FUSED", "GANET_SGA_MERGE4", "GANET_SGA_POINT_BLOCK", "GANET_SGA_STREAMS",
                                          "GANET_LGA_WG", "GANET_SGA_POINT_Q4"};
    for (const char *r : retired)
      if (!strcmp(name, r)) {
        static std::atomic<int> noted{0};
        if (!noted.exchange(1)) fprintf(stderr, "[ganet] ganet_set_option: %s is retired (no effect); see INTEGRATION.md\n", name);
        return GANET_OK;
      }
    return fail(GANET_E_INVALID, "ganet_set_option: unknown option %s", name);
  }
  return GANET_OK;
}

GA_EXPORT int ganet_sga_scan_forward(const float *x, const float *g, float *A, int N, int C, int D,
                                     int H, int W, int dir, void *stream)
{
  if (!x || !g || !A) return fail(GANET_E_INVALID, "ganet_sga_scan_forward: null pointer");
  if (dir < 0 || dir > 3) return fail(GANET_E_INVALID, "ganet_sga_scan_forward: dir %d", dir);
  GA_TRY(check_dims5("ganet_sga_scan_forward", N, C, D, H, W));
  return scan_fwd(x, g, A, N, C, D, H, W, dir, (hipStream_t)stream);
}

GA_EXPORT int ganet_sga_scan_forward_ws(const float *x, const float *g, float *A_ws, int N, int C, int D, int H, int W, int dir,
                                         void *stream)
{
  if (!x || !g || !A_ws) return fail(GANET_E_INVALID, "ganet_sga_scan_forward_ws: null pointer");
  if (dir < 0 || dir > 3) return fail(GANET_E_INVALID, "ganet_sga_scan_forward_ws: dir %d", dir);
  GA_TRY(check_dims5("ganet_sga_scan_forward_ws", N, C, D, H, W));
  const i64 n = (i64)N * C * D * H * W;
  return scan_fwd(x, g, A_ws + dir * n, N, C, D, H, W, dir, (hipStream_t)stream);
}

GA_EXPORT int ganet_sga_backward_scan_ws(const float *g, const uint8_t *mask, const uint16_t *kp, const float *grad_out, float *G_ws,
                                          int N, int C, int D, int H, int W, int dir, void *stream)
{
  if (!g || !mask || !kp || !grad_out || !G_ws) return fail(GANET_E_INVALID, "ganet_sga_backward_scan_ws: null pointer");
  if (dir < 0 || dir > 3) return fail(GANET_E_INVALID, "ganet_sga_backward_scan_ws: dir %d", dir);
  GA_TRY(check_dims5("ganet_sga_backward_scan_ws", N, C, D, H, W));
  const i64 n = (i64)N * C * D * H * W, npix = (i64)N * C * H * W;
  return scan_bwdg(g, mask, kp + dir * npix, grad_out, G_ws + dir * n, N, C, D, H, W, dir, (hipStream_t)stream,
                   dir < 2 && sga_ws_tiled(N, C, D, H, W));
}

GA_EXPORT int ganet_sga_workspace_layout(int N, int C, int D, int H, int W)
{
  if (check_dims5("ganet_sga_workspace_layout", N, C, D, H, W) != GANET_OK) return GANET_E_INVALID;
  return sga_ws_tiled(N, C, D, H, W);
}

GA_EXPORT int ganet_sga_merge(const float *A_ws, float *out, uint8_t *mask, uint16_t *kp, int N, int C, int D, int H, int W,
                               void *stream)
{
  if (!A_ws || !out || !mask || !kp) return fail(GANET_E_INVALID, "ganet_sga_merge: null pointer");
  GA_TRY(check_dims5("ganet_sga_merge", N, C, D, H, W));
  if (D > 65535) return fail(GANET_E_UNSUPPORTED, "ganet_sga_merge: D > 65535");
  const i64 n = (i64)N * C * D * H * W;
  const i64 npix = (i64)N * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  const i64 HWl = (i64)H * W;
  const bool al = aligned16(A_ws) && aligned16(out) && (((uintptr_t)mask & 3) == 0) && (((uintptr_t)kp & 7) == 0);
  if (HWl % 4 == 0 && al && npix / 4 / 64 + 1 < (1ll << 31)) {
    GA_LAUNCH(sga_merge_px4, dim3((unsigned)((npix / 4 + 63) / 64)), dim3(64), st, A_ws, A_ws + n, A_ws + 2 * n,
              A_ws + 3 * n, out, mask, kp, D, HWl, npix);
    return check_launch("sga merge (4 px / lane)");
  }
  GA_LAUNCH((sga_merge_px<uint8_t>), dim3(px_grid(npix)), dim3(256), st, A_ws, A_ws + n, A_ws + 2 * n,
            A_ws + 3 * n, out, mask, kp, D, HWl, npix);
  return check_launch("sga merge");
}

GA_EXPORT int ganet_sga_forward(const float *x, const float *g0, const float *g1, const float *g2,
                                const float *g3, float *A_ws, float *out, uint8_t *mask,
                                uint16_t *kp, int N, int C, int D, int H, int W, void *stream)
{
  if (!x || !g0 || !g1 || !g2 || !g3 || !A_ws || !out || !mask || !kp)
    return fail(GANET_E_INVALID, "ganet_sga_forward: null pointer");
  GA_TRY(check_dims5("ganet_sga_forward", N, C, D, H, W));
  if (D > 65535) return fail(GANET_E_UNSUPPORTED, "ganet_sga_forward: D > 65535");
  const i64 n = (i64)N * C * D * H * W;
  hipStream_t st = (hipStream_t)stream;
  const float *gs[4] = {g0, g1, g2, g3};
  // (the four scans on four streams were measured twice and dropped: 0.596 vs 0.606 ms)
  for (int d = 0; d < 4; d++) GA_TRY(scan_fwd(x, gs[d], A_ws + d * n, N, C, D, H, W, d, st));
  return ganet_sga_merge(A_ws, out, mask, kp, N, C, D, H, W, stream);
}

namespace {
bool infer_fused_ok(const float *x, const float *g0, const float *g1, const float *g2, const float *g3, const float *out,
                    int N, int C, int D, int W)
{
  const bool all_al = aligned16(x) && aligned16(out) && aligned16(g0) && aligned16(g1) && aligned16(g2) && aligned16(g3);
  return all_al && N * C <= 65535 && rowwave_ok(D, W, 2, row_smem_fwd(D)) &&
         colblock_ok(D, W, 0, col_smem_fwd(D));
}
}  // namespace

GA_EXPORT int ganet_sga_forward_infer_scratch(const float *x, const float *g0, const float *g1, const float *g2,
                                              const float *g3, const float *out, int N, int C, int D, int H, int W)
{
  if (!x || !g0 || !g1 || !g2 || !g3 || !out)
    return fail(GANET_E_INVALID, "ganet_sga_forward_infer_scratch: null pointer");
  GA_TRY(check_dims5("ganet_sga_forward_infer_scratch", N, C, D, H, W));
  return infer_fused_ok(x, g0, g1, g2, g3, out, N, C, D, W) ? 0 : 4;
}

GA_EXPORT int ganet_sga_forward_infer(const float *x, const float *g0, const float *g1, const float *g2,
                                      const float *g3, float *A_ws, float *out, const float *bn_scale,
                                      const float *bn_shift, int N, int C, int D, int H, int W,
                                      void *stream)
{
  if (!x || !g0 || !g1 || !g2 || !g3 || !out)
    return fail(GANET_E_INVALID, "ganet_sga_forward_infer: null pointer");
  if ((bn_scale == nullptr) != (bn_shift == nullptr))
    return fail(GANET_E_INVALID, "ganet_sga_forward_infer: bn_scale and bn_shift go together");
  GA_TRY(check_dims5("ganet_sga_forward_infer", N, C, D, H, W));
  const i64 n = (i64)N * C * D * H * W;
  const i64 slice = (i64)D * H * W;
  hipStream_t st = (hipStream_t)stream;
  const float *gs[4] = {g0, g1, g2, g3};
  // Fast form: no directional volume is kept.  `down` writes the output volume, `up` / `right` / `left` take the running
  // maximum in the copy-out of their tiles, `left` also applies the BatchNorm affine + ReLU: 11 V of traffic and four
  // launches instead of 13 V and five (A_ws is not touched).
  if (infer_fused_ok(x, g0, g1, g2, g3, out, N, C, D, W)) {
    GA_TRY(col_fwd(x, g0, out, N * C, D, H, W, 0, st, 0));
    GA_TRY(col_fwd(x, g1, out, N * C, D, H, W, 1, st, 1));
    GA_TRY(row_fwd(x, g2, out, N * C, D, H, W, 2, st, 1));
    return row_fwd(x, g3, out, N * C, D, H, W, 3, st, bn_scale ? 2 : 1, C, bn_scale, bn_shift);
  }
  if (!A_ws) return fail(GANET_E_INVALID, "ganet_sga_forward_infer: this shape needs the A_ws scratch (see ganet_sga_forward_infer_scratch)");
  for (int d = 0; d < 4; d++) GA_TRY(scan_fwd(x, gs[d], A_ws + d * n, N, C, D, H, W, d, st));
  if (slice % 4 == 0 && aligned16(A_ws) && aligned16(out))
    GA_LAUNCH((sga_merge_infer<true>), dim3(ew_grid(n / 4)), dim3(256), st, A_ws, A_ws + n, A_ws + 2 * n, A_ws + 3 * n,
              out, bn_scale, bn_shift, C, slice, n);
  else
    GA_LAUNCH((sga_merge_infer<false>), dim3(ew_grid(n)), dim3(256), st, A_ws, A_ws + n, A_ws + 2 * n, A_ws + 3 * n,
              out, bn_scale, bn_shift, C, slice, n);
  return check_launch("sga merge (inference)");
}

GA_EXPORT int ganet_sga_backward_scan(const float *g, const uint8_t *mask, const uint16_t *kp_dir,
                                      const float *grad_out, float *G, int N, int C, int D, int H,
                                      int W, int dir, void *stream)
{
  if (!g || !mask || !kp_dir || !grad_out || !G)
    return fail(GANET_E_INVALID, "ganet_sga_backward_scan: null pointer");
  if (dir < 0 || dir > 3) return fail(GANET_E_INVALID, "ganet_sga_backward_scan: dir %d", dir);
  GA_TRY(check_dims5("ganet_sga_backward_scan", N, C, D, H, W));
  return scan_bwdg(g, mask, kp_dir, grad_out, G, N, C, D, H, W, dir, (hipStream_t)stream);
}

GA_EXPORT int ganet_sga_backward_dir(const float *x, const float *g, const float *A,
                                     const uint8_t *mask, const uint16_t *kp_dir,
                                     const float *grad_out, float *G_ws, float *grad_x, float *gw,
                                     int N, int C, int D, int H, int W, int dir, int accumulate,
                                     void *stream)
{
  if (!x || !g || !A || !mask || !kp_dir || !grad_out || !G_ws || !grad_x || !gw)
    return fail(GANET_E_INVALID, "ganet_sga_backward_dir: null pointer");
  if (dir < 0 || dir > 3) return fail(GANET_E_INVALID, "ganet_sga_backward_dir: dir %d", dir);
  GA_TRY(check_dims5("ganet_sga_backward_dir", N, C, D, H, W));
  hipStream_t st = (hipStream_t)stream;
  GA_TRY(scan_bwdg(g, mask, kp_dir, grad_out, G_ws, N, C, D, H, W, dir, st));
  PointArgs pa = {};
  pa.G[0] = G_ws; pa.A[0] = A; pa.g[0] = g; pa.gw[0] = gw; pa.dir[0] = dir;
  return bwd_point(x, grad_x, pa, 1, N, C, D, H, W, accumulate ? 1 : 0, st);
}

namespace {
// the per-pixel kernel over all four directions, with the layout of G_down / G_up DECIDED BY THE CALLER (one read of the options
// per backward: ADVICE r4)
int backward_point_impl(const float *x, const float *g0, const float *g1, const float *g2, const float *g3,
                        const float *A_ws, const float *G_ws, float *grad_x, float *gw0, float *gw1, float *gw2,
                        float *gw3, int N, int C, int D, int H, int W, hipStream_t st, int tiled)
{
  const i64 n = (i64)N * C * D * H * W;
  const float *gs[4] = {g0, g1, g2, g3};
  float *gws[4] = {gw0, gw1, gw2, gw3};
  PointArgs pa = {};
  for (int d = 0; d < 4; d++) {
    pa.G[d] = G_ws + d * n; pa.A[d] = A_ws + d * n; pa.g[d] = gs[d]; pa.gw[d] = gws[d]; pa.dir[d] = d;
  }
  return bwd_point(x, grad_x, pa, 4, N, C, D, H, W, 0, st, tiled);
}
}  // namespace

GA_EXPORT int ganet_sga_backward_point(const float *x, const float *g0, const float *g1, const float *g2, const float *g3,
                                        const float *A_ws, const float *G_ws, float *grad_x, float *gw0, float *gw1, float *gw2,
                                        float *gw3, int N, int C, int D, int H, int W, void *stream)
{
  if (!x || !g0 || !g1 || !g2 || !g3 || !A_ws || !G_ws || !grad_x || !gw0 || !gw1 || !gw2 || !gw3)
    return fail(GANET_E_INVALID, "ganet_sga_backward_point: null pointer");
  GA_TRY(check_dims5("ganet_sga_backward_point", N, C, D, H, W));
  // pairs with ganet_sga_backward_scan_ws, which keeps G_down / G_up in the layout ganet_sga_workspace_layout reports
  return backward_point_impl(x, g0, g1, g2, g3, A_ws, G_ws, grad_x, gw0, gw1, gw2, gw3, N, C, D, H, W, (hipStream_t)stream,
                             sga_ws_tiled(N, C, D, H, W));
}

GA_EXPORT int ganet_sga_backward(const float *x, const float *g0, const float *g1, const float *g2,
                                 const float *g3, const float *A_ws, const uint8_t *mask,
                                 const uint16_t *kp, const float *grad_out, float *G_ws,
                                 float *grad_x, float *gw0, float *gw1, float *gw2, float *gw3,
                                 int N, int C, int D, int H, int W, void *stream)
{
  if (!x || !g0 || !g1 || !g2 || !g3 || !A_ws || !mask || !kp || !grad_out || !G_ws || !grad_x ||
      !gw0 || !gw1 || !gw2 || !gw3)
    return fail(GANET_E_INVALID, "ganet_sga_backward: null pointer");
  GA_TRY(check_dims5("ganet_sga_backward", N, C, D, H, W));
  const i64 n = (i64)N * C * D * H * W;
  const i64 npix = (i64)N * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  const float *gs[4] = {g0, g1, g2, g3};
  // The layout of G_down / G_up is private to this call: decided ONCE, from the dimensions, the options and the alignment of what
  // the tiled column kernels touch (a contiguous but 4-byte aligned gradient takes the API layout and the generic scans).
  const int tiled = sga_ws_tiled(N, C, D, H, W) && aligned16(g0) && aligned16(g1) && aligned16(grad_out) && aligned16(G_ws) &&
                    (n % 4 == 0) && (((uintptr_t)mask & 3) == 0);
  for (int d = 0; d < 4; d++)
    GA_TRY(scan_bwdg(gs[d], mask, kp + d * npix, grad_out, G_ws + d * n, N, C, D, H, W, d, st, d < 2 && tiled));
  return backward_point_impl(x, g0, g1, g2, g3, A_ws, G_ws, grad_x, gw0, gw1, gw2, gw3, N, C, D, H, W, st, tiled);
}

GA_EXPORT int ganet_sga_forward_compat(const float *x, const float *g0, const float *g1,
                                       const float *g2, const float *g3, float *temp_out,
                                       float *out, float *mask_f32, int N, int C, int D, int H,
                                       int W, void *stream)
{
  if (!x || !g0 || !g1 || !g2 || !g3 || !temp_out || !out || !mask_f32)
    return fail(GANET_E_INVALID, "ganet_sga_forward_compat: null pointer");
  GA_TRY(check_dims5("ganet_sga_forward_compat", N, C, D, H, W));
  const i64 n = (i64)N * C * D * H * W;
  hipStream_t st = (hipStream_t)stream;
  const float *gs[4] = {g0, g1, g2, g3};
  GA_TRY(scan_fwd(x, gs[0], out, N, C, D, H, W, 0, st));
  for (int d = 1; d < 4; d++) {
    GA_TRY(scan_fwd(x, gs[d], temp_out, N, C, D, H, W, d, st));
    GA_LAUNCH((sga_merge_running<float>), dim3(ew_grid(n)), dim3(256), st, temp_out, out, mask_f32,
              n, d, d == 1 ? 1 : 0);
    GA_TRY(check_launch("sga merge (compat)"));
  }
  return GANET_OK;
}

GA_EXPORT int ganet_sga_backward_compat(const float *x, const float *g0, const float *g1,
                                        const float *g2, const float *g3, float *temp_out,
                                        const float *mask_f32, float *max_idx,
                                        const float *grad_out, float *temp_grad, float *grad_x,
                                        float *gw0, float *gw1, float *gw2, float *gw3, int N,
                                        int C, int D, int H, int W, void *stream)
{
  if (!x || !g0 || !g1 || !g2 || !g3 || !temp_out || !mask_f32 || !max_idx || !grad_out ||
      !temp_grad || !grad_x || !gw0 || !gw1 || !gw2 || !gw3)
    return fail(GANET_E_INVALID, "ganet_sga_backward_compat: null pointer");
  GA_TRY(check_dims5("ganet_sga_backward_compat", N, C, D, H, W));
  if (D > 65535) return fail(GANET_E_UNSUPPORTED, "ganet_sga_backward_compat: D > 65535");
  const i64 npix = (i64)N * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  // scratch roles as in the reference: temp_out = A_dir, temp_grad = adjoint volume,
  // max_idx = first-argmax per pixel (stored as uint16 in the first half of the buffer)
  uint16_t *kp = reinterpret_cast<uint16_t *>(max_idx);
  const float *gs[4] = {g0, g1, g2, g3};
  float *gws[4] = {gw0, gw1, gw2, gw3};
  const int order[4] = {3, 0, 1, 2};   // GANet_kernel.cu:1040-1128
  for (int i = 0; i < 4; i++) {
    const int d = order[i];
    if (d != 3) GA_TRY(scan_fwd(x, gs[d], temp_out, N, C, D, H, W, d, st));
    GA_LAUNCH(sga_argmax_px, dim3(px_grid(npix)), dim3(256), st, temp_out, kp, D, (i64)H * W, npix);
    GA_TRY(check_launch("sga argmax"));
    GA_TRY(scan_bwdg_f32mask(gs[d], mask_f32, kp, grad_out, temp_grad, N, C, D, H, W, d, st));
    PointArgs pa = {};
    pa.G[0] = temp_grad; pa.A[0] = temp_out; pa.g[0] = gs[d]; pa.gw[0] = gws[d]; pa.dir[0] = d;
    // gradInput is accumulated into (caller zero-fills); gw is written (each pixel once per direction)
    GA_TRY(bwd_point(x, grad_x, pa, 1, N, C, D, H, W, 1, st));
  }
  return GANET_OK;
}

GA_EXPORT int ganet_lga_forward(const float *x, const float *f, float *y, int B, int D, int H,
                                int W, int radius, void *stream)
{
  if (!x || !f || !y) return fail(GANET_E_INVALID, "ganet_lga_forward: null pointer");
  if (x == y) return fail(GANET_E_INVALID, "ganet_lga_forward: y must not alias x");
  GA_TRY(check_lga("ganet_lga_forward", B, D, H, W, radius));
  hipStream_t st = (hipStream_t)stream;
  if (radius == 1) return launch_lga_fwd<1>(x, f, y, B, D, H, W, false, st);
  if (radius == 2) return launch_lga_fwd<2>(x, f, y, B, D, H, W, false, st);
  return launch_lga_fwd<3>(x, f, y, B, D, H, W, false, st);
}

namespace {
int lga_apply_paired_impl(const char *who, const float *x, const float *f, float *y, float *edge, int B, int D, int H, int W, int radius,
                          int transposed, int x_paired, int y_paired, void *stream)
{
  if (!x || !f || !y) return fail(GANET_E_INVALID, "%s: null pointer", who);
  if (x == y) return fail(GANET_E_INVALID, "%s: y must not alias x", who);
  GA_TRY(check_lga(who, B, D, H, W, radius));
  if (x_paired && y_paired) return fail(GANET_E_INVALID, "%s: at most one of x_paired / y_paired", who);
  if (radius != 2) return fail(GANET_E_UNSUPPORTED, "%s: radius 2 only", who);
  if (!x_paired && !y_paired) {
    if (edge) return fail(GANET_E_UNSUPPORTED, "%s: the edge-sum side channel belongs to the pair-interleaved forms", who);
    return launch_lga_fwd<2>(x, f, y, B, D, H, W, transposed != 0, (hipStream_t)stream);
  }
  return launch_lga_paired(x, f, y, B, D, H, W, transposed != 0, x_paired != 0, (hipStream_t)stream, edge);
}
}  // namespace

GA_EXPORT int ganet_lga_apply_paired(const float *x, const float *f, float *y, int B, int D, int H, int W, int radius,
                                     int transposed, int x_paired, int y_paired, void *stream)
{
  return lga_apply_paired_impl("ganet_lga_apply_paired", x, f, y, nullptr, B, D, H, W, radius, transposed, x_paired, y_paired, stream);
}

GA_EXPORT int ganet_lga_apply_paired_edges(const float *x, const float *f, float *y, float *edge, int B, int D, int H, int W, int radius,
                                           int transposed, int x_paired, int y_paired, void *stream)
{
  if (!edge) return fail(GANET_E_INVALID, "ganet_lga_apply_paired_edges: null edge buffer");
  return lga_apply_paired_impl("ganet_lga_apply_paired_edges", x, f, y, edge, B, D, H, W, radius, transposed, x_paired, y_paired, stream);
}

GA_EXPORT int ganet_lga_filter_grad_paired(const float *x, const float *gy, float *gf, int B, int D, int H, int W, int radius,
                                           int accumulate_gf, int x_paired, int gy_paired, void *stream)
{
  if (!x || !gy || !gf) return fail(GANET_E_INVALID, "ganet_lga_filter_grad_paired: null pointer");
  GA_TRY(check_lga("ganet_lga_filter_grad_paired", B, D, H, W, radius));
  if (x_paired && gy_paired) return fail(GANET_E_INVALID, "ganet_lga_filter_grad_paired: at most one of x_paired / gy_paired");
  if (radius != 2) return fail(GANET_E_UNSUPPORTED, "ganet_lga_filter_grad_paired: radius 2 only");
  if (!x_paired && !gy_paired) return launch_lga_gf<2>(x, gy, gf, B, D, H, W, accumulate_gf != 0, (hipStream_t)stream);
  return launch_lga_gf_paired(x, gy, gf, B, D, H, W, accumulate_gf != 0, x_paired != 0, (hipStream_t)stream);
}

GA_EXPORT int ganet_lga_forward_regress(const float *x, const float *f, float *y, float *snorm, float *sdy, int B, int D,
                                        int H, int W, int radius, void *stream)
{
  if (!x || !f || !snorm || !sdy) return fail(GANET_E_INVALID, "ganet_lga_forward_regress: null pointer");
  if (x == y) return fail(GANET_E_INVALID, "ganet_lga_forward_regress: y must not alias x");
  GA_TRY(check_lga("ganet_lga_forward_regress", B, D, H, W, radius));
  hipStream_t st = (hipStream_t)stream;
  if (radius == 1) return launch_lga_fwd_regress<1>(x, f, y, snorm, sdy, B, D, H, W, st);
  if (radius == 2) return launch_lga_fwd_regress<2>(x, f, y, snorm, sdy, B, D, H, W, st);
  return launch_lga_fwd_regress<3>(x, f, y, snorm, sdy, B, D, H, W, st);
}

GA_EXPORT int ganet_lga_backward(const float *x, const float *f, const float *gy, float *gx,
                                 float *gf, int B, int D, int H, int W, int radius,
                                 int accumulate_gf, void *stream)
{
  if (!x || !f || !gy || !gx || !gf) return fail(GANET_E_INVALID, "ganet_lga_backward: null pointer");
  if (gx == gy) return fail(GANET_E_INVALID, "ganet_lga_backward: gx must not alias gy");
  GA_TRY(check_lga("ganet_lga_backward", B, D, H, W, radius));
  hipStream_t st = (hipStream_t)stream;
  const int acc = accumulate_gf ? 1 : 0;
  // filter gradient first: it is the only consumer of x, so gx may alias x afterwards
  // (the reference's chained LGA2/LGA3 backward relies on that, functions/GANet.py:197).
  // (the two kernels of a pass on two streams: measured slower, 0.215 -> 0.238 ms, profiles/r2m_*)
  int rc;
  if (radius == 1) { rc = launch_lga_gf<1>(x, gy, gf, B, D, H, W, acc, st); if (rc == GANET_OK) rc = launch_lga_fwd<1>(gy, f, gx, B, D, H, W, true, st); }
  else if (radius == 2) { rc = launch_lga_gf<2>(x, gy, gf, B, D, H, W, acc, st); if (rc == GANET_OK) rc = launch_lga_fwd<2>(gy, f, gx, B, D, H, W, true, st); }
  else { rc = launch_lga_gf<3>(x, gy, gf, B, D, H, W, acc, st); if (rc == GANET_OK) rc = launch_lga_fwd<3>(gy, f, gx, B, D, H, W, true, st); }
  return rc;
}

GA_EXPORT int ganet_cost_volume_forward(const float *x, const float *y, float *cost, int N, int C,
                                        int Dn, int H, int W, void *stream)
{
  if (!x || !y || !cost) return fail(GANET_E_INVALID, "ganet_cost_volume_forward: null pointer");
  if (N <= 0 || C <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_cost_volume_forward: non-positive size");
  const i64 n = (i64)N * 2 * C * Dn * H * W;
  if (W % 4 == 0 && aligned16(x) && aligned16(cost))
    GA_LAUNCH(cost_volume_fwd4, dim3(ew_grid(n / 4)), dim3(256), (hipStream_t)stream, x, y, cost, N, C, Dn, H, W);
  else
    GA_LAUNCH(cost_volume_fwd, dim3(ew_grid(n)), dim3(256), (hipStream_t)stream, x, y, cost, N, C, Dn, H, W);
  return check_launch("cost volume forward");
}

GA_EXPORT int ganet_cost_volume_backward(const float *grad_cost, float *grad_x, float *grad_y,
                                         int N, int C, int Dn, int H, int W, void *stream)
{
  if (!grad_cost || !grad_x || !grad_y)
    return fail(GANET_E_INVALID, "ganet_cost_volume_backward: null pointer");
  if (N <= 0 || C <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_cost_volume_backward: non-positive size");
  const i64 n = (i64)N * C * H * W;
  GA_LAUNCH(cost_volume_bwd, dim3(ew_grid(n)), dim3(256), (hipStream_t)stream, grad_cost, grad_x, grad_y, N, C, Dn, H, W);
  return check_launch("cost volume backward");
}

GA_EXPORT int ganet_disparity_regression_forward(const float *x, float *out, int N, int Dn, int H,
                                                 int W, void *stream)
{
  if (!x || !out) return fail(GANET_E_INVALID, "ganet_disparity_regression_forward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_disparity_regression_forward: non-positive size");
  const i64 HW = (i64)H * W;
  GA_LAUNCH(disp_regression_fwd, dim3(ew_grid((i64)N * HW)), dim3(256), (hipStream_t)stream, x, out, N, Dn, HW);
  return check_launch("disparity regression forward");
}

GA_EXPORT int ganet_disparity_regression_backward(const float *grad_out, float *grad_x, int N,
                                                  int Dn, int H, int W, void *stream)
{
  if (!grad_out || !grad_x)
    return fail(GANET_E_INVALID, "ganet_disparity_regression_backward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_disparity_regression_backward: non-positive size");
  const i64 HW = (i64)H * W;
  if (HW % 4 == 0 && aligned16(grad_out) && aligned16(grad_x))
    GA_LAUNCH(disp_regression_bwd4, dim3(ew_grid((i64)N * (HW / 4))), dim3(256), (hipStream_t)stream, grad_out, grad_x, N, Dn, HW);
  else
    GA_LAUNCH(disp_regression_bwd, dim3(ew_grid((i64)N * Dn * HW)), dim3(256), (hipStream_t)stream, grad_out, grad_x, N, Dn, HW);
  return check_launch("disparity regression backward");
}

// ---- callers' normalisations folded into single kernels ------------------------
namespace {
int check_norm(const char *who, int N, int G, int C, int K, int H, int W)
{
  if (N <= 0 || G <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "%s: non-positive size N=%d G=%d C=%d K=%d H=%d W=%d", who, N, G, C, K, H, W);
  if (G > 4) return fail(GANET_E_UNSUPPORTED, "%s: at most 4 groups, got %d", who, G);
  return GANET_OK;
}
}  // namespace

GA_EXPORT int ganet_l1_normalize_forward(const float *x, float *y0, float *y1, float *y2, float *y3,
                                         int N, int G, int C, int K, int H, int W, void *stream)
{
  GA_TRY(check_norm("ganet_l1_normalize_forward", N, G, C, K, H, W));
  float *ys[4] = {y0, y1, y2, y3};
  if (!x) return fail(GANET_E_INVALID, "ganet_l1_normalize_forward: null pointer");
  NormPtrs p = {};
  for (int g = 0; g < 4; g++) {
    if (g < G && !ys[g]) return fail(GANET_E_INVALID, "ganet_l1_normalize_forward: null output %d", g);
    p.y[g] = ys[g < G ? g : 0];
  }
  const i64 HW = (i64)H * W;
  const dim3 grid(ew_grid((i64)N * G * C * HW)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (K == 5) GA_LAUNCH((l1norm_fwd<5>), grid, block, st, x, p, N, G, C, K, HW);
  else if (K == 75) GA_LAUNCH((l1norm_fwd<75>), grid, block, st, x, p, N, G, C, K, HW);
  else GA_LAUNCH((l1norm_fwd<0>), grid, block, st, x, p, N, G, C, K, HW);
  return check_launch("l1 normalise forward");
}

GA_EXPORT int ganet_l1_normalize_backward(const float *x, const float *gy0, const float *gy1,
                                          const float *gy2, const float *gy3, float *grad_x,
                                          int N, int G, int C, int K, int H, int W, void *stream)
{
  GA_TRY(check_norm("ganet_l1_normalize_backward", N, G, C, K, H, W));
  const float *gs[4] = {gy0, gy1, gy2, gy3};
  if (!x || !grad_x) return fail(GANET_E_INVALID, "ganet_l1_normalize_backward: null pointer");
  NormPtrs p = {};
  for (int g = 0; g < 4; g++) {
    if (g < G && !gs[g]) return fail(GANET_E_INVALID, "ganet_l1_normalize_backward: null gradient %d", g);
    p.gy[g] = gs[g < G ? g : 0];
  }
  const i64 HW = (i64)H * W;
  const dim3 grid(ew_grid((i64)N * G * C * HW)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (K == 5) GA_LAUNCH((l1norm_bwd<5>), grid, block, st, x, p, grad_x, N, G, C, K, HW);
  else if (K == 75) GA_LAUNCH((l1norm_bwd<75>), grid, block, st, x, p, grad_x, N, G, C, K, HW);
  else GA_LAUNCH((l1norm_bwd<0>), grid, block, st, x, p, grad_x, N, G, C, K, HW);
  return check_launch("l1 normalise backward");
}

GA_EXPORT int ganet_norm_disparity_regression_forward(const float *x, float *out, float *snorm, int N,
                                                      int Dn, int H, int W, void *stream)
{
  if (!x || !out || !snorm)
    return fail(GANET_E_INVALID, "ganet_norm_disparity_regression_forward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_norm_disparity_regression_forward: non-positive size");
  const i64 HW = (i64)H * W;
  GA_LAUNCH(norm_disp_regression_fwd, dim3(ew_grid((i64)N * HW)), dim3(256), (hipStream_t)stream, x, out, snorm, N, Dn, HW);
  return check_launch("normalised disparity regression forward");
}

GA_EXPORT int ganet_norm_disparity_regression_backward(const float *x, const float *out,
                                                       const float *snorm, const float *grad_out,
                                                       float *grad_x, int N, int Dn, int H, int W,
                                                       void *stream)
{
  if (!x || !out || !snorm || !grad_out || !grad_x)
    return fail(GANET_E_INVALID, "ganet_norm_disparity_regression_backward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_norm_disparity_regression_backward: non-positive size");
  const i64 HW = (i64)H * W;
  if (HW % 4 == 0 && aligned16(x) && aligned16(out) && aligned16(snorm) && aligned16(grad_out) && aligned16(grad_x))
    GA_LAUNCH(norm_disp_regression_bwd4, dim3(ew_grid((i64)N * (HW / 4))), dim3(256), (hipStream_t)stream, x, out, snorm,
              grad_out, grad_x, N, Dn, HW);
  else
    GA_LAUNCH(norm_disp_regression_bwd, dim3(ew_grid((i64)N * Dn * HW)), dim3(256), (hipStream_t)stream, x, out, snorm,
              grad_out, grad_x, N, Dn, HW);
  return check_launch("normalised disparity regression backward");
}

namespace {
int check_up(const char *who, const void *a, const void *b, int S, int Di, int Hi, int Wi, int Do, int Ho, int Wo)
{
  if (!a || !b) return fail(GANET_E_INVALID, "%s: null pointer", who);
  if (S <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0)
    return fail(GANET_E_INVALID, "%s: non-positive size", who);
  return GANET_OK;
}
UpAxis up_axis(int in, int out)
{
  UpAxis a;
  a.in = in; a.out = out;
  a.scale = (float)in / (float)out;      // ATen: area_pixel_compute_scale<float>(in, out, align_corners = false, nullopt)
  return a;
}
}  // namespace

GA_EXPORT int ganet_trilinear_upsample_forward(const float *x, float *y, int S, int Di, int Hi, int Wi, int Do, int Ho,
                                               int Wo, void *stream)
{
  GA_TRY(check_up("ganet_trilinear_upsample_forward", x, y, S, Di, Hi, Wi, Do, Ho, Wo));
  const i64 total = (i64)S * Do * Ho * Wo;
  GA_LAUNCH(trilinear_up_fwd, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, x, y, (i64)S, up_axis(Di, Do),
            up_axis(Hi, Ho), up_axis(Wi, Wo));
  return check_launch("trilinear upsample forward");
}

GA_EXPORT int ganet_trilinear_upsample_backward(const float *grad_y, float *grad_x, int S, int Di, int Hi, int Wi, int Do,
                                                int Ho, int Wo, void *stream)
{
  GA_TRY(check_up("ganet_trilinear_upsample_backward", grad_y, grad_x, S, Di, Hi, Wi, Do, Ho, Wo));
  const i64 total = (i64)S * Di * Hi * Wi;
  GA_LAUNCH(trilinear_up_bwd, dim3(ew_grid(total)), dim3(256), (hipStream_t)stream, grad_y, grad_x, (i64)S, up_axis(Di, Do),
            up_axis(Hi, Ho), up_axis(Wi, Wo));
  return check_launch("trilinear upsample backward");
}

GA_EXPORT int ganet_softmin_forward(const float *x, float *y, int N, int Dn, int H, int W, void *stream)
{
  if (!x || !y) return fail(GANET_E_INVALID, "ganet_softmin_forward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0) return fail(GANET_E_INVALID, "ganet_softmin_forward: non-positive size");
  const i64 HW = (i64)H * W;
  if (HW % 4 == 0 && aligned16(x) && aligned16(y))
    GA_LAUNCH(softmin_fwd4, dim3(ew_grid((i64)N * HW / 4)), dim3(256), (hipStream_t)stream, x, y, N, Dn, HW);
  else
    GA_LAUNCH(softmin_fwd, dim3(ew_grid((i64)N * HW)), dim3(256), (hipStream_t)stream, x, y, N, Dn, HW);
  return check_launch("softmin forward");
}

GA_EXPORT int ganet_softmin_backward(const float *y, const float *grad_y, float *grad_x, int N, int Dn, int H,
                                     int W, void *stream)
{
  if (!y || !grad_y || !grad_x) return fail(GANET_E_INVALID, "ganet_softmin_backward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0) return fail(GANET_E_INVALID, "ganet_softmin_backward: non-positive size");
  const i64 HW = (i64)H * W;
  if (HW % 4 == 0 && aligned16(y) && aligned16(grad_y) && aligned16(grad_x))
    GA_LAUNCH(softmin_bwd4, dim3(ew_grid((i64)N * HW / 4)), dim3(256), (hipStream_t)stream, y, grad_y, grad_x, N, Dn, HW);
  else
    GA_LAUNCH(softmin_bwd, dim3(ew_grid((i64)N * HW)), dim3(256), (hipStream_t)stream, y, grad_y, grad_x, N, Dn, HW);
  return check_launch("softmin backward");
}

GA_EXPORT int ganet_softmin_regression_forward(const float *x, float *out, float *mx, float *ssum, int N, int Dn,
                                               int H, int W, void *stream)
{
  if (!x || !out || !mx || !ssum) return fail(GANET_E_INVALID, "ganet_softmin_regression_forward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_softmin_regression_forward: non-positive size");
  const i64 HW = (i64)H * W;
  GA_LAUNCH(softmin_regression_fwd, dim3(ew_grid((i64)N * HW)), dim3(256), (hipStream_t)stream, x, out, mx, ssum, N, Dn, HW);
  return check_launch("softmin regression forward");
}

GA_EXPORT int ganet_softmin_regression_backward(const float *x, const float *out, const float *mx,
                                                const float *ssum, const float *grad_out, float *grad_x,
                                                int N, int Dn, int H, int W, void *stream)
{
  if (!x || !out || !mx || !ssum || !grad_out || !grad_x)
    return fail(GANET_E_INVALID, "ganet_softmin_regression_backward: null pointer");
  if (N <= 0 || Dn <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "ganet_softmin_regression_backward: non-positive size");
  const i64 HW = (i64)H * W;
  GA_LAUNCH(softmin_regression_bwd, dim3(ew_grid((i64)N * HW)), dim3(256), (hipStream_t)stream, x, out, mx, ssum,
            grad_out, grad_x, N, Dn, HW);
  return check_launch("softmin regression backward");
}

// ---- SGABlock's residual epilogue (models/GANet_deep.py:270-277) --------------------
namespace {
int check_residual(const char *who, int N, int C, int D, int H, int W)
{
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0)
    return fail(GANET_E_INVALID, "%s: non-positive size N=%d C=%d D=%d H=%d W=%d", who, N, C, D, H, W);
  return GANET_OK;
}
// x: enough 256-thread blocks per slice to fill the chip about four times over, y: the slices
dim3 residual_grid(i64 S, i64 per_slice)
{
  i64 gx = (per_slice + 255) / 256;
  const i64 want = (256 * 16 + S - 1) / S;
  if (gx > want) gx = want;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)(S < 65535 ? S : 65535));
}
}  // namespace

GA_EXPORT int ganet_residual_relu_forward(const float *t, const float *rem, const float *bn_scale, const float *bn_shift,
                                          float *y, int N, int C, int D, int H, int W, void *stream)
{
  GA_TRY(check_residual("ganet_residual_relu_forward", N, C, D, H, W));
  if (!t || !rem || !y) return fail(GANET_E_INVALID, "ganet_residual_relu_forward: null pointer");
  if ((bn_scale == nullptr) != (bn_shift == nullptr))
    return fail(GANET_E_INVALID, "ganet_residual_relu_forward: bn_scale and bn_shift come together");
  const i64 S = (i64)N * C, slice = (i64)D * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (slice % 4 == 0 && aligned16(t) && aligned16(rem) && aligned16(y))
    GA_LAUNCH((residual_relu_fwd<true>), residual_grid(S, slice / 4), dim3(256), st, t, rem, bn_scale, bn_shift, y, S, C, slice);
  else
    GA_LAUNCH((residual_relu_fwd<false>), residual_grid(S, slice), dim3(256), st, t, rem, bn_scale, bn_shift, y, S, C, slice);
  return check_launch("residual + relu forward");
}

GA_EXPORT int ganet_residual_relu_backward(const float *y, const float *grad_y, const float *bn_scale, float *grad_t,
                                           float *grad_rem, int N, int C, int D, int H, int W, void *stream)
{
  GA_TRY(check_residual("ganet_residual_relu_backward", N, C, D, H, W));
  if (!y || !grad_y || !grad_rem) return fail(GANET_E_INVALID, "ganet_residual_relu_backward: null pointer");
  if (bn_scale && !grad_t)
    return fail(GANET_E_INVALID, "ganet_residual_relu_backward: a scaled gradient needs a buffer of its own (grad_t)");
  const i64 S = (i64)N * C, slice = (i64)D * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (slice % 4 == 0 && aligned16(y) && aligned16(grad_y) && aligned16(grad_rem) && aligned16(grad_t))
    GA_LAUNCH((residual_relu_bwd<true>), residual_grid(S, slice / 4), dim3(256), st, y, grad_y, bn_scale, grad_t, grad_rem, S, C, slice);
  else
    GA_LAUNCH((residual_relu_bwd<false>), residual_grid(S, slice), dim3(256), st, y, grad_y, bn_scale, grad_t, grad_rem, S, C, slice);
  return check_launch("residual + relu backward");
}

GA_EXPORT int ganet_selftest_dpp_wave(int *scratch_dev, int *host_out, void *stream)
{
  if (!scratch_dev || !host_out) return fail(GANET_E_INVALID, "ganet_selftest_dpp_wave: null pointer");
  hipStream_t st = (hipStream_t)stream;
  GA_LAUNCH(dpp_probe_wave, dim3(1), dim3(64), st, scratch_dev);
  GA_TRY(check_launch("dpp probe (wave)"));
#if defined(GA_HIPSIM)
  memcpy(host_out, scratch_dev, sizeof(int) * 4 * 64);
#else
  GA_HIP(hipMemcpyAsync(host_out, scratch_dev, sizeof(int) * 4 * 64, hipMemcpyDeviceToHost, st));
  GA_HIP(hipStreamSynchronize(st));
#endif
  int bad = 0, first_pat = -1, first_lane = -1;
  for (int lane = 0; lane < 64; lane++) {
    const int expect[4] = {lane < 63 ? lane + 1 : -1, lane > 0 ? lane - 1 : -1, 63, 2016};
    for (int p = 0; p < 4; p++)
      if (host_out[p * 64 + lane] != expect[p]) {
        if (!bad) { first_pat = p; first_lane = lane; }
        bad++;
      }
  }
  if (bad)
    return fail(GANET_E_RUNTIME, "DPP self-test (wave): %d mismatches, first at pattern %d lane %d (got %d)",
                bad, first_pat, first_lane, host_out[first_pat * 64 + first_lane]);
  return GANET_OK;
}

GA_EXPORT int ganet_selftest_dpp(int *scratch_dev, int *host_out, void *stream)
{
  if (!scratch_dev || !host_out) return fail(GANET_E_INVALID, "ganet_selftest_dpp: null pointer");
  hipStream_t st = (hipStream_t)stream;
  GA_LAUNCH(dpp_probe, dim3(1), dim3(64), st, scratch_dev);
  GA_TRY(check_launch("dpp probe"));
#if defined(GA_HIPSIM)
  memcpy(host_out, scratch_dev, sizeof(int) * 8 * 64);
#else
  GA_HIP(hipMemcpyAsync(host_out, scratch_dev, sizeof(int) * 8 * 64, hipMemcpyDeviceToHost, st));
  GA_HIP(hipStreamSynchronize(st));
#endif
  int bad = 0, first_pat = -1, first_lane = -1;
  for (int lane = 0; lane < 64; lane++) {
    const int r = lane & 15, row = lane & ~15;
    int expect[8];
    expect[0] = lane ^ 1;
    expect[1] = lane ^ 2;
    expect[2] = r < 15 ? lane + 1 : -1;
    expect[3] = r > 0 ? lane - 1 : -1;
    expect[4] = row + 15 - r;
    expect[5] = (lane & ~7) + 7 - (lane & 7);
    // argmax of (lane*37)%64 over the 16-lane row, smallest lane on ties
    int bk = row;
    for (int l = row; l < row + 16; l++)
      if ((l * 37) % 64 > (bk * 37) % 64) bk = l;
    expect[6] = bk;
    expect[7] = 16 * row + 120;
    for (int p = 0; p < 8; p++)
      if (host_out[p * 64 + lane] != expect[p]) {
        if (!bad) { first_pat = p; first_lane = lane; }
        bad++;
      }
  }
  if (bad)
    return fail(GANET_E_RUNTIME, "DPP self-test: %d mismatches, first at pattern %d lane %d (got %d)",
                bad, first_pat, first_lane, host_out[first_pat * 64 + first_lane]);
  return GANET_OK;
}
