// sga_col_kernels.h -- vertical SGA scans (down / up) with LDS-staged column blocks.
//
// Why: the register-only vertical kernels (sga_kernels.h) must choose between wide global
// pieces and parallelism: GD=4 gives 64-byte pieces but only 416 wavefronts of 17 disparities
// per lane at cfg2 -- an ablation shows 2/3 of its 0.097 ms is the serial per-wave instruction
// stream (216 VALU per position), not memory; GD=16 has 4x the waves and 5 disparities per lane
// but touches 16-byte pieces (0.19 ms).  Staging through LDS decouples the two:
//   * a 256-thread block owns 16 neighbouring columns of one (n,c) slice; per batch of SBV = 4
//     rows the whole block copies the [D][4][16] tiles with 16-byte pieces that are 64 B
//     contiguous per (plane, row) -- the GD=4 access pattern;
//   * each of the 4 waves then runs the GD=16 compute layout out of LDS: DPP row r of wave v
//     owns column 4v+r, lane g keeps disparities [5g, 5g+5): 1664 waves, all 64 lanes busy,
//     ~70 VALU per position, DPP-only cross-lane traffic;
//   * the next batch's global loads are issued into registers before the current batch is
//     computed and land in LDS after it, so the memory latency hides behind the compute.
// Arithmetic (fwd_step / bwdg_step) is shared with sga_kernels.h: bit-exact forward.
#pragma once
#include "ga_common.h"
#include "sga_kernels.h"

namespace ga {

struct ColGeom {
  int D, H, W;
  i64 HW;
  int out_mode;     // forward scans: 0  A = tile;  1  A = max(A, tile) (running direction max, inference path)
  int tiled;        // adjoint scans: the result volume G has the private tiled layout below
};

// ---- private tiled layout of the vertical directions' ADJOINT volumes -------------------------------------------------------------
// What limits the column scans at the benchmark size is the pattern of their result stores: per batch a block writes D x 4 runs
// of 64 bytes (16 columns of one plane row) with an 832-byte row pitch -- 3.1 - 3.5 TB/s where a linear fill reaches 4.4 - 5.0
// (profiles/r3c_*, r3d_*).  The adjoint volumes G_dir are PRIVATE to ganet_sga_backward (written by the adjoint scans, read by
// sga_bwd_point, gone afterwards), so the two vertical directions keep theirs tiled:
//     element (s, d, h, w)  ->  ((((s * NCB + w / 16) * NRB + h / 4) * D + d) * 4 + h % 4) * 16 + w % 16,   NCB = W / 16, NRB = H / 4
// i.e. [slice][column block][row batch][d][4 rows][16 columns]: a block's batch is ONE contiguous burst of D * 256 bytes, a
// thread's 16-byte piece lands at 16 * tid.  Same size as the API layout (needs W % 16 == 0 and H % 4 == 0; otherwise the
// volume stays in the API layout).  Measured (profiles/r7b_* ... r7e_*, same box): column adjoint scans 98 -> 84 us each,
// sga_bwd_point (which now reads 64-byte runs of G_down / G_up) 283 -> 291, the whole step -1.1 %.
// The same layout for the directional volumes A_down / A_up of the FORWARD was built and measured as well, and removed: the
// column forward scans gain only 5.5 us each, and the merge, which runs at ~6 TB/s on 1 KB runs, loses 35 us on 64-byte runs
// (53 us with a tile-congruent pixel mapping, whose API-layout streams then break into 256-byte pieces; +2.9 % on the step with
// non-temporal loads, which let a line go before its second row is asked for): whole step +2.0 %.  Reads of a latency-bound
// kernel tolerate the short runs, a kernel near the fabric's rate does not.
GA_DEV i64 col_tiled_off(int s, int ncb, int cb, int H, int D, int d, int row)
{
  return ((((i64)s * ncb + cb) * (H >> 2) + (row >> 2)) * D + d) * 64 + (row & 3) * 16;
}

constexpr int COL_SBV = 4;     // rows (scan positions) per staged batch = one ds_read_b128
constexpr int COL_NC = 16;     // columns per block

// ---- 16 lanes per column: 256-thread blocks (the default for every model shape) ------------------------------------------
#define GA_COL_GDC 16
#define GA_COL_THREADS 256
#define GA_COL_CS 64
#define GA_COL_CIDX(wv, lane) ((wv) * 4 + ((lane) >> 4))
#define GA_COL_LG(lane) ((lane) & 15)
#define GA_COL_FWD_NAME sga_col_fwd
#define GA_COL_BWDG_NAME sga_col_bwdg
#include "sga_col_kernels.inc"
#undef GA_COL_GDC
#undef GA_COL_THREADS
#undef GA_COL_CS
#undef GA_COL_CIDX
#undef GA_COL_LG
#undef GA_COL_FWD_NAME
#undef GA_COL_BWDG_NAME

// ---- one WAVEFRONT per column: 1,024-thread blocks (sga_col_fwd_wide / sga_col_bwdg_wide; D <= 64 * DPL) -------------------
// For inputs with few column blocks and many disparities (SURVEY 8d's stress shape [1,1,192,240,624]: 39 blocks of 16 columns):
// 16 waves per block instead of 4 and 3 disparities of serial work per lane instead of 12-13, with the same 64-byte global
// pieces -- which the register-only wide segment kernels (one column per wave, every lane of a load in another plane) lack.
#define GA_COL_GDC 64
#define GA_COL_THREADS 1024
#define GA_COL_CS 256
#define GA_COL_CIDX(wv, lane) (wv)
#define GA_COL_LG(lane) (lane)
#define GA_COL_FWD_NAME sga_col_fwd_wide
#define GA_COL_BWDG_NAME sga_col_bwdg_wide
#include "sga_col_kernels.inc"
#undef GA_COL_GDC
#undef GA_COL_THREADS
#undef GA_COL_CS
#undef GA_COL_CIDX
#undef GA_COL_LG
#undef GA_COL_FWD_NAME
#undef GA_COL_BWDG_NAME

}  // namespace ga
