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
};

constexpr int COL_SBV = 4;     // rows (scan positions) per staged batch = one ds_read_b128
constexpr int COL_NC = 16;     // columns per block

// grid = (ceil(W/16), S), block = 256.  asc: visit rows 0..H-1 (down), else H-1..0 (up).
// Requires W % 4 == 0 and 16-byte aligned bases.
// dynamic LDS floats: 2 * 16*D*4 (x tile, A tile) + 16*5*4 (guidance).
template <int DPL, bool asc, bool FULL>
__global__ void __launch_bounds__(256)
sga_col_fwd(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ A,
            ColGeom geo)
{
  constexpr int SB = COL_SBV, NC = COL_NC;
  constexpr int NIT = (16 * DPL * SB + 63) / 64;        // copy iterations (D <= 16*DPL)
  GA_DYN_SMEM(smem);
  const int D = geo.D, H = geo.H, W = geo.W;
  float *xt = smem;                      // [NC][D][SB]
  float *at = xt + NC * D * SB;          // [NC][D][SB]
  float *wt = at + NC * D * SB;          // [NC][5][SB]
  const int tid = threadIdx.x;
  // XCD-aware order: neighbouring column blocks share 128-byte lines (a block covers 64 B per
  // plane row), so consecutive blocks of a slice must sit on the same XCD's L2
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bx = lid % gridDim.x, by = lid / gridDim.x;
  const int c0 = bx * NC;
  const i64 sbase = (i64)by * D * geo.HW;
  const i64 gbase = (i64)by * 5 * geo.HW;
  // compute role
  const int lane = tid & 63, wv = tid >> 6;
  const int cidx = wv * 4 + (lane >> 4);
  LaneCtx c;
  c.lg = lane & 15; c.d0 = c.lg * DPL; c.line_ok = c0 + cidx < W; c.s = 0; c.q = 0; c.cap = lane_cap(c.d0, D);
  // copy role: piece = 4 columns (16 B), seg = (plane, row-in-batch)
  const int piece = tid & 3, seg0 = tid >> 2;
  const bool pcol_ok = c0 + 4 * piece < W;
  const int nseg = D * SB;
  const int nb = (H + SB - 1) / SB;

  f4 st[NIT], sw;
#define GA_COL_FETCH(B)                                                            \
  _Pragma("unroll") for (int it = 0; it < NIT; it++) {                             \
    const int seg = it * 64 + seg0;                                                \
    const int d = seg / SB, j = seg - d * SB;                                      \
    const int p = (B) * SB + j;                                                    \
    if (seg < nseg && p < H && pcol_ok) {                                          \
      const int row = asc ? p : H - 1 - p;                                         \
      st[it] = *reinterpret_cast<const f4 *>(x + sbase + (i64)d * geo.HW + (i64)row * W + c0 + 4 * piece); \
    }                                                                              \
  }                                                                                \
  if (seg0 < 5 * SB) {                                                             \
    const int t = seg0 / SB, j = seg0 - t * SB;                                    \
    const int p = (B) * SB + j;                                                    \
    if (p < H && pcol_ok) {                                                        \
      const int row = asc ? p : H - 1 - p;                                         \
      sw = *reinterpret_cast<const f4 *>(g + gbase + (i64)t * geo.HW + (i64)row * W + c0 + 4 * piece); \
    }                                                                              \
  }
#define GA_COL_COMMIT()                                                            \
  _Pragma("unroll") for (int it = 0; it < NIT; it++) {                             \
    const int seg = it * 64 + seg0;                                                \
    const int d = seg / SB, j = seg - d * SB;                                      \
    if (seg < nseg) {                                                              \
      xt[((4 * piece + 0) * D + d) * SB + j] = st[it].x;                           \
      xt[((4 * piece + 1) * D + d) * SB + j] = st[it].y;                           \
      xt[((4 * piece + 2) * D + d) * SB + j] = st[it].z;                           \
      xt[((4 * piece + 3) * D + d) * SB + j] = st[it].w;                           \
    }                                                                              \
  }                                                                                \
  if (seg0 < 5 * SB) {                                                             \
    const int t = seg0 / SB, j = seg0 - t * SB;                                    \
    wt[((4 * piece + 0) * 5 + t) * SB + j] = sw.x;                                 \
    wt[((4 * piece + 1) * 5 + t) * SB + j] = sw.y;                                 \
    wt[((4 * piece + 2) * 5 + t) * SB + j] = sw.z;                                 \
    wt[((4 * piece + 3) * 5 + t) * SB + j] = sw.w;                                 \
  }

  float Ap[DPL], m = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Ap[i] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; it++) { st[it].x = 0.f; st[it].y = 0.f; st[it].z = 0.f; st[it].w = 0.f; }
  sw.x = 0.f; sw.y = 0.f; sw.z = 0.f; sw.w = 0.f;

  // Order inside a batch: compute -> barrier -> commit the NEXT batch's x tile (prefetched into
  // registers before the compute) -> store this batch's A tile -> barrier -> prefetch.  The commit
  // waits on the vector-memory counter, which also counts stores: placed right behind the result
  // stores (as it first was) it waited for THEIR completion, one HBM write round trip per 4 rows.
  GA_COL_FETCH(0)
  GA_COL_COMMIT()
  GA_LDS_BARRIER();
  if (1 < nb) { GA_COL_FETCH(1) }
  for (int b = 0; b < nb; b++) {
    // compute: 4 positions out of LDS
    {
      f4 xv[DPL], wv4[5], ov[DPL];
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        const int d = c.d0 + i < D ? c.d0 + i : D - 1;
        xv[i] = *reinterpret_cast<const f4 *>(xt + (cidx * D + d) * SB);
      }
#pragma unroll
      for (int t = 0; t < 5; t++) wv4[t] = *reinterpret_cast<const f4 *>(wt + (cidx * 5 + t) * SB);
#pragma unroll
      for (int k = 0; k < SB; k++) {
        if (b * SB + k < H) {
          float xs[DPL], w[5];
#pragma unroll
          for (int i = 0; i < DPL; i++) xs[i] = f4_get(xv[i], k);
#pragma unroll
          for (int t = 0; t < 5; t++) w[t] = f4_get(wv4[t], k);
          fwd_step<16, DPL, FULL>(xs, w, Ap, m, b == 0 && k == 0, c, D);
        }
#pragma unroll
        for (int i = 0; i < DPL; i++) f4_set(ov[i], k, Ap[i]);
      }
#pragma unroll
      for (int i = 0; i < DPL; i++)
        if (c.d0 + i < D) *reinterpret_cast<f4 *>(at + (cidx * D + c.d0 + i) * SB) = ov[i];
    }
    GA_LDS_BARRIER();
    if (b + 1 < nb) { GA_COL_COMMIT() }
    // copy out the A tile of this batch
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int seg = it * 64 + seg0;
      const int d = seg / SB, j = seg - d * SB;
      const int p = b * SB + j;
      if (seg < nseg && p < H && pcol_ok) {
        const int row = asc ? p : H - 1 - p;
        f4 o;
        o.x = at[((4 * piece + 0) * D + d) * SB + j];
        o.y = at[((4 * piece + 1) * D + d) * SB + j];
        o.z = at[((4 * piece + 2) * D + d) * SB + j];
        o.w = at[((4 * piece + 3) * D + d) * SB + j];
        float *dst = A + sbase + (i64)d * geo.HW + (i64)row * W + c0 + 4 * piece;
        if (geo.out_mode) {                        // (uniform) running max into the output volume
          const f4 p_ = *reinterpret_cast<const f4 *>(dst);
          o.x = o.x < p_.x ? p_.x : o.x; o.y = o.y < p_.y ? p_.y : o.y;
          o.z = o.z < p_.z ? p_.z : o.z; o.w = o.w < p_.w ? p_.w : o.w;
        }
        *reinterpret_cast<f4 *>(dst) = o;
      }
    }
    GA_LDS_BARRIER();
    if (b + 2 < nb) { GA_COL_FETCH(b + 2) }
  }
#undef GA_COL_FETCH
#undef GA_COL_COMMIT
}

// ---- adjoint scan (backward step 1) over column blocks ---------------------------------------------
// asc: VISIT order rows 0..H-1 (adjoint of `up`), else H-1..0 (adjoint of `down`).
// dynamic LDS: 2 x 16*D*4 floats (gradOut -> G in place, double-buffered) + 16*5*4 floats (guidance)
//              + D*4*16 bytes (mask) + 16*4 ints (kp).
// M16: W % 16 == 0 and a 16-byte aligned mask -- the 16 mask bytes of a (plane, row) piece are ONE
// load.  As four dword loads every wave-wide mask load touched 64 different lines for 4 bytes each,
// 8 such instructions per thread and batch: ~6x the address-coalescing time of the gradOut tile.
template <int DPL, bool asc, bool M16>
__global__ void __launch_bounds__(256)
sga_col_bwdg(const float *__restrict__ g, const uint8_t *__restrict__ mask,
             const uint16_t *__restrict__ kp, const float *__restrict__ gout,
             float *__restrict__ G, ColGeom geo, int dir)
{
  constexpr int SB = COL_SBV, NC = COL_NC;
  constexpr int NIT = (16 * DPL * SB + 63) / 64;
  constexpr int NITM = (16 * DPL * SB + 255) / 256;      // mask: one 16-byte piece per (plane,row)
  GA_DYN_SMEM(smem);
  const int D = geo.D, H = geo.H, W = geo.W;
  float *gt = smem;                                        // [2][NC][D][SB]
  float *wt = gt + 2 * NC * D * SB;                        // [NC][5][SB]
  int *kt = reinterpret_cast<int *>(wt + NC * 5 * SB);     // [NC][SB]
  uint8_t *mt = reinterpret_cast<uint8_t *>(kt + NC * SB); // [D][SB][16]
  const int tid = threadIdx.x;
  // XCD-aware order: neighbouring column blocks share 128-byte lines (a block covers 64 B per
  // plane row), so consecutive blocks of a slice must sit on the same XCD's L2
  const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bx = lid % gridDim.x, by = lid / gridDim.x;
  const int c0 = bx * NC;
  const i64 sbase = (i64)by * D * geo.HW;
  const i64 gbase = (i64)by * 5 * geo.HW;
  const i64 kbase = (i64)by * geo.HW;
  const int lane = tid & 63, wv = tid >> 6;
  const int cidx = wv * 4 + (lane >> 4);
  LaneCtx c;
  c.lg = lane & 15; c.d0 = c.lg * DPL; c.line_ok = c0 + cidx < W; c.s = 0; c.q = 0; c.cap = lane_cap(c.d0, D);
  const int piece = tid & 3, seg0 = tid >> 2;
  const bool pcol_ok = c0 + 4 * piece < W;
  const int nseg = D * SB;
  const int nb = (H + SB - 1) / SB;

  f4 st[NIT], sw;
  uint4 sm[NITM];
  int sk = 0;
#define GA_COL_FETCH(B)                                                            \
  _Pragma("unroll") for (int it = 0; it < NIT; it++) {                             \
    const int seg = it * 64 + seg0;                                                \
    const int d = seg / SB, j = seg - d * SB;                                      \
    const int p = (B) * SB + j;                                                    \
    if (seg < nseg && p < H && pcol_ok) {                                          \
      const int row = asc ? p : H - 1 - p;                                         \
      st[it] = *reinterpret_cast<const f4 *>(gout + sbase + (i64)d * geo.HW + (i64)row * W + c0 + 4 * piece); \
    }                                                                              \
  }                                                                                \
  _Pragma("unroll") for (int it = 0; it < NITM; it++) {                            \
    const int seg = it * 256 + tid;                                                \
    const int d = seg / SB, j = seg - d * SB;                                      \
    const int p = (B) * SB + j;                                                    \
    if (seg < nseg && p < H) {                                                     \
      const int row = asc ? p : H - 1 - p;                                         \
      const uint8_t *mp = mask + sbase + (i64)d * geo.HW + (i64)row * W + c0;      \
      if (M16) {                                                                   \
        sm[it] = *reinterpret_cast<const uint4 *>(mp);                             \
      } else {                                                                     \
        /* four dword loads, unconditional (columns past W re-read the block's first group and are masked): */ \
        /* under a condition each would be followed by s_waitcnt vmcnt(0) */         \
        uint32_t q4[4];                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; e++) {                            \
          const bool in = c0 + 4 * e < W;                                          \
          const uint32_t v = *reinterpret_cast<const uint32_t *>(mp + (in ? 4 * e : 0)); \
          q4[e] = v & (in ? ~0u : 0u);                                             \
        }                                                                          \
        sm[it].x = q4[0]; sm[it].y = q4[1]; sm[it].z = q4[2]; sm[it].w = q4[3];   \
      }                                                                            \
    }                                                                              \
  }                                                                                \
  if (seg0 < 5 * SB) {                                                             \
    const int t = seg0 / SB, j = seg0 - t * SB;                                    \
    const int p = (B) * SB + j;                                                    \
    if (p < H && pcol_ok) {                                                        \
      const int row = asc ? p : H - 1 - p;                                         \
      sw = *reinterpret_cast<const f4 *>(g + gbase + (i64)t * geo.HW + (i64)row * W + c0 + 4 * piece); \
    }                                                                              \
  }                                                                                \
  if (tid < NC * SB) {                                                             \
    const int j = tid / NC, cc = tid - j * NC;                                     \
    const int p = (B) * SB + j;                                                    \
    if (p < H && c0 + cc < W) {                                                    \
      const int row = asc ? p : H - 1 - p;                                         \
      sk = (int)kp[kbase + (i64)row * W + c0 + cc];                                \
    }                                                                              \
  }
#define GA_COL_COMMIT(GT)                                                          \
  _Pragma("unroll") for (int it = 0; it < NIT; it++) {                             \
    const int seg = it * 64 + seg0;                                                \
    const int d = seg / SB, j = seg - d * SB;                                      \
    if (seg < nseg) {                                                              \
      (GT)[((4 * piece + 0) * D + d) * SB + j] = st[it].x;                           \
      (GT)[((4 * piece + 1) * D + d) * SB + j] = st[it].y;                           \
      (GT)[((4 * piece + 2) * D + d) * SB + j] = st[it].z;                           \
      (GT)[((4 * piece + 3) * D + d) * SB + j] = st[it].w;                           \
    }                                                                              \
  }                                                                                \
  _Pragma("unroll") for (int it = 0; it < NITM; it++) {                            \
    const int seg = it * 256 + tid;                                                \
    if (seg < nseg) *reinterpret_cast<uint4 *>(mt + seg * 16) = sm[it];            \
  }                                                                                \
  if (seg0 < 5 * SB) {                                                             \
    const int t = seg0 / SB, j = seg0 - t * SB;                                    \
    wt[((4 * piece + 0) * 5 + t) * SB + j] = sw.x;                                 \
    wt[((4 * piece + 1) * 5 + t) * SB + j] = sw.y;                                 \
    wt[((4 * piece + 2) * 5 + t) * SB + j] = sw.z;                                 \
    wt[((4 * piece + 3) * 5 + t) * SB + j] = sw.w;                                 \
  }                                                                                \
  if (tid < NC * SB) {                                                             \
    const int j = tid / NC, cc = tid - j * NC;                                     \
    kt[cc * SB + j] = sk;                                                          \
  }

  float Gn[DPL], wn[5], sgn = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) Gn[i] = 0.f;
#pragma unroll
  for (int t = 0; t < 5; t++) wn[t] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; it++) { st[it].x = 0.f; st[it].y = 0.f; st[it].z = 0.f; st[it].w = 0.f; }
#pragma unroll
  for (int it = 0; it < NITM; it++) { sm[it].x = 0u; sm[it].y = 0u; sm[it].z = 0u; sm[it].w = 0u; }
  sw.x = 0.f; sw.y = 0.f; sw.z = 0.f; sw.w = 0.f;

  // Batch order as in sga_col_fwd: compute -> barrier -> commit the next batch (prefetched into
  // registers before the compute) -> store this batch's G -> barrier -> prefetch.  G overwrites the
  // gradOut tile in place, so that tile is double-buffered: the next batch can be committed while
  // this batch's result is still being copied out, and the commit's wait on the vector-memory
  // counter never lands right behind the result stores.
  const int TS = NC * D * SB;
  GA_COL_FETCH(0)
  GA_COL_COMMIT(gt)
  GA_LDS_BARRIER();
  if (1 < nb) { GA_COL_FETCH(1) }
  for (int b = 0; b < nb; b++) {
    float *gc = gt + (b & 1) * TS, *gnx = gt + ((b + 1) & 1) * TS;
    {
      f4 gov[DPL], wv4[5], ov[DPL];
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        const int d = c.d0 + i < D ? c.d0 + i : D - 1;
        gov[i] = *reinterpret_cast<const f4 *>(gc + (cidx * D + d) * SB);
      }
#pragma unroll
      for (int t = 0; t < 5; t++) wv4[t] = *reinterpret_cast<const f4 *>(wt + (cidx * 5 + t) * SB);
      int kv[SB];
#pragma unroll
      for (int k = 0; k < SB; k++) kv[k] = kt[cidx * SB + k];
#pragma unroll
      for (int k = 0; k < SB; k++) {
        if (b * SB + k < H) {
          float go[DPL], w[5];
          uint8_t mk[DPL];
#pragma unroll
          for (int i = 0; i < DPL; i++) {
            const int d = c.d0 + i < D ? c.d0 + i : D - 1;
            go[i] = f4_get(gov[i], k);
            mk[i] = mt[(d * SB + k) * 16 + cidx];
          }
#pragma unroll
          for (int t = 0; t < 5; t++) w[t] = f4_get(wv4[t], k);
          bwdg_step<16, DPL, uint8_t>(go, mk, Gn, wn, sgn, w, kv[k], !(b == 0 && k == 0), c, D, dir);
        }
#pragma unroll
        for (int i = 0; i < DPL; i++) f4_set(ov[i], k, Gn[i]);
      }
#pragma unroll
      for (int i = 0; i < DPL; i++)
        if (c.d0 + i < D) *reinterpret_cast<f4 *>(gc + (cidx * D + c.d0 + i) * SB) = ov[i];
    }
    GA_LDS_BARRIER();
    if (b + 1 < nb) { GA_COL_COMMIT(gnx) }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int seg = it * 64 + seg0;
      const int d = seg / SB, j = seg - d * SB;
      const int p = b * SB + j;
      if (seg < nseg && p < H && pcol_ok) {
        const int row = asc ? p : H - 1 - p;
        f4 o;
        o.x = gc[((4 * piece + 0) * D + d) * SB + j];
        o.y = gc[((4 * piece + 1) * D + d) * SB + j];
        o.z = gc[((4 * piece + 2) * D + d) * SB + j];
        o.w = gc[((4 * piece + 3) * D + d) * SB + j];
        *reinterpret_cast<f4 *>(G + sbase + (i64)d * geo.HW + (i64)row * W + c0 + 4 * piece) = o;
      }
    }
    GA_LDS_BARRIER();
    if (b + 2 < nb) { GA_COL_FETCH(b + 2) }
  }
#undef GA_COL_FETCH
#undef GA_COL_COMMIT
}

}  // namespace ga
