// lga_kernels.h -- local guided aggregation (LGA) for gfx950.
//
// What it computes: SURVEY.md Appendix A.3, i.e. the reference's
// lga_filtering_forward / lga_filter_backward / lga_data_backward
// (libs/GANet/src/GANet_kernel.cu:1131-1269): a per-pixel 3 x (2r+1) x (2r+1)
// filter over (disparity, row, col) where an out-of-range neighbour (in ANY of the
// three axes) is replaced by the centre sample.
//
// Design (instead of one CUDA thread per output element doing 75 global RMWs):
//  * one lane per PIXEL, marching over disparity; the pixel's 3K filter taps live
//    in VGPRs for the whole march (they do not depend on d), so the filter volume
//    is read from HBM exactly once;
//  * the input plane tile (+halo, zero outside the image) is staged through LDS in
//    chunks of PB planes, double-buffered, one barrier per chunk; each LDS read
//    feeds three FMAs (the plane contributes to y[d-1], y[d], y[d+1]);
//  * the centre-replacement rule is folded into per-pixel constants: taps that are
//    spatially out of range get weight 0 and their sum multiplies the centre
//    sample; the d = 0 / d = D-1 planes add the in-range part of the missing
//    depth slab.  Border and interior pixels run the same straight-line code.
//  * data-backward is the SAME kernel with transposed weights gathered from the
//    neighbouring pixels' filters (flipped tap), so it inherits the tiling;
//  * filter-backward keeps the 3K partial sums of a pixel in VGPRs across the
//    whole disparity march (the reference does a global RMW per disparity).
//
// MFMA note (north_star asks for it "where it is a true dense contraction"): per
// pixel this is a [D x K] . [K x 3] product whose two operands are BOTH private
// to the pixel; there is no operand shared across pixels, so an MFMA tile would
// run at N = 3 of 16/32 columns (<= 19 % of the fp32 MFMA rate, which on gfx950
// equals the fp32 VALU rate).  The VALU formulation is the faster one; see
// DESIGN.md.
#pragma once
#include "ga_common.h"
#include <type_traits>

namespace ga {

constexpr int LGA_TW = 32;   // tile width  (pixels, = lanes along W)
constexpr int LGA_TH = 8;    // tile height (2 / 4 / 12 / 16 measured slower, profiles/HISTORY.md)
constexpr int LGA_NT = LGA_TW * LGA_TH;   // threads per block of the tile kernels
constexpr int LGA_PB = 4;    // planes per LDS stage
#ifndef LGA_WAVES_PER_SIMD
#define LGA_WAVES_PER_SIMD 3   // register cap for R <= 2: the march is latency-bound, occupancy pays
                               // (R = 3 holds 147 taps per pixel: left uncapped so nothing spills)
#endif

// Aligned-window layout.  The tile's column halo is rounded up to an even width RE, so tile
// column 0 sits at an even LDS offset; a lane whose (2R+1)-wide window starts at an odd tile
// column (par = 1) reads from one column earlier.  Every lane therefore fetches the SAME shape
// -- NP = R+1 eight-byte-aligned ds_read_b64 per tile row (256 B/clk, twice the ds_read2_b32
// rate; neighbouring lanes of a pair read identical addresses, which broadcast) -- and the
// parity is folded into the WEIGHTS once per kernel: w6[k] = w[k - par], zero outside.  Each
// b64 result is a natural even-aligned register pair, which is exactly what v_pk_fma_f32 wants:
//   (slab -1, slab 0) accumulators += (v_k, v_k) * (w6a[k], w6b[k])       2 per pair
//   slab +1 accumulator pair       += (v_2q, v_2q+1) * (w6c[2q], w6c[2q+1])  1 per pair
// = 3*NP packed FMAs per row (45 per plane at R = 2, against 75 scalar FMAs + 25 LDS dwords in
// the first version, profiles/r1b_pmc_summary.txt).
template <int R> struct LgaCfg {
  static constexpr int WS = 2 * R + 1;
  static constexpr int K = WS * WS;
  static constexpr int RE = (R + 1) & ~1;            // even column halo
  static constexpr int NP = R + 1;                   // b64 pairs per window
  static constexpr int NK = 2 * NP;                  // window slots (one is a zero-weight dummy)
  static constexpr int TW2 = LGA_TW + 2 * RE;        // even
  static constexpr int TH2 = LGA_TH + 2 * R;
  static constexpr int PLANE = TW2 * TH2;
  static constexpr int STAGE = PLANE * LGA_PB;
  static constexpr int NLD = (STAGE + LGA_NT - 1) / LGA_NT;    // staged elements per thread
};

struct LgaGeom {
  int D, H, W;
  i64 HW;
};

// Cooperative stage load: planes [d0, d0+PB) of the tile (+halo) -> registers -> LDS.
// Which tile cell a thread copies does not depend on the chunk, so the (plane-in-chunk,
// element offset, in-image) triple is computed ONCE per kernel.  The loads themselves are
// unconditional: an always-valid address, the value ANDed with an all-ones / zero mask.  Written as
// "inside ? load : 0", every load sits in a branch of its own and hipcc, unable to tell whether one
// is still pending, guards the next use of the staging registers with s_waitcnt vmcnt(0) -- right
// behind the prefetch it was meant to overlap (scripts/isa_lint.py finds the pattern).
template <int R> struct LgaStage {
  int off[LgaCfg<R>::NLD];          // element offset from the chunk's first plane (0 if outside the image)
  unsigned msk[LgaCfg<R>::NLD];     // ~0u inside the image, 0 outside
  int pl[LgaCfg<R>::NLD];           // plane within the chunk
};
template <int R>
GA_DEV void lga_stage_init(LgaStage<R> &st, const LgaGeom &geo, int ty0, int tx0)
{
  typedef LgaCfg<R> C;
#pragma unroll
  for (int l = 0; l < C::NLD; l++) {
    const int e = l * LGA_NT + (int)threadIdx.x;
    st.off[l] = 0;
    st.msk[l] = 0u;
    st.pl[l] = 0;
    if (e < C::STAGE) {
      const int pl = e / C::PLANE, rem = e - pl * C::PLANE;
      const int r = rem / C::TW2, cc = rem - r * C::TW2;
      const int i = ty0 + r - R, j = tx0 + cc - C::RE;
      st.pl[l] = pl;
      if (i >= 0 && i < geo.H && j >= 0 && j < geo.W) { st.off[l] = i * geo.W + j; st.msk[l] = ~0u; }
    }
  }
}
template <int R>
GA_DEV void lga_stage_fetch(const float *__restrict__ xb, const LgaGeom &geo, const LgaStage<R> &st,
                            int d0, unsigned (&regs)[LgaCfg<R>::NLD])
{
  typedef LgaCfg<R> C;
#pragma unroll
  for (int l = 0; l < C::NLD; l++) {
    int d = d0 + st.pl[l];
    d = d < geo.D ? d : geo.D - 1;                      // (past the last plane: a copy that is masked below)
    regs[l] = *reinterpret_cast<const unsigned *>(xb + (i64)d * geo.HW + st.off[l]);
  }
}
template <int R>
GA_DEV void lga_stage_commit(float *__restrict__ buf, const LgaGeom &geo, const LgaStage<R> &st, int d0,
                             const unsigned (&regs)[LgaCfg<R>::NLD])
{
  typedef LgaCfg<R> C;
#pragma unroll
  for (int l = 0; l < C::NLD; l++) {
    const int e = l * LGA_NT + (int)threadIdx.x;
    const unsigned m = d0 + st.pl[l] < geo.D ? st.msk[l] : 0u;
    if (e < C::STAGE) reinterpret_cast<unsigned *>(buf)[e] = regs[l] & m;
  }
}

// weights of one pixel in the aligned-window packing; CHECK = false when every tap is in the image
template <int R, bool TRANSPOSED, bool CHECK>
GA_DEV void lga_gather_weights(const float *__restrict__ fb, const LgaGeom &geo, int ic, int jc, int par,
                               f2 (&wab)[LgaCfg<R>::WS][LgaCfg<R>::NK], f2 (&wc)[LgaCfg<R>::WS][LgaCfg<R>::NP],
                               float &cmid, float &sin_m, float &sin_p)
{
  typedef LgaCfg<R> C;
  const float *fp = fb + (i64)ic * geo.W + jc;          // own pixel, tap plane 0
#pragma unroll
  for (int a = -R; a <= R; a++) {
    float wt[3][C::WS];
#pragma unroll
    for (int dd = 0; dd < 3; dd++) {
#pragma unroll
      for (int bb = -R; bb <= R; bb++) {
        const int t = dd * C::K + (a + R) * C::WS + (bb + R);
        bool ok = true;
        if (CHECK) {
          const int i2 = ic + a, j2 = jc + bb;
          ok = i2 >= 0 && i2 < geo.H && j2 >= 0 && j2 < geo.W;
        }
        float own = 0.f;
        if (CHECK || !TRANSPOSED || dd != 1) own = fp[(i64)t * geo.HW];   // interior gX needs own taps only for the d-edge sums
        float wv = own;
        if (TRANSPOSED) {
          // unconditional load (own pixel where the neighbour is outside the image, value dropped below): under a
          // condition every one of the 75 loads of a border tile is followed by s_waitcnt vmcnt(0)
          const int tf = (2 - dd) * C::K + (-a + R) * C::WS + (-bb + R);
          const int noff = ok ? a * geo.W + bb : 0;
          wv = fp[(i64)tf * geo.HW + noff];
        }
        // (masked with AND, not selected: a select lets the compiler sink the load back under the condition)
        const int okm = ok ? -1 : 0;
        wt[dd][bb + R] = i2f(f2i(wv) & okm);
        cmid += i2f(f2i(own) & ~okm);
        if (dd == 0) sin_m += i2f(f2i(own) & okm);
        if (dd == 2) sin_p += i2f(f2i(own) & okm);
      }
    }
    float w6[3][C::NK];
#pragma unroll
    for (int dd = 0; dd < 3; dd++)
#pragma unroll
      for (int k = 0; k < C::NK; k++) {
        const float we = k < C::WS ? wt[dd][k < C::WS ? k : 0] : 0.f;            // par = 0: slot k = tap k
        const float wo = k >= 1 ? wt[dd][k >= 1 ? k - 1 : 0] : 0.f;              // par = 1: slot k = tap k-1
        w6[dd][k] = par ? wo : we;
      }
#pragma unroll
    for (int k = 0; k < C::NK; k++) wab[a + R][k] = mk2(w6[0][k], w6[1][k]);
#pragma unroll
    for (int q = 0; q < C::NP; q++) wc[a + R][q] = mk2(w6[2][2 * q], w6[2][2 * q + 1]);
  }
}

// ---- forward (TRANSPOSED = false) and data-backward (TRANSPOSED = true) ---------
// y[b,d,i,j] = sum_t w_t * xs(d+dd, i+a, j+b)  with centre replacement.
template <int R, bool TRANSPOSED>
__global__ void __launch_bounds__(LGA_NT, (R <= 2 ? LGA_WAVES_PER_SIMD : 1))
lga_apply(const float *__restrict__ x, const float *__restrict__ f, float *__restrict__ y,
          LgaGeom geo)
{
  typedef LgaCfg<R> C;
  __shared__ __attribute__((aligned(16))) float tile[2][C::STAGE];
  const int tx = threadIdx.x % LGA_TW, ty = threadIdx.x / LGA_TW;
  // XCD-aware tile order: hardware puts consecutive block ids on different XCDs (id % 8), each
  // with its own L2; neighbouring tiles share halo lines, so give every XCD a contiguous band
  // of tiles (measured: 3.2x DRAM over-fetch without it, profiles/r1h_pmc_memory_side.txt)
  int bx, by, b;
  {
    const int nb = gridDim.x * gridDim.y * gridDim.z;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nb);
    bx = lid % gridDim.x;
    by = (lid / gridDim.x) % gridDim.y;
    b = lid / (gridDim.x * gridDim.y);
  }
  const int tx0 = bx * LGA_TW, ty0 = by * LGA_TH;
  const int i = ty0 + ty, j = tx0 + tx;
  const bool inb = i < geo.H && j < geo.W;
  const int ic = i < geo.H ? i : geo.H - 1, jc = j < geo.W ? j : geo.W - 1;
  const float *xb = x + (i64)b * geo.D * geo.HW;
  const float *fb = f + (i64)b * 3 * C::K * geo.HW;
  float *yb = y + (i64)b * geo.D * geo.HW;
  const i64 pix = (i64)ic * geo.W + jc;
  const int wcol = tx + C::RE - R;          // tile column where the window starts
  const int par = wcol & 1;                 // 1: the aligned read starts one column earlier
  const int rcol = wcol - par;              // even

  // per-pixel weights (parity-shifted, packed) and centre coefficients.  Tiles whose halo lies
  // entirely inside the image (84 % of them at 240x624) skip every bounds test: the weight
  // set-up was ~30 % of this kernel's VALU instructions (profiles/r1f_pmc_lga_apply_packed.txt
  // vs the 67 VALU per plane of the steady-state loop).
  f2 wab[C::WS][C::NK];                     // (slab -1, slab 0) of window slot k
  f2 wc[C::WS][C::NP];                      // slab +1 of slots (2q, 2q+1)
  float cmid = 0.f, sin_m = 0.f, sin_p = 0.f;
  const bool interior = ty0 >= R && ty0 + LGA_TH + R <= geo.H && tx0 >= R && tx0 + LGA_TW + R <= geo.W;
  if (interior)
    lga_gather_weights<R, TRANSPOSED, false>(fb, geo, ic, jc, par, wab, wc, cmid, sin_m, sin_p);
  else
    lga_gather_weights<R, TRANSPOSED, true>(fb, geo, ic, jc, par, wab, wc, cmid, sin_m, sin_p);

  const int nchunks = (geo.D + LGA_PB - 1) / LGA_PB;
  LgaStage<R> stg;
  lga_stage_init<R>(stg, geo, ty0, tx0);
  unsigned regs[C::NLD];
  lga_stage_fetch<R>(xb, geo, stg, 0, regs);
  lga_stage_commit<R>(tile[0], geo, stg, 0, regs);
  GA_LDS_BARRIER();

  float acc_a = 0.f, acc_b = 0.f;   // partial y[d-1], y[d] while visiting plane d
  float xc_prev = 0.f;
  float *yp = yb + pix;               // output cursor: plane d-1 of the own pixel
  for (int c = 0; c < nchunks; c++) {
    lga_stage_fetch<R>(xb, geo, stg, (c + 1) * LGA_PB, regs);       // (unconditional: past the end it is a masked copy)
    const lds_cptr buf = GA_LDS_CPTR(&tile[0][0]) + (c & 1) * C::STAGE;
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      const int d = c * LGA_PB + pl;
      if (d < geo.D) {
        const lds_cptr pb = buf + pl * C::PLANE + ty * C::TW2 + rcol;
        // Six accumulator chains in a fixed rotation, rows software-pipelined through two register
        // sets.  One wave issues a VALU instruction every ~8 clk at best and a dependent
        // v_pk_fma_f32 needs its predecessor ~14 clk earlier, so with 3 waves per SIMD the chains
        // must be interleaved >= 4 deep to keep the SIMD busy (MI355X, scripts/ubench/valu_rate.hip:
        // 7.1 clk per packed FMA with two chains, 4.9 with four, 4.5 with eight).  The compiler's
        // own schedule groups the FMAs chain by chain (it minimises live registers), hence the
        // scheduling fences.
        f2 s_x[2] = {mk2(0.f, 0.f), mk2(0.f, 0.f)}, s_y[2] = {mk2(0.f, 0.f), mk2(0.f, 0.f)},
           s_p[2] = {mk2(0.f, 0.f), mk2(0.f, 0.f)};
        float xc = 0.f;
        f2 vrow[2][C::NP];
#pragma unroll
        for (int q = 0; q < C::NP; q++) vrow[0][q] = lds_read_b64(pb + 2 * q);
#pragma unroll
        for (int a = 0; a < C::WS; a++) {
          if (a + 1 < C::WS) {
#pragma unroll
            for (int q = 0; q < C::NP; q++) vrow[(a + 1) & 1][q] = lds_read_b64(pb + (a + 1) * C::TW2 + 2 * q);
          }
          GA_SCHED_FENCE();
#pragma unroll
          for (int q = 0; q < C::NP; q++) {
            const f2 vv = vrow[a & 1][q];
            const int ch = (a * C::NP + q) & 1;
            s_x[ch] = fma2(mk2(vv.x, vv.x), wab[a][2 * q], s_x[ch]);
            s_y[ch] = fma2(mk2(vv.y, vv.y), wab[a][2 * q + 1], s_y[ch]);
            s_p[ch] = fma2(vv, wc[a][q], s_p[ch]);
            GA_SCHED_FENCE();
            if (a == R && 2 * q <= R && R <= 2 * q + 1) {
              // centre sample: slot R (par 0) or R+1 (par 1)
              const float c0 = (R & 1) ? vv.y : vv.x;
              xc = par ? xc : c0;
            }
            if (a == R && 2 * q <= R + 1 && R + 1 <= 2 * q + 1) {
              const float c1 = ((R + 1) & 1) ? vv.y : vv.x;
              xc = par ? c1 : xc;
            }
          }
        }
        // (the sums must be complete here: otherwise the FMAs are sunk, as IR, below the store branch)
        GA_KEEP_F2(s_x[0]); GA_KEEP_F2(s_x[1]); GA_KEEP_F2(s_y[0]); GA_KEEP_F2(s_y[1]); GA_KEEP_F2(s_p[0]); GA_KEEP_F2(s_p[1]);
        const f2 t_x = add2(s_x[0], s_x[1]), t_y = add2(s_y[0], s_y[1]), t_p = add2(s_p[0], s_p[1]);
        const float zm = t_x.x + t_y.x;                       // depth slab -1 -> y[d+1]
        const float z0 = t_x.y + t_y.y;                       // depth slab  0 -> y[d]
        const float zp = t_p.x + t_p.y;                       // depth slab +1 -> y[d-1]
        if (d >= 1) {
          const int dy = d - 1;
          float cc = cmid;
          if (dy == 0) cc += sin_m;
          const float r = fmaf(xc_prev, cc, acc_a + zp);
          if (inb) *yp = r;
          yp += geo.HW;
        }
        acc_a = acc_b + z0;
        acc_b = zm;
        xc_prev = xc;
      }
    }
    lga_stage_commit<R>(tile[(c + 1) & 1], geo, stg, (c + 1) * LGA_PB, regs);
    GA_LDS_BARRIER();
  }
  {
    const int dy = geo.D - 1;
    float cc = cmid + sin_p;
    if (dy == 0) cc += sin_m;
    const float r = fmaf(xc_prev, cc, acc_a);
    if (inb) *yp = r;
  }
}

// ---- wave-autonomous kernels: common definitions -----------------------------------------------------------------
// What the counters said about the 256-thread version (profiles/r1i_pmc_summary.txt, lga_apply<2, false>): 2,400 waves on
// 1,024 SIMDs leave every SIMD with 2 or 3 waves and the kernel lasts as long as the 3-wave ones; the four waves of a block
// sit on four differently loaded SIMDs and meet at a barrier every chunk, so all of them run at the pace of the slowest
// (37 % of wave time parked).  In the plane-pair kernels below one WAVE owns a 32 x 2 pixel tile (and a SEGMENT of the
// disparity range), stages its own halo rows into a private LDS ring by LDS-DMA and never meets a barrier.
// (Two earlier families of the same decomposition -- register-staged, and LDS-DMA with column-packed FMAs -- were the
// defaults of round 1 and were removed in round 3; their measurements are in DESIGN.md section 7 and profiles/r1*.)
constexpr int LGAW_TH = 2;       // pixel rows per wave (lanes = LGA_TW x LGAW_TH = 64)
#ifndef LGAW_LA
#define LGAW_LA 2                // row steps of LDS lookahead
#endif

// Work items of the filter gradient = whole tiles (nseg = 1, seg_len = D; the forward / data-backward use LgaSegMix below).
struct LgaSeg {
  int nseg, seg_len, tiles_x, tiles_y;
};

// Emulator hooks of the LDS-DMA kernels.  hipcc does not count an asm global -> LDS copy, so their waits are explicit
// (GA_VMCNT) and derived from the fixed issue order; GA_DMA_MASKED(n): this lane sits out n copy instructions its wave
// issues (the wave's counter counts them all the same) -- nothing on the GPU, book-keeping for the emulator's late-landing
// copy model (tests/hipsim/hipsim.h)
#if defined(GA_HIPSIM)
#define GA_VMCNT(n) hipsim::vmcnt(n)
#define GA_LGKMCNT0() hipsim::lgkmcnt(0)
#define GA_DMA_MASKED(n) hipsim::dma_masked(n)
#else
#define GA_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define GA_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define GA_DMA_MASKED(n) ((void)0)
#endif

// ---- wave-autonomous forward / data-backward, PLANE-PAIR packing ----------------------------------------
// Packing the FMAs along the window's COLUMNS (round 1's kernels, removed) costs a 5-wide window three aligned register pairs,
// one slot of which carries a zero weight -- 45 v_pk_fma_f32 for 75 FMAs per plane (83 %), plus 15 further VALU per plane for
// the parity bookkeeping and the slab reductions.  Here the two halves of a packed FMA are two consecutive PLANES at the same window
// position.  The ring holds plane PAIRS, interleaved [row][col][2] in LDS, so ONE ds_read_b64 at any (8-byte aligned by
// construction) window position returns X = (x[2m], x[2m+1]) and with E = (y[2m], y[2m+1]), O = (y[2m+1], y[2m+2]):
//     E_m     += w( 0)[a,b] * X       (x[2m]   -> y[2m],   x[2m+1] -> y[2m+1])
//     O_{m-1} += w(+1)[a,b] * X       (x[2m]   -> y[2m-1], x[2m+1] -> y[2m]  )
//     O_m     += w(-1)[a,b] * X       (x[2m]   -> y[2m+1], x[2m+1] -> y[2m+2])
// every packed FMA does two useful FMAs (75 per plane pair = 37.5 per plane), the weight is one plain dword broadcast to
// both halves by op_sel (75 weight registers instead of 90, no parity, no dummy slot) and the window needs 25 LDS reads per
// plane pair (12.5 per plane instead of 15).  After step m:  y[2m-1] = E_{m-1}.hi + O_{m-1}.lo,  y[2m] = E_m.lo + O_{m-1}.hi.
// The interleaved layout is produced by the copy engine itself: global_load_lds_dword moves one dword per lane, lane l's
// dword lands at slot + 4 l, and the lane's GLOBAL address is free -- lane l fetches plane (l & 1), cell (l >> 1) of the
// halo'd tile (7 instructions per plane pair at R = 2).  Addresses are clamped into the image instead of masked (taps that
// fall outside the image carry weight 0, so what is staged there does not matter as long as it is finite): no exec masks,
// no ring clear.  Only a plane past the END of the volume (odd D: the last pair has one real plane) must read as zero; that
// one pair zeroes its odd cells and loads the even lanes only.
#ifndef LGAP_NR
#define LGAP_NR 5                // plane-PAIR slots per wave (1792 B each at R = 2)
#endif
#ifndef LGAP_WG_NR
#define LGAP_WG_NR 8      // ring slots (4 KB each) of the workgroup-shared ring: 34 KB per workgroup, three workgroups per CU (see the march of lga_apply_pp.inc)
#endif
#define LGAP_ROW_ASM 1           // radius 2: a window row's 15 packed FMAs as one asm statement (lga_row_fma); 0 = one statement per FMA
// Two dwords of one LDS row pair in ONE instruction, into a register pair: p[O0] and p[O1] (dword offsets, at most 255).  The
// planar staging of lga_apply_pp.inc (GA_PP_IN = 2) keeps the two planes of a pair LGA_TW + 8 dwords apart; written as two
// loads hipcc pairs up neighbouring COLUMNS instead and assembles the plane pairs with 32 v_mov per plane pair.  The asm is
// invisible to the compiler's wait insertion: lds_rows_ready() below is the counted wait that goes with it.
#if !defined(GA_HIPSIM)
template <int O0, int O1> GA_DEV void lds_read2_b32(f2 &r, lds_cptr p)
{
  static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2_b32 offsets are 8 bits");
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(p), "n"(O0), "n"(O1));
}
// wait until at most N LDS reads issued AFTER those of `row` are still in flight (LDS returns in order), and make every later
// use of the row's registers depend on the wait
template <int N> GA_DEV void lds_rows_ready(f2 (&row)[5])
{
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(row[0]), "+v"(row[1]), "+v"(row[2]), "+v"(row[3]), "+v"(row[4]) : "n"(N));
}
#else
// emulator: the read lands in `r` when a counted wait of the lane covers it (tests/hipsim/hipsim.h: late_lds)
template <int O0, int O1> GA_DEV void lds_read2_b32(f2 &r, lds_cptr p) { hipsim::lds_read2(&r.x, p + O0, p + O1); }
template <int N> GA_DEV void lds_rows_ready(f2 (&)[5]) { hipsim::lgkmcnt(N); }
#endif
// the five cells of window row TROW (0, 1, 2) relative to `p`, planes PLD dwords apart, rows ROWF dwords apart
template <int TROW, int ROWF, int PLD> GA_DEV void lds_read2_row5(f2 (&row)[5], lds_cptr p)
{
  lds_read2_b32<TROW * ROWF + 0, TROW * ROWF + 0 + PLD>(row[0], p);
  lds_read2_b32<TROW * ROWF + 1, TROW * ROWF + 1 + PLD>(row[1], p);
  lds_read2_b32<TROW * ROWF + 2, TROW * ROWF + 2 + PLD>(row[2], p);
  lds_read2_b32<TROW * ROWF + 3, TROW * ROWF + 3 + PLD>(row[3], p);
  lds_read2_b32<TROW * ROWF + 4, TROW * ROWF + 4 + PLD>(row[4], p);
}

template <int R> struct LgaPCfg {
  static constexpr int WS = 2 * R + 1;
  static constexpr int TW2 = LGA_TW + 2 * R;               // no alignment padding: a cell is 8 bytes wherever it is
  static constexpr int TH2 = LGAW_TH + 2 * R;
  static constexpr int CELLS = TW2 * TH2;                  // cells per slot, two floats (planes) each
  static constexpr int NDMA = (2 * CELLS + 63) / 64;       // copy instructions per plane pair
  static constexpr int SLOT = NDMA * 64;                   // floats per slot: every lane of every copy owns a dword
};

// M0 holds the LDS base of an LDS-DMA copy.  It is a RESERVED register: hipcc never allocates it, and the only code of its own
// that reads it (LDS-DMA builtins, s_movrel / dynamic register indexing, GWS, s_sendmsg) sets it immediately before the use, so
// a kernel without such constructs need not preserve it around the hand-written copies.  scripts/isa_loop_check.py asserts
// that no instruction outside the copy batches touches m0 in these kernels (a save / restore pair around every
// batch cost two scalar instructions; the march is bound by its instruction count, profiles/r5c_* ... r5f_*).
// ("m0" in the clobber lists instead, ADVICE r4: hipcc answers "inline asm clobber list contains reserved registers: M0 ... clobbering
// them may lead to undefined behaviour" -- for a reserved register the list is not the contract; the ISA check above is.)
#define GA_M0_SAVE_ASM "; (m0 not preserved) %0\n\t"
#define GA_M0_RESTORE_ASM ""
#define GA_M0_RESTORE_TAIL ""

// acc += X * (w, w) with w = the LOW / HIGH half of the register pair W: op_sel picks the half for both results, so 2 n
// loop-invariant weights live in n register pairs (written as fma2(X, mk2(w, w), acc) the optimiser hoists the splat out of
// the loop as a 64-bit value per weight -- 150 registers at R = 2 -- and spills)
template <int HALF> GA_DEV f2 fma2_bcast(f2 X, f2 W, f2 acc)
{
#if defined(GA_HIPSIM)
  const float w = HALF ? W.y : W.x;
  return fma2(X, mk2(w, w), acc);
#else
  if (HALF) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(X), "v"(W));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(X), "v"(W));
  return acc;
#endif
}

// X * (w, w): starts an accumulator chain without a zero-initialised register pair
template <int HALF> GA_DEV f2 mul2_bcast(f2 X, f2 W)
{
#if defined(GA_HIPSIM)
  const float w = HALF ? W.y : W.x;
  return mk2(X.x * w, X.y * w);
#else
  f2 r;
  if (HALF) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(X), "v"(W));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(X), "v"(W));
  return r;
#endif
}

// ---- one window ROW of the plane-pair forward / data-backward (5 x 5 window) as ONE asm statement ---------------------------
// The 15 packed FMAs of a window row -- 5 taps x (E_m, O_{m-1}, O_m), see above -- in one statement.  Why: a v_pk_*_f32 result
// needs one wait state before a dependent VALU read, and hipcc's hazard recogniser cannot see into inline asm: between two asm
// statements of which the second reads a register the first wrote it inserts `s_nop 0` even when a dozen other asm statements
// lie in between (it counts them as zero wait states).  With one statement per FMA that was 19 s_nop per plane pair, 12 % of
// the steady body's instructions, in kernels that are bound by what a wave ISSUES, of any kind; inside one statement the order
// below keeps dependent FMAs six instructions apart and none is needed.
//   Taps of the row LAST column first (its LDS read is issued last: one wait covers the row).  Column bb uses weight half
//   (P + bb) & 1 of pair (P + bb) >> 1 of the three pairs the caller passes for a slab, P = parity of the row's first tap index:
//   slab -1 (tap 5a + bb, chain c) and slab +1 (50 + 5a + bb, chain p): P = a & 1; slab 0 (25 + 5a + bb, chain e): P = ~a & 1.
//   Taps alternate between two accumulator chains (n = 5a + b2, b2 = 4 - bb); (e1, p1, c1) = the chain of the row's first tap.
//   FIRST (row 0): the first two taps start the chains with v_pk_mul_f32.  WAIT >= 0: the statement begins with
//   s_waitcnt lgkmcnt(WAIT) -- the counted wait that belongs to the planar staging's asm ds_read2_b32 (see lds_rows_ready).
// The three op tables below were generated from that rule (the GA_HIPSIM branch restates it in plain C; both are held to the
// oracle by tests/test_sim_lga.py and tests/test_gpu_parity.py).
template <bool A_ODD, bool FIRST, int WAIT>
GA_DEV void lga_row_fma(f2 &e1, f2 &p1, f2 &c1, f2 &e2, f2 &p2, f2 &c2, const f2 (&X)[5], const f2 *We, const f2 *Wp, const f2 *Wc)
{
#if defined(GA_HIPSIM)
  if (WAIT >= 0) hipsim::lgkmcnt(WAIT);
  const int Pe = A_ODD ? 0 : 1, Pp = A_ODD ? 1 : 0, Pc = Pp;
  f2 *acc[2][3] = {{&e1, &p1, &c1}, {&e2, &p2, &c2}};
  const f2 *W[3] = {We, Wp, Wc};
  const int P[3] = {Pe, Pp, Pc};
  for (int b2 = 0; b2 < 5; b2++) {
    const int bb = 4 - b2;
    for (int s3 = 0; s3 < 3; s3++) {
      const f2 wp = W[s3][(P[s3] + bb) >> 1];
      const float w = ((P[s3] + bb) & 1) ? wp.y : wp.x;
      f2 &a = *acc[b2 & 1][s3];
      if (FIRST && b2 < 2) a = mk2(X[bb].x * w, X[bb].y * w);
      else a = fma2(X[bb], mk2(w, w), a);
    }
  }
#else
#define GA_RF_FMA0(acc, x, w) "v_pk_fma_f32 %" #acc ", %" #x ", %" #w ", %" #acc " op_sel_hi:[1,0,1]\n\t"
#define GA_RF_FMA1(acc, x, w) "v_pk_fma_f32 %" #acc ", %" #x ", %" #w ", %" #acc " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define GA_RF_MUL0(acc, x, w) "v_pk_mul_f32 %" #acc ", %" #x ", %" #w " op_sel_hi:[1,0]\n\t"
#define GA_RF_MUL1(acc, x, w) "v_pk_mul_f32 %" #acc ", %" #x ", %" #w " op_sel:[0,1] op_sel_hi:[1,1]\n\t"
// operands: 0..2 = e1 p1 c1, 3..5 = e2 p2 c2, 6..10 = X[0..4], 11..13 = We[0..2], 14..16 = Wp[0..2], 17..19 = Wc[0..2], 20 = WAIT
#define GA_RF_ROW_EVEN \
  GA_RF_FMA1(0, 10, 13) GA_RF_FMA0(1, 10, 16) GA_RF_FMA0(2, 10, 19) \
  GA_RF_FMA0(3, 9, 13) GA_RF_FMA1(4, 9, 15) GA_RF_FMA1(5, 9, 18) \
  GA_RF_FMA1(0, 8, 12) GA_RF_FMA0(1, 8, 15) GA_RF_FMA0(2, 8, 18) \
  GA_RF_FMA0(3, 7, 12) GA_RF_FMA1(4, 7, 14) GA_RF_FMA1(5, 7, 17) \
  GA_RF_FMA1(0, 6, 11) GA_RF_FMA0(1, 6, 14) GA_RF_FMA0(2, 6, 17)
#define GA_RF_ROW_ODD \
  GA_RF_FMA0(0, 10, 13) GA_RF_FMA1(1, 10, 16) GA_RF_FMA1(2, 10, 19) \
  GA_RF_FMA1(3, 9, 12) GA_RF_FMA0(4, 9, 16) GA_RF_FMA0(5, 9, 19) \
  GA_RF_FMA0(0, 8, 12) GA_RF_FMA1(1, 8, 15) GA_RF_FMA1(2, 8, 18) \
  GA_RF_FMA1(3, 7, 11) GA_RF_FMA0(4, 7, 15) GA_RF_FMA0(5, 7, 18) \
  GA_RF_FMA0(0, 6, 11) GA_RF_FMA1(1, 6, 14) GA_RF_FMA1(2, 6, 17)
#define GA_RF_ROW_FIRST \
  GA_RF_MUL1(0, 10, 13) GA_RF_MUL0(1, 10, 16) GA_RF_MUL0(2, 10, 19) \
  GA_RF_MUL0(3, 9, 13) GA_RF_MUL1(4, 9, 15) GA_RF_MUL1(5, 9, 18) \
  GA_RF_FMA1(0, 8, 12) GA_RF_FMA0(1, 8, 15) GA_RF_FMA0(2, 8, 18) \
  GA_RF_FMA0(3, 7, 12) GA_RF_FMA1(4, 7, 14) GA_RF_FMA1(5, 7, 17) \
  GA_RF_FMA1(0, 6, 11) GA_RF_FMA0(1, 6, 14) GA_RF_FMA0(2, 6, 17)
#define GA_RF_IN "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]), "v"(X[4]), "v"(We[0]), "v"(We[1]), "v"(We[2]), "v"(Wp[0]), "v"(Wp[1]), \
                 "v"(Wp[2]), "v"(Wc[0]), "v"(Wc[1]), "v"(Wc[2]), "n"(WAIT >= 0 ? WAIT : 0)
  static_assert(!(FIRST && A_ODD), "the pair's first row is row 0");
  // (volatile only where the statement carries the wait: the wait orders it against the asm LDS reads, which the compiler
  // cannot see as producers of X)
  if constexpr (FIRST) {
    if constexpr (WAIT >= 0)
      asm volatile("s_waitcnt lgkmcnt(%20)\n\t" GA_RF_ROW_FIRST : "=&v"(e1), "=&v"(p1), "=&v"(c1), "=&v"(e2), "=&v"(p2), "=&v"(c2) : GA_RF_IN);
    else
      asm(GA_RF_ROW_FIRST : "=&v"(e1), "=&v"(p1), "=&v"(c1), "=&v"(e2), "=&v"(p2), "=&v"(c2) : GA_RF_IN);
  } else if constexpr (A_ODD) {
    if constexpr (WAIT >= 0)
      asm volatile("s_waitcnt lgkmcnt(%20)\n\t" GA_RF_ROW_ODD : "+v"(e1), "+v"(p1), "+v"(c1), "+v"(e2), "+v"(p2), "+v"(c2) : GA_RF_IN);
    else
      asm(GA_RF_ROW_ODD : "+v"(e1), "+v"(p1), "+v"(c1), "+v"(e2), "+v"(p2), "+v"(c2) : GA_RF_IN);
  } else {
    if constexpr (WAIT >= 0)
      asm volatile("s_waitcnt lgkmcnt(%20)\n\t" GA_RF_ROW_EVEN : "+v"(e1), "+v"(p1), "+v"(c1), "+v"(e2), "+v"(p2), "+v"(c2) : GA_RF_IN);
    else
      asm(GA_RF_ROW_EVEN : "+v"(e1), "+v"(p1), "+v"(c1), "+v"(e2), "+v"(p2), "+v"(c2) : GA_RF_IN);
  }
#undef GA_RF_IN
#undef GA_RF_ROW_EVEN
#undef GA_RF_ROW_ODD
#undef GA_RF_ROW_FIRST
#undef GA_RF_FMA0
#undef GA_RF_FMA1
#undef GA_RF_MUL0
#undef GA_RF_MUL1
#endif
}

// all ND copies of one plane pair: copy k moves lane l's dword from base + off[k] (bytes) to slot + 256 k + 4 l, scalar base +
// 32-bit lane offset (the offsets are the same for every pair, only the base moves).  M0 holds the LDS base of the batch
// (a write to M0 needs one wait state before the copy that uses it).
// LGAP_IMM_OFFSET = 1: the instruction's immediate offset is added to BOTH addresses of an LDS-DMA copy (global and LDS), so
// ONE M0 value serves the whole batch: copy k carries offset:256 k and its lane offsets are stored 256 k lower (the caller
// passes `base` LGAP_BIAS bytes low and offsets LGAP_BIAS bytes high so that they stay non-negative) -- ND + 4 instructions
// per batch instead of 3 ND + 3.  LGAP_IMM_OFFSET = 0 advances M0 between the copies instead.
#ifndef LGAP_IMM_OFFSET
#define LGAP_IMM_OFFSET 1
#endif
constexpr unsigned LGAP_BIAS = LGAP_IMM_OFFSET ? 4096u : 0u;
GA_DEV unsigned lga_pp_off(unsigned byte_off, int k) { return byte_off + LGAP_BIAS - (LGAP_IMM_OFFSET ? 256u * (unsigned)k : 0u); }
#if LGAP_IMM_OFFSET
#define GA_PP_COPY(n, k) "global_load_lds_dword %" #n ", %2 offset:" #k "\n\t"
#else
#define GA_PP_COPY(n, k) "s_nop 0\n\tglobal_load_lds_dword %" #n ", %2\n\ts_add_u32 m0, m0, 0x100\n\t"
#endif
template <int ND> GA_DEV void lga_dma4p_all(const float *base, const unsigned (&o)[ND], float *slot, int lane)
{
#if defined(GA_HIPSIM)
  for (int k = 0; k < ND; k++)
    hipsim::dma_issue(slot + k * 64 + lane,
                      reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + o[k] + (LGAP_IMM_OFFSET ? 256 * k : 0)), 1);
#else
  (void)lane;
  __builtin_assume(slot != nullptr);      // (a generic -> LDS address cast otherwise carries a null test: two scalar instructions per batch)
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) float *)slot;
  unsigned keep;
  static_assert(ND == 5 || ND == 7 || ND == 10, "copy batch written out for R = 1, 2, 3");
  if constexpr (ND == 5)
    asm volatile(GA_M0_SAVE_ASM "s_mov_b32 m0, %1\n\ts_nop 0\n\t" GA_PP_COPY(3, 0) GA_PP_COPY(4, 256) GA_PP_COPY(5, 512) GA_PP_COPY(6, 768)
                 GA_PP_COPY(7, 1024) GA_M0_RESTORE_TAIL
                 : "=&s"(keep) : "s"(dst), "s"(base), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]) : "memory", "scc");
  else if constexpr (ND == 7)
    asm volatile(GA_M0_SAVE_ASM "s_mov_b32 m0, %1\n\ts_nop 0\n\t" GA_PP_COPY(3, 0) GA_PP_COPY(4, 256) GA_PP_COPY(5, 512) GA_PP_COPY(6, 768)
                 GA_PP_COPY(7, 1024) GA_PP_COPY(8, 1280) GA_PP_COPY(9, 1536) GA_M0_RESTORE_TAIL
                 : "=&s"(keep) : "s"(dst), "s"(base), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]) : "memory", "scc");
  else
    asm volatile(GA_M0_SAVE_ASM "s_mov_b32 m0, %1\n\ts_nop 0\n\t" GA_PP_COPY(3, 0) GA_PP_COPY(4, 256) GA_PP_COPY(5, 512) GA_PP_COPY(6, 768)
                 GA_PP_COPY(7, 1024) GA_PP_COPY(8, 1280) GA_PP_COPY(9, 1536) GA_PP_COPY(10, 1792) GA_PP_COPY(11, 2048) GA_PP_COPY(12, 2304)
                 GA_M0_RESTORE_TAIL
                 : "=&s"(keep) : "s"(dst), "s"(base), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]),
                   "v"(o[8]), "v"(o[9]) : "memory", "scc");
#endif
}

// weights of one pixel for the plane-pair kernels: tap t = (dd * WS + a) * WS + b in half (t & 1) of register pair t >> 1,
// zero where the tap leaves the image; cmid / sin_m / sin_p as in lga_gather_weights (centre coefficient of the spatially
// replaced taps, in-range sums of the two outer depth slabs).  One depth slab at a time (scheduling fence in between): left
// alone, the compiler issues all 125 loads of the transposed gather first and spills their destinations.
// OWN_SUMS = false (transposed only): cmid / sin_m / sin_p are given (the forward pass of the same filters wrote them,
// lga_apply_pp.inc: `edge`), the first of the two sweeps below is skipped.
template <int R, bool TRANSPOSED, bool CHECK, bool OWN_SUMS = true>
GA_DEV void lga_gather_pairs(const float *__restrict__ fb, const LgaGeom &geo, int ic, int jc,
                             f2 (&wq)[(3 * (2 * R + 1) * (2 * R + 1) + 1) / 2], float &cmid, float &sin_m, float &sin_p)
{
  constexpr int WS = 2 * R + 1, K = WS * WS, NT = 3 * K;
  // every address = (uniform tap-plane pointer) + (32-bit per-lane pixel offset): scalar base + one offset register per
  // distinct neighbour instead of a 64-bit address per load
  const unsigned pix32 = (unsigned)(ic * geo.W + jc);
  // in-image test of tap (a, bb) as an all-ones / zero word: row mask AND column mask, ten per-lane words for a border tile
  // instead of 75 predicates (which do not fit the scalar registers); an AND with such a word is not a select the optimiser
  // could sink a load under
  int mrow[WS], mcol[WS];
#pragma unroll
  for (int k = 0; k < WS; k++) {
    mrow[k] = (!CHECK || (ic + k - R >= 0 && ic + k - R < geo.H)) ? -1 : 0;
    mcol[k] = (!CHECK || (jc + k - R >= 0 && jc + k - R < geo.W)) ? -1 : 0;
  }
  auto ok_mask = [&](int a, int bb) { return CHECK ? (mrow[a + R] & mcol[bb + R]) : -1; };
  // the pixel's OWN tap t feeds the three sums (and, untransposed, is the weight); loads are unconditional and masked with AND
  // (a load under a condition costs an s_waitcnt vmcnt(0) each)
  auto own_tap = [&](int t) -> float {
    const int dd = t / K, a = (t % K) / WS - R, bb = t % WS - R;
    const float own = stream_load<(GA_NT_LOADS & 16) != 0>(fb + (i64)t * geo.HW + pix32);
    const int okm = ok_mask(a, bb);
    cmid += i2f(f2i(own) & ~okm);
    if (dd == 0) sin_m += i2f(f2i(own) & okm);
    if (dd == 2) sin_p += i2f(f2i(own) & okm);
    return i2f(f2i(own) & okm);
  };
  static_assert(OWN_SUMS || TRANSPOSED, "untransposed: the own taps ARE the weights");
  if (TRANSPOSED && OWN_SUMS) {
    // Two sweeps, so that at most ~75 loads are in flight with the 76 weight registers not yet live in the first: (1) the own
    // taps, reduced to the three sums (an interior pixel needs only the two outer slabs: its centre coefficient is zero);
    // (2) the weights proper, tap (-dd, -a, -b) of the neighbour at (+a, +b).  In one sweep the ~125 destinations plus the
    // weights exceed the 168 registers of three waves per SIMD (13 - 21 spill operations per lane).
    // One depth slab at a time: its K loads are issued back to back into an array, THEN reduced (written as load-and-add per
    // tap the sums are one dependent chain and hipcc issues load, s_waitcnt vmcnt(0), add, load, ...: 75 serial round trips in
    // a border tile, +20 % on the whole data-backward pass, profiles/r7a_*).
#pragma unroll
    for (int dd = 0; dd < 3; dd++) {
      if (CHECK || dd != 1) {
        float ow[K];
#pragma unroll
        for (int k = 0; k < K; k++) ow[k] = stream_load<(GA_NT_LOADS & 16) != 0>(fb + (i64)(dd * K + k) * geo.HW + pix32);
        GA_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < K; k++) {
          const int okm = ok_mask(k / WS - R, k % WS - R);
          cmid += i2f(f2i(ow[k]) & ~okm);
          if (dd == 0) sin_m += i2f(f2i(ow[k]) & okm);
          if (dd == 2) sin_p += i2f(f2i(ow[k]) & okm);
        }
      }
    }
    GA_SCHED_FENCE();
  }
  auto tap = [&](int t) -> float {
    if (!TRANSPOSED) return own_tap(t);
    const int dd = t / K, a = (t % K) / WS - R, bb = t % WS - R;
    const int okm = ok_mask(a, bb);
    const int tf = (2 - dd) * K + (-a + R) * WS + (-bb + R);
    const int noff = (a * geo.W + bb) & okm;              // (own pixel where the neighbour is outside the image, value dropped)
    const float wv = stream_load<(GA_NT_LOADS & 16) != 0>(fb + (i64)tf * geo.HW + (unsigned)((int)pix32 + noff));
    return i2f(f2i(wv) & okm);
  };
#pragma unroll
  for (int t = 0; t < NT; t += 2) {
    const float lo = tap(t);
    const float hi = t + 1 < NT ? tap(t + 1) : 0.f;
    wq[t >> 1] = mk2(lo, hi);
    if ((!TRANSPOSED || CHECK) && (t + 2) % K < 2) GA_SCHED_FENCE();            // (about once per depth slab)
  }
}

// both 16-byte copies of one pair of a pair-interleaved volume (lga_apply_pp.inc, GA_PP_IN): copy k moves lane l's 16 bytes
// from base + o[k] to slot + 1024 k + 16 l; one M0, the second copy through the immediate offset (see lga_dma4p_all)
GA_DEV void lga_dma16p_pair(const float *base, const unsigned (&o)[2], float *slot, int lane)
{
  static_assert(LGAP_IMM_OFFSET == 1, "written for the immediate-offset form");
#if defined(GA_HIPSIM)
  for (int k = 0; k < 2; k++)
    hipsim::dma_issue(slot + k * 256 + 4 * lane, reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + o[k] + 1024 * k), 4);
#else
  (void)lane;
  __builtin_assume(slot != nullptr);      // (a generic -> LDS address cast otherwise carries a null test: two scalar instructions per batch)
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) float *)slot;
  unsigned keep;
  asm volatile(GA_M0_SAVE_ASM "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %3, %2 offset:0\n\tglobal_load_lds_dwordx4 %4, %2 offset:1024\n\t"
               GA_M0_RESTORE_TAIL
               : "=&s"(keep) : "s"(dst), "s"(base), "v"(o[0]), "v"(o[1]) : "memory", "scc");
#endif
}

// ONE 16-byte copy per lane: lane l's 16 bytes from base + off to slot + 16 l (the workgroup-shared ring of lga_apply_pp.inc,
// GA_PP_IN = 3: `slot` is this wave's quarter of the ring slot)
GA_DEV void lga_dma16p_one(const float *base, unsigned off, float *slot, int lane)
{
#if defined(GA_HIPSIM)
  hipsim::dma_issue(slot + 4 * lane, reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off), 4);
#else
  (void)lane;
  __builtin_assume(slot != nullptr);
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) float *)slot;
  unsigned keep;
  asm volatile(GA_M0_SAVE_ASM "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2 offset:0\n\t" GA_M0_RESTORE_TAIL
               : "=&s"(keep) : "s"(dst), "s"(base), "v"(off) : "memory", "scc");
#endif
}

// ---- item list of the forward / data-backward kernels: whole tiles first, then tiles cut into depth segments -----------------
// These kernels are bound by VALU issue and the waves of a SIMD share one VALU, so a pass lasts as long as the SIMD with the
// most resident work: 2,400 tiles on 1,024 SIMDs are 2 or 3 tiles per SIMD and the pass takes the time of 3 (the FMA-only
// ablation runs exactly 3 x 97 x 75 packed FMAs x 4 clk = 36 us), although the average is 2.34.  Cutting EVERY tile in
// two was measured slower (each item gathers its 75 taps and fills its pipeline again).  Here only the T mod S tiles
// beyond a whole number per SIMD are cut, into nsub segments with T mod S x nsub <= S: items [0, n_whole) are whole tiles --
// dispatched first, q per SIMD -- and the rest are the segments, at most one per SIMD (measured: forward pass 0.103 ->
// 0.0955 ms, profiles/r3a_*).  The same list describes the plain forms: n_whole = all tiles (nothing cut), or n_whole = 0 with
// every tile cut into nsub equal segments (few tiles; GANET_LGA_SEGS).  Segments start on even planes.
struct LgaSegMix {
  int tiles_x, tiles_y;
  int n_whole;        // whole tiles (items [0, n_whole)), then nsub items per remaining tile
  int nsub, sub_len;  // sub_len even
};
GA_DEV void lga_decode_item_mix(const LgaSegMix &sg, int D, int &bx, int &by, int &b, int &d_lo, int &d_hi)
{
  int item = (int)blockIdx.x;
  if (item < sg.n_whole) {
    item = xcd_remap(item, sg.n_whole);
    d_lo = 0;
    d_hi = D;
  } else {
    const int idx = item - sg.n_whole;
    const int seg = idx % sg.nsub;
    item = sg.n_whole + idx / sg.nsub;
    d_lo = seg * sg.sub_len;
    d_hi = d_lo + sg.sub_len < D ? d_lo + sg.sub_len : D;
  }
  bx = item % sg.tiles_x; item /= sg.tiles_x;
  by = item % sg.tiles_y;
  b = item / sg.tiles_y;
}
// ---- lga_apply_pp (API layout in, API layout out) and its pair-interleaved forms: lga_apply_pp.inc --------------------------
#define GA_PP_NAME lga_apply_pp
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 0
#define GA_PP_OUT 0
#define GA_PP_SLOT PC::SLOT
#define GA_PP_NDC ND
#define GA_PP_Y(d) yb[(i64)(d) * geo.HW + pix]
#include "lga_apply_pp.inc"

// API layout in and out, the input staged planar by 16-byte copies (W % 4 == 0, 16-byte aligned x; radius 2)
#define GA_PP_NAME lga_apply_pp_x
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 2
#define GA_PP_OUT 0
#define GA_PP_SLOT 512
#define GA_PP_NDC 2
#define GA_PP_Y(d) yb[(i64)(d) * geo.HW + pix]
#include "lga_apply_pp.inc"

// API layout in and out, ONE planar ring per 256-thread workgroup (32 x 8 tile, W % 4 == 0, 16-byte aligned x; radius 2; GANET_LGA_WAVE = 2)
#define GA_PP_NAME lga_apply_pp_wx
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 3
#define GA_PP_OUT 0
#define GA_PP_SLOT 1024
#define GA_PP_NDC 1
#define GA_PP_Y(d) yb[(i64)(d) * geo.HW + pix]
#include "lga_apply_pp.inc"

#define GA_PP_Y_PAIRED(d) yb[((i64)((d) >> 1) * geo.HW + pix) * 2 + ((d) & 1)]
// API layout in, pair-interleaved out (first pass of an LGA2; data-backward of its second pass)
#define GA_PP_NAME lga_apply_pp_po
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 0
#define GA_PP_OUT 1
#define GA_PP_SLOT PC::SLOT
#define GA_PP_NDC ND
#define GA_PP_Y(d) GA_PP_Y_PAIRED(d)
#include "lga_apply_pp.inc"
// the same with the API-layout input staged planar by 16-byte copies (W % 4 == 0, 16-byte aligned x)
#define GA_PP_NAME lga_apply_pp_xo
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 2
#define GA_PP_OUT 1
#define GA_PP_SLOT 512
#define GA_PP_NDC 2
#define GA_PP_Y(d) GA_PP_Y_PAIRED(d)
#include "lga_apply_pp.inc"

// the same with ONE planar ring per 256-thread workgroup (GANET_LGA_WAVE = 2)
#define GA_PP_NAME lga_apply_pp_wxo
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 3
#define GA_PP_OUT 1
#define GA_PP_SLOT 1024
#define GA_PP_NDC 1
#define GA_PP_Y(d) GA_PP_Y_PAIRED(d)
#include "lga_apply_pp.inc"
// pair-interleaved in, API layout out (second pass of an LGA2; data-backward of its first pass)
#define GA_PP_NAME lga_apply_pp_pi
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 1
#define GA_PP_OUT 0
#define GA_PP_SLOT 512
#define GA_PP_NDC 2
#define GA_PP_Y(d) yb[(i64)(d) * geo.HW + pix]
#include "lga_apply_pp.inc"
// the same through ONE ring per 256-thread workgroup (GANET_LGA_WAVE = 2)
#define GA_PP_NAME lga_apply_pp_wpi
#define GA_PP_SEG_T LgaSegMix
#define GA_PP_DECODE lga_decode_item_mix
#define GA_PP_IN 4
#define GA_PP_OUT 0
#define GA_PP_SLOT 1024
#define GA_PP_NDC 1
#define GA_PP_Y(d) yb[(i64)(d) * geo.HW + pix]
#include "lga_apply_pp.inc"
#undef GA_PP_Y_PAIRED

// one 4-byte global -> LDS copy per lane, scalar base + 32-bit lane offset: lane l's dword lands at slot + 4 * l
GA_DEV void lga_dma4s(const float *base, unsigned off, float *slot, int lane)
{
#if defined(GA_HIPSIM)
  hipsim::dma_issue(slot + lane, reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off), 1);
#else
  (void)lane;
  __builtin_assume(slot != nullptr);      // (a generic -> LDS address cast otherwise carries a null test: two scalar instructions per batch)
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) float *)slot;
  unsigned keep;
  asm volatile(GA_M0_SAVE_ASM "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %3, %2" GA_M0_RESTORE_ASM
               : "=&s"(keep) : "s"(dst), "s"(base), "v"(off) : "memory");
#endif
}

// acc += (x, x) * G with x = the LOW / HIGH half of X (op_sel on the first operand)
template <int HALF> GA_DEV f2 fma2_xbcast(f2 X, f2 G, f2 acc)
{
#if defined(GA_HIPSIM)
  const float x = HALF ? X.y : X.x;
  return fma2(mk2(x, x), G, acc);
#else
  if (HALF) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(X), "v"(G));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(X), "v"(G));
  return acc;
#endif
}

// one window ROW of the plane-pair filter gradient (5 x 5 window) as ONE asm statement, for the reason given at lga_row_fma:
//   P[bb] += (X.x, X.x) * Ga;  Q[bb] += X * Gc;  then  P[bb] += (X.y, X.y) * (Gn.y, Gn.x)   (the two updates of a P five
//   instructions apart), last column first.  Gn = (g[2q+1], g[2q+2]) as the gy ring holds it: the multiplier of the odd plane,
//   (g[2q+2], g[2q+1]), is Gn with its halves swapped by op_sel.
// WAIT >= 0: preceded by s_waitcnt lgkmcnt(WAIT), the counted wait of the planar staging's asm LDS reads.
template <int WAIT>
GA_DEV void lga_row_fg(f2 (&Pr)[5], f2 (&Qr)[5], const f2 (&X)[5], f2 Ga, f2 Gn, f2 Gc)
{
#if defined(GA_HIPSIM)
  if (WAIT >= 0) hipsim::lgkmcnt(WAIT);
  for (int b2 = 0; b2 < 5; b2++) {
    const int bb = 4 - b2;
    Pr[bb] = fma2(mk2(X[bb].x, X[bb].x), Ga, Pr[bb]);
    Qr[bb] = fma2(X[bb], Gc, Qr[bb]);
  }
  for (int b2 = 0; b2 < 5; b2++) {
    const int bb = 4 - b2;
    Pr[bb] = fma2(mk2(X[bb].y, X[bb].y), mk2(Gn.y, Gn.x), Pr[bb]);
  }
#else
// operands: 0..4 = P[0..4], 5..9 = Q[0..4], 10..14 = X[0..4], 15 = Ga, 16 = Gn, 17 = Gc, 18 = WAIT
#define GA_FG_PX(p, x) "v_pk_fma_f32 %" #p ", %" #x ", %15, %" #p " op_sel_hi:[0,1,1]\n\t"
#define GA_FG_PY(p, x) "v_pk_fma_f32 %" #p ", %" #x ", %16, %" #p " op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
#define GA_FG_Q(q, x) "v_pk_fma_f32 %" #q ", %" #x ", %17, %" #q "\n\t"
#define GA_FG_ROW GA_FG_PX(4, 14) GA_FG_Q(9, 14) GA_FG_PX(3, 13) GA_FG_Q(8, 13) GA_FG_PX(2, 12) GA_FG_Q(7, 12) GA_FG_PX(1, 11) GA_FG_Q(6, 11) \
                  GA_FG_PX(0, 10) GA_FG_Q(5, 10) GA_FG_PY(4, 14) GA_FG_PY(3, 13) GA_FG_PY(2, 12) GA_FG_PY(1, 11) GA_FG_PY(0, 10)
#define GA_FG_OPS : "+v"(Pr[0]), "+v"(Pr[1]), "+v"(Pr[2]), "+v"(Pr[3]), "+v"(Pr[4]), "+v"(Qr[0]), "+v"(Qr[1]), "+v"(Qr[2]), "+v"(Qr[3]), "+v"(Qr[4]) \
                   : "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]), "v"(X[4]), "v"(Ga), "v"(Gn), "v"(Gc), "n"(WAIT >= 0 ? WAIT : 0)
  if constexpr (WAIT >= 0) asm volatile("s_waitcnt lgkmcnt(%18)\n\t" GA_FG_ROW GA_FG_OPS);
  else asm(GA_FG_ROW GA_FG_OPS);
#undef GA_FG_PX
#undef GA_FG_PY
#undef GA_FG_Q
#undef GA_FG_ROW
#undef GA_FG_OPS
#endif
}

// The same for N = 3 or N = 2 neighbouring cells of a window row (lga_filter_grad_pair.inc splits the centre row between its two
// waves: columns 0 .. 2 and 3, 4).  Dependent updates of a P stay >= 4 instructions apart.
template <int WAIT, int N>
GA_DEV void lga_row_fg_n(f2 *Pr, f2 *Qr, const f2 *X, f2 Ga, f2 Gn, f2 Gc)
{
  static_assert(N == 2 || N == 3, "partial rows of 2 or 3 cells");
#if defined(GA_HIPSIM)
  if (WAIT >= 0) hipsim::lgkmcnt(WAIT);
  for (int b2 = 0; b2 < N; b2++) {
    const int bb = N - 1 - b2;
    Pr[bb] = fma2(mk2(X[bb].x, X[bb].x), Ga, Pr[bb]);
    Qr[bb] = fma2(X[bb], Gc, Qr[bb]);
  }
  for (int b2 = 0; b2 < N; b2++) {
    const int bb = N - 1 - b2;
    Pr[bb] = fma2(mk2(X[bb].y, X[bb].y), mk2(Gn.y, Gn.x), Pr[bb]);
  }
#else
#define GA_FGN_PX(p, x, ga) "v_pk_fma_f32 %" #p ", %" #x ", %" #ga ", %" #p " op_sel_hi:[0,1,1]\n\t"
#define GA_FGN_PY(p, x, gn) "v_pk_fma_f32 %" #p ", %" #x ", %" #gn ", %" #p " op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
#define GA_FGN_Q(q, x, gc) "v_pk_fma_f32 %" #q ", %" #x ", %" #gc ", %" #q "\n\t"
  if constexpr (N == 3) {
    // operands: 0..2 = P, 3..5 = Q, 6..8 = X, 9 = Ga, 10 = Gn, 11 = Gc, 12 = WAIT
#define GA_FGN_ROW GA_FGN_PX(2, 8, 9) GA_FGN_Q(5, 8, 11) GA_FGN_PX(1, 7, 9) GA_FGN_Q(4, 7, 11) GA_FGN_PX(0, 6, 9) GA_FGN_Q(3, 6, 11) \
                   GA_FGN_PY(2, 8, 10) GA_FGN_PY(1, 7, 10) GA_FGN_PY(0, 6, 10)
#define GA_FGN_OPS : "+v"(Pr[0]), "+v"(Pr[1]), "+v"(Pr[2]), "+v"(Qr[0]), "+v"(Qr[1]), "+v"(Qr[2]) \
                   : "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(Ga), "v"(Gn), "v"(Gc), "n"(WAIT >= 0 ? WAIT : 0)
    if constexpr (WAIT >= 0) asm volatile("s_waitcnt lgkmcnt(%12)\n\t" GA_FGN_ROW GA_FGN_OPS);
    else asm(GA_FGN_ROW GA_FGN_OPS);
#undef GA_FGN_ROW
#undef GA_FGN_OPS
  } else {
    // operands: 0..1 = P, 2..3 = Q, 4..5 = X, 6 = Ga, 7 = Gn, 8 = Gc, 9 = WAIT
#define GA_FGN_ROW GA_FGN_PX(1, 5, 6) GA_FGN_Q(3, 5, 8) GA_FGN_PX(0, 4, 6) GA_FGN_Q(2, 4, 8) GA_FGN_PY(1, 5, 7) GA_FGN_PY(0, 4, 7)
#define GA_FGN_OPS : "+v"(Pr[0]), "+v"(Pr[1]), "+v"(Qr[0]), "+v"(Qr[1]) : "v"(X[0]), "v"(X[1]), "v"(Ga), "v"(Gn), "v"(Gc), "n"(WAIT >= 0 ? WAIT : 0)
    if constexpr (WAIT >= 0) asm volatile("s_waitcnt lgkmcnt(%9)\n\t" GA_FGN_ROW GA_FGN_OPS);
    else asm(GA_FGN_ROW GA_FGN_OPS);
#undef GA_FGN_ROW
#undef GA_FGN_OPS
  }
#undef GA_FGN_PX
#undef GA_FGN_PY
#undef GA_FGN_Q
#endif
}

// ---- filter backward, PLANE-PAIR packing -------------------------------------------------------------------
// Same staging as lga_apply_pp (plane pairs X = (x[2m], x[2m+1]) interleaved in LDS, one ds_read_b64 per window position),
// the lane's own gy values through a second ring (one dword copy per plane, [plane][lane]).  With g_k = gy[k] of the pixel:
//     P[a,b] = (gf(-1), gf(0))[a,b]     += (x[2m],   x[2m]  ) * (g_{2m+1}, g_{2m}  )
//                                       += (x[2m+1], x[2m+1]) * (g_{2m+2}, g_{2m+1})
//     Q[a,b] = gf(+1)[a,b] as (lo, hi)  += (x[2m],   x[2m+1]) * (g_{2m-1}, g_{2m}  )
// 75 packed FMAs per plane PAIR, all useful (the column-packed kernel needs 45 per plane), 100 accumulator registers
// (90 there), 25 LDS window reads per pair.  Copies per step, in this order: gy(2q+1), gy(2q+2), then the ND copies of x
// pair q -- so once x pair q has landed, gy up to plane 2q+2 has too.
// WPS = waves per SIMD the register budget is set for, LA = window rows of LDS look-ahead.  The explicit s_waitcnt vmcnt(n)
// bookkeeping of the march is only valid while the compiler adds NO vector-memory operation of its own to the loop -- a
// register spill is one (scratch_load / scratch_store count in vmcnt) -- so the (WPS, LA) pairs offered by the launcher are
// the ones scripts/isa_lint.py finds spill-free inside the loop: (3, 0) and (2, 2) at R = 2.
#define GA_FG_NAME lga_filter_grad_pp
#define GA_FG_XP 0
#define GA_FG_GYP 0
#define GA_FG_SLOT PC::SLOT
#define GA_FG_NDC ND
#include "lga_filter_grad_pp.inc"
// x staged planar by 16-byte copies (W % 4 == 0, 16-byte aligned x; radius 2)
#define GA_FG_NAME lga_filter_grad_pp_x
#define GA_FG_XP 2
#define GA_FG_GYP 0
#define GA_FG_SLOT 512
#define GA_FG_NDC 2
#include "lga_filter_grad_pp.inc"
// x pair-interleaved (the filter gradient of the second pass of an LGA2, whose x is the private intermediate)
#define GA_FG_NAME lga_filter_grad_pp_xp
#define GA_FG_XP 1
#define GA_FG_GYP 0
#define GA_FG_SLOT 512
#define GA_FG_NDC 2
#include "lga_filter_grad_pp.inc"

// gy pair-interleaved, x in the API layout (the filter gradient of the first pass of an LGA2)
#define GA_FG_NAME lga_filter_grad_pp_gyp
#define GA_FG_XP 0
#define GA_FG_GYP 1
#define GA_FG_SLOT PC::SLOT
#define GA_FG_NDC ND
#include "lga_filter_grad_pp.inc"
// the same with the API-layout x staged planar by 16-byte copies (W % 4 == 0, 16-byte aligned x)
#define GA_FG_NAME lga_filter_grad_pp_gypx
#define GA_FG_XP 2
#define GA_FG_GYP 1
#define GA_FG_SLOT 512
#define GA_FG_NDC 2
#include "lga_filter_grad_pp.inc"

// the taps of a tile split over a wave pair (lga_filter_grad_pair.inc): the two filter gradients of an LGA2's backward
#define GA_FGP_NAME lga_filter_grad_pair_xp
#define GA_FGP_XP 1
#define GA_FGP_GYP 0
#include "lga_filter_grad_pair.inc"

// ---- filter backward --------------------------------------------------------------
// gf[b,t,i,j] (+)= sum_d gy[b,d,i,j] * xs(d+dd, i+a, j+b)   (centre replacement)
template <int R>
__global__ void __launch_bounds__(LGA_NT, (R <= 2 ? LGA_WAVES_PER_SIMD : 1))
lga_filter_grad(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gf,
                LgaGeom geo, int accumulate)
{
  typedef LgaCfg<R> C;
  __shared__ __attribute__((aligned(16))) float tile[2][C::STAGE];
  const int tx = threadIdx.x % LGA_TW, ty = threadIdx.x / LGA_TW;
  // XCD-aware tile order: hardware puts consecutive block ids on different XCDs (id % 8), each
  // with its own L2; neighbouring tiles share halo lines, so give every XCD a contiguous band
  // of tiles (measured: 3.2x DRAM over-fetch without it, profiles/r1h_pmc_memory_side.txt)
  int bx, by, b;
  {
    const int nb = gridDim.x * gridDim.y * gridDim.z;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nb);
    bx = lid % gridDim.x;
    by = (lid / gridDim.x) % gridDim.y;
    b = lid / (gridDim.x * gridDim.y);
  }
  const int tx0 = bx * LGA_TW, ty0 = by * LGA_TH;
  const int i = ty0 + ty, j = tx0 + tx;
  const bool inb = i < geo.H && j < geo.W;
  const int ic = i < geo.H ? i : geo.H - 1, jc = j < geo.W ? j : geo.W - 1;
  const float *xb = x + (i64)b * geo.D * geo.HW;
  const float *gyb = gy + (i64)b * geo.D * geo.HW;
  float *gfb = gf + (i64)b * 3 * C::K * geo.HW;
  const i64 pix = (i64)ic * geo.W + jc;
  const int wcol = tx + C::RE - R;
  const int par = wcol & 1;
  const int rcol = wcol - par;

  // partial sums per window slot: sab[a][k] = (slab -1, slab 0), sc[a][q] = slab +1 of (2q, 2q+1)
  f2 sab[C::WS][C::NK], sc[C::WS][C::NP];
#pragma unroll
  for (int a = 0; a < C::WS; a++) {
#pragma unroll
    for (int k = 0; k < C::NK; k++) sab[a][k] = mk2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < C::NP; q++) sc[a][q] = mk2(0.f, 0.f);
  }
  float gc = 0.f;                 // sum_d gy[d] * x[d][centre]
  float e_lo = 0.f, e_hi = 0.f;   // gy[0]*x[0][c], gy[D-1]*x[D-1][c]

  const int nchunks = (geo.D + LGA_PB - 1) / LGA_PB;
  LgaStage<R> stg;
  lga_stage_init<R>(stg, geo, ty0, tx0);
  unsigned regs[C::NLD];
  lga_stage_fetch<R>(xb, geo, stg, 0, regs);
  lga_stage_commit<R>(tile[0], geo, stg, 0, regs);
  GA_LDS_BARRIER();

  // gy of the own pixel at planes d-1, d, d+1 (rolling); the next chunk's values are fetched
  // one chunk ahead so the march never waits on a dependent global load
  float g_m = 0.f, g_0 = gyb[pix];
  unsigned gnext[LGA_PB];
  auto gy_fetch = [&](int dn) {          // gy[dn] of the own pixel, 0 past the last plane; unconditional load
    const int dc = dn < geo.D ? dn : geo.D - 1;
    const unsigned v = *reinterpret_cast<const unsigned *>(gyb + (i64)dc * geo.HW + pix);
    return v & (dn < geo.D ? ~0u : 0u);
  };
#pragma unroll
  for (int pl = 0; pl < LGA_PB; pl++) gnext[pl] = gy_fetch(pl + 1);
  for (int c = 0; c < nchunks; c++) {
    lga_stage_fetch<R>(xb, geo, stg, (c + 1) * LGA_PB, regs);       // (unconditional: past the end it is a masked copy)
    float gcur[LGA_PB];
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      gcur[pl] = i2f((int)gnext[pl]);
      gnext[pl] = gy_fetch((c + 1) * LGA_PB + pl + 1);
    }
    const lds_cptr buf = GA_LDS_CPTR(&tile[0][0]) + (c & 1) * C::STAGE;
#pragma unroll
    for (int pl = 0; pl < LGA_PB; pl++) {
      const int d = c * LGA_PB + pl;
      if (d < geo.D) {
        const float g_p = gcur[pl];                       // gy[d+1] (0 past the end)
        const lds_cptr pb = buf + pl * C::PLANE + ty * C::TW2 + rcol;
        // plane d pairs with gy[d+1] for slab -1, gy[d] for slab 0, gy[d-1] for slab +1
        const f2 g01 = mk2(g_p, g_0), gmm = mk2(g_m, g_m);
        float xc = 0.f;
#pragma unroll
        for (int a = 0; a < C::WS; a++) {
#pragma unroll
          for (int q = 0; q < C::NP; q++) {
            const f2 vv = lds_read_b64(pb + a * C::TW2 + 2 * q);
            sab[a][2 * q] = fma2(mk2(vv.x, vv.x), g01, sab[a][2 * q]);
            sab[a][2 * q + 1] = fma2(mk2(vv.y, vv.y), g01, sab[a][2 * q + 1]);
            sc[a][q] = fma2(vv, gmm, sc[a][q]);
            if (a == R && 2 * q <= R && R <= 2 * q + 1) {
              const float c0 = (R & 1) ? vv.y : vv.x;
              xc = par ? xc : c0;
            }
            if (a == R && 2 * q <= R + 1 && R + 1 <= 2 * q + 1) {
              const float c1 = ((R + 1) & 1) ? vv.y : vv.x;
              xc = par ? c1 : xc;
            }
          }
        }
        const float e = g_0 * xc;
        gc += e;
        if (d == 0) e_lo = e;
        if (d == geo.D - 1) e_hi = e;
        g_m = g_0;
        g_0 = g_p;
      }
    }
    lga_stage_commit<R>(tile[(c + 1) & 1], geo, stg, (c + 1) * LGA_PB, regs);
    GA_LDS_BARRIER();
  }

  if (inb) {
#pragma unroll
    for (int dd = 0; dd < 3; dd++) {
      // accumulate mode: the K old values of this depth slab are loaded together, then added and stored
      // (written as `*dst = accumulate ? *dst + r : r` every tap is load -> s_waitcnt vmcnt(0) -> store:
      // 75 serial round trips per lane, +30 us on the second pass of an LGA2 backward)
      float old[C::K];
#pragma unroll
      for (int k = 0; k < C::K; k++) old[k] = 0.f;
      if (accumulate) {
#pragma unroll
        for (int k = 0; k < C::K; k++) old[k] = gfb[(i64)(dd * C::K + k) * geo.HW + pix];
      }
#pragma unroll
      for (int a = -R; a <= R; a++) {
#pragma unroll
        for (int bb = -R; bb <= R; bb++) {
          const int t = dd * C::K + (a + R) * C::WS + (bb + R);
          const int i2 = i + a, j2 = j + bb;
          const bool ok = i2 >= 0 && i2 < geo.H && j2 >= 0 && j2 < geo.W;
          // tap bb lives in window slot k = (bb + R) + par
          const int ke = bb + R, ko = bb + R + 1, ra = a + R;
          float re, ro;
          if (dd == 0) { re = sab[ra][ke].x; ro = sab[ra][ko].x; }
          else if (dd == 1) { re = sab[ra][ke].y; ro = sab[ra][ko].y; }
          else {
            re = (ke & 1) ? sc[ra][ke >> 1].y : sc[ra][ke >> 1].x;
            ro = (ko & 1) ? sc[ra][ko >> 1].y : sc[ra][ko >> 1].x;
          }
          float r = par ? ro : re;
          if (dd == 0) r += e_lo;
          if (dd == 2) r += e_hi;
          if (!ok) r = gc;
          gfb[(i64)t * geo.HW + pix] = old[(a + R) * C::WS + (bb + R)] + r;
        }
      }
    }
  }
}

}  // namespace ga
