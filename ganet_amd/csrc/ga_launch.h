// ga_launch.h -- what the translation units of libganet_hip.so share on the host side: the launch
// macros (HIP / CPU emulator), the row-kernel tiling constants and the launchers that live in their
// own translation unit.
#pragma once
#include "ga_common.h"
#include "sga_row_kernels.h"
#include <stdio.h>

#if defined(GA_HIPSIM)
#define GA_LAUNCH(kern, grid, block, stream, ...) \
  hipsim::launch((grid), (block), 0, [=]() { kern(__VA_ARGS__); })
#define GA_LAUNCH_SMEM(kern, grid, block, smem, stream, ...) \
  hipsim::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define GA_LAUNCH_SMEM_BIG GA_LAUNCH_SMEM
#define GA_EXPORT extern "C"
#else
// hipGetLastError() is per-thread state shared with the host framework.  An error some earlier, unrelated HIP call left
// there must not be blamed on THIS launch -- and must not vanish either: ga_note_stale_error() reports it on stderr
// (once per thread and error code) before the state is cleared.
inline void ga_note_stale_error()
{
  if (hipPeekAtLastError() == hipSuccess) return;
  const hipError_t e = hipGetLastError();
  static thread_local hipError_t last_reported = hipSuccess;
  if (e != last_reported) {
    last_reported = e;
    fprintf(stderr, "libganet_hip: HIP error \"%s\" was already pending on this thread before a ganet launch "
                    "(left by an earlier call of the host framework); cleared\n", hipGetErrorString(e));
  }
}
#define GA_LAUNCH_SMEM(kern, grid, block, smem, stream, ...) \
  do { ga_note_stale_error(); hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__); } while (0)
#define GA_LAUNCH(kern, grid, block, stream, ...) \
  do { ga_note_stale_error(); hipLaunchKernelGGL(kern, (grid), (block), 0, (stream), __VA_ARGS__); } while (0)
// more than 64 KB of dynamic LDS (gfx950 has 160 KB per CU): the limit of the kernel has to be raised first
#define GA_LAUNCH_SMEM_BIG(kern, grid, block, smem, stream, ...)                                                         \
  do {                                                                                                                   \
    ga_note_stale_error();                                                                                               \
    if ((smem) > 64 * 1024)                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem)); \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__);                                            \
  } while (0)
#define GA_EXPORT extern "C" __attribute__((visibility("default")))
#endif

namespace ga {

// rows per wavefront (LN) x positions per staged batch (SBH): LDS per wave bounds residency
#ifndef GA_ROW_LN_F
#define GA_ROW_LN_F 1
#endif
constexpr int ROW_SBH_F = 32, ROW_PAD_F = 4, ROW_LN_F = GA_ROW_LN_F;   // forward: the A tile overwrites the x tile (9.8 KB per row at D=65)
constexpr size_t ROW_SMEM_MAX = 64 * 1024;

inline size_t row_smem_fwd(int D)
{
  return sizeof(float) * ROW_LN_F * ((size_t)D * RowCfg<ROW_SBH_F, ROW_PAD_F>::RS + 5 * ROW_SBH_F);
}

// disparities per lane of the single DPP row that carries the recurrence (D <= 16 * DPL)
#define GA_ROW_DPLS(X) X(1) X(2) X(3) X(5) X(9) X(13)

inline int row_dpl(int D)
{
  int best = 0;
#define X(P) if (best == 0 && 16 * (P) >= D) best = (P);
  GA_ROW_DPLS(X)
#undef X
  return best;
}

// adjoint scan: 1 tile + mask per row
#ifndef GA_ROW_LN_B
#define GA_ROW_LN_B 1
#endif
constexpr int ROW_SBH_B = 32, ROW_PAD_B = 4, ROW_LN_B = GA_ROW_LN_B;
inline size_t row_smem_bwdg(int D)
{
  return sizeof(float) * ROW_LN_B * ((size_t)(D + 1) * RowCfg<ROW_SBH_B, ROW_PAD_B>::RS + 5 * ROW_SBH_B + ROW_SBH_B / 2);
}

// sga_row_tu.hip: horizontal scans (direction 2 = right, 3 = left), forward and adjoint; the caller checks the launch
void launch_row_bwdg(const float *g, const uint8_t *mask, const uint16_t *kp, const float *gout, float *G,
                     int S, int D, int H, int W, int dir, hipStream_t st);
void launch_row_fwd(const float *x, const float *g, float *A, int S, int D, int H, int W, int dir, hipStream_t st,
                    int out_mode = 0, int C = 1, const float *scale = nullptr, const float *shift = nullptr);

}  // namespace ga
