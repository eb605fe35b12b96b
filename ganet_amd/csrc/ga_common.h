// ga_common.h -- shared device helpers for the gfx950 guided-aggregation kernels.
//
// Built two ways from the SAME kernel source:
//   * hipcc --offload-arch=gfx950            -> libganet_hip.so (the product)
//   * g++ -DGA_HIPSIM -include hipsim.h       -> tests/hipsim/libganet_sim.so, a
//     lockstep wave64 emulator used ONLY by the CPU test-suite to check kernel
//     logic (indexing, DPP lane patterns, reductions) where no GPU exists.
// Wavefront = 64 lanes everywhere; cross-lane traffic is DPP within 16-lane rows.
#pragma once

#if defined(GA_HIPSIM)
#include "hipsim.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdint.h>

#define GA_DEV __device__ __forceinline__

namespace ga {

typedef long long i64;

#if defined(GA_HIPSIM)
struct alignas(16) f4 { float x, y, z, w; };
#else
typedef float4 f4;
#endif

// dynamic LDS: one 16-byte aligned region per workgroup (cdna_hip_programming.md G17)
#if defined(GA_HIPSIM)
#define GA_DYN_SMEM(name) float *name = reinterpret_cast<float *>(hipsim::S().dyn_smem)
#else
extern __shared__ __attribute__((aligned(16))) float ga_dyn_smem_raw[];
#define GA_DYN_SMEM(name) float *name = ga::ga_dyn_smem_raw
#endif

// packed fp32 pair: v_pk_fma_f32 does two FMAs per lane per issue on gfx950 (the 157 TF fp32
// vector peak assumes it); hipcc emits it for ext_vector_type(2) elementwise fma.
#if defined(GA_HIPSIM)
struct f2 { float x, y; };
GA_DEV f2 fma2(f2 a, f2 b, f2 c) { f2 r; r.x = fmaf(a.x, b.x, c.x); r.y = fmaf(a.y, b.y, c.y); return r; }
GA_DEV f2 add2(f2 a, f2 b) { f2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
#else
typedef float f2 __attribute__((ext_vector_type(2)));
GA_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
GA_DEV f2 add2(f2 a, f2 b) { return a + b; }
#endif
GA_DEV f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
// optimisation fence on a packed value: it has to be complete HERE (no sinking into a later branch)
#if defined(GA_HIPSIM)
#define GA_KEEP_F2(v) ((void)0)
#define GA_OPAQUE_S(v) ((void)0)
#define GA_OPAQUE_V(v) ((void)0)
#define GA_OPAQUE_VF(v) ((void)0)
#define GA_SCHED_FENCE() ((void)0)
#else
#define GA_KEEP_F2(v) asm volatile("" : "+v"(v))
#define GA_OPAQUE_S(v) asm volatile("" : "+s"(v))   // uniform value the optimiser may not reason about
#define GA_OPAQUE_V(v) asm volatile("" : "+v"(v))   // the same for a per-lane value
#define GA_OPAQUE_VF(v) asm("" : "+v"(v))          // ... without `volatile`: only hides where the value came from (may be moved, dropped if unused)
// nothing is scheduled across this point: used to pin a hand-chosen instruction interleaving
#define GA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// one 8-byte LDS read that stays a ds_read_b64 (256 B/clk/CU): left alone, the compiler fuses
// neighbouring pairs into ds_read2_b64, which the LDS serves at half that rate
// (MI355X_MICROARCH.md, LDS table).  volatile = "do not merge"; it adds no waits.  The pointer
// must carry the LDS address space explicitly (a volatile access through a generic pointer is
// not re-inferred and would become a flat_load).
#if defined(GA_HIPSIM)
typedef const float *lds_cptr;
#define GA_LDS_CPTR(p) (p)
GA_DEV f2 lds_read_b64(lds_cptr p) { return mk2(p[0], p[1]); }
#else
typedef const __attribute__((address_space(3))) float *lds_cptr;
#define GA_LDS_CPTR(p) ((ga::lds_cptr)(p))
GA_DEV f2 lds_read_b64(lds_cptr p)
{
  return *(const volatile __attribute__((address_space(3))) f2 *)p;
}
#endif

// Cache policy.  Result stores of the streaming kernels -- directional / adjoint volumes, merged output and mask, the input
// gradient, the final outputs of an LGA2 chain: written once, read much later or not at all -- are NON-TEMPORAL
// (`global_store ... nt`), and so are the loads of the scans' inputs and of the directional volumes in the merge: the data
// streams through the L2 instead of displacing what the kernels around it re-read (the filter taps are read by every LGA pass,
// an interleaved intermediate by the very next kernel, a tile's halo by the neighbouring tiles).
// Measured with the whole step captured into a hipGraph, all variants on the same buffers (scripts/ab_step.py: +-0.2 %;
// profiles/r3l_*, r3m_*, r4a_* ... r4f_*): stores -3.0 % (nt0 -> 199 on that box; -4.1 ... -4.9 % on earlier ones), loads 5
// instead of none -2.2 %, the input gradient's store a further -0.3 %.  NOT: the per-pixel gradient kernel's loads (+0.7 %:
// its forward volumes are read at two pixel offsets), every LGA output (+0.8 ... +1.4 %: the interleaved intermediate is
// re-read at once), the LGA tap gather (+3 %), the LGA kernels' LDS-DMA copies (+6 ... +10 %).
// Bit masks for A/B builds (scripts/build_variants.py, -DGA_NT_STORES=n -DGA_NT_LOADS=n):
//   stores: 1 column scans, 2 row scans, 4 merge (output volume), 128 merge (direction mask), 8 per-pixel gradients (gradX),
//           16 LGA apply, 32 LGA filter gradient, 64 LGA apply from an interleaved input only (the final outputs of an LGA2 chain)
//   loads:  1 merge, 2 per-pixel gradients' G / A, 4 scan inputs, 8 per-pixel gradients' x, 16 LGA filter taps
#ifndef GA_NT_STORES
#define GA_NT_STORES 207
#endif
#ifndef GA_NT_LOADS
#define GA_NT_LOADS 5
#endif
typedef float ga_f4v __attribute__((ext_vector_type(4)));
template <bool ON, typename T> GA_DEV void stream_store(T *p, const T &v)
{
#if !defined(GA_HIPSIM)
  if constexpr (ON) { __builtin_nontemporal_store(v, p); return; }
#endif
  *p = v;
}
template <bool ON> GA_DEV void stream_store(f4 *p, const f4 &v)      // (the builtin wants a native vector type, not HIP's float4 class)
{
#if !defined(GA_HIPSIM)
  if constexpr (ON) {
    ga_f4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<ga_f4v *>(p));
    return;
  }
#endif
  *p = v;
}

template <bool ON, typename T> GA_DEV T stream_load(const T *p)
{
#if !defined(GA_HIPSIM)
  if constexpr (ON) return __builtin_nontemporal_load(p);
#endif
  return *p;
}
template <bool ON> GA_DEV f4 stream_load(const f4 *p)
{
#if !defined(GA_HIPSIM)
  if constexpr (ON) {
    const ga_f4v t = __builtin_nontemporal_load(reinterpret_cast<const ga_f4v *>(p));
    f4 o; o.x = t.x; o.y = t.y; o.z = t.z; o.w = t.w;
    return o;
  }
#endif
  return *p;
}

GA_DEV float f4_get(const f4 &v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
GA_DEV void f4_set(f4 &v, int k, float a)
{
  if (k == 0) v.x = a; else if (k == 1) v.y = a; else if (k == 2) v.z = a; else v.w = a;
}

GA_DEV int lane_id()
{
#if defined(GA_HIPSIM)
  return hipsim::lane_id();
#else
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#endif
}

// a value the program knows to be the same in every lane of the wavefront, told to the compiler (v_readfirstlane -> an SGPR)
#if defined(GA_HIPSIM)
#define GA_UNIFORM_I(v) (v)
#else
#define GA_UNIFORM_I(v) __builtin_amdgcn_readfirstlane(v)
#endif

// Workgroup barrier for hand-offs that go through LDS only.  __syncthreads() is a fence + barrier, and
// the fence drains the vector-memory counter as well: every global load still in flight (the prefetch
// of the next tile) and every result store is waited for at each barrier.  Here only the LDS queue
// is drained (cdna_hip_programming.md, "raw s_barrier + lgkmcnt(0) only").
#if defined(GA_HIPSIM)
#define GA_LDS_BARRIER() __syncthreads()
#else
#define GA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// hand-off between the lanes of ONE wavefront through LDS (kernels whose workgroup is a single wave)
#if defined(GA_HIPSIM)
#define GA_WAVE_SYNC() hipsim::wave_sync()     // emulator: a barrier over the caller's own wavefront
#else
// no instruction: the lanes of a wave run in lockstep and its LDS queue is in order; this only
// stops the compiler from moving LDS accesses across the hand-off
#define GA_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// ---- DPP (data-parallel primitives) ---------------------------------------
// dpp_ctrl encodings (LLVM SIDefines.h DppCtrl): quad_perm 0x00-0xFF,
// row_shl:n 0x100+n, row_shr:n 0x110+n, row_mirror 0x140, row_half_mirror 0x141.
// row_shr:1 -> lane i receives lane i-1 of its 16-lane row; row_shl:1 -> lane i+1.
// A lane whose source falls outside the row keeps `old` (bound_ctrl = false).
enum : int {
  DPP_QP_XOR1 = 0xB1,   // quad_perm [1,0,3,2]
  DPP_QP_XOR2 = 0x4E,   // quad_perm [2,3,0,1]
  DPP_QP_BCAST0 = 0x00, // quad_perm [0,0,0,0]: every lane of a quad receives lane 0's value
  DPP_QP_BCAST1 = 0x55,
  DPP_QP_BCAST2 = 0xAA,
  DPP_QP_BCAST3 = 0xFF,
  DPP_ROW_SHL1 = 0x101,
  DPP_ROW_SHR1 = 0x111,
  DPP_ROW_MIRROR = 0x140,
  DPP_ROW_HALF_MIRROR = 0x141,
  // whole-wavefront shifts (gfx9 family incl. gfx950; gone in gfx10+): lane i <- lane i+1 / i-1 across the 16-lane rows;
  // the first / last lane of the WAVE has no source and keeps `old`.  Used by the 64-lanes-per-scanline scans.
  DPP_WAVE_SHL1 = 0x130,
  DPP_WAVE_SHR1 = 0x138
};

#if !defined(GA_HIPSIM) && defined(GA_NO_DPP)
// Debug variant (libganet_hip_nodpp.so): same lane patterns through ds_bpermute.
template <int CTRL> GA_DEV int dpp_src_lane(int lane, bool &valid)
{
  const int row = lane & ~15, r = lane & 15;
  valid = true;
  if (CTRL >= 0 && CTRL <= 0xFF) return (lane & ~3) | ((CTRL >> (2 * (lane & 3))) & 3);      // any quad_perm
  if (CTRL == DPP_ROW_SHL1) { valid = r < 15; return lane + 1; }
  if (CTRL == DPP_ROW_SHR1) { valid = r > 0; return lane - 1; }
  if (CTRL == DPP_ROW_MIRROR) return row + (15 - r);
  if (CTRL == DPP_WAVE_SHL1) { valid = lane < 63; return lane + 1; }
  if (CTRL == DPP_WAVE_SHR1) { valid = lane > 0; return lane - 1; }
  /* DPP_ROW_HALF_MIRROR */ return (lane & ~7) + (7 - (lane & 7));
}
template <int CTRL> GA_DEV int dpp_i(int old, int src)
{
  bool valid;
  const int sl = dpp_src_lane<CTRL>(lane_id(), valid);
  const int v = __builtin_amdgcn_ds_bpermute((valid ? sl : 0) << 2, src);
  return valid ? v : old;
}
#elif !defined(GA_HIPSIM)
template <int CTRL> GA_DEV int dpp_i(int old, int src)
{
  return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false);
}
GA_DEV int f2i_(float f) { return __builtin_bit_cast(int, f); }
GA_DEV float i2f_(int i) { return __builtin_bit_cast(float, i); }
#else
template <int CTRL> GA_DEV int dpp_i(int old, int src) { return hipsim::update_dpp(old, src, CTRL); }
#endif

// shift patterns whose source-less lanes read ZERO (bound_ctrl): with no `old` operand the compiler folds the move into the
// consumer (v_fmac_f32_dpp) instead of emitting v_mov_b32 0 + v_mov_b32_dpp + op
#if !defined(GA_HIPSIM) && !defined(GA_NO_DPP)
template <int CTRL> GA_DEV float dpp_zero_f(float src) { return i2f_(__builtin_amdgcn_mov_dpp(f2i_(src), CTRL, 0xF, 0xF, true)); }
#else
template <int CTRL> GA_DEV float dpp_zero_f(float src) { union { float f; int i; } a, b; a.f = src; b.i = dpp_i<CTRL>(0, a.i); return b.f; }
#endif

// full-permutation patterns (xor / mirror): every lane has an in-row source, so `old` is
// irrelevant; the undef-old form lets the compiler fold the move into the consumer
// (v_max_f32_dpp / v_add_f32_dpp) instead of emitting v_mov_b32_dpp + op.
#if !defined(GA_HIPSIM) && !defined(GA_NO_DPP)
template <int CTRL> GA_DEV int dpp_perm_i(int src) { return __builtin_amdgcn_mov_dpp(src, CTRL, 0xF, 0xF, true); }
#else
template <int CTRL> GA_DEV int dpp_perm_i(int src) { return dpp_i<CTRL>(src, src); }
#endif

GA_DEV int f2i(float f) { union { float f; int i; } u; u.f = f; return u.i; }
GA_DEV float i2f(int i) { union { float f; int i; } u; u.i = i; return u.f; }
template <int CTRL> GA_DEV float dpp_f(float old, float src) { return i2f(dpp_i<CTRL>(f2i(old), f2i(src))); }
template <int CTRL> GA_DEV float dpp_perm_f(float src) { return i2f(dpp_perm_i<CTRL>(f2i(src))); }

// value held by lane `l` of the wavefront, as a wave-uniform scalar
GA_DEV float wave_readlane_f(float v, int l)
{
#if defined(GA_HIPSIM)
  return i2f(hipsim::readlane(f2i(v), l));
#else
  return i2f(__builtin_amdgcn_readlane(f2i(v), l));
#endif
}

// ---- segment ops: a "segment" is GD consecutive lanes (GD in 1,2,4,8,16, or the whole wave: 64) that
// together own one scanline; lg = lane % GD.
// value held by the previous / next lane of the segment; segment ends keep `old`
template <int GD> GA_DEV float seg_from_prev(float old, float src, int lg)
{
  if (GD == 1) return old;
  if (GD == 64) return dpp_f<DPP_WAVE_SHR1>(old, src);     // the wave IS the segment: lane 0 has no source and keeps `old`
  const float r = dpp_f<DPP_ROW_SHR1>(old, src);
  if (GD == 16) return r;          // a 16-lane segment IS a DPP row: its lane 0 has no source and keeps `old`
  return lg == 0 ? old : r;
}
template <int GD> GA_DEV float seg_from_next(float old, float src, int lg)
{
  if (GD == 1) return old;
  if (GD == 64) return dpp_f<DPP_WAVE_SHL1>(old, src);
  const float r = dpp_f<DPP_ROW_SHL1>(old, src);
  if (GD == 16) return r;
  return lg == GD - 1 ? old : r;
}
// the same with 0 for the segment's first / last lane (the adjoint scans)
template <int GD> GA_DEV float seg_from_prev0(float src, int lg)
{
  if (GD == 64) return dpp_zero_f<DPP_WAVE_SHR1>(src);
  if (GD == 16) return dpp_zero_f<DPP_ROW_SHR1>(src);
  return seg_from_prev<GD>(0.f, src, lg);
}
template <int GD> GA_DEV float seg_from_next0(float src, int lg)
{
  if (GD == 64) return dpp_zero_f<DPP_WAVE_SHL1>(src);
  if (GD == 16) return dpp_zero_f<DPP_ROW_SHL1>(src);
  return seg_from_next<GD>(0.f, src, lg);
}
// max of two values that are known not to be signalling NaNs (results of arithmetic): fmaxf()
// makes hipcc canonicalise each operand first (an extra v_max_f32 x, x per element)
GA_DEV float vmax_raw(float a, float b)
{
#if defined(GA_HIPSIM)
  return fmaxf(a, b);
#else
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#endif
}
// all-lanes max over the segment.  On the GPU this is written out as v_max_f32_dpp: from
// fmaxf(v, dpp(v)) hipcc makes v_mov_b32_dpp + v_max_f32 x, x (quieting a possible signalling NaN of
// the moved bits) + v_max_f32, three instructions per butterfly step on the serial path of every scan
// position.  (s_nop 1: a DPP read needs two wait states after the VALU write of its source.)
template <int GD> GA_DEV float seg_allmax(float v)
{
#if defined(GA_HIPSIM) || defined(GA_NO_DPP)
  if (GD >= 2) v = fmaxf(v, dpp_perm_f<DPP_QP_XOR1>(v));
  if (GD >= 4) v = fmaxf(v, dpp_perm_f<DPP_QP_XOR2>(v));
  if (GD >= 8) v = fmaxf(v, dpp_perm_f<DPP_ROW_HALF_MIRROR>(v));
  if (GD >= 16) v = fmaxf(v, dpp_perm_f<DPP_ROW_MIRROR>(v));
#else
  if (GD >= 2) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
  if (GD >= 4) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
  if (GD >= 8) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
  if (GD >= 16) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
#endif
  if (GD == 64) {     // every lane of a row now holds the row's maximum
#if defined(GA_HIPSIM) || defined(GA_NO_DPP)
    const float r0 = wave_readlane_f(v, 0), r1 = wave_readlane_f(v, 16), r2 = wave_readlane_f(v, 32), r3 = wave_readlane_f(v, 48);
    v = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
#else
    // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3, one read of lane 63 (see seg_allsum)
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    v = wave_readlane_f(v, 63);
#endif
  }
  return v;
}
template <int GD> GA_DEV float seg_allsum(float v)
{
  if (GD >= 2) v += dpp_perm_f<DPP_QP_XOR1>(v);
  if (GD >= 4) v += dpp_perm_f<DPP_QP_XOR2>(v);
  if (GD >= 8) v += dpp_perm_f<DPP_ROW_HALF_MIRROR>(v);
  if (GD >= 16) v += dpp_perm_f<DPP_ROW_MIRROR>(v);
  if (GD == 64) {     // every lane of a row holds the row's sum: (r0 + r1) + (r2 + r3)
#if defined(GA_HIPSIM) || defined(GA_NO_DPP)
    const float r0 = wave_readlane_f(v, 0), r1 = wave_readlane_f(v, 16), r2 = wave_readlane_f(v, 32), r3 = wave_readlane_f(v, 48);
    v = (r0 + r1) + (r2 + r3);
#else
    // row_bcast:15 into rows 1 and 3 (r1 + r0, r3 + r2), row_bcast:31 into rows 2 and 3 (row 3: (r3 + r2) + (r0 + r1)), one read
    // of lane 63: two DPP adds and one v_readlane instead of four v_readlane and three adds
    // (written out: from the builtin hipcc makes v_mov_b32 0 + v_mov_b32_dpp + v_add_f32 per step)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    v = wave_readlane_f(v, 63);
#endif
  }
  return v;
}
// (max value, smallest index attaining it) over the segment: the reference's
// strict-'<' first-argmax (GANet_kernel.cu:60-62, 122-123)
template <int CTRL> GA_DEV void argmax_merge(float &v, int &k)
{
  const float ov = dpp_perm_f<CTRL>(v);
  const int ok = dpp_perm_i<CTRL>(k);
  const bool take = (ov > v) || (ov == v && ok < k);
  v = take ? ov : v;
  k = take ? ok : k;
}
template <int GD> GA_DEV void seg_argmax(float &v, int &k)
{
  if (GD >= 2) argmax_merge<DPP_QP_XOR1>(v, k);
  if (GD >= 4) argmax_merge<DPP_QP_XOR2>(v, k);
  if (GD >= 8) argmax_merge<DPP_ROW_HALF_MIRROR>(v, k);
  if (GD >= 16) argmax_merge<DPP_ROW_MIRROR>(v, k);
}

template <int GD> GA_DEV int seg_allmin_i(int v)
{
  int o;
  if (GD >= 2) { o = dpp_perm_i<DPP_QP_XOR1>(v); v = o < v ? o : v; }
  if (GD >= 4) { o = dpp_perm_i<DPP_QP_XOR2>(v); v = o < v ? o : v; }
  if (GD >= 8) { o = dpp_perm_i<DPP_ROW_HALF_MIRROR>(v); v = o < v ? o : v; }
  if (GD >= 16) { o = dpp_perm_i<DPP_ROW_MIRROR>(v); v = o < v ? o : v; }
  return v;
}

// MI355X: workgroup b runs on XCD b % 8 (observed, speed only).  Give each XCD a
// contiguous range of logical blocks so neighbouring tiles share one L2.
GA_DEV int xcd_remap(int b, int nb)
{
  const int per = nb >> 3;
  if (per == 0 || b >= (per << 3)) return b;
  return (b & 7) * per + (b >> 3);
}

}  // namespace ga
