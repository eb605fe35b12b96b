// Micro-benchmark: what does one extra instruction of each kind cost a wave that is streaming
// v_pk_fma_f32?  (per-wave issue interval on gfx950; decides what to trim from a low-occupancy kernel)
// build: hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/issue_mix.hip -o scripts/ubench/issue_mix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define FMA8()                                                                        \
  _Pragma("unroll") for (int c = 0; c < 8; c++)                                       \
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(aa), "v"(bb));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
  extern __shared__ float pad[];
  if (iters < 0) out[0] = pad[threadIdx.x];
  f2 acc[8];
#pragma unroll
  for (int c = 0; c < 8; c++) { acc[c].x = threadIdx.x + c; acc[c].y = c; }
  f2 aa; aa.x = a; aa.y = b;
  f2 bb; bb.x = b; bb.y = a;
  int sreg = 0;
  float vx = a;
  for (int i = 0; i < iters; i++) {
    // 4 groups of 8 packed FMAs per trip, so the loop's own s_add/s_cmp/s_cbranch is ~10 %
#pragma unroll
    for (int g = 0; g < 4; g++) {
      FMA8();
      if (MODE == 1) { asm volatile("s_waitcnt lgkmcnt(0)\ns_waitcnt lgkmcnt(0)\ns_waitcnt lgkmcnt(0)\ns_waitcnt lgkmcnt(0)\ns_waitcnt lgkmcnt(0)\ns_waitcnt lgkmcnt(0)\ns_waitcnt lgkmcnt(0)\ns_waitcnt lgkmcnt(0)"); }
      if (MODE == 2) { asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0"); }
      if (MODE == 3) { asm volatile("s_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1" : "+s"(sreg)); }
      if (MODE == 4) { asm volatile("v_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0" : "+v"(vx)); }
      if (MODE == 5) { asm volatile("ds_read_b64 %0, %1\nds_read_b64 %0, %1\nds_read_b64 %0, %1\nds_read_b64 %0, %1\nds_read_b64 %0, %1\nds_read_b64 %0, %1\nds_read_b64 %0, %1\nds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(bb) : "v"((threadIdx.x & 63) * 8)); }
      if (MODE == 6) { asm volatile("ds_read2_b64 %0, %2 offset1:1\nds_read2_b64 %1, %2 offset0:2 offset1:3\nds_read2_b64 %0, %2 offset1:1\nds_read2_b64 %1, %2 offset0:2 offset1:3\ns_waitcnt lgkmcnt(0)" : "=v"(*(float __attribute__((ext_vector_type(4))) *)&acc[6]), "=v"(*(float __attribute__((ext_vector_type(4))) *)&acc[4]) : "v"((threadIdx.x & 63) * 8)); }
    }
  }
  float s = sreg + vx;
#pragma unroll
  for (int c = 0; c < 8; c++) s += acc[c].x + acc[c].y;
  out[(blockIdx.x % 2048) * 256 + threadIdx.x] = s + bb.x;
}

template <int MODE> void run(const char *name, int bpc)
{
  float *out; hipMalloc(&out, 2048 * 256 * sizeof(float));
  const int iters = 1000, grid = 256 * bpc * 16;
  const size_t lds = (size_t)(160 * 1024 / bpc) - 1024;
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256, lds>>>(out, 50, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 256, lds>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double groups = (double)iters * 4 * bpc * 16;          // groups per SIMD
  printf("%-34s waves/SIMD=%d  %.1f clk per group (8 pk_fma + extras) @2.4GHz\n", name, bpc, ms * 1e6 * 2.4 / groups);
  hipFree(out);
}

int main()
{
  for (int bpc : {1, 3}) {
    if (bpc == 1) {
      run<0>("8 pk_fma", 1); run<1>("+ 8 s_waitcnt(ready)", 1); run<2>("+ 8 s_nop", 1); run<3>("+ 8 s_add_u32", 1);
      run<4>("+ 8 v_mov_b32", 1); run<5>("+ 8 ds_read_b64 + wait", 1); run<6>("+ 4 ds_read2_b64 + wait", 1);
    } else {
      run<0>("8 pk_fma", 3); run<1>("+ 8 s_waitcnt(ready)", 3); run<2>("+ 8 s_nop", 3); run<3>("+ 8 s_add_u32", 3);
      run<4>("+ 8 v_mov_b32", 3); run<5>("+ 8 ds_read_b64 + wait", 3); run<6>("+ 4 ds_read2_b64 + wait", 3);
    }
  }
  return 0;
}
