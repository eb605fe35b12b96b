// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (which one is the fp32 VALU peak?)
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int CH>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
  extern __shared__ float occupancy_pad[];      // dynamic LDS sized so that exactly `waves/SIMD` blocks fit a CU
  if (iters < 0) out[0] = occupancy_pad[threadIdx.x];
  if (MODE == 0) {
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += acc[c];
    out[(blockIdx.x % 2048) * 256 + threadIdx.x] = s;
  } else {
    f2 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { acc[c].x = threadIdx.x + c; acc[c].y = c; }
    f2 aa; aa.x = a; aa.y = b;
    f2 bb; bb.x = b; bb.y = a;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(aa), "v"(bb));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[c]) : "v"(aa), "v"(bb));
      }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += acc[c].x + acc[c].y;
    out[(blockIdx.x % 2048) * 256 + threadIdx.x] = s;
  }
}

template <int MODE, int CH> void run(const char *name, int blocks_per_cu)
{
  float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
  const int iters = 4000, grid = 256 * blocks_per_cu * 16;     // 16 rounds: placement and tail effects average out
  const size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 1024;
  hipFuncSetAttribute((const void *)k<MODE, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, CH><<<grid, 256, lds>>>(out, 100, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, CH><<<grid, 256, lds>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: blocks_per_cu waves, each iters*CH instructions
  const double inst = (double)iters * CH * blocks_per_cu * 16;
  const double ns_per = ms * 1e6 / inst;
  const double flops = (MODE == 0 ? 2.0 : 4.0) * 64 * inst * 1024 / (ms * 1e-3) / 1e12;
  printf("%-28s chains=%2d waves/SIMD=%d  %.3f ns/inst/SIMD (%.2f clk @2.4GHz)  %.1f TFLOP/s\n", name, CH, blocks_per_cu, ns_per, ns_per * 2.4, flops);
  hipFree(out);
}

int main()
{
  for (int rep = 0; rep < 2; rep++) {
    run<0, 8>("v_fma_f32", 1); run<0, 8>("v_fma_f32", 2); run<0, 8>("v_fma_f32", 3); run<0, 8>("v_fma_f32", 4); run<0, 8>("v_fma_f32", 8);
    run<0, 2>("v_fma_f32", 1); run<0, 2>("v_fma_f32", 3); run<0, 1>("v_fma_f32", 3); run<0, 4>("v_fma_f32", 3);
    run<1, 8>("v_pk_fma_f32", 1); run<1, 8>("v_pk_fma_f32", 2); run<1, 8>("v_pk_fma_f32", 3); run<1, 8>("v_pk_fma_f32", 4); run<1, 8>("v_pk_fma_f32", 8);
    run<1, 2>("v_pk_fma_f32", 1); run<1, 2>("v_pk_fma_f32", 3); run<1, 1>("v_pk_fma_f32", 3); run<1, 4>("v_pk_fma_f32", 3);
    run<2, 8>("v_pk_fma_f32 op_sel bcast", 1); run<2, 8>("v_pk_fma_f32 op_sel bcast", 3);
  }
  return 0;
}
