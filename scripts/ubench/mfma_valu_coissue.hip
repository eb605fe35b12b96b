// Micro-benchmark: do the f32 MFMA pipe (v_mfma_f32_4x4x1_16B_f32) and the f32 VALU pipe (v_pk_fma_f32) of one SIMD run
// concurrently on gfx950?  (An LGA kernel whose waves alternate between a VALU formulation and an MFMA formulation
// would then have up to twice the fp32 rate.)  Modes: all waves VALU, all waves MFMA, half/half (by wave parity on a SIMD).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_coissue.hip -o scripts/ubench/mfma_valu_coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 VALU only, 1 MFMA only, 2 mixed: blocks with even index VALU, odd MFMA
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
  const bool use_mfma = MODE == 1 || (MODE == 2 && (blockIdx.x & 1));
  if (!use_mfma) {
    f2 acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) { acc[c].x = threadIdx.x + c; acc[c].y = c; }
    f2 aa; aa.x = a; aa.y = b;
    f2 bb; bb.x = b; bb.y = a;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < 8; c++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(aa), "v"(bb));
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) s += acc[c].x + acc[c].y;
    out[(blockIdx.x % 2048) * 256 + threadIdx.x] = s;
  } else {
    f4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = f4{(float)threadIdx.x, (float)c, 0.f, 1.f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
    out[(blockIdx.x % 2048) * 256 + threadIdx.x] = s;
  }
}

template <int MODE> void run(const char *name, int blocks_per_cu)
{
  float *out; hipMalloc(&out, 2048 * 256 * sizeof(float));
  const int iters = 4000, grid = 256 * blocks_per_cu * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(out, 100, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // useful MACs: pk_fma 128 per wave-instruction, mfma 4x4x1 x16 blocks = 256 per wave-instruction
  const double waves = (double)grid * 4;
  double macs;
  if (MODE == 0) macs = waves * iters * 8 * 128.0;
  else if (MODE == 1) macs = waves * iters * 8 * 256.0;
  else macs = waves / 2 * iters * 8 * (128.0 + 256.0);
  printf("%-34s %8.3f ms  %7.1f TFLOP/s  (%s)\n", name, ms, 2 * macs / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
  hipFree(out);
}

int main()
{
  for (int bpc = 1; bpc <= 2; bpc++) {
    printf("blocks of 256 threads per CU: %d (waves per SIMD: %d)\n", bpc, bpc);
    run<0>("v_pk_fma_f32 only", bpc);
    run<1>("v_mfma_f32_4x4x1_16B_f32 only", bpc);
    run<2>("half the blocks each", bpc);
  }
  run<0>("v_pk_fma_f32 only, 8 blocks/CU", 8);
  run<1>("mfma only, 8 blocks/CU", 8);
  run<2>("half/half, 8 blocks/CU", 8);
  return 0;
}
