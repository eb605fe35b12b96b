// Micro-benchmark: ds_read_b64 at 8-byte aligned addresses vs at addresses that are 4 mod 8 (gfx950): correctness and rate.
// (Would let the LGA kernels read exact 5-wide windows instead of 6-wide aligned ones with a zero-weight dummy slot.)
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_unaligned.hip -o scripts/ubench/lds_unaligned.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OFF>   // byte offset added to every lane's address: 0 (aligned) or 4
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
  __shared__ __attribute__((aligned(16))) float buf[4096 + 64];
  for (int i = threadIdx.x; i < 4096 + 64; i += 256) buf[i] = (float)i;
  __syncthreads();
  const __attribute__((address_space(3))) char *base = (const __attribute__((address_space(3))) char *)buf;
  f2 acc = {0.f, 0.f};
  unsigned a = (threadIdx.x % 64) * 8 + OFF + (threadIdx.x / 64) * 1024;      // consecutive lanes, 8 B apart
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      f2 v;
      asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(u * 512));
      acc += v;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + 2.f * acc.y;
}

template <int OFF> float run(const char *name, float *ref)
{
  float *out; hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OFF><<<256 * 16, 256>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OFF><<<256 * 16, 256>>>(out, 2000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  // expected for lane l (first wave), one iteration set: sum over u of (idx, idx+1) with idx = (l*8 + OFF + u*512)/4
  double ex = 0, ey = 0;
  for (int u = 0; u < 16; u++) { const int idx = (5 * 8 + OFF + u * 512) / 4; ex += idx; ey += idx + 1; }
  const double want = 2000.0 * (ex + 2 * ey);
  printf("%-22s %8.3f ms   lane 5: got %.0f want %.0f  (%s)\n", name, ms, h[5], want, hipGetErrorString(hipGetLastError()));
  *ref = ms; hipFree(out); return ms;
}

int main()
{
  float a, b;
  run<0>("ds_read_b64 aligned", &a);
  run<4>("ds_read_b64 at 4 mod 8", &b);
  printf("ratio unaligned/aligned = %.2f\n", b / a);
  return 0;
}
