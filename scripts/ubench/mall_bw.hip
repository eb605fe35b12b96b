// Micro-benchmark: how fast is a read that hits the Infinity Cache (MALL, 256 MiB) compared with one that goes to HBM?
//   hot read    : the same S-byte buffer summed again and again (fits the cache or not)
//   write->read : kernel A fills S bytes, kernel B reads them back (producer -> consumer through the memory-side cache)
//   copy        : S bytes read + S bytes written
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/mall_bw.hip -o scripts/ubench/mall_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef long long i64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_sum(const float4 *a, float *out, i64 n4)
{
  float s = 0.f;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_fill(float4 *a, i64 n4, float v)
{
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) a[i] = make_float4(v, v, v, v);
}
__global__ void __launch_bounds__(256) k_copy(const float4 *a, float4 *b, i64 n4)
{
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) b[i] = a[i];
}

int main()
{
  const i64 MAXB = 2048ll << 20;
  float4 *a, *b; float *out;
  CK(hipMalloc(&a, MAXB)); CK(hipMalloc(&b, MAXB)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, MAXB)); CK(hipMemset(b, 0, MAXB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  printf("size_MB  hot_read_TBs  write_then_read_TBs(read part)  fill_TBs  copy_TBs(R+W)\n");
  const int sizes[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048};
  for (int mb : sizes) {
    const i64 bytes = (i64)mb << 20, n4 = bytes / 16;
    const int reps = mb <= 256 ? 40 : 10;
    float ms;
    // hot read
    k_sum<<<grid, 256>>>(a, out, n4);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) k_sum<<<grid, 256>>>(a, out, n4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double hot = bytes * (double)reps / (ms * 1e-3) / 1e12;
    // fill alone
    k_fill<<<grid, 256>>>(b, n4, 1.f);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) k_fill<<<grid, 256>>>(b, n4, (float)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double fill_ms = ms / reps;
    // fill + read back
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) { k_fill<<<grid, 256>>>(b, n4, (float)r); k_sum<<<grid, 256>>>(b, out, n4); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double rb_ms = ms / reps - fill_ms;
    // copy
    k_copy<<<grid, 256>>>(a, b, n4);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) k_copy<<<grid, 256>>>(a, b, n4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double cp = 2.0 * bytes * reps / (ms * 1e-3) / 1e12;
    printf("%6d  %10.2f  %10.2f  %10.2f  %10.2f\n", mb, hot, bytes / (rb_ms * 1e-3) / 1e12, bytes / (fill_ms * 1e-3) / 1e12, cp);
    fflush(stdout);
  }
  return 0;
}
