// Micro-benchmark: what HBM rate do "4 volumes in, 1 volume out" kernels reach on MI355X as a function of the
// ORDER in which the [S][D][HW] volumes are walked?  (sga_merge / sga_bwd_point march over d per pixel.)
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/stream_patterns.hip -o scripts/ubench/stream_patterns.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef long long i64;
constexpr int S = 32, D = 65, HW = 80 * 208;

__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }

// linear order, 16 B per lane, grid-stride
__global__ void __launch_bounds__(256) k_lin(const float4 *a0, const float4 *a1, const float4 *a2, const float4 *a3, float4 *o, i64 n4)
{
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256)
    o[i] = max4(max4(a0[i], a1[i]), max4(a2[i], a3[i]));
}
// one block per contiguous chunk (no grid-stride): chunk = 256 * 16 B * U
template <int U>
__global__ void __launch_bounds__(256) k_chunk(const float4 *a0, const float4 *a1, const float4 *a2, const float4 *a3, float4 *o, i64 n4)
{
  const i64 base = (i64)blockIdx.x * 256 * U + threadIdx.x;
  float4 v[U][4];
#pragma unroll
  for (int u = 0; u < U; u++) { const i64 i = base + u * 256; if (i < n4) { v[u][0] = a0[i]; v[u][1] = a1[i]; v[u][2] = a2[i]; v[u][3] = a3[i]; } }
#pragma unroll
  for (int u = 0; u < U; u++) { const i64 i = base + u * 256; if (i < n4) o[i] = max4(max4(v[u][0], v[u][1]), max4(v[u][2], v[u][3])); }
}
// march: a lane owns PX4 groups of 4 consecutive pixels (PX4 * 64 * 16 B contiguous per wave and plane), loops over d, DU planes in flight
template <int PX4, int DU, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_march(const float *a0, const float *a1, const float *a2, const float *a3, float *o)
{
  const i64 nq = (i64)S * HW / 4 / PX4;
  const i64 q = (i64)blockIdx.x * BLOCK + threadIdx.x;
  if (q >= nq) return;
  const int wave = (int)(q / 64), lane = (int)(q % 64);
  const i64 p0 = ((i64)wave * 64 * PX4 + lane) * 4;         // first pixel; group g adds 256 pixels
  const i64 s = p0 / HW, pix = p0 - s * HW;                  // (HW % (256 * PX4) != 0: a wave may straddle slices; fine for a rate probe when PX4 * 256 divides HW ... 16640 = 65 * 256)
  const i64 vb = s * D * HW + pix;
  for (int dc = 0; dc < D; dc += DU) {
    float4 v[DU][PX4][4];
#pragma unroll
    for (int u = 0; u < DU; u++) {
      const int d = dc + u < D ? dc + u : D - 1;
#pragma unroll
      for (int g = 0; g < PX4; g++) {
        const i64 off = vb + (i64)d * HW + g * 256;
        v[u][g][0] = *(const float4 *)(a0 + off); v[u][g][1] = *(const float4 *)(a1 + off);
        v[u][g][2] = *(const float4 *)(a2 + off); v[u][g][3] = *(const float4 *)(a3 + off);
      }
    }
#pragma unroll
    for (int u = 0; u < DU; u++) {
      const int d = dc + u;
      if (d < D) {
#pragma unroll
        for (int g = 0; g < PX4; g++)
          *(float4 *)(o + vb + (i64)d * HW + g * 256) = max4(max4(v[u][g][0], v[u][g][1]), max4(v[u][g][2], v[u][g][3]));
      }
    }
  }
}
// march, one pixel per lane (4-byte requests), as the current kernels
template <int DU>
__global__ void __launch_bounds__(256) k_march1(const float *a0, const float *a1, const float *a2, const float *a3, float *o)
{
  const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
  if (p >= (i64)S * HW) return;
  const i64 s = p / HW, pix = p - s * HW, vb = s * D * HW + pix;
  for (int dc = 0; dc < D; dc += DU) {
    float v[DU][4];
#pragma unroll
    for (int u = 0; u < DU; u++) { const int d = dc + u < D ? dc + u : D - 1; const i64 off = vb + (i64)d * HW; v[u][0] = a0[off]; v[u][1] = a1[off]; v[u][2] = a2[off]; v[u][3] = a3[off]; }
#pragma unroll
    for (int u = 0; u < DU; u++) { const int d = dc + u; if (d < D) o[vb + (i64)d * HW] = fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])); }
  }
}

// nine volumes in, one out, one pixel per lane marching over d (the shape of sga_bwd_point): a[4..7] are read at a
// neighbouring pixel (-W, +W, -1, +1) like the forward volumes of the previous scan position
template <int DU, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_march9(const float *x, const float *g0, const float *g1, const float *g2, const float *g3,
                                                        const float *a0, const float *a1, const float *a2, const float *a3, float *o)
{
  const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
  if (p >= (i64)S * HW) return;
  const i64 s = p / HW, pix = p - s * HW, vb = s * D * HW + pix;
  const int W = 208;
  const int h = (int)(pix / W), w = (int)(pix % W);
  const int o0 = h > 0 ? -W : 0, o1 = h < 79 ? W : 0, o2 = w > 0 ? -1 : 0, o3 = w < W - 1 ? 1 : 0;
  float acc = 0.f;
  for (int dc = 0; dc < D; dc += DU) {
    float v[DU][9];
#pragma unroll
    for (int u = 0; u < DU; u++) {
      const int d = dc + u < D ? dc + u : D - 1;
      const i64 off = vb + (i64)d * HW;
      v[u][0] = x[off]; v[u][1] = g0[off]; v[u][2] = g1[off]; v[u][3] = g2[off]; v[u][4] = g3[off];
      v[u][5] = a0[off + o0]; v[u][6] = a1[off + o1]; v[u][7] = a2[off + o2]; v[u][8] = a3[off + o3];
    }
#pragma unroll
    for (int u = 0; u < DU; u++) {
      const int d = dc + u;
      if (d < D) {
        float r = v[u][0];
#pragma unroll
        for (int q = 1; q < 9; q++) r = fmaf(r, 0.5f, v[u][q]);
        acc += r;
        o[vb + (i64)d * HW] = r + acc;
      }
    }
  }
}

template <typename F> void run(const char *name, F launch)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = 5.0 * S * D * HW * 4;
  printf("%-28s %.4f ms  %.2f TB/s   %s\n", name, ms, bytes / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main()
{
  const i64 n = (i64)S * D * HW;
  float *a[4], *o;
  for (int i = 0; i < 4; i++) { hipMalloc(&a[i], n * 4); hipMemset(a[i], i, n * 4); }
  hipMalloc(&o, n * 4);
  const i64 n4 = n / 4;
  run("lin grid=256*8", [&] { k_lin<<<256 * 8, 256>>>((float4 *)a[0], (float4 *)a[1], (float4 *)a[2], (float4 *)a[3], (float4 *)o, n4); });
  run("lin grid=256*32", [&] { k_lin<<<256 * 32, 256>>>((float4 *)a[0], (float4 *)a[1], (float4 *)a[2], (float4 *)a[3], (float4 *)o, n4); });
  run("chunk U=1", [&] { k_chunk<1><<<(unsigned)((n4 + 255) / 256), 256>>>((float4 *)a[0], (float4 *)a[1], (float4 *)a[2], (float4 *)a[3], (float4 *)o, n4); });
  run("chunk U=4", [&] { k_chunk<4><<<(unsigned)((n4 + 1023) / 1024), 256>>>((float4 *)a[0], (float4 *)a[1], (float4 *)a[2], (float4 *)a[3], (float4 *)o, n4); });
  run("march 1px DU=1", [&] { k_march1<1><<<(unsigned)(((i64)S * HW + 255) / 256), 256>>>(a[0], a[1], a[2], a[3], o); });
  run("march 1px DU=2", [&] { k_march1<2><<<(unsigned)(((i64)S * HW + 255) / 256), 256>>>(a[0], a[1], a[2], a[3], o); });
  run("march 1px DU=4", [&] { k_march1<4><<<(unsigned)(((i64)S * HW + 255) / 256), 256>>>(a[0], a[1], a[2], a[3], o); });
#define M(PX4, DU, B) run("march4 PX4=" #PX4 " DU=" #DU " B=" #B, [&] { const i64 nq = (i64)S * HW / 4 / PX4; k_march<PX4, DU, B><<<(unsigned)((nq + B - 1) / B), B>>>(a[0], a[1], a[2], a[3], o); });
  M(1, 1, 64) M(1, 2, 64) M(1, 4, 64) M(1, 1, 256) M(1, 2, 256) M(1, 4, 256) M(2, 1, 64) M(2, 2, 64) M(5, 1, 64)
  {
    float *b[9];
    for (int i = 0; i < 9; i++) { hipMalloc(&b[i], n * 4); hipMemset(b[i], i, n * 4); }
    const unsigned grid = (unsigned)(((i64)S * HW + 255) / 256);
    printf("nine in / one out (x2 for the byte count shown: multiply TB/s by 2)\n");
#define M9(DU, WV) run("march9 1px DU=" #DU " waves=" #WV, [&] { k_march9<DU, WV><<<grid, 256>>>(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], o); });
    M9(1, 8) M9(2, 8) M9(2, 5) M9(4, 4) M9(4, 8)
  }
  return 0;
}
