// Restated launch order of the reference's host launchers.  They contain no
// arithmetic, only this sequence (cited lines: GANet_kernel.cu).  The kernel
// bodies themselves are the reference's, compiled from where they lie.
template <class F> static void launch(long n, F f)
{
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) {
    blockDim.x = 1; threadIdx.x = 0; blockIdx.x = (int)i;
    f();
  }
}

extern "C" {

int ref_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// sga_kernel_forward, :935-998.  mask must be zero-filled by the caller
// (functions/GANet.py:16), exactly as in the reference.
void ref_sga_forward(const float *x, const float *g0, const float *g1, const float *g2,
                     const float *g3, float *tmp, float *out, float *mask,
                     int num, int channel, int depth, int height, int width)
{
  const int wsize = 5;
  const long N = (long)num * channel * depth * height * width;
  int n = num * channel * width;
  memcpy(tmp, x, sizeof(float) * N);
  launch(n, [&] { sga_down_forward(n, g0, height, width, depth, wsize, tmp); });
  memcpy(out, tmp, sizeof(float) * N);
  memcpy(tmp, x, sizeof(float) * N);
  launch(n, [&] { sga_up_forward(n, g1, height, width, depth, wsize, tmp); });
  launch(N, [&] { Max((int)N, tmp, out, mask, 1); });
  n = num * channel * height;
  memcpy(tmp, x, sizeof(float) * N);
  launch(n, [&] { sga_right_forward(n, g2, height, width, depth, wsize, tmp); });
  launch(N, [&] { Max((int)N, tmp, out, mask, 2); });
  memcpy(tmp, x, sizeof(float) * N);
  launch(n, [&] { sga_left_forward(n, g3, height, width, depth, wsize, tmp); });
  launch(N, [&] { Max((int)N, tmp, out, mask, 3); });
}

// one directional scan only (used to cross-check per-direction volumes)
void ref_sga_scan(const float *x, const float *g, float *tmp, int num, int channel,
                  int depth, int height, int width, int dir)
{
  const int wsize = 5;
  const long N = (long)num * channel * depth * height * width;
  memcpy(tmp, x, sizeof(float) * N);
  int n = num * channel * (dir < 2 ? width : height);
  if (dir == 0) launch(n, [&] { sga_down_forward(n, g, height, width, depth, wsize, tmp); });
  if (dir == 1) launch(n, [&] { sga_up_forward(n, g, height, width, depth, wsize, tmp); });
  if (dir == 2) launch(n, [&] { sga_right_forward(n, g, height, width, depth, wsize, tmp); });
  if (dir == 3) launch(n, [&] { sga_left_forward(n, g, height, width, depth, wsize, tmp); });
}

// sga_kernel_backward, :1000-1129.  tmp holds A_left on entry (F6);
// gradInput / grad0..3 zero-filled by the caller (functions/GANet.py:33-37).
void ref_sga_backward(const float *x, const float *g0, const float *g1, const float *g2,
                      const float *g3, float *tmp, const float *mask, float *idx,
                      const float *grad_out, float *top_grad, float *grad_input,
                      float *grad0, float *grad1, float *grad2, float *grad3,
                      int num, int channel, int depth, int height, int width)
{
  const int wsize = 5;
  const long N = (long)num * channel * depth * height * width;
  const long P = (long)num * channel * width * height;
  const int step = height * width;
  int n;
  // left (uses saved tmp)
  n = num * channel * height;
  memset(top_grad, 0, sizeof(float) * N);
  launch(N, [&] { get_temp_grad((int)N, grad_out, mask, top_grad, 3); });
  launch(P, [&] { MaxDepth((int)P, tmp, step, depth, idx); });
  launch(n, [&] { sga_left_data_backward(n, g3, top_grad, idx, height, width, depth, wsize, grad_input); });
  launch(P, [&] { sga_left_weight_backward((int)P, x, tmp, top_grad, idx, height, width, depth, wsize, grad3); });
  // down
  n = num * channel * width;
  memcpy(tmp, x, sizeof(float) * N);
  launch(n, [&] { sga_down_forward(n, g0, height, width, depth, wsize, tmp); });
  memset(top_grad, 0, sizeof(float) * N);
  launch(N, [&] { get_temp_grad((int)N, grad_out, mask, top_grad, 0); });
  launch(P, [&] { MaxDepth((int)P, tmp, step, depth, idx); });
  launch(n, [&] { sga_down_data_backward(n, g0, top_grad, idx, height, width, depth, wsize, grad_input); });
  launch(P, [&] { sga_down_weight_backward((int)P, x, tmp, top_grad, idx, height, width, depth, wsize, grad0); });
  // up
  memcpy(tmp, x, sizeof(float) * N);
  launch(n, [&] { sga_up_forward(n, g1, height, width, depth, wsize, tmp); });
  memset(top_grad, 0, sizeof(float) * N);
  launch(N, [&] { get_temp_grad((int)N, grad_out, mask, top_grad, 1); });
  launch(P, [&] { MaxDepth((int)P, tmp, step, depth, idx); });
  launch(n, [&] { sga_up_data_backward(n, g1, top_grad, idx, height, width, depth, wsize, grad_input); });
  launch(P, [&] { sga_up_weight_backward((int)P, x, tmp, top_grad, idx, height, width, depth, wsize, grad1); });
  // right
  n = num * channel * height;
  memcpy(tmp, x, sizeof(float) * N);
  launch(n, [&] { sga_right_forward(n, g2, height, width, depth, wsize, tmp); });
  memset(top_grad, 0, sizeof(float) * N);
  launch(N, [&] { get_temp_grad((int)N, grad_out, mask, top_grad, 2); });
  launch(P, [&] { MaxDepth((int)P, tmp, step, depth, idx); });
  launch(n, [&] { sga_right_data_backward(n, g2, top_grad, idx, height, width, depth, wsize, grad_input); });
  launch(P, [&] { sga_right_weight_backward((int)P, x, tmp, top_grad, idx, height, width, depth, wsize, grad2); });
}

// lga_forward :1271-1296 / lga3d_forward :1324-1338 (same kernel; `batch`
// is N for the 4-D form and N*C for the 5-D form).  y zero-filled by caller.
void ref_lga_forward(const float *x, const float *f, float *y, int batch, int channel,
                     int height, int width, int radius)
{
  const long n = (long)batch * channel * height * width;
  launch(n, [&] { lga_filtering_forward((int)n, x, f, height, width, channel, radius, y); });
}

// lga_backward :1299-1322 / lga3d_backward :1341-1364.
void ref_lga_backward(const float *x, const float *f, const float *gy, float *gx, float *gf,
                      int batch, int channel, int height, int width, int radius)
{
  const int ws = 2 * radius + 1;
  long n = (long)batch * 3 * ws * ws * height * width;
  launch(n, [&] { lga_filter_backward((int)n, x, gy, height, width, channel, radius, gf); });
  n = (long)batch * channel * height * width;
  memset(gx, 0, sizeof(float) * n);
  launch(n, [&] { lga_data_backward((int)n, f, gy, height, width, channel, radius, gx); });
}

}  // extern "C"
