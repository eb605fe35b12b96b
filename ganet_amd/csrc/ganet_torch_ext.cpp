// ganet_torch_ext.cpp -- the pybind module named `GANet` that the reference's Python layer imports as
// `from ..build.lib import GANet` (libs/GANet/functions/GANet.py:3): the six functions of
// libs/GANet/src/GANet_cuda.cpp:67-75 with the same names, argument order, caller-allocated buffer contract
// and return value (1), forwarding to the C ABI of libganet_hip.so (include/ganet_hip.h) on torch's CURRENT
// HIP stream.  With this module under libs/GANet/build/lib/ the reference's own functions/GANet.py and
// modules/GANet.py run unmodified on MI355X (mode B).
//
// Differences from the reference's binding, all on the safe side: dtype / device / contiguity / shape are
// checked (the reference checks nothing, SURVEY 8b "Preconditions"), runtime errors raise, and the device of
// `input` is made current for the call (the reference relies on the Python-side torch.cuda.device_of).
#include <torch/extension.h>

// PyTorch-ROCm presents its HIP devices as device type "cuda": the guard / stream accessors that accept such a device
// are the *MasqueradingAsCUDA forms (the plain c10::hip::HIPGuard rejects it: "non-HIP DeviceType: cuda")
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "../../include/ganet_hip.h"

namespace {

void check_all(const char *who, std::initializer_list<at::Tensor> ts)
{
  const at::Tensor &first = *ts.begin();
  for (const at::Tensor &t : ts) {
    TORCH_CHECK(t.is_cuda(), who, ": tensors must live on a HIP device (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, who, ": tensors must be float32");
    TORCH_CHECK(t.is_contiguous(), who, ": tensors must be contiguous");
    TORCH_CHECK(t.device() == first.device(), who, ": all tensors must live on one device");
  }
}

void check_rc(const char *who, int rc) { TORCH_CHECK(rc == GANET_OK, who, ": ", ganet_last_error()); }

void *cur_stream() { return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

float *p(const at::Tensor &t) { return t.data_ptr<float>(); }

// GANet_cuda.cpp:39-48
int sga_cuda_forward(at::Tensor input, at::Tensor guidance_down, at::Tensor guidance_up, at::Tensor guidance_right,
                     at::Tensor guidance_left, at::Tensor temp_out, at::Tensor output, at::Tensor mask)
{
  check_all("sga_cuda_forward", {input, guidance_down, guidance_up, guidance_right, guidance_left, temp_out, output, mask});
  TORCH_CHECK(input.dim() == 5, "sga_cuda_forward: input must be [N,C,D,H,W]");
  for (const at::Tensor &g : {guidance_down, guidance_up, guidance_right, guidance_left})
    TORCH_CHECK(g.dim() == 5 && g.size(0) == input.size(0) && g.size(1) == input.size(1) && g.size(2) == 5 &&
                    g.size(3) == input.size(3) && g.size(4) == input.size(4),
                "sga_cuda_forward: guidance must be [N,C,5,H,W]");
  for (const at::Tensor &t : {temp_out, output, mask})
    TORCH_CHECK(t.sizes() == input.sizes(), "sga_cuda_forward: temp_out / output / mask must have input's shape");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
  check_rc("sga_cuda_forward",
           ganet_sga_forward_compat(p(input), p(guidance_down), p(guidance_up), p(guidance_right), p(guidance_left),
                                    p(temp_out), p(output), p(mask), (int)input.size(0), (int)input.size(1),
                                    (int)input.size(2), (int)input.size(3), (int)input.size(4), cur_stream()));
  return 1;
}

// GANet_cuda.cpp:50-64
int sga_cuda_backward(at::Tensor input, at::Tensor guidance_down, at::Tensor guidance_up, at::Tensor guidance_right,
                      at::Tensor guidance_left, at::Tensor temp_out, at::Tensor mask, at::Tensor max_idx,
                      at::Tensor gradOutput, at::Tensor temp_grad, at::Tensor gradInput, at::Tensor grad_down,
                      at::Tensor grad_up, at::Tensor grad_right, at::Tensor grad_left)
{
  check_all("sga_cuda_backward", {input, guidance_down, guidance_up, guidance_right, guidance_left, temp_out, mask,
                                  max_idx, gradOutput, temp_grad, gradInput, grad_down, grad_up, grad_right, grad_left});
  TORCH_CHECK(input.dim() == 5, "sga_cuda_backward: input must be [N,C,D,H,W]");
  for (const at::Tensor &t : {temp_out, mask, gradOutput, temp_grad, gradInput})
    TORCH_CHECK(t.sizes() == input.sizes(), "sga_cuda_backward: volume arguments must have input's shape");
  for (const at::Tensor &g : {grad_down, grad_up, grad_right, grad_left})
    TORCH_CHECK(g.sizes() == guidance_down.sizes(), "sga_cuda_backward: guidance gradients must be [N,C,5,H,W]");
  TORCH_CHECK(max_idx.numel() == input.size(0) * input.size(1) * input.size(3) * input.size(4),
              "sga_cuda_backward: max_idx must be [N,C,H,W]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
  check_rc("sga_cuda_backward",
           ganet_sga_backward_compat(p(input), p(guidance_down), p(guidance_up), p(guidance_right), p(guidance_left),
                                     p(temp_out), p(mask), p(max_idx), p(gradOutput), p(temp_grad), p(gradInput),
                                     p(grad_down), p(grad_up), p(grad_right), p(grad_left), (int)input.size(0),
                                     (int)input.size(1), (int)input.size(2), (int)input.size(3), (int)input.size(4),
                                     cur_stream()));
  return 1;
}

struct LgaDims {
  int B, D, H, W;
};

LgaDims lga_dims(const char *who, const at::Tensor &input, const at::Tensor &filters, int radius, bool five_d)
{
  TORCH_CHECK(input.dim() == (five_d ? 5 : 4), who, five_d ? ": input must be [N,C,D,H,W]" : ": input must be [N,D,H,W]");
  TORCH_CHECK(filters.dim() == input.dim(), who, ": filters must have input's rank");
  const int64_t taps = 3 * (2 * radius + 1) * (2 * radius + 1);
  const int k = input.dim();
  TORCH_CHECK(filters.size(k - 3) == taps && filters.size(k - 2) == input.size(k - 2) && filters.size(k - 1) == input.size(k - 1),
              who, ": filters must hold 3*(2r+1)^2 taps per pixel");
  for (int i = 0; i < k - 3; i++) TORCH_CHECK(filters.size(i) == input.size(i), who, ": filters / input batch mismatch");
  LgaDims d;
  d.B = five_d ? (int)(input.size(0) * input.size(1)) : (int)input.size(0);
  d.D = (int)input.size(k - 3);
  d.H = (int)input.size(k - 2);
  d.W = (int)input.size(k - 1);
  return d;
}

int lga_fwd(const char *who, at::Tensor input, at::Tensor filters, at::Tensor output, int radius, bool five_d)
{
  check_all(who, {input, filters, output});
  TORCH_CHECK(output.sizes() == input.sizes(), who, ": output must have input's shape");
  const LgaDims d = lga_dims(who, input, filters, radius, five_d);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
  // output is overwritten: equal to the reference's `+=` into the zero-filled buffer its caller hands over
  check_rc(who, ganet_lga_forward(p(input), p(filters), p(output), d.B, d.D, d.H, d.W, radius, cur_stream()));
  return 1;
}

int lga_bwd(const char *who, at::Tensor input, at::Tensor filters, at::Tensor gradOutput, at::Tensor gradInput,
            at::Tensor gradFilters, int radius, bool five_d)
{
  check_all(who, {input, filters, gradOutput, gradInput, gradFilters});
  TORCH_CHECK(gradOutput.sizes() == input.sizes() && gradInput.sizes() == input.sizes(), who,
              ": gradOutput / gradInput must have input's shape");
  TORCH_CHECK(gradFilters.sizes() == filters.sizes(), who, ": gradFilters must have filters' shape");
  const LgaDims d = lga_dims(who, input, filters, radius, five_d);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
  // gradFilters accumulated into, gradInput overwritten (GANet_kernel.cu:1299-1322); gradInput may alias input
  check_rc(who, ganet_lga_backward(p(input), p(filters), p(gradOutput), p(gradInput), p(gradFilters), d.B, d.D, d.H, d.W,
                                   radius, 1, cur_stream()));
  return 1;
}

// GANet_cuda.cpp:5-37
int lga_cuda_forward(at::Tensor input, at::Tensor filters, at::Tensor output, const int radius)
{
  return lga_fwd("lga_cuda_forward", input, filters, output, radius, false);
}
int lga_cuda_backward(at::Tensor input, at::Tensor filters, at::Tensor gradOutput, at::Tensor gradInput,
                      at::Tensor gradFilters, const int radius)
{
  return lga_bwd("lga_cuda_backward", input, filters, gradOutput, gradInput, gradFilters, radius, false);
}
int lga3d_cuda_forward(at::Tensor input, at::Tensor filters, at::Tensor output, const int radius)
{
  return lga_fwd("lga3d_cuda_forward", input, filters, output, radius, true);
}
int lga3d_cuda_backward(at::Tensor input, at::Tensor filters, at::Tensor gradOutput, at::Tensor gradInput,
                        at::Tensor gradFilters, const int radius)
{
  return lga_bwd("lga3d_cuda_backward", input, filters, gradOutput, gradInput, gradFilters, radius, true);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
  m.doc() = "GA-Net guided aggregation on MI355X (libganet_hip.so) behind the reference's pybind surface";
  m.def("lga_cuda_forward", &lga_cuda_forward, "lga forward (HIP, gfx950)");
  m.def("lga_cuda_backward", &lga_cuda_backward, "lga backward (HIP, gfx950)");
  m.def("lga3d_cuda_forward", &lga3d_cuda_forward, "lga3d forward (HIP, gfx950)");
  m.def("lga3d_cuda_backward", &lga3d_cuda_backward, "lga3d backward (HIP, gfx950)");
  m.def("sga_cuda_forward", &sga_cuda_forward, "sga forward (HIP, gfx950)");
  m.def("sga_cuda_backward", &sga_cuda_backward, "sga backward (HIP, gfx950)");
  m.def("abi_version", []() { return ganet_abi_version(); }, "C-ABI version of the libganet_hip.so in use");
}
