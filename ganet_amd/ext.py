"""Stand-in for the reference's pybind module `GANet` (libs/GANet/src/GANet_cuda.cpp:67-75,
imported there as `from ..build.lib import GANet`): the same six functions, the same
caller-allocated buffer contract, each returning 1, forwarding to the C ABI of
libganet_hip.so.  A caller that keeps the reference's own functions/GANet.py can do
`from ganet_amd import ext as GANet` and change nothing else."""
import torch

from . import _native


def _s():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("GANet ext: tensors must be contiguous fp32 HIP tensors")


def sga_cuda_forward(input, g0, g1, g2, g3, temp_out, output, mask):
    _chk(input, g0, g1, g2, g3, temp_out, output, mask)
    N, C, D, H, W = input.shape
    with torch.cuda.device_of(input):
        _native.lib().call("ganet_sga_forward_compat", input.data_ptr(), g0.data_ptr(), g1.data_ptr(),
                           g2.data_ptr(), g3.data_ptr(), temp_out.data_ptr(), output.data_ptr(),
                           mask.data_ptr(), N, C, D, H, W, _s())
    return 1


def sga_cuda_backward(input, g0, g1, g2, g3, temp_out, mask, max_idx, gradOutput, temp_grad, gradInput,
                      grad0, grad1, grad2, grad3):
    _chk(input, g0, g1, g2, g3, temp_out, mask, max_idx, gradOutput, temp_grad, gradInput, grad0, grad1,
         grad2, grad3)
    N, C, D, H, W = input.shape
    with torch.cuda.device_of(input):
        _native.lib().call("ganet_sga_backward_compat", input.data_ptr(), g0.data_ptr(), g1.data_ptr(),
                           g2.data_ptr(), g3.data_ptr(), temp_out.data_ptr(), mask.data_ptr(),
                           max_idx.data_ptr(), gradOutput.data_ptr(), temp_grad.data_ptr(),
                           gradInput.data_ptr(), grad0.data_ptr(), grad1.data_ptr(), grad2.data_ptr(),
                           grad3.data_ptr(), N, C, D, H, W, _s())
    return 1


def _lga_dims(input):
    if input.dim() == 5:
        return input.shape[0] * input.shape[1], input.shape[2], input.shape[3], input.shape[4]
    return tuple(input.shape)


def lga_cuda_forward(input, filters, output, radius):
    """output is overwritten (== the reference's `+=` into its zero-filled buffer)."""
    _chk(input, filters, output)
    B, D, H, W = _lga_dims(input)
    with torch.cuda.device_of(input):
        _native.lib().call("ganet_lga_forward", input.data_ptr(), filters.data_ptr(), output.data_ptr(),
                           B, D, H, W, radius, _s())
    return 1


def lga_cuda_backward(input, filters, gradOutput, gradInput, gradFilters, radius):
    """gradInput overwritten, gradFilters accumulated into -- as in GANet_kernel.cu:1299-1322.
    gradInput may alias `input` (the reference's chained backward does that)."""
    _chk(input, filters, gradOutput, gradInput, gradFilters)
    B, D, H, W = _lga_dims(input)
    with torch.cuda.device_of(input):
        _native.lib().call("ganet_lga_backward", input.data_ptr(), filters.data_ptr(), gradOutput.data_ptr(),
                           gradInput.data_ptr(), gradFilters.data_ptr(), B, D, H, W, radius, 1, _s())
    return 1


lga3d_cuda_forward = lga_cuda_forward
lga3d_cuda_backward = lga_cuda_backward
