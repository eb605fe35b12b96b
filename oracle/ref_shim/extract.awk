# Prints the reference's __global__ kernel bodies only (no host launchers,
# which need <<<>>> and the CUDA runtime): GANet_kernel.cu from `Max` up to
# (not including) sga_kernel_forward, and from lga_filtering_forward up to
# (not including) lga_forward.
/^__global__ void Max/                   { on = 1 }
/^void sga_kernel_forward/               { on = 0 }
/^__global__ void lga_filtering_forward/ { on = 1 }
/^void lga_forward/                      { on = 0 }
on { print }
