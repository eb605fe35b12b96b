// Host shim that lets the reference's own __global__ kernel BODIES
// (/root/reference/libs/GANet/src/GANet_kernel.cu, piped in at build time by
// oracle/Makefile -- never copied into this repository) compile with g++ and
// run one "CUDA thread" per loop iteration.  Valid because no reference kernel
// uses shared memory, atomics or any inter-thread communication
// (SURVEY.md section 2.1): thread i is an independent sequential program.
// TEST INFRASTRUCTURE ONLY (see oracle/ganet_oracle.c header).
#include <cstring>
#include <cstdlib>
#ifdef _OPENMP
#include <omp.h>
#endif
#define __global__
struct ref_dim3 { int x, y, z; };
static thread_local ref_dim3 blockIdx, blockDim, threadIdx;
