// Stub for the CUDA half of the reference's curope extension, so that its CPU half
// (/root/reference/croco/models/curope/curope.cpp, compiled from where it lies) links on a machine without CUDA.
// TEST INFRASTRUCTURE ONLY (oracle/build_ref.py).  No reference code is copied into this repository.
#include <torch/extension.h>

void rope_2d_cuda(torch::Tensor, const torch::Tensor, const float, const float) {
  TORCH_CHECK(false, "oracle/_ref/curope_ref is the reference's CPU path only");
}
