// Geometry of a DCNv3 call, shared by the gather kernel (dcnv3.hip) and the LDS-tiled kernel (dcnv3_tiled.hip).
#pragma once
namespace vllm {
struct Dcnv3Geo {
    int N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
};

// ONE definition of a sampling location for every DCNv3 kernel (forward gather / tiled / pipelined, backward gather / windowed):
//   loc = p0 + (i * dilation + offset) * offset_scale          (dcnv3_im2col_cuda.cuh:256-259)
// with the product ROUNDED ON ITS OWN.  Under hipcc's default -ffp-contract=fast the backend may or may not fuse the multiply into
// the add, kernel by kernel; floor(loc) picks the cell and grad_offset is discontinuous across a cell border, so two kernels that
// disagree in the last bit of loc can disagree about a whole cell for a location next to an integer.  The empty asm
// makes the product opaque (zero instructions), as mul_rn of msda_sample.hpp does for the MSDA kernels.
// What this guarantees and what it does not: cell parity next to integer locations holds between THIS library's
// kernels and against the test suite's numpy / C restatement of the reference (it rounds the product on its own too).  The reference's
// compiled CUDA extension is built by nvcc with its default -fmad=true, which may contract p0 + (i*d + off) * scale into one FMA:
// for a location within 1 ulp of an integer that build can pick the neighbouring cell (grad_offset then differs by a whole cell's
// slope).  No reference build exists here to compare against (CUDA only); the half-precision fixture keeps its
// locations away from integers for that reason.
#ifdef __HIPCC__
template <typename T>
__device__ __forceinline__ T dcn_mul_rn(T a, T b)
{
    T p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
template <typename T>
__device__ __forceinline__ T dcn_loc(T p0, T i_dil, T off, T scale) { return p0 + dcn_mul_rn<T>(i_dil + off, scale); }
#endif
}  // namespace vllm
