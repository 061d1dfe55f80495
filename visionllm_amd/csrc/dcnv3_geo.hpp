// Geometry of a DCNv3 call, shared by the gather kernel (dcnv3.hip) and the LDS-tiled kernel (dcnv3_tiled.hip).
#pragma once
namespace vllm {
struct Dcnv3Geo {
    int N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
};
}  // namespace vllm
