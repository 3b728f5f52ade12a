// average_images.hip -- motion-blur sub-frame averaging in linear light, on the GPU.
//
// Reference: average_images (src/main.rs:645-722), a CPU loop over RGBA8 frames using two LUTs:
// S_TO_L[c] = c*c (gamma-2 decode to 0..65025), integer mean over the N sub-frames, then
// L_TO_S[l] = (u8)(sqrt(l as f32) + 0.5) (gamma-2 encode, rounded), alpha = 255.
// This is byte/integer work and HBM-bound: per output pixel it reads 4 N bytes and writes 4.
//
// gfx950 mapping: one lane handles 4 consecutive pixels = one 16-byte load per sub-frame
// (64 lanes x 16 B = 1 KiB per wave-instruction, fully coalesced), accumulates 12 u32 sums in
// registers, and issues one 16-byte store.  The sub-frame loop is unrolled x4 so at least four
// independent 16-byte loads per lane are in flight; streaming (non-temporal) loads keep the frames
// from displacing anything useful in L2.  No LDS, no atomics.  The LUTs are arithmetic here:
// c*c is exact in u32, and sqrt of an integer <= 65025 in binary32 + 0.5 truncated is what the
// reference's table holds (sqrtf is correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt).
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#define PTL_MAX_SUBFRAMES 64

typedef unsigned int ptl_u32x4 __attribute__((ext_vector_type(4)));  // native vector: what the nontemporal builtins accept

struct ptl_frame_list {
    const ptl_u32x4* frame[PTL_MAX_SUBFRAMES];
};

__device__ __forceinline__ void ptl_accumulate(unsigned int (&sum)[12], ptl_u32x4 p) {
    const unsigned int w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned int r = w[k] & 0xffu, g = (w[k] >> 8) & 0xffu, b = (w[k] >> 16) & 0xffu;
        sum[3 * k + 0] += r * r;
        sum[3 * k + 1] += g * g;
        sum[3 * k + 2] += b * b;
    }
}

__device__ __forceinline__ unsigned int ptl_l_to_s(unsigned int linear) {
    return (unsigned int)(__builtin_sqrtf((float)linear) + 0.5f);  // truncation, like `as u8` on a value <= 255.5
}

extern "C" __global__ void __launch_bounds__(256)
ptl_average_images_kernel(ptl_frame_list frames, int n, ptl_u32x4* __restrict__ out, long n_vec) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) {
        unsigned int sum[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        int f = 0;
        for (; f + 4 <= n; f += 4) {  // four independent 16-byte loads in flight per lane
            const ptl_u32x4 a = __builtin_nontemporal_load(frames.frame[f + 0] + i);
            const ptl_u32x4 b = __builtin_nontemporal_load(frames.frame[f + 1] + i);
            const ptl_u32x4 c = __builtin_nontemporal_load(frames.frame[f + 2] + i);
            const ptl_u32x4 d = __builtin_nontemporal_load(frames.frame[f + 3] + i);
            ptl_accumulate(sum, a);
            ptl_accumulate(sum, b);
            ptl_accumulate(sum, c);
            ptl_accumulate(sum, d);
        }
        for (; f < n; ++f) ptl_accumulate(sum, __builtin_nontemporal_load(frames.frame[f] + i));
        unsigned int px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int r = ptl_l_to_s(sum[3 * k + 0] / (unsigned)n);
            const unsigned int g = ptl_l_to_s(sum[3 * k + 1] / (unsigned)n);
            const unsigned int b = ptl_l_to_s(sum[3 * k + 2] / (unsigned)n);
            px[k] = r | (g << 8) | (b << 16) | 0xff000000u;
        }
        const ptl_u32x4 packed = {px[0], px[1], px[2], px[3]};
        __builtin_nontemporal_store(packed, out + i);
    }
}
