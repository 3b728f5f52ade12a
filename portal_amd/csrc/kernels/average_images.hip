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
// independent 16-byte loads per lane are in flight.  No LDS, no atomics.  The LUTs are arithmetic here:
// c*c is exact in u32, the integer mean is a multiply-high by a per-launch constant, and sqrt + 0.5 truncated equals
// the reference's table entry for every l <= 65025 (see ptl_l_to_s).
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#define PTL_MAX_SUBFRAMES 64

typedef unsigned int ptl_u32x4 __attribute__((ext_vector_type(4)));  // native vector: what the nontemporal builtins accept

struct ptl_frame_list {  // up to 64 sub-frames: the pointers travel in the kernel arguments
    const ptl_u32x4* frame[PTL_MAX_SUBFRAMES];
};
struct ptl_frame_table {  // 65..256 sub-frames: a pointer table in device memory (scalar loads)
    const ptl_u32x4* const* frame;
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

// L_TO_S[l] = ((l as f32).sqrt() + 0.5) as u8 for l <= 65025.  The hardware v_sqrt_f32 (1 ulp) is enough: the value only
// changes where sqrt(l) + 0.5 crosses an integer, i.e. near l = k*k + k + 1/4, and the nearest integers l = k*k + k and
// k*k + k + 1 keep sqrt(l) at least 1/(8k+4) >= 4.9e-4 away from k + 0.5 -- against an ulp of 3e-5 at 256.  (Checked for
// every l in tests/test_gpu_parity.py::test_average_images_every_linear_value.)
__device__ __forceinline__ unsigned int ptl_l_to_s(unsigned int linear) {
    return (unsigned int)(__builtin_amdgcn_sqrtf((float)linear) + 0.5f);  // truncation, like `as u8` on a value <= 255.5
}

// sum / n for sum <= 65025 * n and 2 <= n <= 256 as one multiply-high: with m = floor(2^32 / n) + 1,
// m*n - 2^32 = e in (0, n], and floor(sum * m / 2^32) == floor(sum / n) whenever sum * e < 2^32
// (65025 * 256 * 256 = 4.26e9 < 2^32 = 4.29e9: n = 256 is the last one that fits; checked for every n in tests/test_host_logic.py).
// A runtime `/` would be ~30 VALU instructions, twelve times per lane -- more than the whole rest of the kernel.
__device__ __forceinline__ unsigned int ptl_div_n(unsigned int sum, unsigned int magic) {
    return magic ? __umulhi(sum, magic) : sum;  // magic == 0 encodes n == 1
}

// Tuning knobs (tools/average_variants.py builds the alternatives; the defaults are what ships):
#ifndef PTL_AVG_UNROLL
#define PTL_AVG_UNROLL 4  // independent 16-byte loads in flight per lane and sub-frame group
#endif
#ifndef PTL_AVG_NT
#define PTL_AVG_NT 0      // 1: non-temporal (streaming) loads and store.  Measured slower (5.3 vs 6.1 TB/s at 4K, N = 4): the
                          // sub-frames were written by the tracer a moment ago and part of them is still in the 256 MB MALL
#endif
#ifndef PTL_AVG_VPT
#define PTL_AVG_VPT 1     // 16-byte vectors per lane per grid-stride step
#endif

__device__ __forceinline__ ptl_u32x4 ptl_stream_load(const ptl_u32x4* p) {
#if PTL_AVG_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

template <class Frames>
__device__ __forceinline__ void ptl_average_one(const Frames& frames, int n, unsigned int magic, ptl_u32x4* __restrict__ out, long i) {
    unsigned int sum[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    int f = 0;
    for (; f + PTL_AVG_UNROLL <= n; f += PTL_AVG_UNROLL) {
        ptl_u32x4 v[PTL_AVG_UNROLL];
#pragma unroll
        for (int k = 0; k < PTL_AVG_UNROLL; ++k) v[k] = ptl_stream_load(frames.frame[f + k] + i);
#pragma unroll
        for (int k = 0; k < PTL_AVG_UNROLL; ++k) ptl_accumulate(sum, v[k]);
    }
    for (; f < n; ++f) ptl_accumulate(sum, ptl_stream_load(frames.frame[f] + i));
    unsigned int px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned int r = ptl_l_to_s(ptl_div_n(sum[3 * k + 0], magic));
        const unsigned int g = ptl_l_to_s(ptl_div_n(sum[3 * k + 1], magic));
        const unsigned int b = ptl_l_to_s(ptl_div_n(sum[3 * k + 2], magic));
        px[k] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
    const ptl_u32x4 packed = {px[0], px[1], px[2], px[3]};
#if PTL_AVG_NT
    __builtin_nontemporal_store(packed, out + i);
#else
    out[i] = packed;
#endif
}

// One pixel with 4-byte accesses: the up-to-three pixels behind the last whole 16-byte vector of a frame.
template <class Frames>
__device__ __forceinline__ void ptl_average_tail_pixel(const Frames& frames, int n, unsigned int magic, unsigned int* __restrict__ out, long px) {
    unsigned int r = 0u, g = 0u, b = 0u;
    for (int f = 0; f < n; ++f) {
        const unsigned int w = reinterpret_cast<const unsigned int*>(frames.frame[f])[px];
        const unsigned int cr = w & 0xffu, cg = (w >> 8) & 0xffu, cb = (w >> 16) & 0xffu;
        r += cr * cr;
        g += cg * cg;
        b += cb * cb;
    }
    out[px] = ptl_l_to_s(ptl_div_n(r, magic)) | (ptl_l_to_s(ptl_div_n(g, magic)) << 8) | (ptl_l_to_s(ptl_div_n(b, magic)) << 16) | 0xff000000u;
}

template <class Frames>
__device__ __forceinline__ void ptl_average_all(const Frames& frames, int n, ptl_u32x4* __restrict__ out, long n_px) {
    const long stride = (long)gridDim.x * 256;
    const long n_vec = n_px >> 2;  // whole 4-pixel vectors
    const unsigned int magic = n > 1 ? 0xffffffffu / (unsigned)n + 1u : 0u;  // wave-uniform: one division on the scalar side of things
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride * PTL_AVG_VPT) {
#pragma unroll
        for (int v = 0; v < PTL_AVG_VPT; ++v)
            if (i + v * stride < n_vec) ptl_average_one(frames, n, magic, out, i + v * stride);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_px & 3)) ptl_average_tail_pixel(frames, n, magic, reinterpret_cast<unsigned int*>(out), n_vec * 4 + threadIdx.x);
}

extern "C" __global__ void __launch_bounds__(256)
ptl_average_images_kernel(ptl_frame_list frames, int n, ptl_u32x4* __restrict__ out, long n_px) {
    ptl_average_all(frames, n, out, n_px);
}

// The reference takes any motion_blur_frames (src/main.rs:1766); beyond 64 the pointers no longer fit the kernel arguments.
extern "C" __global__ void __launch_bounds__(256)
ptl_average_images_table_kernel(const ptl_u32x4* const* table, int n, ptl_u32x4* __restrict__ out, long n_px) {
    ptl_average_all(ptl_frame_table{table}, n, out, n_px);
}
