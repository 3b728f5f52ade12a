// fb_store.hip -- the framebuffer store pattern of ptl_render_kernel in isolation.
//
// BASELINE.json asks for ">= 60 % of the HBM roofline on the framebuffer write".  Inside the
// trace kernel the store is well under 1 % of the time (the kernel is FP32-VALU-bound, DESIGN.md
// section 2.1), so this micro-benchmark measures what the store *instruction pattern* sustains when
// nothing else runs.  Same launch geometry and entry signature as the trace kernel (it is loaded
// through the same C ABI, ptl_kernel_compile), selected with -DPTL_FB_VARIANT=
//   0  RGBA8 through the 32x8 LDS transpose: two 128-byte rows per wave       (what the renderer does)
//   1  RGBA32F, one float4 per lane: 128 bytes per 8x8-tile row                (the parity buffer)
//   2  RGBA8 straight from the 8x8 tile: 32-byte row segments                  (no transpose, for contrast)
//   3  linear streaming store, 16 B per lane, grid-stride                      (upper reference)
//   4  RGBA8 through the 32x8 LDS transpose, 16 B per lane: wave 0 of the workgroup stores the block's eight 128-byte rows
//   5  the same store from a PERSISTENT grid (the launch covers `width/32 x height/8` workgroups of a small fake frame; the real frame
//      size comes through the uniforms real_w_u / real_h_u and every workgroup walks over blocks): launch ramp and tail taken out
//   6  linear 16 B per lane with nontemporal stores from a persistent grid
//   7  nothing (the floor of the timing method: one empty launch between two events)
// Built ahead of time by `make kernels` (hipcc --genco, gfx950) as a compile check and at run time
// by tools/fb_store_bench.py through hiprtc.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#ifndef PTL_FB_VARIANT
#define PTL_FB_VARIANT 0
#endif

namespace glsl {
struct ptl_uniform_block { int seed_u; int pad_u; int real_w_u; int real_h_u; };
__constant__ ptl_uniform_block ptl_u;
}  // namespace glsl

extern "C" __global__ void __launch_bounds__(256)
ptl_render_kernel(unsigned int* __restrict__ out_rgba8, float* __restrict__ out_rgba32f, int width, int height,
                  int rb_phase, int rb_stride, unsigned long long* __restrict__ segments) {
    const int t = (int)threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int lx = wave * 8 + (lane & 7), ly = lane >> 3;
    const int px = (int)blockIdx.x * 32 + lx;
    const int py = (int)blockIdx.y * 8 + ly;
    const unsigned int v = (unsigned)glsl::ptl_u.seed_u ^ (unsigned)(py * width + px);
#if PTL_FB_VARIANT == 0
    __shared__ unsigned int tile[8][32 + 1];
    tile[ly][lx] = v;
    __syncthreads();
    const int row = t >> 5, col = t & 31;
    const int gx = (int)blockIdx.x * 32 + col, gy = (int)blockIdx.y * 8 + row;
    if (gx < width && gy < height) out_rgba8[(long)gy * width + gx] = tile[row][col];
#elif PTL_FB_VARIANT == 1
    if (px < width && py < height)
        reinterpret_cast<float4*>(out_rgba32f)[(long)py * width + px] = make_float4((float)v, (float)px, (float)py, 1.0f);
#elif PTL_FB_VARIANT == 2
    if (px < width && py < height) out_rgba8[(long)py * width + px] = v;
#elif PTL_FB_VARIANT == 4
    __shared__ __attribute__((aligned(16))) unsigned int tile[8][32];
    tile[ly][lx] = v;
    __syncthreads();
    if (wave == 0) {  // 64 lanes x 16 B = the block's 1 KB: lane -> (row, 16-byte segment)
        const int row = lane >> 3, seg = lane & 7;
        const int gx = (int)blockIdx.x * 32 + seg * 4, gy = (int)blockIdx.y * 8 + row;
        if (gx + 3 < width && gy < height) *reinterpret_cast<uint4*>(out_rgba8 + (long)gy * width + gx) = *reinterpret_cast<const uint4*>(&tile[row][seg * 4]);
    }
#elif PTL_FB_VARIANT == 5
    __shared__ __attribute__((aligned(16))) unsigned int tile[8][32];
    const int W = glsl::ptl_u.real_w_u, H = glsl::ptl_u.real_h_u;
    const int bx_n = W / 32, by_n = H / 8;
    const int n_blocks = bx_n * by_n, n_wg = (int)(gridDim.x * gridDim.y);
    for (int b = (int)(blockIdx.y * gridDim.x + blockIdx.x); b < n_blocks; b += n_wg) {
        const int bx = b % bx_n, by = b / bx_n;
        tile[ly][lx] = v + (unsigned)b;
        __syncthreads();
        if (wave == 0) {
            const int row = lane >> 3, seg = lane & 7;
            *reinterpret_cast<uint4*>(out_rgba8 + (long)(by * 8 + row) * W + bx * 32 + seg * 4) = *reinterpret_cast<const uint4*>(&tile[row][seg * 4]);
        }
        __syncthreads();
    }
#elif PTL_FB_VARIANT == 6
    const long n_vec = (long)glsl::ptl_u.real_w_u * glsl::ptl_u.real_h_u / 4;
    const long stride = (long)gridDim.x * gridDim.y * 256;
    for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t; i < n_vec; i += stride) {
        unsigned int* p = out_rgba8 + 4 * i;
        __builtin_nontemporal_store(v, p);
        __builtin_nontemporal_store(v + 1, p + 1);
        __builtin_nontemporal_store(v + 2, p + 2);
        __builtin_nontemporal_store(v + 3, p + 3);
    }
#elif PTL_FB_VARIANT == 7
    if (v == 0x7fffffffu && width < 0) out_rgba8[0] = v;  // (never true: keeps `v` alive, stores nothing)
#else
    const long n_vec = (long)width * height / 4;  // RGBA8 frame as uint4
    const long stride = (long)gridDim.x * gridDim.y * 256;
    for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t; i < n_vec; i += stride)
        reinterpret_cast<uint4*>(out_rgba8)[i] = make_uint4(v, v + 1, v + 2, v + 3);
#endif
}
