// fb_store.hip -- the framebuffer store pattern of ptl_render_kernel in isolation.
//
// BASELINE.json asks for ">= 60 % of the HBM roofline on the framebuffer write".  Inside the
// trace kernel the store is ~0.1 % of the time (the kernel is FP32-VALU-bound),
// so this micro-benchmark measures what the store *instruction pattern* sustains when nothing
// else runs: the same 32x8 block / 8x8-tile-per-wave mapping, RGBA8 through the LDS transpose
// (two 128-byte rows per wave) and RGBA32F as one float4 per lane (128 bytes per tile row).
#include <hip/hip_runtime.h>

extern "C" __global__ void __launch_bounds__(256)
ptl_fb_store_rgba8(unsigned int* __restrict__ out, int width, int height, unsigned int seed) {
    __shared__ unsigned int tile[8][32 + 1];
    const int t = (int)threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int lx = wave * 8 + (lane & 7), ly = lane >> 3;
    tile[ly][lx] = seed ^ (unsigned)((blockIdx.y * 8 + ly) * width + blockIdx.x * 32 + lx);
    __syncthreads();
    const int row = t >> 5, col = t & 31;
    const int gx = (int)blockIdx.x * 32 + col, gy = (int)blockIdx.y * 8 + row;
    if (gx < width && gy < height) out[(long)gy * width + gx] = tile[row][col];
}

extern "C" __global__ void __launch_bounds__(256)
ptl_fb_store_rgba32f(float4* __restrict__ out, int width, int height, float seed) {
    const int t = (int)threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int px = (int)blockIdx.x * 32 + wave * 8 + (lane & 7);
    const int py = (int)blockIdx.y * 8 + (lane >> 3);
    if (px < width && py < height) out[(long)py * width + px] = make_float4(seed, (float)px, (float)py, 1.0f);
}

// Same bytes with the textbook streaming pattern (grid-stride, 16 B per lane, fully linear),
// as the upper reference for the two kernels above.
extern "C" __global__ void __launch_bounds__(256)
ptl_fb_store_linear(float4* __restrict__ out, long n_vec, float seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long)gridDim.x * blockDim.x)
        out[i] = make_float4(seed, (float)i, 0.0f, 1.0f);
}
