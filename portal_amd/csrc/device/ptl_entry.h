// ptl_entry.h -- launchable entry points appended after the filled-in trace template.
//
// Takes the place of the reference's full-screen quad draw (src/main.rs:1424-1425): one
// invocation of glsl::shade_pixel per framebuffer pixel.
//
// gfx950 mapping: a 256-thread workgroup renders a 32x8 pixel block as four 8x8 tiles,
// one per 64-lane wavefront, so the lanes of a wave trace a compact bundle of rays and
// the bounce loop's exit test is wave-uniform as soon as all 64 rays have finished
// (the loop back-edge is a scalar branch on `exec == 0`; no explicit ballot is needed,
// the compiler's structurizer produces exactly that).  Row blocks (8 rows) are the
// sharding unit: a launch renders blocks phase, phase+stride, ... so N GPUs interleave.

#if defined(PTL_NO_TELEPORT_ENTRY) && !defined(PTL_NO_DERIVE_ENTRY)
#define PTL_NO_DERIVE_ENTRY  // a hand-written kernel without a scene has neither a teleport query nor derived uniforms
#endif

#if PTL_DEVICE_BUILD

#ifdef PTL_WAVES_PER_EU
#define PTL_LAUNCH_BOUNDS __launch_bounds__(256, PTL_WAVES_PER_EU)
#else
#define PTL_LAUNCH_BOUNDS __launch_bounds__(256)
#endif

// A build can be split in two modules (host/kernel.cpp `ptl_kernel::split`): -DPTL_RENDER_MODULE leaves the camera-teleport entry out,
// -DPTL_TELEPORT_MODULE the render entry; the prologue entry is in both.  Without either define the module has all three.
#if !defined(PTL_TELEPORT_MODULE)
extern "C" __global__ void PTL_LAUNCH_BOUNDS
ptl_render_kernel(unsigned int* __restrict__ out_rgba8,   // packed rows of this shard, or null
                  float* __restrict__ out_rgba32f,        // same, 4 floats per pixel, or null
                  int width, int height,                   // full frame size
                  int rb_phase, int rb_stride,             // row-block interleave
                  unsigned long long* __restrict__ segment_counter,
                  int in_place) {                          // 1: out_* address the full frame (maybe a peer GPU's), rows go where they belong
#ifndef PTL_DIRECT_STORE
    __shared__ unsigned int tile[8][32 + 1];
#endif
    const int t = (int)threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int lx = wave * 8 + (lane & 7);
    const int ly = lane >> 3;
#ifdef PTL_XCD_SWIZZLE
    // Experiment (measured: no gain): workgroups are dealt round-robin to the 8 XCDs; this remap gives every XCD a
    // contiguous band of the frame instead of every 8th block, for L2 locality of texture taps.
    int block_x = (int)blockIdx.x, block_y = (int)blockIdx.y;
    {
        const int total = (int)(gridDim.x * gridDim.y);
        if ((total & 7) == 0) {
            const int id = block_y * (int)gridDim.x + block_x;
            const int remapped = (id & 7) * (total >> 3) + (id >> 3);
            block_x = remapped % (int)gridDim.x;
            block_y = remapped / (int)gridDim.x;
        }
    }
#else
    const int block_x = (int)blockIdx.x, block_y = (int)blockIdx.y;
#endif
    const int local_block = block_y;
    const int block_px = (int)blockDim.x >> 3;  // 32 pixels per row block for the shipped 256-thread workgroup (experiments launch 64 / 128)
    const int px = block_x * block_px + lx;
    const int py = (rb_phase + local_block * rb_stride) * 8 + ly;
    const bool live = px < width && py < height;

#ifdef PTL_COUNT_SEGMENTS
    ptl_segments_lds[t] = 0u;
#endif
#if defined(PTL_MATERIAL_TABLE) && PTL_MATERIAL_TABLE == 1
    {  // stage the Simple materials' constants in LDS once per workgroup (codegen.cpp, materials): two ds_read_b128 per lane fetch a material's nine values
        for (int k = (int)threadIdx.x; k < PTL_MATERIAL_TABLE_WORDS; k += (int)blockDim.x) glsl::ptl_material_table[k] = glsl::ptl_material_table_init[k];
        __syncthreads();
    }
#endif
#ifdef PTL_UNIFORMS_IN_LDS
    {  // stage the scene constants (portal matrices, uniforms) in LDS once per workgroup
        const unsigned int* src = reinterpret_cast<const unsigned int*>(&glsl::ptl_u);
        unsigned int* dst = reinterpret_cast<unsigned int*>(&glsl::ptl_lds_u);
        for (int i = t; i < (int)(sizeof(glsl::ptl_uniform_block) / 4); i += 256) dst[i] = src[i];
        __syncthreads();
    }
#endif
    glsl::vec4 c = glsl::vec4(0.0f);
    if (live) c = glsl::shade_pixel(glsl::vec2((float)px + 0.5f, (float)py + 0.5f));

    const int out_block = in_place ? rb_phase + local_block * rb_stride : local_block;  // wave-uniform
    const long shard_row = (long)out_block * 8 + ly;
    if (out_rgba32f != nullptr && live) {
        float4 v = make_float4(c.x, c.y, c.z, c.w);
        *reinterpret_cast<float4*>(out_rgba32f + 4 * (shard_row * width + px)) = v;  // 8 lanes x 16 B = one 128 B line per tile row
    }
#ifdef PTL_DIRECT_STORE
    if (out_rgba8 != nullptr && live) out_rgba8[shard_row * width + px] = glsl::pack_rgba8(c);  // 8 lanes x 4 B = 32 B per tile row
#else
    if (blockDim.x != 256) {  // narrower experimental workgroups: plain tile store
        if (out_rgba8 != nullptr && live) out_rgba8[shard_row * width + px] = glsl::pack_rgba8(c);
    } else if (out_rgba8 != nullptr) {
        tile[ly][lx] = glsl::pack_rgba8(c);
        __syncthreads();
        const int row = t >> 5, col = t & 31;  // each wave now owns two full 32-pixel rows = 2 x 128 B
        const int gx = block_x * 32 + col;
        const int gy = (rb_phase + local_block * rb_stride) * 8 + row;
        if (gx < width && gy < height) out_rgba8[((long)out_block * 8 + row) * width + gx] = tile[row][col];
    }
#endif
#ifdef PTL_COUNT_SEGMENTS
    if (segment_counter != nullptr) {
        unsigned int n = ptl_segments_lds[t];
        for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
        if (lane == 0) atomicAdd(segment_counter, (unsigned long long)n);
    }
#endif
}
#endif  // !PTL_TELEPORT_MODULE

// The camera-teleport query of src/main.rs:1361-1409: one thread instead of the reference's 2x3-pixel
// draw with float-in-RGBA8 packing.  (A hand-written kernel without a scene can opt out.)
#if !defined(PTL_NO_TELEPORT_ENTRY) && !defined(PTL_RENDER_MODULE)
extern "C" __global__ void __launch_bounds__(64) ptl_teleport_kernel(float* __restrict__ out6) {
#ifdef PTL_UNIFORMS_IN_LDS
    {
        const unsigned int* src = reinterpret_cast<const unsigned int*>(&glsl::ptl_u);
        unsigned int* dst = reinterpret_cast<unsigned int*>(&glsl::ptl_lds_u);
        for (int i = (int)threadIdx.x; i < (int)(sizeof(glsl::ptl_uniform_block) / 4); i += 64) dst[i] = src[i];
        __syncthreads();
    }
#endif
#ifdef PTL_COUNT_SEGMENTS
    ptl_segments_lds[threadIdx.x] = 0u;
#endif
#if defined(PTL_MATERIAL_TABLE) && PTL_MATERIAL_TABLE == 1
    {  // stage the Simple materials' constants in LDS once per workgroup (codegen.cpp, materials): two ds_read_b128 per lane fetch a material's nine values
        for (int k = (int)threadIdx.x; k < PTL_MATERIAL_TABLE_WORDS; k += (int)blockDim.x) glsl::ptl_material_table[k] = glsl::ptl_material_table_init[k];
        __syncthreads();
    }
#endif
    if (threadIdx.x == 0 && blockIdx.x == 0) glsl::teleport_external_ray_entry(out6);
}
#endif

// Prologue: evaluates the derived uniforms (ptl_tracer::derive) into the tail of the uniform block.  The host launches it on
// the render stream after every upload of the block; `block` is the address of ptl_u (a __constant__ object cannot be
// written through its own name).  One thread: a few dozen planes, ~100 instructions each.
#ifndef PTL_NO_DERIVE_ENTRY
extern "C" __global__ void __launch_bounds__(64) ptl_derive_kernel(glsl::ptl_uniform_block* block) {
#ifdef PTL_UNIFORMS_IN_LDS
    {
        const unsigned int* src = reinterpret_cast<const unsigned int*>(&glsl::ptl_u);
        unsigned int* dst = reinterpret_cast<unsigned int*>(&glsl::ptl_lds_u);
        for (int i = (int)threadIdx.x; i < (int)(sizeof(glsl::ptl_uniform_block) / 4); i += 64) dst[i] = src[i];
        __syncthreads();
    }
#endif
#ifdef PTL_COUNT_SEGMENTS
    ptl_segments_lds[threadIdx.x] = 0u;
#endif
    if (threadIdx.x == 0 && blockIdx.x == 0) glsl::derive_uniforms(block);
}
#endif

#else  // host build of the same source (oracle/host_build): rows [row_begin, row_end) of the frame

#include <cstdint>
#include <cstring>

extern "C" void* ptl_host_uniform_block(unsigned long* size) {
    if (size) *size = sizeof(glsl::ptl_u);
    return &glsl::ptl_u;
}

#ifndef PTL_NO_TELEPORT_ENTRY
extern "C" void ptl_host_teleport(float* out6) {
#ifndef PTL_NO_DERIVE_ENTRY
    glsl::derive_uniforms(&glsl::ptl_u);
#endif
    glsl::teleport_external_ray_entry(out6);
}
#else
extern "C" void ptl_host_teleport(float*) {}
#endif

// Renders the listed pixel rows x columns [col_begin, col_end) of a width x height frame into
// out_* (output row i = rows[i]) with `threads` OpenMP threads.  Returns the number of
// bounce-loop trips when compiled with PTL_COUNT_SEGMENTS, else 0.
extern "C" unsigned long long ptl_host_render(uint8_t* out_rgba8, float* out_rgba32f, int width, int height,
                                              const int* rows, int n_rows, int col_begin, int col_end, int threads) {
    unsigned long long segments = 0;
    (void)width;
    (void)height;
    (void)threads;
#ifndef PTL_NO_DERIVE_ENTRY
    glsl::derive_uniforms(&glsl::ptl_u);  // the prologue the device build runs as ptl_derive_kernel
#endif
    const int cols = col_end - col_begin;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : segments)
    for (int i = 0; i < n_rows; ++i) {
        const int py = rows[i];
#ifdef PTL_COUNT_SEGMENTS
        ptl_segments_tls = 0;
#endif
        for (int px = col_begin; px < col_end; ++px) {
            glsl::vec4 c = glsl::shade_pixel(glsl::vec2((float)px + 0.5f, (float)py + 0.5f));
            long idx = (long)i * cols + (px - col_begin);
            if (out_rgba32f) {
                float v[4] = {c.x, c.y, c.z, c.w};
                std::memcpy(out_rgba32f + 4 * idx, v, sizeof v);
            }
            if (out_rgba8) {
                uint32_t p = glsl::pack_rgba8(c);
                std::memcpy(out_rgba8 + 4 * idx, &p, 4);
            }
        }
#ifdef PTL_COUNT_SEGMENTS
        segments += ptl_segments_tls;
#endif
    }
    return segments;
}

#endif
