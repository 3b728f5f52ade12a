// mfma_probe.hip -- can the idle matrix pipe take over `mat4 * vec4` without changing a bit?
//
// The numerics contract (portal_amd/csrc/device/ptl_glsl.h, `operator*(mat4, vec4)`) fixes a matrix-vector product as
//     y_i = fma(M[3][i], v.w, fma(M[2][i], v.z, fma(M[1][i], v.y, M[0][i] * v.x)))
// `v_mfma_f32_4x4x1_16b_f32` computes, in each of its 16 blocks of four lanes, D[i][j] = A[i] * B[j] + C[i][j].  Four of
// them chained over k with A_k[i] = M[k][i] (the same matrix in every block), B_k[j] = component k of lane j's own vector
// and C = -0.0f give lane j its own y in its four result registers.  This program answers, on the GPU:
//   layout     which (A lane, B lane) pair lands in which (lane, register) of D
//   exact      is the chain bit-equal to the contract's fmaf chain (random / denormal / huge / signed zero / inf / NaN)
//   exec       what an MFMA does under a partial EXEC mask: does it read A from inactive lanes, does it write D there
//   rate       cycles per instruction of the MFMA chain, of the FMA chain, and of both together (do the pipes overlap)
// Build: hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize tools/mfma_probe.hip -o /tmp/mfma_probe ; run on an MI355X.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(2);                                                                 \
        }                                                                                 \
    } while (0)

__device__ inline f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// ---- layout ------------------------------------------------------------------------------------------------------
__global__ void layout_kernel(float* out) {  // out[lane * 4 + reg]
    const int lane = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    f4 d = mfma4((float)(lane + 1), (float)(1000 * (lane + 1)), c);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}

// ---- exactness ---------------------------------------------------------------------------------------------------
// mats: n_mat matrices, column-major (GLSL), vecs: one vec4 per (matrix, lane).  One wave per matrix.
__global__ void exact_kernel(const float* mats, const float* vecs, float* y_fma, float* y_mfma) {
    const int lane = threadIdx.x, m = blockIdx.x;
    const float* M = mats + 16 * m;  // M[k*4+i] = column k, row i
    const float* v = vecs + 4 * (m * 64 + lane);
    float x = v[0], y = v[1], z = v[2], w = v[3];
    float r[4];
    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(M[12 + i], w, __builtin_fmaf(M[8 + i], z, __builtin_fmaf(M[4 + i], y, M[i] * x)));
    const int i = lane & 3;
    f4 c = {-0.f, -0.f, -0.f, -0.f};
    c = mfma4(M[0 + i], x, c);
    c = mfma4(M[4 + i], y, c);
    c = mfma4(M[8 + i], z, c);
    c = mfma4(M[12 + i], w, c);
    for (int q = 0; q < 4; ++q) {
        y_fma[4 * (m * 64 + lane) + q] = r[q];
        y_mfma[4 * (m * 64 + lane) + q] = c[q];
    }
}

// ---- EXEC --------------------------------------------------------------------------------------------------------
// mask: lanes that execute the MFMA.  A is valid (= lane&3 + 1) in the active lanes and `poison` in the others;
// B = 10 * (lane + 1); D starts as a sentinel in every lane.
__global__ void exec_kernel(unsigned long long mask, float poison, float* out) {
    const int lane = threadIdx.x;
    const bool active = (mask >> lane) & 1ull;
    float a = active ? (float)((lane & 3) + 1) : poison;
    float b = 10.0f * (float)(lane + 1);
    f4 d = {-777.f, -777.f, -777.f, -777.f};
    asm volatile("" : "+v"(a), "+v"(b), "+v"(d));
    if (active) {
        f4 c = {0.f, 0.f, 0.f, 0.f};
        d = mfma4(a, b, c);
    }
    asm volatile("" : "+v"(d));
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}

// ---- A operand fetched from LDS under a forced full EXEC mask -------------------------------------------------------
// What the tracer needs: the matrix rows come from LDS into ALL lanes although only some lanes are active.
__global__ void lds_full_exec_kernel(unsigned long long mask, const float* mat, const float* vecs, float* out) {
    __shared__ __attribute__((aligned(16))) float rows[16];  // row-major: rows[i*4+k] = M[k][i]
    __shared__ unsigned int lane_addr[64];
    const int lane = threadIdx.x;
    if (lane < 16) rows[(lane & 3) * 4 + (lane >> 2)] = mat[lane];
    lane_addr[lane] = (unsigned)(lane & 3) * 16u;
    __syncthreads();
    const bool active = (mask >> lane) & 1ull;
    float x = vecs[4 * lane + 0], y = vecs[4 * lane + 1], z = vecs[4 * lane + 2], w = vecs[4 * lane + 3];
    f4 d = {-777.f, -777.f, -777.f, -777.f};
    if (active) {
        f4 a;
        unsigned long long saved;
        unsigned int addr;
        const unsigned int table = (unsigned int)(uintptr_t)lane_addr;  // LDS byte address of the per-lane offset table
        const unsigned int base = (unsigned int)(uintptr_t)rows;
        asm volatile(
            "s_mov_b64 %[saved], exec\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_mov_b32 m0, %[table]\n\t"
            "s_nop 0\n\t"  // SALU write of M0 -> LDS add-TID instruction: one wait state
            "ds_read_addtid_b32 %[addr]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 %[addr], %[base], %[addr]\n\t"
            "ds_read_b128 %[a], %[addr]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b64 exec, %[saved]"
            : [a] "=&v"(a), [saved] "=&s"(saved), [addr] "=&v"(addr)
            : [table] "s"(table), [base] "s"(base)
            : "memory", "m0");
        f4 c = {-0.f, -0.f, -0.f, -0.f};
        c = mfma4(a[0], x, c);
        c = mfma4(a[1], y, c);
        c = mfma4(a[2], z, c);
        c = mfma4(a[3], w, c);
        d = c;
    }
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}

// ---- rate --------------------------------------------------------------------------------------------------------
// mode 1: FMA chains only, 2: MFMA chains only, 3: both.  Two independent mat*vec per iteration in each pipe.
template <int MODE>
__global__ void __launch_bounds__(512) rate_kernel(const float* mats, float* out, int iters, long long* cycles) {
    const int lane = threadIdx.x & 63;
    float M[16];
    for (int i = 0; i < 16; ++i) { M[i] = mats[i]; asm volatile("" : "+v"(M[i])); }  // VGPR operands: v_fma v, v, v, v
    float a0 = mats[0 + (lane & 3)], a1 = mats[4 + (lane & 3)], a2 = mats[8 + (lane & 3)], a3 = mats[12 + (lane & 3)];
    float x = 1.0f + 1e-3f * threadIdx.x, y = 0.5f, z = 0.25f, w = 1.0f;
    float p = 0.3f + 1e-3f * threadIdx.x, q = 0.7f, s = 0.1f, t = 0.0f;
    f4 u = {x, y, z, w}, v = {p, q, s, t};
    long long t0 = __builtin_readcyclecounter();
    const bool do_fma = (MODE & 1) || (MODE == 4 && ((threadIdx.x >> 8) & 1) == 0);
    const bool do_mfma = (MODE & 2) || (MODE == 4 && ((threadIdx.x >> 8) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_fma) {
            float r0 = __builtin_fmaf(M[12], w, __builtin_fmaf(M[8], z, __builtin_fmaf(M[4], y, M[0] * x)));
            float r1 = __builtin_fmaf(M[13], w, __builtin_fmaf(M[9], z, __builtin_fmaf(M[5], y, M[1] * x)));
            float r2 = __builtin_fmaf(M[14], w, __builtin_fmaf(M[10], z, __builtin_fmaf(M[6], y, M[2] * x)));
            float r3 = __builtin_fmaf(M[15], w, __builtin_fmaf(M[11], z, __builtin_fmaf(M[7], y, M[3] * x)));
            float s0 = __builtin_fmaf(M[12], t, __builtin_fmaf(M[8], s, __builtin_fmaf(M[4], q, M[0] * p)));
            float s1 = __builtin_fmaf(M[13], t, __builtin_fmaf(M[9], s, __builtin_fmaf(M[5], q, M[1] * p)));
            float s2 = __builtin_fmaf(M[14], t, __builtin_fmaf(M[10], s, __builtin_fmaf(M[6], q, M[2] * p)));
            float s3 = __builtin_fmaf(M[15], t, __builtin_fmaf(M[11], s, __builtin_fmaf(M[7], q, M[3] * p)));
            x = r0; y = r1; z = r2; w = r3; p = s0; q = s1; s = s2; t = s3;
        }
        if (do_mfma) {
            f4 c = {-0.f, -0.f, -0.f, -0.f}, e = {-0.f, -0.f, -0.f, -0.f};
            c = mfma4(a0, u[0], c); e = mfma4(a0, v[0], e);
            c = mfma4(a1, u[1], c); e = mfma4(a1, v[1], e);
            c = mfma4(a2, u[2], c); e = mfma4(a2, v[2], e);
            c = mfma4(a3, u[3], c); e = mfma4(a3, v[3], e);
            u = c; v = e;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w + p + q + s + t + u[0] + u[1] + u[2] + u[3] + v[0] + v[1] + v[2] + v[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

template <int MODE>
static void run_rate(const float* d_mats, float* d_out, long long* d_cyc, int blocks, int iters, const char* name) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, d_mats, d_out, iters, d_cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, d_mats, d_out, iters, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc = 0;
    CHECK(hipMemcpy(&cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost));
    const double waves_per_simd = (double)blocks * 8 / 1024.0;
    const int fma_per_iter = (MODE & 1) ? 32 : (MODE == 4 ? 16 : 0), mfma_per_iter = (MODE & 2) ? 8 : (MODE == 4 ? 4 : 0);  // MODE 4: per-wave average
    std::printf("{\"probe\": \"rate\", \"mode\": \"%s\", \"blocks\": %d, \"waves_per_simd\": %.1f, \"iters\": %d, \"ms\": %.4f, "
                "\"cycles_per_iter_wave0\": %.2f, \"fma_per_iter\": %d, \"mfma_per_iter\": %d, "
                "\"simd_ns_per_iter_per_wave\": %.3f}\n",
                name, blocks, waves_per_simd, iters, ms, (double)cyc / iters, fma_per_iter, mfma_per_iter,
                ms * 1e6 / iters / waves_per_simd);
}

int main() {
    // layout
    float* d_out;
    CHECK(hipMalloc(&d_out, 1 << 24));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<float> lay(256);
    CHECK(hipMemcpy(lay.data(), d_out, 1024, hipMemcpyDeviceToHost));
    int layout_ok = 1;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int a_lane = (lane & ~3) + r, b_lane = lane;  // expectation: D[reg r] of lane L = A[block(L), r] * B[L]
            const float want = (float)(a_lane + 1) * (float)(1000 * (b_lane + 1));
            if (lay[lane * 4 + r] != want) layout_ok = 0;
        }
    std::printf("{\"probe\": \"layout\", \"d_reg_r_of_lane_L_is_A_of_lane_4floor(L/4)+r_times_B_of_lane_L\": %s, \"lane5\": [%.0f, %.0f, %.0f, %.0f]}\n",
                layout_ok ? "true" : "false", lay[20], lay[21], lay[22], lay[23]);

    // exactness
    const int n_mat = 4096;
    std::mt19937_64 rng(20260925);
    std::vector<float> mats(16 * n_mat), vecs(4 * 64 * n_mat);
    auto special = [&](int cls) -> float {
        std::uniform_real_distribution<float> u(-1.f, 1.f);
        switch (cls) {
            case 0: return u(rng) * 4.f;                                                          // ordinary
            case 1: return from_bits((uint32_t)(rng() & 0x807fffffu));                            // subnormal / zero, either sign
            case 2: return std::ldexp(u(rng), (int)(rng() % 250) - 125);                          // whole exponent range
            case 3: return from_bits((uint32_t)rng());                                            // raw bits (NaN, inf included)
            case 4: { const float t[8] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, 3.4028235e38f, 1.17549435e-38f}; return t[rng() % 8]; }
            default: return std::ldexp(u(rng), -126 + (int)(rng() % 30) - 15);                    // around the subnormal boundary
        }
    };
    for (int m = 0; m < n_mat; ++m) {
        const int cls_m = (m / 6) % 6, cls_v = m % 6;
        for (int i = 0; i < 16; ++i) mats[16 * m + i] = special(cls_m);
        for (int i = 0; i < 256; ++i) vecs[256 * m + i] = special(cls_v);
    }
    float *d_m, *d_v, *d_yf, *d_ym;
    CHECK(hipMalloc(&d_m, mats.size() * 4)); CHECK(hipMalloc(&d_v, vecs.size() * 4));
    CHECK(hipMalloc(&d_yf, vecs.size() * 4)); CHECK(hipMalloc(&d_ym, vecs.size() * 4));
    CHECK(hipMemcpy(d_m, mats.data(), mats.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_v, vecs.data(), vecs.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(exact_kernel, dim3(n_mat), dim3(64), 0, 0, d_m, d_v, d_yf, d_ym);
    std::vector<float> yf(vecs.size()), ym(vecs.size());
    CHECK(hipMemcpy(yf.data(), d_yf, yf.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ym.data(), d_ym, ym.size() * 4, hipMemcpyDeviceToHost));
    long mism[6][6] = {}, nan_payload = 0, total = 0, n_nan = 0, n_sub = 0, host_mism = 0;
    for (size_t i = 0; i < yf.size(); ++i) {
        const int m = (int)(i / 256);
        ++total;
        if (std::isnan(yf[i]) && std::isnan(ym[i])) { ++n_nan; if (bits(yf[i]) != bits(ym[i])) ++nan_payload; continue; }
        if (yf[i] != 0.f && std::fabs(yf[i]) < 1.17549435e-38f) ++n_sub;
        if (bits(yf[i]) != bits(ym[i])) ++mism[(m / 6) % 6][m % 6];
        // the contract evaluated on the host (x86 fmaf is the IEEE operation)
        const float* M = &mats[16 * m];
        const float* v = &vecs[4 * (i / 4)];
        const int q = (int)(i % 4);
        const float h = std::fmaf(M[12 + q], v[3], std::fmaf(M[8 + q], v[2], std::fmaf(M[4 + q], v[1], M[q] * v[0])));
        if (!(std::isnan(h) && std::isnan(yf[i])) && bits(h) != bits(yf[i])) ++host_mism;
    }
    long all_mism = 0;
    for (auto& row : mism) for (long c : row) all_mism += c;
    std::printf("{\"probe\": \"exact\", \"results\": %ld, \"mfma_vs_fma_bit_mismatches\": %ld, \"both_nan\": %ld, \"nan_payload_differs\": %ld, "
                "\"subnormal_results\": %ld, \"gpu_fma_vs_host_fmaf_mismatches\": %ld, \"by_class_matrix_x_vector\": [",
                total, all_mism, n_nan, nan_payload, n_sub, host_mism);
    for (int a = 0; a < 6; ++a) { std::printf("%s[", a ? ", " : ""); for (int b = 0; b < 6; ++b) std::printf("%s%ld", b ? ", " : "", mism[a][b]); std::printf("]"); }
    std::printf("]}\n");

    // EXEC
    const unsigned long long masks[4] = {0xffffffffffffffffull, 0x00000000000000f0ull, 0x5555555555555555ull, 0x0000000000000020ull};
    for (unsigned long long mask : masks) {
        hipLaunchKernelGGL(exec_kernel, dim3(1), dim3(64), 0, 0, mask, 5000.0f, d_out);
        CHECK(hipMemcpy(lay.data(), d_out, 1024, hipMemcpyDeviceToHost));
        int active_ok = 1, active_saw_poison = 0, inactive_written = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const bool act = (mask >> lane) & 1ull;
                const float got = lay[lane * 4 + r];
                const float b = 10.0f * (lane + 1);
                if (act) {
                    const bool a_active = (mask >> ((lane & ~3) + r)) & 1ull;
                    if (got == (float)(r + 1) * b) continue;
                    active_ok = 0;
                    if (!a_active && got == 5000.0f * b) active_saw_poison = 1;
                } else if (got != -777.f) inactive_written = 1;
            }
        std::printf("{\"probe\": \"exec\", \"mask\": \"%016llx\", \"active_lanes_all_as_if_A_valid_everywhere\": %s, "
                    "\"active_lanes_read_A_from_inactive_lanes\": %s, \"inactive_lanes_D_overwritten\": %s, \"lane5\": [%.0f, %.0f, %.0f, %.0f], \"lane4\": [%.0f, %.0f, %.0f, %.0f]}\n",
                    mask, active_ok ? "true" : "false", active_saw_poison ? "true" : "false", inactive_written ? "true" : "false",
                    lay[20], lay[21], lay[22], lay[23], lay[16], lay[17], lay[18], lay[19]);
    }

    // LDS rows under a forced full EXEC
    for (unsigned long long mask : masks) {
        hipLaunchKernelGGL(lds_full_exec_kernel, dim3(1), dim3(64), 0, 0, mask, d_m, d_v, d_out);
        CHECK(hipMemcpy(lay.data(), d_out, 1024, hipMemcpyDeviceToHost));
        int ok = 1, untouched = 1;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const bool act = (mask >> lane) & 1ull;
                if (act) { if (bits(lay[lane * 4 + r]) != bits(yf[lane * 4 + r])) ok = 0; }
                else if (lay[lane * 4 + r] != -777.f) untouched = 0;
            }
        std::printf("{\"probe\": \"lds_rows_full_exec\", \"mask\": \"%016llx\", \"active_lanes_bit_equal_to_fma_chain\": %s, \"inactive_lanes_keep_their_value\": %s}\n",
                    mask, ok ? "true" : "false", untouched ? "true" : "false");
    }

    // rate
    long long* d_cyc;
    CHECK(hipMalloc(&d_cyc, 8));
    std::vector<float> ident = {0.5f, 0.1f, 0, 0, -0.1f, 0.5f, 0, 0, 0, 0, 0.5f, 0, 0.01f, 0.02f, 0.03f, 0.5f};
    CHECK(hipMemcpy(d_m, ident.data(), 64, hipMemcpyHostToDevice));
    for (int blocks : {256, 512, 1024}) {
        run_rate<1>(d_m, d_out, d_cyc, blocks, 20000, "fma");
        run_rate<2>(d_m, d_out, d_cyc, blocks, 20000, "mfma");
        run_rate<3>(d_m, d_out, d_cyc, blocks, 20000, "fma+mfma");
        run_rate<4>(d_m, d_out, d_cyc, blocks, 20000, "waves 0-3 fma, waves 4-7 mfma (one of each per SIMD)");
    }
    return 0;
}
