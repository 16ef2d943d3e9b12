// tools/micro/valu_rate.hip — issue rate of the vector instructions k_bvh's node step is made of, with 1..8 waves per SIMD.
// Answers what "VALU busy" in profiles/*_pmc.csv is a share OF: how many cycles one wave64 instruction holds its SIMD when
// several waves are resident (MI355X_MICROARCH.md: v_fma_f32 4 cycles for one wave alone).  Each wave runs a loop of 32
// independent instructions of one kind; s_memtime stamps the loop.  Prints cycles per instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { OP_FMA = 0, OP_PK_FMA, OP_PERM, OP_CNDMASK, OP_MIN3, OP_CMP, OP_PK_FMA_F16, OP_MINU, OP_CND64, OP_CND_INDEP, OP_CMP64, OP_ADD, OP_CND_NOCHAIN64, OP_COUNT };
static const char* names[OP_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_perm_b32", "v_cndmask_b32", "v_min3_f32", "v_cmp_lt_f32", "v_pk_fma_f16", "v_min_u32", "v_cndmask_b32_e64 (sgpr mask, chained)", "v_cndmask_b32 (vcc, dest not a source)", "v_cmp_lt_f32_e64 -> sgpr pair", "v_add_u32", "v_cndmask_b32_e64 (sgpr mask, dest not a source)"};

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint64_t* cycles, float* sink, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = 1.0000001f, c = 1e-9f;
    const v2 pb = {b, b}, pc = {c, c};
    uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == OP_FMA) {
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (OP == OP_PK_FMA) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (OP == OP_PERM) {
            REP8(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x4B000000u), "v"(0x07040100u));)
        } else if (OP == OP_CNDMASK) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if (OP == OP_MIN3) {
            REP8(asm volatile("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (OP == OP_CMP) {
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if (OP == OP_PK_FMA_F16) {
            REP8(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x3C003C00u), "v"(0x00010001u));)
        } else if (OP == OP_CND64) {
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (OP == OP_CND_INDEP) {
            REP8(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        } else if (OP == OP_CMP64) {
            REP8(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (OP == OP_ADD) {
            REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x11u));)
        } else if (OP == OP_CND_NOCHAIN64) {
            REP8(asm volatile("v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %4, %5, s[20:21]\n v_cndmask_b32_e64 %2, %4, %5, s[22:23]\n v_cndmask_b32_e64 %3, %4, %5, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else {
            REP8(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0xFFFFFFF0u));)
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    const float s = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y + (float)(u0 ^ u1 ^ u2 ^ u3);
    if (s == 123.456f) *sink = s;
}

template <int OP>
static int run(int waves_per_simd, uint64_t* d_cycles, float* sink)
{
    const int iters = 4096, per_iter = 32;
    // one 256-thread block = 4 waves = one wave per SIMD of a CU; W blocks per CU -> W waves per SIMD (256 CUs)
    const int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_cycles, sink, iters);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (auto v : h) sum += (double)v;
    const double per_wave = sum / h.size() / ((double)iters * per_iter);   // cycles one wave needs per instruction
    std::printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_wave\": %.3f, \"cycles_per_inst_per_simd\": %.3f}\n",
                names[OP], waves_per_simd, per_wave, per_wave / waves_per_simd);
    return 0;
}

int main()
{
    uint64_t* d_cycles = nullptr; float* sink = nullptr;
    CHECK(hipMalloc((void**)&d_cycles, 8ull * 256 * 8 * 4));
    CHECK(hipMalloc((void**)&sink, 4));
    for (int w : {1, 2, 4, 8}) {
        if (run<OP_FMA>(w, d_cycles, sink) || run<OP_PK_FMA>(w, d_cycles, sink) || run<OP_PERM>(w, d_cycles, sink) || run<OP_CNDMASK>(w, d_cycles, sink) ||
            run<OP_MIN3>(w, d_cycles, sink) || run<OP_CMP>(w, d_cycles, sink) || run<OP_PK_FMA_F16>(w, d_cycles, sink) || run<OP_MINU>(w, d_cycles, sink) ||
            run<OP_CND64>(w, d_cycles, sink) || run<OP_CND_INDEP>(w, d_cycles, sink) || run<OP_CMP64>(w, d_cycles, sink) || run<OP_ADD>(w, d_cycles, sink) || run<OP_CND_NOCHAIN64>(w, d_cycles, sink)) return 1;
    }
    return 0;
}
