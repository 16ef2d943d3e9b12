// tools/micro/fetch_calib.hip — calibrates rocprofv3's FETCH_SIZE on THIS repo's access patterns.
// Three kernels read a known number of bytes from a table far larger than the 256-MiB Infinity Cache:
//   k_calib_stream    wide coalesced read, 16 B per lane                (the pattern MI355X_MICROARCH.md calibrates: reports 1/2)
//   k_calib_gather16  one random 16-byte record per lane                (a lone quarter of a node / triangle record)
//   k_calib_gather64  one random 64-byte-aligned record per lane as 4 x 16 B loads  (k_bvh's node fetch)
//   k_calib_gather48  one random 48-byte record per lane as 3 x 16 B loads, 48-byte stride  (k_bvh's triangle fetch)
// Run under `rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib`; it prints the true bytes per kernel as JSON; FETCH_SIZE
// (KiB) per kernel / true bytes gives the factor tools/summarize_profile.py applies.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void k_calib_stream(const float4* t, size_t n16, float* sink)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const float4 v = t[i]; acc += v.x + v.w; }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_calib_gather16(const float4* t, uint32_t n16, uint32_t per_thread, float* sink)
{
    float acc = 0.0f;
    uint32_t h = mix((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 1u);
    for (uint32_t k = 0; k < per_thread; ++k) { h = mix(h + k); const float4 v = t[h % n16]; acc += v.x + v.w; }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_calib_gather64(const float4* t, uint32_t n64, uint32_t per_thread, float* sink)
{
    float acc = 0.0f;
    uint32_t h = mix((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 7u);
    for (uint32_t k = 0; k < per_thread; ++k) {
        h = mix(h + k);
        const float4* p = t + (size_t)(h % n64) * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_calib_gather48(const float4* t, uint32_t n48, uint32_t per_thread, float* sink)
{
    float acc = 0.0f;
    uint32_t h = mix((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 13u);
    for (uint32_t k = 0; k < per_thread; ++k) {
        h = mix(h + k);
        const float4* p = t + (size_t)(h % n48) * 3;
        const float4 a = p[0], b = p[1], c = p[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) *sink = acc;
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 3072;   // table size: 12 x the Infinity Cache
    const size_t bytes = mib << 20;
    float4* t = nullptr; float* sink = nullptr;
    CHECK(hipMalloc((void**)&t, bytes));
    CHECK(hipMalloc((void**)&sink, 4));
    CHECK(hipMemset(t, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const uint32_t blocks = 256 * 8, threads = 256, per = 64;
    const size_t n16 = bytes / 16;
    const double gathers = (double)blocks * threads * per;
    hipLaunchKernelGGL(k_calib_stream, dim3(blocks), dim3(threads), 0, 0, t, n16, sink);
    hipLaunchKernelGGL(k_calib_gather16, dim3(blocks), dim3(threads), 0, 0, t, (uint32_t)(n16 > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : n16), per, sink);
    hipLaunchKernelGGL(k_calib_gather64, dim3(blocks), dim3(threads), 0, 0, t, (uint32_t)(bytes / 64), per, sink);
    hipLaunchKernelGGL(k_calib_gather48, dim3(blocks), dim3(threads), 0, 0, t, (uint32_t)(bytes / 48), per, sink);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    std::printf("{\"table_mib\": %zu, \"true_bytes\": {\"k_calib_stream\": %.0f, \"k_calib_gather16\": %.0f, \"k_calib_gather64\": %.0f, \"k_calib_gather48\": %.0f}, "
                "\"records\": %.0f, \"note\": \"gather16: bytes the lanes asked for (16 B each); a 64-byte line per record is 4x that\"}\n",
                mib, (double)bytes, gathers * 16, gathers * 64, gathers * 48, gathers);
    return 0;
}
