// Does `x % n` (24-bit x, run-time n) come out right on the device?   hipcc --offload-arch=gfx950 -O3 tools/urem_probe.hip -o tools/urem_probe && ./tools/urem_probe
// (Round 4: k_window_region's first class function, `(tag * 0x9e3779b1u >> 8) % n_pass`, lost keys on the GPU for n_pass = 11 - this probe asks the compiler's urem directly.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ void k_urem(const uint32_t* tags, uint32_t n_tags, uint32_t n, uint32_t* out, uint32_t* quot)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n24 = (n + 0x1fffu) >> 13;      // (as in the kernel: the compiler can see that the divisor fits 24 bits, and takes its float-reciprocal 24-bit expansion; the caller passes n * 8192)
    if (i < n_tags) { out[i] = (tags[i] * 0x9e3779b1u >> 8) % n24; quot[i] = (tags[i] * 0x9e3779b1u >> 8) / n24; }
}
int main()
{
    const uint32_t N = 1u << 22;
    std::vector<uint32_t> tags(N), got(N), gotq(N);
    uint64_t s = 0x243f6a8885a308d3ull;
    for (auto& t : tags) { s = s * 6364136223846793005ull + 1442695040888963407ull; t = (uint32_t)(s >> 32); }
    uint32_t *d_t, *d_o, *d_q;
    hipMalloc(&d_t, N * 4); hipMalloc(&d_o, N * 4); hipMalloc(&d_q, N * 4);
    hipMemcpy(d_t, tags.data(), N * 4, hipMemcpyHostToDevice);
    for (uint32_t n = 1; n <= 24; ++n) {
        hipLaunchKernelGGL(k_urem, dim3(N / 256), dim3(256), 0, 0, d_t, N, n * 8192u, d_o, d_q);
        hipMemcpy(got.data(), d_o, N * 4, hipMemcpyDeviceToHost); hipMemcpy(gotq.data(), d_q, N * 4, hipMemcpyDeviceToHost);
        size_t badq = 0; for (uint32_t i = 0; i < N; ++i) if (gotq[i] != (tags[i] * 0x9e3779b1u >> 8) / n) ++badq;
        size_t bad = 0, out_of_range = 0; uint32_t ex_t = 0, ex_g = 0;
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t want = (tags[i] * 0x9e3779b1u >> 8) % n;
            if (got[i] != want) { if (!bad) { ex_t = tags[i]; ex_g = got[i]; } ++bad; if (got[i] >= n) ++out_of_range; }
        }
        printf("n = %2u: quotients that differ %zu; remainders: %zu of %u differ from the host's %% (%zu of them >= n)%s", n, badq, bad, N, out_of_range, bad ? "" : "\n");
        if (bad) printf("   e.g. tag %08x: x = %u, device %u, host %u\n", ex_t, ex_t * 0x9e3779b1u >> 8, ex_g, (ex_t * 0x9e3779b1u >> 8) % n);
    }
    return 0;
}
