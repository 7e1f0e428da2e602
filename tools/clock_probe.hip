// Shader clock while another kernel runs: one wave on its own stream reads s_memtime (shader cycles) and s_memrealtime (constant reference clock)
// around a sleep loop of `ms` milliseconds; cycles / ticks x reference rate = the clock the chip sustained during that window.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/clock_probe.hip -o tools/libclock_probe.so      (driver: tools/dp_clock.py)
#include <hip/hip_runtime.h>
__global__ void k_probe(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
extern "C" int probe_clock(int device, double ms, double* ghz)
{
    if (hipSetDevice(device) != hipSuccess) return 1;
    int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) return 2;
    static hipStream_t s = nullptr; static unsigned long long* d = nullptr;
    if (!s) { if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 3; if (hipMalloc(&d, 16) != hipSuccess) return 4; }
    k_probe<<<1, 64, 0, s>>>(d, (unsigned long long)(ms * khz));
    unsigned long long h[2];
    if (hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 5;
    *ghz = (double)h[0] / (double)h[1] * khz * 1e-6;
    return 0;
}
