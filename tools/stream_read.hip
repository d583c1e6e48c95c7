// Known-good reference for the table walk's ceiling: plain read-only streaming of a buffer
// with 16-byte-per-lane loads (8 in flight per lane, 256-lane workgroups), timed with
// per-dispatch HIP events.  Usage: stream_read <MB> [reps]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void stream_read(const uint4 *__restrict__ p, size_t n, unsigned *out)
{
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (base + j * 256 < n) ? p[base + j * 256] : make_uint4(0, 0, 0, 0);
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    if (s == 0x12345678u) out[blockIdx.x] = s;      // practically never; keeps the loads alive
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atoll(argv[1]) : 105;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    const size_t bytes = mb << 20, n = bytes / 16;
    uint4 *p; unsigned *out;
    hipMalloc(&p, bytes); hipMalloc(&out, 4 << 20);
    hipMemset(p, 1, bytes);
    const unsigned grid = (unsigned)((n + 2047) / 2048);
    std::vector<float> us;
    for (int r = 0; r < reps + 5; ++r) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipExtLaunchKernelGGL(stream_read, dim3(grid), dim3(256), 0, 0, a, b, 0, p, n, out);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r >= 5) us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    const float med = us[us.size() / 2];
    printf("%zu MB: median %.2f us = %.2f TB/s (min %.2f us)\n", mb, med, bytes / med / 1e6, us[0]);
    return 0;
}
