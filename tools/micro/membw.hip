// Ceiling probe: how fast can this GPU read an N-byte table once?  Forms: (0) dwordx4 per lane, grid-sized;
// (1) one dword per 20-byte record (the walk's own access shape), 8 records per lane; (2) LDS-DMA 16 B per lane
// into a per-wave ring, ptr dwords read back from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_x4(const uint4 *__restrict__ p, size_t n16, unsigned *out)
{
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        size_t k = i + (size_t)j * 256;
        if (k < n16) { uint4 v = p[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

__global__ __launch_bounds__(256) void read_x4_nt(const uint4 *__restrict__ p, size_t n16, unsigned *out)
{
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        size_t k = i + (size_t)j * 256;
        if (k < n16) {
            const unsigned *q = reinterpret_cast<const unsigned *>(p + k);
            acc += __builtin_nontemporal_load(q) ^ __builtin_nontemporal_load(q + 1) ^ __builtin_nontemporal_load(q + 2) ^
                   __builtin_nontemporal_load(q + 3);
        }
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

__global__ __launch_bounds__(256) void read_ptr_nt(const int *__restrict__ p, size_t nrec, unsigned *out)
{
    size_t base = (size_t)blockIdx.x * 256 * 8;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        size_t k = base + (size_t)j * 256 + threadIdx.x;
        if (k < nrec) acc ^= (unsigned)__builtin_nontemporal_load(p + k * 5 + 3);
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

// the walk's own shape: kN records per lane, 256 * kN per workgroup (the frame's walk takes 4)
template <int kN>
__global__ __launch_bounds__(256) void read_ptr_nt_n(const int *__restrict__ p, size_t nrec, unsigned *out)
{
    size_t base = (size_t)blockIdx.x * 256 * kN;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < kN; ++j) {
        size_t k = base + (size_t)j * 256 + threadIdx.x;
        if (k < nrec) acc ^= (unsigned)__builtin_nontemporal_load(p + k * 5 + 3);
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}
// ... behind a kernel-argument block of the frame launch's size (about 1 KB: two FrameParams, two DevPtrs, PipeArgs), every
// word of which is read before the first table load, as the frame kernel's role selection does
struct FatArgs { unsigned w[256]; };
template <int kN>
__global__ __launch_bounds__(256) void read_ptr_nt_fat(const int *__restrict__ p, size_t nrec, unsigned *out, const FatArgs fat)
{
    unsigned sel = 0;
#pragma unroll
    for (int i = 0; i < 256; i += 16) sel += fat.w[i];
    size_t base = (size_t)(blockIdx.x + sel) * 256 * kN;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < kN; ++j) {
        size_t k = base + (size_t)j * 256 + threadIdx.x;
        if (k < nrec) acc ^= (unsigned)__builtin_nontemporal_load(p + k * 5 + 3);
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

__global__ __launch_bounds__(256) void read_ptr(const int *__restrict__ p, size_t nrec, unsigned *out)
{
    size_t base = (size_t)blockIdx.x * 256 * 8;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        size_t k = base + (size_t)j * 256 + threadIdx.x;
        if (k < nrec) acc ^= (unsigned)p[k * 5 + 3];
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

// each wave: chunks of 5120 B (256 records) by 5 LDS-DMA instructions, double-buffered
template <int kNt>
__global__ __launch_bounds__(256) void read_ldsdma(const char *__restrict__ p, size_t nchunks, unsigned *out)
{
    __shared__ __attribute__((aligned(16))) char ring[4][2][5120];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t wavesTotal = (size_t)gridDim.x * 4;
    size_t c = (size_t)blockIdx.x * 4 + wave;
    unsigned acc = 0;
    int buf = 0;
    auto issue = [&](size_t chunk, int b) {
        const char *src = p + chunk * 5120 + lane * 16;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                             (__attribute__((address_space(3))) void *)(&ring[wave][b][j * 1024]), 16, 0, kNt ? 2 : 0);
    };
    if (c < nchunks) issue(c, 0);
    for (; c < nchunks; c += wavesTotal) {
        const size_t nxt = c + wavesTotal;
        if (nxt < nchunks) {
            issue(nxt, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int *rec = reinterpret_cast<const int *>(&ring[wave][buf][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= (unsigned)rec[(lane + 64 * j) * 5 + 3];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        buf ^= 1;
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

// one-shot forms: a workgroup handles kPer chunks per wave, all issued up front into kPer buffers
template <int kNt, int kPer>
__global__ __launch_bounds__(256) void read_ldsdma_oneshot(const char *__restrict__ p, size_t nchunks, unsigned *out)
{
    __shared__ __attribute__((aligned(16))) char ring[4][kPer][5120];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
        const size_t c = ((size_t)blockIdx.x * kPer + r) * 4 + wave;
        if (c < nchunks) {
            const char *src = p + c * 5120 + lane * 16;
#pragma unroll
            for (int j = 0; j < 5; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                                 (__attribute__((address_space(3))) void *)(&ring[wave][r][j * 1024]), 16, 0, kNt ? 2 : 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
        const int *rec = reinterpret_cast<const int *>(&ring[wave][r][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= (unsigned)rec[(lane + 64 * j) * 5 + 3];
    }
    if (acc == 0x12345678u) atomicAdd(out, 1u);
}

int main(int argc, char **argv)
{
    size_t mb = argc > 1 ? atoi(argv[1]) : 419;
    size_t bytes = (mb * 1000000 / 5120) * 5120;
    char *d; unsigned *out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(d, 0xff, bytes)); CK(hipMemset(out, 0, 4));
    // something to evict between runs is NOT used: the table itself exceeds the cache for 419 MB
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        std::vector<float> t;
        for (int r = 0; r < 30; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-28s %6zu MB  med %7.2f us  min %7.2f us  -> %.2f TB/s (med)\n", name, bytes / 1000000, t[15] * 1e3, t[0] * 1e3,
               bytes / (t[15] * 1e-3) / 1e12);
    };
    const size_t n16 = bytes / 16, nrec = bytes / 20, nchunks = bytes / 5120;
    timeit("dwordx4 grid", [&] { read_x4<<<dim3((n16 + 2047) / 2048), 256>>>((const uint4 *)d, n16, out); });
    timeit("ptr dword /20B x8", [&] { read_ptr<<<dim3((nrec + 2047) / 2048), 256>>>((const int *)d, nrec, out); });
    timeit("dwordx4 nt grid", [&] { read_x4_nt<<<dim3((n16 + 2047) / 2048), 256>>>((const uint4 *)d, n16, out); });
    timeit("ptr dword /20B x8 nt", [&] { read_ptr_nt<<<dim3((nrec + 2047) / 2048), 256>>>((const int *)d, nrec, out); });
    timeit("ptr dword /20B x4 nt", [&] { read_ptr_nt_n<4><<<dim3((nrec + 1023) / 1024), 256>>>((const int *)d, nrec, out); });
    timeit("ptr dword /20B x2 nt", [&] { read_ptr_nt_n<2><<<dim3((nrec + 511) / 512), 256>>>((const int *)d, nrec, out); });
    timeit("ptr dword /20B x16 nt", [&] { read_ptr_nt_n<16><<<dim3((nrec + 4095) / 4096), 256>>>((const int *)d, nrec, out); });
    { FatArgs fat; for (auto &w : fat.w) w = 0;
      timeit("ptr x4 nt, 1 KB kernarg", [&] { read_ptr_nt_fat<4><<<dim3((nrec + 1023) / 1024), 256>>>((const int *)d, nrec, out, fat); });
      timeit("ptr x8 nt, 1 KB kernarg", [&] { read_ptr_nt_fat<8><<<dim3((nrec + 2047) / 2048), 256>>>((const int *)d, nrec, out, fat); }); }
    for (int g : {512, 768, 1024, 2048})
        { char nm[64]; snprintf(nm, 64, "lds-dma grid %d", g);
          timeit(nm, [&] { read_ldsdma<0><<<dim3(g), 256>>>(d, nchunks, out); }); }
    for (int g : {768, 1024})
        { char nm[64]; snprintf(nm, 64, "lds-dma nt grid %d", g);
          timeit(nm, [&] { read_ldsdma<1><<<dim3(g), 256>>>(d, nchunks, out); }); }
    timeit("lds-dma nt one-shot x1", [&] { read_ldsdma_oneshot<1, 1><<<dim3((nchunks + 3) / 4), 256>>>(d, nchunks, out); });
    timeit("lds-dma nt one-shot x2", [&] { read_ldsdma_oneshot<1, 2><<<dim3((nchunks + 7) / 8), 256>>>(d, nchunks, out); });
    timeit("lds-dma    one-shot x2", [&] { read_ldsdma_oneshot<0, 2><<<dim3((nchunks + 7) / 8), 256>>>(d, nchunks, out); });
    for (int g : {256, 512, 1536})
        { char nm[64]; snprintf(nm, 64, "lds-dma nt grid %d", g);
          timeit(nm, [&] { read_ldsdma<1><<<dim3(g), 256>>>(d, nchunks, out); }); }
    return 0;
}
