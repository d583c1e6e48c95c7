// Drives the C++ CameraTracking facade the way Application.cpp:73-76 would drive the reference
// tracker: maps of two frames on the device, Align(), getTransform().
//   tracking_demo <input.bin> <target.bin> <targetNormals.bin>   (640*480 float4 each)
// prints the 16 row-major floats of the transform and the summed residual of the last round.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "CameraTracking.h"

static vh_float4 *upload(const char *path, size_t n)
{
    std::vector<vh_float4> h(n);
    FILE *f = std::fopen(path, "rb");
    if (!f || std::fread(h.data(), sizeof(vh_float4), n, f) != n) return nullptr;
    std::fclose(f);
    vh_float4 *d = nullptr;
    if (hipMalloc((void **)&d, n * sizeof(vh_float4)) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), n * sizeof(vh_float4), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const size_t n = 640 * 480;
    vh_float4 *d_input = upload(argv[1], n), *d_target = upload(argv[2], n), *d_normals = upload(argv[3], n);
    if (!d_input || !d_target || !d_normals) return 3;
    CameraTracking tracker(640, 480);            // Application.cpp:32
    tracker.Align(d_input, nullptr, d_target, d_normals, nullptr, nullptr);
    const float4x4 T = tracker.getTransform();
    for (int i = 0; i < 16; ++i) std::printf("%.9g ", T.entries[i]);
    std::printf("\nerror=%.9g\n", tracker.lastError());
    return 0;
}
