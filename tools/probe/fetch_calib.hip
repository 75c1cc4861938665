// FETCH_SIZE calibration for gfx950 (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Each kernel below reads a 1 GiB buffer (4 x the Infinity Cache) exactly once with ONE access shape -- the
// shapes the four kernels of a decoder time step use -- so that
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/probe/fetch_calib
// gives bytes-per-FETCH_SIZE-unit for each shape (tools/pmc_traffic.py reads the table).
//   calib_b4 / calib_b8 / calib_b16      coalesced 4 / 8 / 16 bytes per lane (16: the LDS-DMA weight streams)
//   calib_seg128                         128-byte row segments, 512-byte row stride: K_e / K_b2 read 32 of the 128
//                                        processed-memory floats of every position (4 bytes per lane, 32 lanes per row)
//   calib_seg256                         256-byte row segments, 1 KB row stride: K_c / K_b1 read one 128-channel column
//                                        group of a bf16 memory row (16 bytes per lane, 16 lanes per row)
//   calib_lds16                          global_load_lds_dwordx4 (the LSTM / dgrad weight stream itself)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/fetch_calib.hip -o tools/probe/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void calib_b4(const float* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void calib_b8(const float2* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float2 v = p[i]; acc += v.x + v.y; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void calib_b16(const float4* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
// rows of 128 floats; workgroup (slice s = blockIdx.x & 3) reads floats [32 s, 32 s + 32) of its rows: 32 lanes per row
__global__ void calib_seg128(const float* p, size_t rows, float* sink) {
    const int s = blockIdx.x & 3;
    const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;            // 8 rows per 256-thread pass
    float acc = 0.f;
    for (size_t r = (size_t)(blockIdx.x >> 2) * 8 + sub; r < rows; r += (size_t)(gridDim.x >> 2) * 8) acc += p[r * 128 + 32 * s + lane];
    if (acc == 123.456f) sink[0] = acc;
}
// rows of 512 bf16 (1 KB); workgroup (group g = blockIdx.x & 3) reads bytes [256 g, 256 g + 256) of its rows: 16 lanes x 16 B
__global__ void calib_seg256(const float4* p, size_t rows, float* sink) {
    const int g = blockIdx.x & 3;
    const int lane = threadIdx.x & 15, sub = threadIdx.x >> 4;            // 16 rows per 256-thread pass
    float acc = 0.f;
    for (size_t r = (size_t)(blockIdx.x >> 2) * 16 + sub; r < rows; r += (size_t)(gridDim.x >> 2) * 16) { float4 v = p[r * 64 + 16 * g + lane]; acc += v.x + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
// the LDS-DMA stream: every wave instruction moves 64 x 16 B = 1 KB lane-linear into LDS
__global__ void calib_lds16(const float4* p, size_t n, float* sink) {
    __shared__ float4 buf[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + wave * 64; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4* src = p + i + lane;
        __builtin_amdgcn_global_load_lds((const void*)src, (void __attribute__((address_space(3)))*)&buf[wave][0], 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (buf[wave][lane].x == 123.456f) sink[0] = 1.f;
}

// WRITE_SIZE calibration: 1 GiB written once, 16 bytes per lane coalesced
__global__ void calib_w16(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t bytes = 1ull << 30;
    void* buf; float* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_b4, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(calib_b8, dim3(4096), dim3(256), 0, 0, (const float2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(calib_b16, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_seg128, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / 512, sink);
        hipLaunchKernelGGL(calib_seg256, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 1024, sink);
        hipLaunchKernelGGL(calib_lds16, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_w16, dim3(4096), dim3(256), 0, 0, (float4*)buf, bytes / 16);
        CK(hipDeviceSynchronize());
    }
    printf("fetch_calib: 7 kernels x 2 launches, %zu bytes read per launch\n", bytes);
    return 0;
}
