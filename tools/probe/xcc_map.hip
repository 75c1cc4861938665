// Build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/probe/xcc_map tools/probe/xcc_map.hip ; run on the GPU box.
// Prints which XCD (XCC_ID) and CU each workgroup of a 3-D grid lands on: tools only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        out[2 * L] = xcc;
        out[2 * L + 1] = hwid;
    }
}
int main() {
    dim3 grid(20, 32, 3);
    const int total = grid.x * grid.y * grid.z;
    unsigned* d;
    hipMalloc(&d, total * 8);
    hipLaunchKernelGGL(probe, grid, dim3(256), 0, 0, d);
    std::vector<unsigned> h(total * 2);
    hipMemcpy(h.data(), d, total * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 48; ++i) printf("L=%d xcc=0x%x (id %u) hwid=0x%x\n", i, h[2 * i], h[2 * i] & 0xf, h[2 * i + 1]);
    int match = 0;
    for (int i = 0; i < total; ++i) match += ((h[2 * i] & 0xf) == (unsigned)(i & 7));
    printf("xcc_id == L %% 8 for %d of %d workgroups\n", match, total);
    return 0;
}
