// TEST INFRASTRUCTURE ONLY — executes "kernels" compiled against tests/hip_emu/hip/hip_runtime.h on the CPU:
// one workgroup at a time, one OS thread per work-item (a persistent pool), pthread barriers for __syncthreads and
// for the lock-step wave exchange behind __shfl_xor.  Slow and only meant for the small shapes of the CPU tests.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <vector>
#include "../../include/tacotron2_amd.h"

thread_local emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
const int MAXT = 1024;
pthread_barrier_t g_block_barrier;          // sized to the current block
int g_nthreads = 0;
std::vector<unsigned char> g_slots(MAXT * 16);
emu_idx g_block;
const std::function<void()>* g_body = nullptr;

void* run_item(void* arg) {
    const int id = (int)(intptr_t)arg;
    threadIdx.x = id % blockDim.x;
    threadIdx.y = (id / blockDim.x) % blockDim.y;
    threadIdx.z = id / (blockDim.x * blockDim.y);
    blockIdx = g_block;
    (*g_body)();
    return nullptr;
}
}  // namespace

void emu_syncthreads() { pthread_barrier_wait(&g_block_barrier); }

// Every work-item of the block calls this together (the emulated kernels shuffle under uniform control flow), so a
// block-wide barrier is a valid (stronger) stand-in for the wave's lock step.
void emu_exchange(const void* mine, void* partner, size_t bytes, int mask) {
    const int flat = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    if (bytes > 16) abort();
    memcpy(&g_slots[flat * 16], mine, bytes);
    pthread_barrier_wait(&g_block_barrier);
    const int wave = flat / 64, lane = flat % 64;
    int src = wave * 64 + (lane ^ mask);
    if (src >= g_nthreads) src = flat;
    memcpy(partner, &g_slots[src * 16], bytes);
    pthread_barrier_wait(&g_block_barrier);
}

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int n = block.x * block.y * block.z;
    if (n < 1 || n > MAXT) abort();
    gridDim = grid;
    blockDim = block;
    g_nthreads = n;
    g_body = &body;
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    std::vector<pthread_t> th(n);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_block = emu_idx{bx, by, bz};
                pthread_barrier_init(&g_block_barrier, nullptr, n);
                // Items that return early never reach a barrier; the emulated kernels that synchronise keep every
                // item alive until the last barrier (checked by the tests finishing at all).
                for (int i = 0; i < n; ++i) pthread_create(&th[i], &attr, run_item, (void*)(intptr_t)i);
                for (int i = 0; i < n; ++i) pthread_join(th[i], nullptr);
                pthread_barrier_destroy(&g_block_barrier);
            }
    pthread_attr_destroy(&attr);
}

// the pieces of api.hip the kernels' launchers refer to
static thread_local char g_err[512] = "";
extern "C" void t2amd_set_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* t2amd_last_error(void) { return g_err; }
extern "C" int t2amd_validate_only_flag_(void) { return 0; }
extern "C" int t2amd_emulated(void) { return 1; }
