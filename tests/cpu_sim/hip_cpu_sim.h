// Minimal host execution model for simple HIP kernels (TEST INFRASTRUCTURE, tests/test_kernel_cpu_sim.py): a block is
// blockDim.x real threads, __syncthreads() is a barrier, threadIdx / blockIdx are thread-local, dynamic shared memory is one
// heap buffer per block.  Blocks run one after another.  Good for kernels made of plain loads, FMAs and one or two barriers
// (no MFMA, no cross-lane builtins); the kernel TEXT is extracted from the .hip source between [cpu-sim:begin/end] markers,
// so what runs here is the code that is compiled for gfx950, not a copy.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <thread>
#include <vector>

struct SimDim3 { unsigned x, y, z; };
static thread_local SimDim3 threadIdx, blockIdx;
static SimDim3 blockDim, gridDim;
static pthread_barrier_t sim_barrier;
static char* sim_smem = nullptr;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
static inline void __syncthreads() { pthread_barrier_wait(&sim_barrier); }

typedef _Float16 half_t;
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
#define MV_ACT_NONE 0
#define MV_ACT_SILU 1
static inline float mv_silu(float x) { return x / (1.0f + expf(-x)); }

// launch: grid (gx, gy), block bx threads, smem bytes; `body` is called once per simulated thread
static void sim_launch(unsigned gx, unsigned gy, unsigned bx, size_t smem, const std::function<void()>& body) {
    gridDim = {gx, gy, 1};
    blockDim = {bx, 1, 1};
    sim_smem = (char*)aligned_alloc(64, ((smem + 63) / 64 + 1) * 64);
    for (unsigned by = 0; by < gy; ++by)
        for (unsigned bxi = 0; bxi < gx; ++bxi) {
            pthread_barrier_init(&sim_barrier, nullptr, bx);
            std::vector<std::thread> ts;
            ts.reserve(bx);
            for (unsigned t = 0; t < bx; ++t)
                ts.emplace_back([&, t]() {
                    threadIdx = {t, 0, 0};
                    blockIdx = {bxi, by, 0};
                    body();
                });
            for (auto& th : ts) th.join();
            pthread_barrier_destroy(&sim_barrier);
        }
    free(sim_smem);
    sim_smem = nullptr;
}
