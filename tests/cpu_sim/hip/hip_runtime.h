// Host stand-in for <hip/hip_runtime.h> (TEST INFRASTRUCTURE, tests/test_kernel_cpu_sim.py): lets the UNMODIFIED kernel
// sources of musev_amd/csrc be compiled for x86 and executed thread-per-lane, so that the index arithmetic, predicates,
// LDS layouts, barrier protocol and MFMA fragment layouts of a kernel can be checked against a torch reference without a
// GPU.  What it models:
//   * a block = blockDim.x OS threads; __syncthreads / s_barrier = pthread barrier; one 64-thread barrier per wave;
//   * v_mfma_f32_16x16x32_f16 with the CDNA3/4 lane layout (A: lane l -> row l%16, k 8*(l/16)..+7; B alike with the column;
//     D: lane l -> rows 4*(l/16)..+3 of column l%16), fp32 accumulation;
//   * buffer descriptors: raw_ptr_buffer_load_lds / raw_buffer_load_b128 with the range check on the VGPR offset (an offset
//     >= num_records reads zero) and a hard failure if an in-range lane would touch bytes outside the allocation;
//   * LDS-DMA completion time: SIM_DEFER=1 delays every LDS-DMA write until the issuing thread's next s_waitcnt vmcnt(N) /
//     __syncthreads (the LATEST legal landing), SIM_DEFER=0 performs it at issue (the EARLIEST) -- a kernel has to be right
//     under both;
//   * dynamic shared memory = one global buffer (blocks run one after another); static __shared__ arrays = function statics.
//   * all simulator state is C++17 `inline` (one instance per process): the kernel sources include this header from several
//     translation units, and common.h's non-static inline helpers are merged by the linker across them.
// What it cannot model: timing, bank conflicts, occupancy, alignment faults, anything about performance.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <deque>
#include <functional>
#include <thread>
#include <vector>

struct uint4 { unsigned x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 8, hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "sim"; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) {  // SIM_CUS: lets small problems meet the one-round-grid rules
    const char* e = getenv("SIM_CUS");
    *v = e ? atoi(e) : 256;
    return hipSuccess;
}

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static  // `__shared__ T buf[N];` inside a kernel: one function-local static buffer, blocks run one at a time
#define __expf expf

inline thread_local dim3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;
alignas(64) inline char sim_smem_buf[160 * 1024];  // the block's LDS: the tests' source transform turns `extern __shared__ T name[];` into `T* name = (T*)sim_smem_buf;`
inline pthread_barrier_t sim_block_bar;
inline pthread_barrier_t sim_wave_bar[16];
inline unsigned sim_wave_lanes[16];  // threads of each wave of the running block (the last wave may be partial)
inline int sim_defer = 0;

typedef _Float16 sim_half8 __attribute__((ext_vector_type(8)));
typedef float sim_float4 __attribute__((ext_vector_type(4)));
typedef unsigned sim_u32x4 __attribute__((ext_vector_type(4)));

// ---- pending LDS-DMA writes of the calling thread (SIM_DEFER) ----
struct SimDma { unsigned char data[16]; void* dst; };
inline thread_local std::deque<SimDma> sim_dma;
static inline void sim_retire(size_t keep) {
    while (sim_dma.size() > keep) {
        if (sim_dma.front().dst) memcpy(sim_dma.front().dst, sim_dma.front().data, 16);
        sim_dma.pop_front();
    }
}
static inline void sim_waitcnt_vm(int n) { sim_retire((size_t)n); }
static inline void __syncthreads() { sim_retire(0); pthread_barrier_wait(&sim_block_bar); }  // s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier
static inline void sim_s_barrier() { pthread_barrier_wait(&sim_block_bar); }
static inline void sim_wave_barrier() { pthread_barrier_wait(&sim_wave_bar[threadIdx.x >> 6]); }
#define __builtin_amdgcn_s_barrier sim_s_barrier
#define __builtin_amdgcn_wave_barrier sim_wave_barrier
#define __builtin_amdgcn_fence(...) ((void)0)
#define MV_KEEP_ONE_REGISTER(x) ((void)0)
#define MV_FMA_SCALAR(a, b, c) fmaf((a), (b), (c))
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T> static inline T sim_atomic_fetch_add(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
#define __hip_atomic_fetch_add(p, v, order, scope) sim_atomic_fetch_add((p), (v))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

static void sim_dma_write(void* dst, const void* src16) {
    if (sim_defer) {
        SimDma d;
        if (src16) memcpy(d.data, src16, 16); else memset(d.data, 0, 16);
        d.dst = dst;
        sim_dma.push_back(d);
    } else if (src16) memcpy(dst, src16, 16); else memset(dst, 0, 16);
}

// A lane that is switched off (EXEC) for an LDS-DMA instruction: the hardware's vmcnt counts the wave's INSTRUCTION, so the lane's
// per-thread queue gets an entry that writes nothing (otherwise its counted waits would run ahead of its wave's).
static inline void sim_dma_lane_off() {
    if (sim_defer) {
        SimDma d;
        d.dst = nullptr;
        sim_dma.push_back(d);
    }
}
#define MV_DMA_LANE_OFF() sim_dma_lane_off()

// ---- buffer descriptors ----
struct SimRsrc { const char* base; unsigned num; };
#define __amdgpu_buffer_rsrc_t SimRsrc
static inline SimRsrc sim_make_rsrc(void* p, int, unsigned num, unsigned) { return SimRsrc{(const char*)p, num}; }
#define __builtin_amdgcn_make_buffer_rsrc sim_make_rsrc
static inline const char* sim_buf_addr(const SimRsrc& r, int voff, int soff, int ioff) {
    const unsigned vo = (unsigned)voff + (unsigned)ioff;
    if (vo >= r.num) return nullptr;  // range check on the VGPR (+ instruction) offset: reads zero
    const unsigned long total = (unsigned long)vo + (unsigned)soff;
    if (r.num != 0x7fffffffu && total + 16 > r.num) {  // an in-range lane must stay inside the bytes the descriptor covers
        fprintf(stderr, "SIM: buffer load overruns its descriptor: voffset %u + soffset %u + 16 > %u\n", vo, (unsigned)soff, r.num);
        abort();
    }
    return r.base + total;
}
static inline void sim_buffer_load_lds(SimRsrc r, __attribute__((address_space(3))) void* lds, int size, int voff, int soff, int ioff, int) {
    if (size != 16) abort();
    sim_dma_write((char*)(void*)lds + 16 * (threadIdx.x & 63), sim_buf_addr(r, voff, soff, ioff));
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds sim_buffer_load_lds
static inline sim_u32x4 sim_buffer_load_b128(SimRsrc r, int voff, int soff, int) {
    sim_u32x4 v = {0, 0, 0, 0};
    const char* a = sim_buf_addr(r, voff, soff, 0);
    if (a) memcpy(&v, a, 16);
    return v;
}
#define __builtin_amdgcn_raw_buffer_load_b128 sim_buffer_load_b128
static inline void sim_global_load_lds(const __attribute__((address_space(1))) void* g, __attribute__((address_space(3))) void* lds, int size, int off, int) {
    if (size != 16) abort();
    sim_dma_write((char*)(void*)lds + 16 * (threadIdx.x & 63), (const char*)(const void*)g + off);
}
#define __builtin_amdgcn_global_load_lds sim_global_load_lds

// ---- MFMA 16x16x32 f16 ----
inline _Float16 sim_mfma_a[16][64][8], sim_mfma_b[16][64][8];
static inline sim_float4 sim_mfma_16x16x32_f16(sim_half8 a, sim_half8 b, sim_float4 c, int, int, int) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = 0; j < 8; ++j) {
        sim_mfma_a[wave][lane][j] = a[j];
        sim_mfma_b[wave][lane][j] = b[j];
    }
    sim_wave_barrier();
    const int col = lane & 15, g = lane >> 4;
    sim_float4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k)
            acc += (float)sim_mfma_a[wave][(k >> 3) * 16 + row][k & 7] * (float)sim_mfma_b[wave][(k >> 3) * 16 + col][k & 7];
        d[r] += acc;
    }
    sim_wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 sim_mfma_16x16x32_f16
// ---- MFMA 16x16x16 f16: lane (row = l & 15, group g = l >> 4) holds k = 4g .. 4g+3 of its A / B row ----
typedef _Float16 sim_half4 __attribute__((ext_vector_type(4)));
static inline sim_float4 sim_mfma_16x16x16_f16(sim_half4 a, sim_half4 b, sim_float4 c, int, int, int) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = 0; j < 4; ++j) {
        sim_mfma_a[wave][lane][j] = a[j];
        sim_mfma_b[wave][lane][j] = b[j];
    }
    sim_wave_barrier();
    const int col = lane & 15, g = lane >> 4;
    sim_float4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = 0.f;
        for (int k = 0; k < 16; ++k)
            acc += (float)sim_mfma_a[wave][(k >> 2) * 16 + row][k & 3] * (float)sim_mfma_b[wave][(k >> 2) * 16 + col][k & 3];
        d[r] += acc;
    }
    sim_wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x16f16 sim_mfma_16x16x16_f16

// ---- cross-lane operations (one exchange buffer per wave, two wave barriers per operation) ----
inline double sim_xchg[16][64];  // 8-byte slots: float, int and double payloads
template <typename T>
static inline T sim_lane_read(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "exchange slot");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    memcpy(&sim_xchg[wave][lane], &v, sizeof(T));
    sim_wave_barrier();
    T r;
    memcpy(&r, &sim_xchg[wave][src_lane & 63], sizeof(T));
    sim_wave_barrier();
    return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int) { return sim_lane_read(v, (int)(threadIdx.x & 63) ^ mask); }
template <typename T> static inline T __shfl(T v, int src, int) { return sim_lane_read(v, src); }
template <typename T> static inline T __shfl_down(T v, int off, int) { const int l = (int)(threadIdx.x & 63); return sim_lane_read(v, l + off < 64 ? l + off : l); }
static inline int __any(int pred) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    sim_xchg[wave][lane] = pred ? 1.0 : 0.0;
    sim_wave_barrier();
    int r = 0;
    for (int i = 0; i < (int)sim_wave_lanes[wave]; ++i) r |= sim_xchg[wave][i] != 0.0;
    sim_wave_barrier();
    return r;
}
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
// ds_read_b64_tr_b16 as measured by mv_probe_tr16 on the MI355X (tests/kernel_cases.case_tr16_probe): the 16 lanes of a group
// each address 4 consecutive 16-bit elements; lane i of the group receives element (i & 3) of the chunks of lanes
// 4j + (i >> 2), j = 0..3 -- i.e. column i of the 4 x 16 block the group addresses row-wise
typedef short sim_short4 __attribute__((ext_vector_type(4)));
inline short sim_tr[16][64][4];
static inline sim_short4 sim_ds_read_tr16_b64(__attribute__((address_space(3))) sim_short4* p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    memcpy(sim_tr[wave][lane], (const void*)p, 8);
    sim_wave_barrier();
    const int g = lane >> 4, i = lane & 15;
    sim_short4 r;
    for (int j = 0; j < 4; ++j) r[j] = sim_tr[wave][16 * g + 4 * j + (i >> 2)][i & 3];
    sim_wave_barrier();
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16 sim_ds_read_tr16_b64
#define __builtin_amdgcn_exp2f exp2f
// v_dot2_f32_f16: c + a.x b.x + a.y b.y in fp32
typedef _Float16 sim_half2 __attribute__((ext_vector_type(2)));
static inline float sim_fdot2(sim_half2 a, sim_half2 b, float c, bool) { return c + (float)a[0] * (float)b[0] + (float)a[1] * (float)b[1]; }
#define __builtin_amdgcn_fdot2 sim_fdot2
#define __builtin_amdgcn_s_setprio(x) ((void)0)
typedef __fp16 sim_fp16x2 __attribute__((ext_vector_type(2)));
static inline uint16_t sim_rtz_half_bits(float x) {  // v_cvt_pkrtz_f16_f32: round toward zero
    _Float16 h = (_Float16)x;
    uint16_t b;
    memcpy(&b, &h, 2);
    if (fabsf((float)h) > fabsf(x)) b -= 1;  // one ulp toward zero (same sign, smaller magnitude)
    return b;
}
static inline sim_fp16x2 sim_cvt_pkrtz(float a, float b) {
    const uint16_t bits[2] = {sim_rtz_half_bits(a), sim_rtz_half_bits(b)};
    sim_fp16x2 r;
    memcpy(&r, bits, 4);
    return r;
}
#define __builtin_amdgcn_cvt_pkrtz sim_cvt_pkrtz

// ---- launch ----
// One pool of blockDim.x threads per launch; the blocks of the grid run one after another on it (a barrier between blocks),
// so a grid of thousands of small blocks costs a barrier per block instead of a thread creation per simulated thread.
inline pthread_barrier_t sim_pool_bar;
static void sim_launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    if (smem_bytes > sizeof(sim_smem_buf) || block.x == 0 || block.x > 1024 || block.y != 1 || block.z != 1) { fprintf(stderr, "SIM: bad launch\n"); abort(); }
    const unsigned n_waves = (block.x + 63) / 64;
    gridDim = grid;
    blockDim = block;
    const char* e = getenv("SIM_DEFER");
    sim_defer = e ? atoi(e) : 0;
    if (getenv("SIM_TRACE")) fprintf(stderr, "SIM launch grid %u x %u x %u block %u smem %zu\n", grid.x, grid.y, grid.z, block.x, smem_bytes);
    pthread_barrier_init(&sim_block_bar, nullptr, block.x);
    pthread_barrier_init(&sim_pool_bar, nullptr, block.x);
    for (unsigned w = 0; w < n_waves; ++w) {
        sim_wave_lanes[w] = (block.x - 64 * w < 64) ? block.x - 64 * w : 64;
        pthread_barrier_init(&sim_wave_bar[w], nullptr, sim_wave_lanes[w]);
    }
    const unsigned long n_blocks = (unsigned long)grid.x * grid.y * grid.z;
    std::vector<std::thread> ts;
    ts.reserve(block.x);
    for (unsigned t = 0; t < block.x; ++t)
        ts.emplace_back([&, t]() {
            for (unsigned long b = 0; b < n_blocks; ++b) {
                if (smem_bytes) {  // poison the block's LDS: reading it before writing shows up as garbage, not as a stale tile
                    if (t == 0) memset(sim_smem_buf, 0xCD, smem_bytes);
                    pthread_barrier_wait(&sim_pool_bar);
                }
                threadIdx = dim3(t, 0, 0);
                blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long)grid.x * grid.y)));
                sim_dma.clear();
                body();
                sim_retire(0);
                pthread_barrier_wait(&sim_pool_bar);  // every simulated thread of block b is done before block b + 1 starts
            }
        });
    for (auto& th : ts) th.join();
    pthread_barrier_destroy(&sim_block_bar);
    pthread_barrier_destroy(&sim_pool_bar);
    for (unsigned w = 0; w < n_waves; ++w) pthread_barrier_destroy(&sim_wave_bar[w]);
}
#define hipLaunchKernelGGL(kernel, grid, block, smem_bytes, stream, ...) \
    sim_launch(dim3(grid), dim3(block), (smem_bytes), [&]() { kernel(__VA_ARGS__); })
