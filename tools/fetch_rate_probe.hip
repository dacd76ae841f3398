// fetch_rate_probe.hip -- what one CU can pull out of the L2 / the infinity cache per clock, by load flavour (run on the MI355X):
//
//     hipcc --offload-arch=gfx950 -O3 -o tools/scratch/fetch_rate_probe tools/fetch_rate_probe.hip && tools/scratch/fetch_rate_probe
//
// Every matrix kernel of the step (implicit GEMM, fused feed-forward, fused temporal sub-block) stages its operands at 10-15 bytes per
// clock and CU while the L2 is good for ~56.  This probe separates the candidates: the LDS-DMA path against plain 16-byte
// loads into registers (with and without the ds_write that a register-staged loader needs), every CU streaming the SAME bytes (a
// weight matrix) against its OWN bytes, with few or many requests in flight.  Timing only -- the data is summed so that nothing is
// optimised away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Args {
    const char* src;      // source bytes
    unsigned* sink;       // [blocks * threads]
    unsigned span;        // bytes each block walks (a multiple of 16 KiB * waves)
    unsigned stride;      // byte distance between the blocks' regions (0 = every block reads the same bytes)
    int iters;            // passes over the span
    unsigned rot;         // same bytes: block b starts its walk rot * b pieces into the span (the CUs ask for different lines at any moment)
    const char* stream;   // if set: beside the walk every wave also reads its OWN 1 KiB of fresh bytes per `stream_every` pieces (an activation stream through the same L2)
    int stream_every;
};

// MODE 0: LDS-DMA (buffer_load_dwordx4 ... lds), DEPTH pieces of 1 KiB in flight per wave
// MODE 1: buffer_load_dwordx4 into registers, DEPTH loads in flight per wave, summed
// MODE 2: MODE 1 + ds_write_b128 of every loaded value into the wave's LDS slot
template <int MODE, int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = a.src + (size_t)blockIdx.x * a.stride;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, a.span, 0x00020000);
    char* slot0 = lds + wave * (DEPTH * 1024);
    unsigned acc = 0;
    const unsigned pieces = a.span / 1024 / WAVES;   // 1-KiB pieces this wave walks per pass: piece p of wave w = bytes (p * WAVES + w) * 1024
    const unsigned rot = (a.rot * blockIdx.x) % pieces;
    const uint4v* strm = a.stream ? reinterpret_cast<const uint4v*>(a.stream) + ((size_t)blockIdx.x * WAVES + wave) * (size_t)(64 * 256) + lane : nullptr;   // 256 KiB of its own per wave
    unsigned sidx = 0;
    for (int it = 0; it < a.iters; ++it) {
        if constexpr (MODE == 0) {
            for (unsigned p0 = 0; p0 < pieces; p0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    unsigned pp = p0 + d + rot;
                    pp = pp >= pieces ? pp - pieces : pp;
                    const unsigned so = (pp * WAVES + wave) * 1024u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(slot0 + d * 1024), 16, lane * 16, (int)so, 0, 0);
                }
                if (strm && (p0 / DEPTH) % a.stream_every == 0) {   // (wave-uniform) fresh bytes: 1 KiB per wave
                    const uint4v t = strm[(size_t)(sidx & 255) * 64];
                    ++sidx;
                    acc += t[0];
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            for (unsigned p0 = 0; p0 < pieces; p0 += DEPTH) {
                uint4v v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const unsigned so = ((p0 + d) * WAVES + wave) * 1024u;
                    v[d] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (int)so, 0));
                }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if constexpr (MODE == 2) *reinterpret_cast<uint4v*>(slot0 + d * 1024 + lane * 16) = v[d];
                    acc += v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3];
                }
            }
        }
    }
    if constexpr (MODE != 1) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<unsigned*>(slot0 + lane * 4);
    }
    a.sink[blockIdx.x * blockDim.x + tid] = acc;
}

// ---- the weight-stream RING of the fused kernels (ffn.hip / tsa.hip) in isolation: 8 waves, 16-KiB tiles ([128 rows][64 halfs], the
// swizzle on the source address), RING stages, per tile: counted vmcnt wait + one raw s_barrier + the issue of the tile RING - 1 ahead
// (2 pieces per wave) + NREAD fragment reads (ds_read_b128) + NMFMA MFMAs on them.  ADDR 0: a tile is 16 KiB of consecutive bytes;
// ADDR 1: the rows of a tile are ROWB bytes apart (a K-slice of a row-major weight matrix, as the kernels read it) ----
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
struct RingArgs {
    const char* src;
    float* sink;
    unsigned span;      // bytes of the shared matrix
    int tiles;          // tiles per block
    unsigned rot;       // block b starts rot * b tiles in
    unsigned rowb;      // ADDR 1: bytes between the rows of the matrix (640 = W1 of the feed-forward)
};

template <int RING, int NREAD, int NMFMA, int ADDR, bool BARRIER>
__global__ __launch_bounds__(512) void ring_probe(const RingArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.span, 0x00020000);
    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);
    unsigned off[2];
    const unsigned ntiles_src = ADDR == 0 ? a.span / 16384u : (a.span / a.rowb / 128u) * (a.rowb / 128u);
    const unsigned kt_per_row = a.rowb / 128u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned row = 8 * (wave + 8 * q) + lrow;
        off[q] = ADDR == 0 ? row * 128u + lsl * 16u : row * a.rowb + lsl * 16u;
    }
    unsigned cur = (a.rot * blockIdx.x) % ntiles_src;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        unsigned so;
        if (ADDR == 0) so = cur * 16384u;
        else so = (cur / kt_per_row) * 128u * a.rowb + (cur % kt_per_row) * 128u;
        cur = cur + 1 == ntiles_src ? 0 : cur + 1;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + stage * 16384 + (wave + 8 * q) * 1024), 16, (int)off[q], (int)so, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) issue(t);
    float4v acc[4] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
    half8v xr[2];
    for (int e = 0; e < 8; ++e) { xr[0][e] = (half_t)(0.001f * lane); xr[1][e] = (half_t)(0.002f * lane); }
    float facc = 0.f;
    int stage = 0, istage = RING - 1;
    const int swz = l15 & 7;
    for (int t = 0; t < a.tiles; ++t) {
        // steady state: RING - 2 younger tiles may stay in flight (the tail of the walk over-waits: timing only)
        if (t + RING - 2 < a.tiles) {
            if constexpr (RING == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (RING == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (RING == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + RING - 1 < a.tiles) issue(istage);
        istage = istage == RING - 1 ? 0 : istage + 1;
        const half_t* tile = reinterpret_cast<const half_t*>(lds + stage * 16384);
        stage = stage == RING - 1 ? 0 : stage + 1;
        half8v wf[NREAD > 0 ? NREAD : 1];
#pragma unroll
        for (int j = 0; j < NREAD; ++j) {
            const int kk = j & 1, jj = (j >> 1) & 7;
            wf[j] = *reinterpret_cast<const half8v*>(tile + (16 * jj + l15) * 64 + ((((kk * 4 + g) ^ swz)) << 3));
        }
        if constexpr (NMFMA > 0) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[m % (NREAD > 0 ? NREAD : 1)], xr[(m >> 2) & 1], acc[m & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NREAD; ++j) facc += (float)wf[j][0];
        }
    }
    a.sink[blockIdx.x * 512 + tid] = facc + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int RING, int NREAD, int NMFMA, int ADDR, bool BARRIER>
static void run_ring(const RingArgs& a, int blocks, double ghz, int cus) {
    const int smem = RING * 16384;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_probe<RING, NREAD, NMFMA, ADDR, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ring_probe<RING, NREAD, NMFMA, ADDR, BARRIER>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ring_probe<RING, NREAD, NMFMA, ADDR, BARRIER>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rounds = (double)((blocks + cus - 1) / cus);
    const double cyc_per_tile = ms * 1e-3 * ghz * 1e9 / (a.tiles * rounds);
    printf("ring %d stages, %2d reads, %2d MFMAs per wave and tile, %s, %s, rot %3u, blocks %4d: %7.3f ms  %6.0f cycles per tile and block  %5.1f B/clk/CU  %5.2f TB/s  MFMA pipe %4.2f\n",
           RING, NREAD, NMFMA, ADDR ? "strided rows" : "contiguous  ", BARRIER ? "barrier   " : "no barrier", a.rot, blocks, ms, cyc_per_tile, 16384.0 / cyc_per_tile,
           (double)blocks * a.tiles * 16384.0 / (ms * 1e-3) * 1e-12, NMFMA * 2 * 16.0 / cyc_per_tile);
    fflush(stdout);
}

// the same ring with the fragment reads of tile t + 1 issued AHEAD of the MFMAs of tile t (two fragment sets in registers): the wait for
// tile t + 1 and the barrier move one tile earlier, the LDS reads run in the shadow of the matrix pipe
// SPREAD 1: the two LDS-DMA issues of a step sit BETWEEN the MFMAs (after 1/4 and 3/4 of them) instead of right behind the barrier;
// SPREAD 2: additionally no fragment prefetch (reads of tile t right behind the barrier, as the kernels do today)
template <int RING, int NREAD, int NMFMA, int SPREAD = 0>
__global__ __launch_bounds__(512) void ring_probe_pipe(const RingArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.span, 0x00020000);
    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);
    unsigned off[2];
    const unsigned kt_per_row = a.rowb / 128u;
    const unsigned ntiles_src = (a.span / a.rowb / 128u) * kt_per_row;
#pragma unroll
    for (int q = 0; q < 2; ++q) off[q] = (8 * (wave + 8 * q) + lrow) * a.rowb + lsl * 16u;
    unsigned cur = (a.rot * blockIdx.x) % ntiles_src;
    unsigned so_next = 0;
    auto issue_q = [&](int stage, int q) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + stage * 16384 + (wave + 8 * q) * 1024), 16, (int)off[q], (int)so_next, 0, 0);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        so_next = (cur / kt_per_row) * 128u * a.rowb + (cur % kt_per_row) * 128u;
        cur = cur + 1 == ntiles_src ? 0 : cur + 1;
    };
    auto issue = [&](int stage) __attribute__((always_inline)) {
        advance();
        issue_q(stage, 0);
        issue_q(stage, 1);
    };
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) issue(t);
    float4v acc[4] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
    half8v xr[2];
    for (int e = 0; e < 8; ++e) { xr[0][e] = (half_t)(0.001f * lane); xr[1][e] = (half_t)(0.002f * lane); }
    int stage = 0, istage = RING - 1;
    const int swz = l15 & 7;
    auto read_frags = [&](half8v (&wf)[NREAD], int st) __attribute__((always_inline)) {
        const half_t* tile = reinterpret_cast<const half_t*>(lds + st * 16384);
#pragma unroll
        for (int j = 0; j < NREAD; ++j) {
            const int kk = j & 1, jj = (j >> 1) & 7;
            wf[j] = *reinterpret_cast<const half8v*>(tile + (16 * jj + l15) * 64 + ((((kk * 4 + g) ^ swz)) << 3));
        }
    };
    half8v wa[NREAD], wb[NREAD];
    // tile 0 -> wa
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (RING - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(wa, 0);
    auto body = [&](half8v (&wcur)[NREAD], half8v (&wnext)[NREAD], int t) __attribute__((always_inline)) {
        // tile t + 1 has landed when RING - 3 younger tiles are in flight (tiles t + 2 .. t + RING - 2 were issued before this point)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (RING - 3)) : "memory");
        __builtin_amdgcn_s_barrier();   // tile t + 1 visible to every wave; every wave has read tile t (its fragments are in registers): stage of tile t is free
        asm volatile("" ::: "memory");
        const bool more = t + RING - 1 < a.tiles;
        const int ist = istage;
        if constexpr (SPREAD == 0) {
            if (more) issue(ist);   // into the stage tile t - 0 just left ... (stage bookkeeping below)
        } else {
            advance();
        }
        istage = istage == RING - 1 ? 0 : istage + 1;
        stage = stage == RING - 1 ? 0 : stage + 1;
        read_frags(wnext, stage);
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the reads below the MFMAs: the loop is then the unpipelined one again)
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) {
            if constexpr (SPREAD != 0) {
                if (m == NMFMA / 4) { __builtin_amdgcn_sched_barrier(0); if (more) issue_q(ist, 0); __builtin_amdgcn_sched_barrier(0); }
                if (m == (3 * NMFMA) / 4) { __builtin_amdgcn_sched_barrier(0); if (more) issue_q(ist, 1); __builtin_amdgcn_sched_barrier(0); }
            }
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur[m % NREAD], xr[(m >> 2) & 1], acc[m & 3], 0, 0, 0);
        }
    };
    for (int t = 0; t + 1 < a.tiles; t += 2) {
        body(wa, wb, t);
        body(wb, wa, t + 1);
    }
    a.sink[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)wa[0][0];
}

template <int RING, int NREAD, int NMFMA, int SPREAD = 0>
static void run_ring_pipe(const RingArgs& a, int blocks, double ghz, int cus) {
    const int smem = RING * 16384;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_probe_pipe<RING, NREAD, NMFMA, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ring_probe_pipe<RING, NREAD, NMFMA, SPREAD>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ring_probe_pipe<RING, NREAD, NMFMA, SPREAD>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rounds = (double)((blocks + cus - 1) / cus);
    const double cyc_per_tile = ms * 1e-3 * ghz * 1e9 / (a.tiles * rounds);
    printf("PIPELINED%s ring %d stages, %2d reads, %2d MFMAs per wave and tile, strided rows, rot %3u, blocks %4d: %7.3f ms  %6.0f cycles per tile and block  %5.1f B/clk/CU  MFMA pipe %4.2f\n",
           SPREAD ? " + DMA issues spread among the MFMAs" : "", RING, NREAD, NMFMA, a.rot, blocks, ms, cyc_per_tile, 16384.0 / cyc_per_tile, NMFMA * 2 * 16.0 / cyc_per_tile);
    fflush(stdout);
}

// the ring with TWO tiles per step: one counted wait + one barrier + four LDS-DMA issues per 32 KiB (half the synchronisation per byte)
template <int RING, int NREAD, int NMFMA>
__global__ __launch_bounds__(512) void ring_probe2(const RingArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.span, 0x00020000);
    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);
    unsigned off[2];
    const unsigned kt_per_row = a.rowb / 128u;
    const unsigned ntiles_src = (a.span / a.rowb / 128u) * kt_per_row;
#pragma unroll
    for (int q = 0; q < 2; ++q) off[q] = (8 * (wave + 8 * q) + lrow) * a.rowb + lsl * 16u;
    unsigned cur = (a.rot * blockIdx.x) % ntiles_src;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const unsigned so = (cur / kt_per_row) * 128u * a.rowb + (cur % kt_per_row) * 128u;
        cur = cur + 1 == ntiles_src ? 0 : cur + 1;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + stage * 16384 + (wave + 8 * q) * 1024), 16, (int)off[q], (int)so, 0, 0);
    };
    static_assert(RING % 2 == 0, "pairs of stages");
#pragma unroll
    for (int t = 0; t < RING - 2; ++t) issue(t);
    float4v acc[4] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
    half8v xr[2];
    for (int e = 0; e < 8; ++e) { xr[0][e] = (half_t)(0.001f * lane); xr[1][e] = (half_t)(0.002f * lane); }
    int stage = 0, istage = RING - 2;
    const int swz = l15 & 7;
    for (int t = 0; t < a.tiles; t += 2) {
        // tiles t, t + 1 have landed when the RING - 4 tiles behind them may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (RING - 4)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + RING - 2 < a.tiles) { issue(istage); issue(istage + 1); }
        istage = istage + 2 == RING ? 0 : istage + 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const half_t* tile = reinterpret_cast<const half_t*>(lds + (stage + u) * 16384);
            half8v wf[NREAD];
#pragma unroll
            for (int j = 0; j < NREAD; ++j) {
                const int kk = j & 1, jj = (j >> 1) & 7;
                wf[j] = *reinterpret_cast<const half8v*>(tile + (16 * jj + l15) * 64 + ((((kk * 4 + g) ^ swz)) << 3));
            }
#pragma unroll
            for (int m = 0; m < NMFMA; ++m)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[m % NREAD], xr[(m >> 2) & 1], acc[m & 3], 0, 0, 0);
        }
        stage = stage + 2 == RING ? 0 : stage + 2;
    }
    a.sink[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int RING, int NREAD, int NMFMA>
static void run_ring2(const RingArgs& a, int blocks, double ghz, int cus) {
    const int smem = RING * 16384;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_probe2<RING, NREAD, NMFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ring_probe2<RING, NREAD, NMFMA>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ring_probe2<RING, NREAD, NMFMA>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rounds = (double)((blocks + cus - 1) / cus);
    const double cyc_per_tile = ms * 1e-3 * ghz * 1e9 / (a.tiles * rounds);
    printf("TWO TILES PER BARRIER ring %d stages, %2d reads, %2d MFMAs per wave and tile, strided rows, rot %3u, blocks %4d: %7.3f ms  %6.0f cycles per tile and block  %5.1f B/clk/CU  MFMA pipe %4.2f\n",
           RING, NREAD, NMFMA, a.rot, blocks, ms, cyc_per_tile, 16384.0 / cyc_per_tile, NMFMA * 2 * 16.0 / cyc_per_tile);
    fflush(stdout);
}

template <int MODE, int DEPTH, int WAVES>
static double run(const char* name, const Args& a, int blocks, double clock_ghz, int cus) {
    const int smem = WAVES * DEPTH * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, DEPTH, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE, DEPTH, WAVES>), dim3(blocks), dim3(64 * WAVES), smem, 0, a);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE, DEPTH, WAVES>), dim3(blocks), dim3(64 * WAVES), smem, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)blocks * a.span * a.iters;
    const double tbs = bytes / (ms * 1e-3) * 1e-12;
    const int active = blocks < cus ? blocks : cus;
    printf("%-34s depth %2d waves %d blocks %4d span %5u KiB %s: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU (at %.2f GHz, %d CUs)\n", name, DEPTH, WAVES, blocks,
           a.span / 1024, a.stride ? "own bytes " : "same bytes", ms, tbs, bytes / (ms * 1e-3) / (clock_ghz * 1e9) / active, clock_ghz, active);
    fflush(stdout);
    return tbs;
}

int main(int argc, char**) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    const size_t total = (size_t)1 << 30;
    char* src;
    unsigned* sink;
    CK(hipMalloc(&src, total));
    CK(hipMemset(src, 1, total));
    CK(hipMalloc(&sink, (size_t)4096 * 1024 * 4));
    for (int same = 1; same >= 0 && argc < 2; --same) {
        for (unsigned span_kib : {2048u, 512u}) {
            Args a;
            a.src = src; a.sink = sink; a.span = span_kib * 1024u;
            a.stride = same ? 0u : a.span;
            a.rot = 0; a.stream = nullptr; a.stream_every = 1;
            a.iters = same ? (span_kib == 2048 ? 8 : 32) : (span_kib == 2048 ? 2 : 8);   // own bytes: 512 MiB / 128 MiB in all -- the second fits the infinity cache
            for (int blocks : {cus, 2 * cus}) {
                run<0, 2, 8>("LDS-DMA", a, blocks, ghz, cus);
                run<0, 8, 8>("LDS-DMA", a, blocks, ghz, cus);
                run<0, 16, 4>("LDS-DMA", a, blocks, ghz, cus);
                run<1, 2, 8>("load to registers", a, blocks, ghz, cus);
                run<1, 8, 8>("load to registers", a, blocks, ghz, cus);
                run<1, 8, 4>("load to registers", a, blocks, ghz, cus);
                run<2, 8, 8>("load to registers + ds_write", a, blocks, ghz, cus);
            }
        }
    }
    // the fused feed-forward's situation: every CU walks the same 2.4 MB of weights, each from its own starting point, and one 1-KiB piece
    // in ten is fresh activation data
    if (argc < 2) {
        Args a;
        a.src = src; a.sink = sink; a.span = 2400u * 1024u; a.stride = 0; a.iters = 8; a.rot = 0; a.stream = nullptr; a.stream_every = 1;
        printf("-- 2400 KiB of shared bytes, in phase\n");
        run<0, 2, 8>("LDS-DMA", a, cus, ghz, cus);
        a.rot = 131;
        printf("-- the same, every block from its own starting point\n");
        run<0, 2, 8>("LDS-DMA", a, cus, ghz, cus);
        run<0, 8, 8>("LDS-DMA", a, cus, ghz, cus);
        a.stream = src + ((size_t)256 << 20);
        for (int ev : {8, 2, 1}) {
            a.stream_every = ev;
            printf("-- + 1 KiB of fresh bytes per wave every %d x depth pieces (counted bytes: the shared walk only)\n", ev);
            run<0, 2, 8>("LDS-DMA", a, cus, ghz, cus);
        }
    }
    {
        RingArgs a;
        a.src = src; a.sink = reinterpret_cast<float*>(sink); a.span = 2560u * 640u; a.tiles = 1600; a.rot = 7; a.rowb = 640;
        printf("-- the weight-stream ring in isolation (shared 1.6 MB matrix [2560][320] halfs)\n");
        run_ring<8, 0, 0, 0, true>(a, cus, ghz, cus);
        run_ring<8, 0, 0, 1, true>(a, cus, ghz, cus);
        run_ring<8, 0, 0, 1, false>(a, cus, ghz, cus);
        run_ring<3, 0, 0, 1, true>(a, cus, ghz, cus);
        run_ring<8, 8, 0, 1, true>(a, cus, ghz, cus);
        run_ring<8, 16, 0, 1, true>(a, cus, ghz, cus);
        run_ring<8, 8, 16, 1, true>(a, cus, ghz, cus);
        run_ring<8, 8, 16, 0, true>(a, cus, ghz, cus);
        run_ring<8, 8, 16, 1, false>(a, cus, ghz, cus);
        run_ring<4, 8, 16, 1, true>(a, cus, ghz, cus);
        run_ring<8, 8, 32, 1, true>(a, cus, ghz, cus);
        run_ring<8, 16, 32, 1, true>(a, cus, ghz, cus);
        run_ring<8, 8, 16, 1, true>(a, 2 * cus, ghz, cus);
        a.rot = 0;
        run_ring<8, 8, 16, 1, true>(a, cus, ghz, cus);
        a.rot = 7;
        run_ring_pipe<8, 8, 16>(a, cus, ghz, cus);
        run_ring_pipe<6, 8, 16>(a, cus, ghz, cus);
        run_ring_pipe<8, 8, 32>(a, cus, ghz, cus);
        run_ring_pipe<8, 16, 32>(a, cus, ghz, cus);
        run_ring_pipe<8, 4, 16>(a, cus, ghz, cus);
        run_ring_pipe<8, 8, 16, 1>(a, cus, ghz, cus);
        run_ring_pipe<8, 8, 32, 1>(a, cus, ghz, cus);
        run_ring_pipe<8, 4, 16, 1>(a, cus, ghz, cus);
        run_ring2<8, 8, 16>(a, cus, ghz, cus);
        run_ring2<8, 4, 16>(a, cus, ghz, cus);
        run_ring2<8, 8, 32>(a, cus, ghz, cus);
    }
    return 0;
}