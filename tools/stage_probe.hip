// stage_probe.hip -- the implicit GEMM's two-stage K loop in isolation: how should the operand tiles reach the LDS?  (run on the MI355X)
//
//     hipcc --offload-arch=gfx950 -O3 -o tools/scratch/stage_probe tools/stage_probe.hip && tools/scratch/stage_probe
//
// profiles/r06r_ring_probe.log found the LDS side of a tile loop additive: the LDS-DMA of a 16-KiB tile costs ~340-430 cycles of the CU's
// LDS (38-48 B/clk) and the fragment reads (128 B/clk) do not overlap it.  A ds_write_b128 out of registers writes 128 B/clk.  This probe
// runs gemm2_kernel's SCHED-0 loop shape (8 waves, 2 LDS stages, per K step: fetch tile k+1, read fragments + MFMA on tile k, barrier) with
// the tile of a K step split into PD 1-KiB pieces per wave by LDS-DMA and PR pieces per wave through registers (buffer_load_b128 issued
// ahead of the MFMAs, ds_write_b128 behind them), at the read / MFMA mix of the 256 x 256 (8 pieces, 24 reads, 64 MFMAs per wave and
// K step) and 256 x 320 (9, 26, 80) tiles.  Timing only; every CU walks the same 1.6 MB (L2-resident) matrix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct StageArgs {
    const char* src;
    float* sink;
    unsigned span;   // bytes of the shared matrix (a multiple of the K step's bytes)
    int steps;       // K steps per block
    unsigned rot;    // block b starts rot * b K steps in
    // REAL: the A half of a K step's pieces are the block's OWN rows of a [mtiles * 256][krow] matrix (8 blocks of an XCD share an
    // m-tile, as the n-tiles of the product do; every krow / 64 steps the block moves on to another m-tile), the other half are rows
    // of a shared [256 (320)][krow] weight matrix; rows are krow halfs apart (a 1-KiB piece = 8 rows x 128 B, swizzled on the source)
    const char* amat;
    unsigned krow;     // halfs per row
    unsigned mtiles;
};

template <int PD, int PR, int NREAD, int NMFMA, int REAL = 0, int M32 = 0>
__global__ __launch_bounds__(512, 2) void stage_probe(const StageArgs a) {
    constexpr int P = PD + PR;                 // 1-KiB pieces per wave and K step
    constexpr int STEP = 8 * P * 1024;         // bytes per K step and block
    constexpr int ROWS = 64 * P;               // 128-byte rows per stage
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.span, 0x00020000);
    const unsigned nsteps_src = a.span / STEP;
    unsigned cur = (a.rot * blockIdx.x) % nsteps_src;
    const unsigned lane_off = (unsigned)lane * 16u;
    uint4v stg[PR > 0 ? PR : 1];
    // REAL: per-piece lane offsets (row * pitch + swizzled slot), the K walk is the scalar offset
    constexpr int PA = P / 2;   // A pieces per wave; the rest are weight pieces
    const unsigned pitch = a.krow * 2u;
    const unsigned steps_per_row = a.krow / 64u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, 64u * P * pitch, 0x00020000);
    unsigned voff[P];
    if constexpr (REAL) {
        const unsigned lrow = lane >> 3, lsl = (lane & 7) ^ lrow;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const unsigned row = 8u * (wave + 8u * (q < PA ? q : q - PA)) + lrow;
            voff[q] = row * pitch + lsl * 16u;
        }
    }
    unsigned mt = ((blockIdx.x & 7) * 4u + ((blockIdx.x >> 3) >> 3)) % a.mtiles, kstep = 0;
    auto fetch = [&](int stage) __attribute__((always_inline)) {
        if constexpr (REAL) {
            static_assert(!REAL || PR == 0, "REAL: LDS-DMA only");
            const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.amat + (size_t)mt * 256u * pitch), 0, 256u * pitch, 0x00020000);
            const unsigned so = kstep * 128u;
#pragma unroll
            for (int q = 0; q < P; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(q < PA ? rA : rW, (__attribute__((address_space(3))) void*)(lds + stage * STEP + (wave + 8 * q) * 1024), 16,
                                                         (int)voff[q], (int)so, 0, 0);
            if (++kstep == steps_per_row) {
                kstep = 0;
                mt = mt + 32u >= a.mtiles ? mt + 32u - a.mtiles : mt + 32u;
            }
            return;
        }
        const unsigned so = cur * STEP;
        cur = cur + 1 == nsteps_src ? 0 : cur + 1;
#pragma unroll
        for (int q = 0; q < PD; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + stage * STEP + (wave + 8 * q) * 1024), 16,
                                                     (int)((wave + 8 * q) * 1024u + lane_off), (int)so, 0, 0);
#pragma unroll
        for (int q = 0; q < PR; ++q)
            stg[q] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((wave + 8 * (PD + q)) * 1024u + lane_off), (int)so, 0);
    };
    auto commit = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PR; ++q)
            *reinterpret_cast<uint4v*>(lds + stage * STEP + (wave + 8 * (PD + q)) * 1024 + lane_off) = stg[q];
    };
    float4v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
    // M32: the same FLOPs as NMFMA / 2 v_mfma_f32_32x32x16_f16 (half the operand registers read per FLOP)
    typedef float float16v __attribute__((ext_vector_type(16)));
    float16v acc32[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
    half8v xr[2];
    for (int e = 0; e < 8; ++e) { xr[0][e] = (half_t)(0.001f * lane); xr[1][e] = (half_t)(0.002f * lane); }
    const int swz = l15 & 7;
    auto mma = [&](int stage) __attribute__((always_inline)) {
        const half_t* tile = reinterpret_cast<const half_t*>(lds + stage * STEP);
        // two halves like mma_tile's kk loop: NREAD / 2 fragment reads, then NMFMA / 2 MFMAs on them
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8v wf[NREAD > 1 ? NREAD / 2 : 1];
#pragma unroll
            for (int j = 0; j < NREAD / 2; ++j) {
                const int row = (16 * (j + wave) + l15) % ROWS;
                wf[j] = *reinterpret_cast<const half8v*>(tile + row * 64 + ((((kk * 4 + g) ^ swz)) << 3));
            }
            if constexpr (NMFMA > 0 && NREAD > 1 && M32) {
#pragma unroll
                for (int m = 0; m < NMFMA / 4; ++m)
                    acc32[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[m % (NREAD / 2)], xr[(m >> 2) & 1], acc32[m & 3], 0, 0, 0);
            } else if constexpr (NMFMA > 0 && NREAD > 1) {
#pragma unroll
                for (int m = 0; m < NMFMA / 2; ++m)
                    acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[m % (NREAD / 2)], xr[(m >> 3) & 1], acc[m & 7], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < NREAD / 2; ++j) acc[j & 7][0] += (float)wf[j][0];   // (keeps the reads alive)
            }
        }
    };
    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    commit(0);
    __syncthreads();
    for (int t = 0; t < a.steps - 1; ++t) {
        const int s = t & 1;
        fetch(s ^ 1);
        mma(s);
        if constexpr (PR > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            commit(s ^ 1);
        }
        __syncthreads();
    }
    mma((a.steps - 1) & 1);
    float f = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f += acc[i][i & 3];
#pragma unroll
    for (int i = 0; i < 4; ++i) f += acc32[i][i];
    a.sink[blockIdx.x * 512 + tid] = f;
}

template <int PD, int PR, int NREAD, int NMFMA, int REAL = 0, int M32 = 0>
static void run_stage(const StageArgs& a0, int blocks, double ghz, int cus, const char* note = "") {
    constexpr int STEP = 8 * (PD + PR) * 1024;
    StageArgs a = a0;
    a.span = (a0.span / STEP) * STEP;
    const int smem = 2 * STEP;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_probe<PD, PR, NREAD, NMFMA, REAL, M32>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stage_probe<PD, PR, NREAD, NMFMA, REAL, M32>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stage_probe<PD, PR, NREAD, NMFMA, REAL, M32>), dim3(blocks), dim3(512), smem, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rounds = (double)((blocks + cus - 1) / cus);
    const double cyc = ms * 1e-3 * ghz * 1e9 / (a.steps * rounds);
    printf("%d KiB per K step: %d pieces per wave by LDS-DMA + %d through registers, %2d reads, %2d MFMAs per wave and step, blocks %4d: %7.3f ms  %6.0f cycles per K step  "
           "MFMA pipe %4.2f  staged %5.1f B/clk/CU  %s\n",
           8 * (PD + PR), PD, PR, NREAD, NMFMA, blocks, ms, cyc, NMFMA * 2 * 16.0 / cyc, (double)STEP / cyc, note);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    char* src;
    float* sink;
    CK(hipMalloc(&src, (size_t)64 << 20));
    CK(hipMemset(src, 1, (size_t)64 << 20));
    CK(hipMalloc(&sink, (size_t)1024 * 512 * 4));
    StageArgs a;
    a.src = src; a.sink = sink; a.span = 2560u * 640u; a.steps = 400; a.rot = 3;
    a.amat = nullptr; a.krow = 2560; a.mtiles = 128;
    printf("-- 256 x 256 tile's mix (64 KiB per K step, 24 fragment reads, 64 MFMAs per wave)\n");
    run_stage<8, 0, 24, 64>(a, cus, ghz, cus);
    run_stage<4, 4, 24, 64>(a, cus, ghz, cus);
    run_stage<2, 6, 24, 64>(a, cus, ghz, cus);
    run_stage<0, 8, 24, 64>(a, cus, ghz, cus);
    printf("-- 256 x 320 tile's mix (72 KiB per K step, 26 fragment reads, 80 MFMAs per wave)\n");
    run_stage<9, 0, 26, 80>(a, cus, ghz, cus);
    run_stage<5, 4, 26, 80>(a, cus, ghz, cus);
    run_stage<0, 9, 26, 80>(a, cus, ghz, cus);
    printf("-- no MFMAs (the LDS side alone)\n");
    run_stage<8, 0, 24, 0>(a, cus, ghz, cus);
    run_stage<0, 8, 24, 0>(a, cus, ghz, cus);
    run_stage<8, 0, 0, 0>(a, cus, ghz, cus);
    run_stage<0, 8, 0, 0>(a, cus, ghz, cus);
    printf("-- no fragment reads (staging + MFMAs)\n");
    run_stage<8, 0, 2, 64>(a, cus, ghz, cus);
    run_stage<0, 8, 2, 64>(a, cus, ghz, cus);
    // ---- the same loop on the product's addressing and on real bytes ----
    {
        const size_t abytes = (size_t)128 * 256 * 2560 * 2;   // 128 m-tiles x 256 rows x 2560 halfs = 168 MB
        char* amat;
        CK(hipMalloc(&amat, abytes));
        a.amat = amat;
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (rnd == 0) {
                CK(hipMemset(amat, 1, abytes));
                CK(hipMemset(src, 1, (size_t)64 << 20));
            } else {
                // uniform fp16 values in (-1, 1): the switching activity of real activations / weights
                std::vector<unsigned short> h(abytes / 2);
                unsigned long long st = 88172645463325252ull;
                for (size_t i = 0; i < h.size(); ++i) {
                    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
                    const float f = ((st >> 11) & 0xFFFFFF) / 8388608.0f - 1.0f;
                    _Float16 hf = (_Float16)f;
                    unsigned short u;
                    __builtin_memcpy(&u, &hf, 2);
                    h[i] = u;
                }
                CK(hipMemcpy(amat, h.data(), abytes, hipMemcpyHostToDevice));
                CK(hipMemcpy(src, h.data(), (size_t)64 << 20, hipMemcpyHostToDevice));
            }
            const char* note = rnd ? "random fp16 bytes" : "constant bytes";
            printf("-- %s: contiguous shared tiles (as above), then the product's addressing (own A rows of a 168 MB matrix + shared weight rows, rows 5 KiB apart)\n", note);
            run_stage<8, 0, 24, 64>(a, cus, ghz, cus, note);
            run_stage<9, 0, 26, 80>(a, cus, ghz, cus, note);
            run_stage<8, 0, 24, 64, 1>(a, cus, ghz, cus, note);
            run_stage<9, 0, 26, 80, 1>(a, cus, ghz, cus, note);   // (9 pieces: 4 A + 5 W = the 256 x 320 tile)
            run_stage<8, 0, 24, 64, 1>(a, 2 * cus, ghz, cus, note);
            run_stage<8, 0, 24, 64, 1, 1>(a, cus, ghz, cus, rnd ? "random fp16 bytes, 32 x 32 x 16 MFMAs (same FLOPs)" : "constant bytes, 32 x 32 x 16 MFMAs (same FLOPs)");
            run_stage<9, 0, 26, 80, 1, 1>(a, cus, ghz, cus, rnd ? "random fp16 bytes, 32 x 32 x 16 MFMAs (same FLOPs)" : "constant bytes, 32 x 32 x 16 MFMAs (same FLOPs)");
            run_stage<8, 0, 12, 64, 1>(a, cus, ghz, cus, rnd ? "random fp16 bytes, half the fragment reads" : "constant bytes, half the fragment reads");
            run_stage<8, 0, 24, 0, 1>(a, cus, ghz, cus, rnd ? "random fp16 bytes, no MFMAs" : "constant bytes, no MFMAs");
        }
    }
    return 0;
}
