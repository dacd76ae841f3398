// timeline_ts.hip -- a one-lane kernel that stores the constant-rate wall clock (100 MHz) into a slot: tools/gpu_timeline.py launches one
// behind every C-ABI launch of the step, on the launch's own stream, so that the hipGraph replay of the two-stream step leaves a
// timeline (rocprofv3 serialises the streams under tracing: profiles/r06b_trace_overlap.json).  Tool code, not part of libmusev_hip.
//     hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/scratch/libmvts.so tools/timeline_ts.hip
#include <hip/hip_runtime.h>
__global__ void mvts_kernel(unsigned long long* slots, int idx) {
    if (threadIdx.x == 0) slots[idx] = wall_clock64();
}
extern "C" int mvts_record(void* slots, int idx, void* stream) {
    hipLaunchKernelGGL(mvts_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)slots, idx);
    return (int)hipGetLastError();
}
