// fetch_calib.hip — what rocprofv3's FETCH_SIZE reports for k_nfm_fwd's IQ access pattern, on a known byte count.
// MI355X_MICROARCH.md calibrates the counter only for wide coalesced reads (16 B / lane: FETCH_SIZE = half the bytes) and calls every other
// access width uncalibrated.  k_nfm_fwd's workers read their frames with 8-byte buffer loads, lane = frame (8 KB apart), nine consecutive
// samples per lane and chunk — this kernel does exactly that over a batch read exactly once (and a wide coalesced copy of the same batch beside it):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -o fc -- /tmp/fetch_calib
// -> FETCH_SIZE of k_strided / k_wide against the 512 MiB each of them reads (tools/prof_round.sh prints the ratio into the round summary).
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 1024;          // samples per frame (complex64)
constexpr long NF = 65536;       // frames: 512 MiB

// one wavefront per tile of 64 frames, lane = frame; per "chunk" of 8 samples: nine 8-byte loads (the ninth overlaps the next chunk), as the workers do
__global__ __launch_bounds__(64) void k_strided(const float2 *__restrict__ iq, float *__restrict__ out)
{
    const long f = (long)blockIdx.x * 64 + threadIdx.x;
    const float2 *x = iq + f * N;
    float acc = 0.0f;
    for (int c = 0; c + 9 <= N; c += 8) {
        float2 v[9];
#pragma unroll
        for (int e = 0; e < 9; e++) v[e] = x[c + e];
#pragma unroll
        for (int e = 0; e < 9; e++) acc += v[e].x * v[e].y;
    }
    out[f] = acc;
}

__global__ __launch_bounds__(256) void k_wide(const float4 *__restrict__ iq, float *__restrict__ out, long n4)
{
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = iq[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    float2 *iq;
    float *out;
    const size_t bytes = (size_t)NF * N * sizeof(float2);
    hipMalloc(&iq, bytes);
    hipMalloc(&out, NF * sizeof(float));
    hipMemset(iq, 0, bytes);
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(k_strided, dim3(NF / 64), dim3(64), 0, 0, iq, out);
        hipLaunchKernelGGL(k_wide, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4 *>(iq), out, (long)(bytes / 16));
    }
    hipDeviceSynchronize();
    printf("fetch_calib: k_strided and k_wide each read %zu bytes (%.1f MiB) per launch\n", bytes, bytes / 1048576.0);
    return 0;
}
