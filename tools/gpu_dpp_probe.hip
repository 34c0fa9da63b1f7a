// GPU probe: quad-permute DPP broadcast as used by gc_zstd_dec_seqv_kernel (hipcc --offload-arch=gfx950 tools/gpu_dpp_probe.hip -o /tmp/dpp_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int K> __device__ __forceinline__ uint32_t zv_quad(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, true); }
__global__ void probe(uint32_t* out)
{
    const uint32_t lane = threadIdx.x, v = lane * 10u + 1u;
    out[lane] = zv_quad<0>(v); out[64 + lane] = zv_quad<1>(v); out[128 + lane] = zv_quad<2>(v);
}
int main()
{
    uint32_t* d; uint32_t h[192];
    hipMalloc(&d, sizeof(h)); hipLaunchKernelGGL(probe, 1, 64, 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 3; k++) for (int l = 0; l < 64; l++) if (h[k * 64 + l] != (uint32_t)(((l & 60) + k) * 10 + 1)) bad++;
    printf("dpp quad broadcast: %d wrong of 192; lane 5: %u %u %u\n", bad, h[5], h[69], h[133]);
    return 0;
}
