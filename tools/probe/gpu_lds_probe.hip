// tools/probe/gpu_lds_probe.hip -- measurement only (not product): what LDS operations on random slots cost a CU on gfx950.
// One wave per workgroup with a private 16 KiB table (as W4's long table); W workgroups per CU.  Prints cycles of CU time per wave-instruction.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/gpu_lds_probe.hip -o gpurun_out/lds_probe && ./gpurun_out/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
template <int OP, int PAD>
__global__ void __launch_bounds__(64) probe(uint32_t* out, uint32_t seed)
{
    __shared__ uint32_t tab[4096 + PAD / 4];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) tab[i] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * 64u + lane) * 2654435761u + seed, acc = 0;
    for (uint32_t it = 0; it < ITERS; it += 8) {
        uint32_t r[8];
#pragma unroll
        for (int d = 0; d < 8; d++) {
            x = x * 1664525u + 1013904223u;
            uint32_t slot = OP == 3 || OP == 7 ? ((lane + 64u * ((x >> 20) & 63u)) & 4095u) : (x >> 20);      // 3, 7: conflict-free (lane-strided)
            const uint32_t val = ((it + d + 1u) << 8) | lane;
            if (OP == 0 || OP == 3) r[d] = atomicMax(&tab[slot], val);                       // returning ds_max
            else if (OP == 1) { atomicMax(&tab[slot], val); r[d] = 0; }                       // non-returning
            else if (OP == 2 || OP == 7) { r[d] = tab[slot]; __builtin_amdgcn_wave_barrier(); tab[slot] = val; }   // plain read + plain write
            else if (OP == 4) { r[d] = tab[slot]; }                                            // plain read only
            else if (OP == 5) { tab[slot] = val; r[d] = 0; }                                   // plain write only
            else if (OP == 6) { r[d] = atomicAdd(&tab[slot], 1u); }                            // returning add
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int d = 0; d < 8; d++) acc += r[d];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc + tab[lane];
}
template <int OP, int PAD> void run(const char* name, int wgPerCu, uint32_t* d)
{
    const int grid = 256 * wgPerCu * 4;       // four rounds of resident workgroups
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<OP, PAD><<<grid, 64>>>(d, 1u); hipDeviceSynchronize();
    hipEventRecord(a); probe<OP, PAD><<<grid, 64>>>(d, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double instrPerCu = (double)grid / 256.0 * ITERS;
    printf("%-34s lds/wg %6d B  %2d wg/CU  %8.3f ms  %7.1f cycles of CU time per wave-op (at 2.4 GHz)\n", name, 16384 + PAD, wgPerCu, ms, ms * 1e-3 * 2.4e9 / instrPerCu);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 1 << 22);
    // PAD sizes the LDS footprint so that exactly wgPerCu workgroups fit a CU (160 KiB)
    run<0, 8192>("ds_max_rtn random", 6, d);      // 24 KiB like W4: 6 per CU
    run<0, 0>("ds_max_rtn random", 10, d);        // 16 KiB: 10 per CU
    run<0, 65536>("ds_max_rtn random", 2, d);
    run<3, 8192>("ds_max_rtn conflict-free", 6, d);
    run<1, 8192>("ds_max (no return) random", 6, d);
    run<6, 8192>("ds_add_rtn random", 6, d);
    run<2, 8192>("ds_read + ds_write random", 6, d);
    run<7, 8192>("ds_read + ds_write conflict-free", 6, d);
    run<4, 8192>("ds_read random", 6, d);
    run<5, 8192>("ds_write random", 6, d);
    return 0;
}
