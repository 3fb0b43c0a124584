// Micro-benchmark: issue rate vs dependent-chain latency of fp64 VALU instructions on gfx950,
// one wavefront per SIMD (the occupancy of the lane-mapped sequential kernels).
//   hipcc --offload-arch=gfx950 -O2 valu_latency.hip -o valu_latency && ./valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)

template <int CHAINS, int OP>
__global__ __launch_bounds__(64) void k_chain(double *out, int iters, double a, double b, long long *cyc) {
    double v[CHAINS];
#pragma unroll
    for(int c = 0; c < CHAINS; c++) v[c] = a + threadIdx.x * 1e-9 + c;
    long long t0 = __builtin_readcyclecounter();
    for(int i = 0; i < iters; i++) {
#pragma unroll
        for(int r = 0; r < 16; r++) {
#pragma unroll
            for(int c = 0; c < CHAINS; c++) {
                if(OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(b), "v"(a));
                if(OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(b));
                if(OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(b));
                if(OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[c]));
                if(OP == 4) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[c]));
                if(OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float *)&v[c]) : "v"((float)b), "v"((float)a));
                if(OP == 6) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(v[c]) : "v"(b) : "vcc");
                if(OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(int *)&v[c]) : "v"(1) : "vcc");
                if(OP == 8) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(v[c]));
                if(OP == 9) asm volatile("v_sqrt_f64 %0, %0" : "+v"(v[c]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for(int c = 0; c < CHAINS; c++) s += v[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if(threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS, int OP>
int run(const char *name, int blocks, int waves_per_block = 1) {
    double *out;
    long long *cyc, hc;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * waves_per_block * 8));
    CHECK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<CHAINS, OP>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, 10, 1.0, 1.0000001, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<CHAINS, OP>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, iters, 1.0, 1.0000001, cyc);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHECK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * 16 * CHAINS;
    printf("%-14s chains %d blocks %5d x %d waves: %7.2f ns/instr/wave  %6.2f counter ticks/instr  (%.3f ms)\n", name, CHAINS, blocks, waves_per_block,
           ms * 1e6 / n, (double)hc / n, ms);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int B = p.multiProcessorCount * 4;  // one wave per SIMD
    run<1, 0>("fma_f64", B); run<2, 0>("fma_f64", B); run<4, 0>("fma_f64", B); run<8, 0>("fma_f64", B);
    run<1, 1>("add_f64", B); run<4, 1>("add_f64", B);
    run<1, 2>("mul_f64", B); run<4, 2>("mul_f64", B);
    run<1, 3>("rcp_f64", B); run<4, 3>("rcp_f64", B);
    run<1, 4>("rsq_f64", B); run<4, 4>("rsq_f64", B);
    run<1, 9>("sqrt_f64", B); run<4, 9>("sqrt_f64", B);
    run<1, 5>("fma_f32", B); run<4, 5>("fma_f32", B);
    run<1, 6>("div_scale_f64", B); run<4, 6>("div_scale_f64", B);
    run<1, 7>("cndmask_b32", B); run<4, 7>("cndmask_b32", B);
    run<1, 8>("ldexp_f64", B); run<4, 8>("ldexp_f64", B);
    // more waves per SIMD, dependent chain
    run<1, 0>("fma_f64", B * 2); run<1, 0>("fma_f64", B * 4); run<1, 0>("fma_f64", B * 8);
    run<4, 0>("fma_f64", B * 2); run<4, 0>("fma_f64", B * 4);
    return 0;
}
