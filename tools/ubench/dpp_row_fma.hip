// Micro-benchmark for the row-broadcast product of the wave mapping (gfx950):
//   acc += (value of lane S of my 16-lane row) * (my own value)
// as ONE instruction, v_fmac_f64_dpp ... row_newbcast:S, against the two-instruction form
// (v_mov_b64_dpp + v_fma_f64) and against an LDS broadcast read (all 16 lanes of a row read
// the same address) + v_fma_f64.  Also checks the semantics of row_newbcast on 64-bit operands
// and whether a VALU write directly in front of the DPP read needs wait states.
//   hipcc --offload-arch=gfx950 -O2 dpp_row_fma.hip -o dpp_row_fma && ./dpp_row_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)

#define FMAC_DPP(acc, a, b, S) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #S " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b))
#define MOV_DPP(dst, a, S) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #S " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(a))

// MODE 0: fmac_dpp, 8 independent accumulators x 16 source lanes per trip
// MODE 1: mov_dpp + fma
// MODE 2: LDS broadcast read (ds_read_b64, one address per row) + fma
// MODE 3: fmac_dpp, ONE accumulator (dependent chain)
template <int MODE>
__global__ __launch_bounds__(64) void k_prod(double *out, int iters, const double *in, long long *cyc) {
    __shared__ double sh[64 * 8];
    const int lane = threadIdx.x;
    double a[8], b[16], c[8];
#pragma unroll
    for(int i = 0; i < 8; i++) { a[i] = in[lane * 8 + i]; c[i] = 0.0; }
#pragma unroll
    for(int i = 0; i < 16; i++) b[i] = in[512 + lane * 16 + i];
#pragma unroll
    for(int i = 0; i < 8; i++) sh[i * 64 + lane] = a[i];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for(int it = 0; it < iters; it++) {
#define ROUND(S)                                                                        \
        _Pragma("unroll") for(int r = 0; r < 8; r++) {                                  \
            if(MODE == 0) FMAC_DPP(c[r], a[r], b[S], S);                                \
            if(MODE == 1) { double t; MOV_DPP(t, a[r], S); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[r]) : "v"(t), "v"(b[S])); } \
            if(MODE == 2) { const double t = sh[r * 64 + (lane & 48) + S]; c[r] += t * b[S]; } \
            if(MODE == 3) FMAC_DPP(c[0], a[r], b[S], S);                                \
        }
        ROUND(0) ROUND(1) ROUND(2) ROUND(3) ROUND(4) ROUND(5) ROUND(6) ROUND(7)
        ROUND(8) ROUND(9) ROUND(10) ROUND(11) ROUND(12) ROUND(13) ROUND(14) ROUND(15)
#undef ROUND
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for(int i = 0; i < 8; i++) s += c[i];
    out[blockIdx.x * 64 + lane] = s;
    if(lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// semantics + hazard check: r = (a*1.0 written by a VALU right before) read through DPP
template <int NOPS>
__global__ void k_sem(const double *in, double *out) {
    const int lane = threadIdx.x;
    double a = in[lane], acc = 0.0, one = 1.0;
    if(NOPS == 0) asm volatile("v_mul_f64 %1, %1, %3\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(a) : "v"(one), "v"(one));
    if(NOPS == 2) asm volatile("v_mul_f64 %1, %1, %3\n s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(a) : "v"(one), "v"(one));
    out[lane] = acc;
}

template <int MODE>
int run(const char *name, int blocks, const double *din) {
    double *out;
    long long *cyc, hc;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * 8));
    CHECK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_prod<MODE>), dim3(blocks), dim3(64), 0, 0, out, 10, din, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_prod<MODE>), dim3(blocks), dim3(64), 0, 0, out, iters, din, cyc);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHECK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * 128;
    printf("%-26s blocks %5d: %7.2f ns per multiply-add per wave, %6.2f counter ticks  (%.3f ms)\n", name, blocks, ms * 1e6 / n, (double)hc / n, ms);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    std::vector<double> h(512 + 1024);
    for(size_t i = 0; i < h.size(); i++) h[i] = 1.0 + 1e-3 * (double)(i % 97);
    double *din;
    CHECK(hipMalloc(&din, h.size() * 8));
    CHECK(hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice));

    // semantics: out[lane] must be in[(lane & 48) + 5]
    {
        std::vector<double> hi(64), ho(64);
        for(int i = 0; i < 64; i++) hi[i] = 100.0 + i;
        double *di, *dout;
        CHECK(hipMalloc(&di, 64 * 8)); CHECK(hipMalloc(&dout, 64 * 8));
        CHECK(hipMemcpy(di, hi.data(), 64 * 8, hipMemcpyHostToDevice));
        for(int nops = 0; nops <= 2; nops += 2) {
            if(nops == 0) hipLaunchKernelGGL((k_sem<0>), dim3(1), dim3(64), 0, 0, di, dout);
            else hipLaunchKernelGGL((k_sem<2>), dim3(1), dim3(64), 0, 0, di, dout);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ho.data(), dout, 64 * 8, hipMemcpyDeviceToHost));
            int bad = 0;
            for(int i = 0; i < 64; i++) if(ho[i] != hi[(i & 48) + 5]) bad++;
            printf("row_newbcast:5 on f64, VALU write + %d wait states before the DPP read: %d of 64 lanes wrong (lane 0 got %g, lane 17 got %g)\n", nops, bad, ho[0], ho[17]);
        }
        hipFree(di); hipFree(dout);
    }

    const int B = p.multiProcessorCount * 4;
    for(int w = 1; w <= 8; w *= 2) {
        run<0>("fmac_dpp 8 chains", B * w, din);
        run<1>("mov_dpp + fma 8 chains", B * w, din);
        run<2>("lds bcast + fma 8 chains", B * w, din);
        run<3>("fmac_dpp 1 chain", B * w, din);
    }
    return 0;
}
