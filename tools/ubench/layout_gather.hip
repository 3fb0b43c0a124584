// Micro-benchmark: cost of the per-step nominal loads of a roll-out lane under three access patterns, with the
// arithmetic of a step emulated by a dependent fp64 chain.
//   tiled      [step][tile][component][lane]   lanes of a wavefront = consecutive trajectories (coalesced rows)
//   tiled+perm the same array, lanes = a random subset of trajectories (what a compacted list of pending
//              trajectories looks like): every lane reads 8 bytes of a different 512-byte row
//   aos        [trajectory][step][component]   each lane reads its own 128 contiguous bytes per step
//   hipcc --offload-arch=gfx950 -O2 layout_gather.hip -o layout_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)
constexpr int W = 16;  // doubles per step and trajectory (x, u, l, L of CarParking)

template <int MODE, int WR>
__global__ __launch_bounds__(64) void k(double *a, const int *__restrict__ perm, double *out, int B, int Bp, int N, int work) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if(t >= B) return;
    const int b = (MODE == 1) ? perm[t] : t;
    double acc = 1.0 + blockIdx.y;
    double cur[W], nxt[W];
    auto load = [&](double *dst, int kk) {
        if(MODE == 2) {
            const double *p = a + ((size_t)b * N + kk) * W;
#pragma unroll
            for(int i = 0; i < W; i++) dst[i] = p[i];
        } else {
            const double *p = a + (size_t)kk * W * Bp + (size_t)(b >> 6) * (W * 64) + (b & 63);
#pragma unroll
            for(int i = 0; i < W; i++) dst[i] = p[i * 64];
        }
    };
    load(cur, 0);
    __builtin_amdgcn_s_waitcnt(0);
    for(int kk = 0; kk < N; kk++) {
        if(kk + 1 < N) load(nxt, kk + 1);
        double s = 0;
#pragma unroll
        for(int i = 0; i < W; i++) s += cur[i];
        for(int j = 0; j < work; j++) acc = acc * 0.999999 + s * 1e-9;  // dependent chain = the step's arithmetic
        if(WR) {  // the winner pass: 6 of the 16 doubles are rewritten in place
            if(MODE == 2) {
                double *p = a + ((size_t)b * N + kk) * W;
#pragma unroll
                for(int i = 0; i < 6; i++) p[i] = acc + i;
            } else {
                double *p = a + (size_t)kk * W * Bp + (size_t)(b >> 6) * (W * 64) + (b & 63);
#pragma unroll
                for(int i = 0; i < 6; i++) p[i * 64] = acc + i;
            }
        }
#pragma unroll
        for(int i = 0; i < W; i++) cur[i] = nxt[i];
    }
    out[(size_t)blockIdx.y * Bp + t] = acc;
}

int main() {
    const int Bp = 65536, N = 500;
    double *a, *out; int *perm;
    CHECK(hipMalloc(&a, (size_t)Bp * N * W * 8));
    CHECK(hipMemset(a, 0, (size_t)Bp * N * W * 8));
    CHECK(hipMalloc(&out, (size_t)Bp * 8 * 8));
    std::vector<int> p(Bp);
    for(int i = 0; i < Bp; i++) p[i] = i;
    std::mt19937 rng(1);
    CHECK(hipMalloc(&perm, Bp * 4));
    for(int frac : {100, 15, 4}) {          // % of trajectories in the list
        for(int sorted : {0, 1}) {
            std::vector<int> q = p;
            std::shuffle(q.begin(), q.end(), rng);
            const int B = Bp * frac / 100;
            q.resize(B);
            if(sorted) std::sort(q.begin(), q.end());
            CHECK(hipMemcpy(perm, q.data(), B * 4, hipMemcpyHostToDevice));
            for(int alphas : {1, 5}) {
                for(int work : {100, 400}) {
                    float ms[3];
                    for(int mode = 0; mode < 3; mode++) {
                        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                        dim3 grid((B + 63) / 64, alphas);
                        for(int rep = 0; rep < 2; rep++) {
                            hipEventRecord(e0);
                            if(mode == 0) hipLaunchKernelGGL((k<0, 0>), grid, dim3(64), 0, 0, a, perm, out, B, Bp, N, work);
                            if(mode == 1) hipLaunchKernelGGL((k<1, 0>), grid, dim3(64), 0, 0, a, perm, out, B, Bp, N, work);
                            if(mode == 2) hipLaunchKernelGGL((k<2, 0>), grid, dim3(64), 0, 0, a, perm, out, B, Bp, N, work);
                            hipEventRecord(e1);
                            CHECK(hipDeviceSynchronize());
                        }
                        hipEventElapsedTime(&ms[mode], e0, e1);
                    }
                    printf("list %3d%% %s  alphas %d  work %3d fma/step:  tiled(first B) %.3f ms   tiled+list %.3f ms   aos %.3f ms\n",
                           frac, sorted ? "sorted  " : "shuffled", alphas, work, ms[0], ms[1], ms[2]);
                }
            }
        }
    }
    // the winner pass: whole batch, one wavefront per 64 trajectories, 6 of 16 doubles rewritten in place
    for(int work : {100, 400}) {
        float ms[2];
        for(int mode = 0; mode < 2; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for(int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if(mode == 0) hipLaunchKernelGGL((k<0, 1>), dim3(Bp / 64, 1), dim3(64), 0, 0, a, perm, out, Bp, Bp, N, work);
                if(mode == 1) hipLaunchKernelGGL((k<2, 1>), dim3(Bp / 64, 1), dim3(64), 0, 0, a, perm, out, Bp, Bp, N, work);
                hipEventRecord(e1);
                CHECK(hipDeviceSynchronize());
            }
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
        printf("read 16 + write 6 doubles per step, work %3d: tiled %.3f ms   aos %.3f ms\n", work, ms[0], ms[1]);
    }
    return 0;
}
