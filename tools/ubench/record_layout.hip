// Micro-benchmark (round 6): how the general derivative record (5 548 doubles = 86.7 pieces of 512 B per step, 47.9 KB with the
// rest of trajEl_t) should lie in HBM for its two users:
//   the producer  k_derivs_wave<false>: a wavefront = 64 consecutive steps of a trajectory; with the entries of a piece collected in
//                 LDS it writes, per piece, 64 x 512 B — one 512-byte store instruction per step
//   the consumer  k_backward_wave<false>: a wavefront = a trajectory, step after step (descending), 86 loads of 512 B per step and then
//                 the step's arithmetic (emulated: a dependent fp64 chain)
// Layouts: T = steps per tile; the 512-byte piece p of step k lies at
//     tile(k / T) + p * (T * 512) + (k % T) * 512           tile = T * REC bytes, REC = PIECES * 512
//   T = 1 is the reference's array of structs (what the kernels use today), T = 64 makes the producer's 64 stores of a piece one burst of 32 KB.
// Prints the achieved GB/s of each (layout, role).
//   hipcc --offload-arch=gfx950 -O2 record_layout.hip -o record_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)

constexpr int PIECES = 88;            // 512-byte pieces per record (45 056 B; the product's record has 86.7 + the head)
constexpr size_t REC = PIECES * 512;  // bytes per step

__device__ __forceinline__ size_t piece_at(int T, int N, int b, int k, int p) {
    const size_t traj = (size_t)b * N * REC;
    return traj + (size_t)(k / T) * T * REC + (size_t)p * T * 512 + (size_t)(k % T) * 512;
}

// producer: wavefront w owns steps [64 w', 64 w' + 64) of trajectory b; lane l stores entry l of the piece, one step per instruction
__global__ __launch_bounds__(256) void k_write(char *buf, int T, int N, int B, int work, double seed) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int per = N / 64;
    const int b = wave / per, k0 = (wave % per) * 64;
    if(b >= B) return;
    double acc = seed + lane;
    for(int p = 0; p < PIECES; p++) {
        for(int j = 0; j < work; j++) acc = acc * 0.999999 + 1e-9;  // the generated code between two flushes
#pragma unroll 8
        for(int s = 0; s < 64; s++) {
            double *dst = reinterpret_cast<double *>(buf + piece_at(T, N, b, k0 + s, p)) + lane;
            *dst = acc + s;
        }
    }
}

// the producer again, one wavefront per workgroup with `lds` bytes of dynamic LDS each: as many wavefronts per CU as that leaves
// (33 KB: four, the product kernel's occupancy)
// stagger: 0 none; 1: wavefront w begins with (w mod 4) quarters of a piece's arithmetic, so that the four wavefronts of a CU are a
// quarter period apart; 2: a piece's 64 stores in four instalments between quarters of the NEXT piece's arithmetic (what a second
// ring would allow)
__global__ __launch_bounds__(64) void k_write_occ(char *buf, int T, int N, int B, int work, double seed, int stagger) {
    extern __shared__ double pad[];
    const int wave = blockIdx.x, lane = threadIdx.x & 63;
    const int per = N / 64;
    const int b = wave / per, k0 = (wave % per) * 64;
    if(b >= B) return;
    double acc = seed + lane;
    pad[lane] = acc;
    if(stagger == 1)
        for(int j = 0; j < (wave & 3) * (work / 4); j++) acc = acc * 0.999999 + 1e-9;
    for(int p = 0; p < PIECES; p++) {
        if(stagger == 2) {
            const double prev = acc;
            for(int q = 0; q < 4; q++) {
                for(int j = 0; j < work / 4; j++) acc = acc * 0.999999 + 1e-9;
                if(p > 0)
#pragma unroll 8
                    for(int s = 16 * q; s < 16 * q + 16; s++) {
                        double *dst = reinterpret_cast<double *>(buf + piece_at(T, N, b, k0 + s, p - 1)) + lane;
                        *dst = prev + s;
                    }
            }
            acc += pad[(lane + p) & 63];
            continue;
        }
        for(int j = 0; j < work; j++) acc = acc * 0.999999 + 1e-9;
        acc += pad[(lane + p) & 63];
#pragma unroll 8
        for(int s = 0; s < 64; s++) {
            double *dst = reinterpret_cast<double *>(buf + piece_at(T, N, b, k0 + s, p)) + lane;
            *dst = acc + s;
        }
    }
}

// consumer: one wavefront per trajectory (persistent over a queue), steps N-1 .. 0, 86 loads of 512 B each then `work` dependent FMAs
template <int PF>
__global__ __launch_bounds__(64, 2) void k_read(const char *buf, int T, int N, int B, int work, int *queue, double *out) {
    const int lane = threadIdx.x & 63;
    double total = 0.0;
    for(;;) {
        int b = 0;
        if(lane == 0) b = atomicAdd(queue, 1);
        b = __builtin_amdgcn_readfirstlane(b);
        if(b >= B) break;
        double acc = 1.0;
        unsigned carried = 0;  // the touches of the previous step: consumed a step later, so that nothing waits for them
        for(int k = N - 1; k >= 0; k--) {
            total += (carried == 0x7fffffffu) ? 1.0 : 0.0;
            double s = 0.0;
#pragma unroll 8
            for(int p = 0; p < PIECES; p++) s += reinterpret_cast<const double *>(buf + piece_at(T, N, b, k, p))[lane];
            if(PF && k > 0) {  // touch every line of the next step's record: one dword per 128-byte line, result unused until the end
                unsigned d = 0;
                for(int q = 0; q < (PIECES * 4 + 63) / 64; q++) {
                    const int line = q * 64 + lane;
                    if(line < PIECES * 4)
                        d += *reinterpret_cast<const unsigned *>(buf + piece_at(T, N, b, k - 1, line >> 2) + (line & 3) * 128);
                }
                carried = d;
            }
            for(int j = 0; j < work; j++) acc = acc * 0.999999 + s * 1e-9;
        }
        total += acc;
    }
    out[blockIdx.x * 64 + lane] = total;
}

int main(int argc, char **argv) {
    const int N = argc > 2 ? atoi(argv[2]) : 128;
    const int B = argc > 1 ? atoi(argv[1]) : 4096;  // 4 096 trajectories x 128 steps x 45 KB = 23.6 GB
    const size_t bytes = (size_t)B * N * REC;
    char *buf;
    int *queue;
    double *out;
    CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMalloc((void **)&queue, 4));
    CHECK(hipMalloc((void **)&out, 4096 * 64 * 8));
    CHECK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("record %zu B/step, %d trajectories x %d steps = %.1f GB\n", REC, B, N, bytes / 1e9);
    const int Ts[] = {1, 4, 16, 64};
    for(int work : {0, 400}) {
        for(int T : Ts) {
            const int waves = B * (N / 64);
            for(int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_write, dim3((waves + 3) / 4), dim3(256), 0, 0, buf, T, N, B, work, 1.0);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if(rep) printf("write T=%2d work=%4d: %7.2f ms  %7.1f GB/s\n", T, work, ms, bytes / ms / 1e6);
            }
        }
    }
    for(int stagger = 0; stagger < 3; stagger++)
    for(int lds : {0, 16 * 1024, 33 * 1024, 60 * 1024}) {
        for(int work : {0, 400, 1200}) {
            if(stagger && (work == 0 || lds < 30000)) continue;
            const int waves = B * (N / 64);
            for(int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_write_occ, dim3(waves), dim3(64), lds, 0, buf, 1, N, B, work, 1.0, stagger);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if(rep) printf("write T= 1 work=%4d stagger=%d, one wavefront per workgroup with %5d B of LDS: %7.2f ms  %7.1f GB/s\n", work, stagger, lds, ms, bytes / ms / 1e6);
            }
        }
    }
    for(int work : {0, 2500}) {
        for(int pf = 0; pf < 2; pf++) {
            for(int T : Ts) {
                for(int rep = 0; rep < 2; rep++) {
                    CHECK(hipMemset(queue, 0, 4));
                    CHECK(hipEventRecord(e0));
                    if(pf) hipLaunchKernelGGL(k_read<1>, dim3(2048), dim3(64), 0, 0, buf, T, N, B, work, queue, out);
                    else hipLaunchKernelGGL(k_read<0>, dim3(2048), dim3(64), 0, 0, buf, T, N, B, work, queue, out);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if(rep) printf("read  T=%2d work=%4d prefetch=%d: %7.2f ms  %7.1f GB/s\n", T, work, pf, ms, bytes / ms / 1e6);
                }
            }
        }
    }
    CHECK(hipGetLastError());
    return 0;
}
