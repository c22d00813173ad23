// Dev microbenchmark: HBM write throughput of the filter's store pattern vs a wave-linear one.
// 1024 wavefronts (256 blocks x 256 threads), T steps, each wave writes 3200 B per array per step into
// time-major arrays [T][waves][3200 B].  MODE 0: per 16-lane group runs of 256 B at g*800 + m*256 (current
// BlockIO mapping); MODE 1: wave-linear lane*16 + m*1024; NARR arrays written per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE, int NARR>
__global__ void k(v2d* a0, v2d* a1, v2d* a2, v2d* a3, int T, long nw, int spin)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    v2d* arr[4] = {a0, a1, a2, a3};
    v2d val = {1.0 * lane, 2.0};
    double acc = lane;
    for (int t = 0; t < T; ++t) {
        for (int s = 0; s < spin; ++s) acc = fma(acc, 1.0000001, 1e-9);  // stand-in for compute
        val.x = acc;
#pragma unroll
        for (int ar = 0; ar < NARR; ++ar) {
            v2d* base = arr[ar] + ((long)t * nw + wave) * 200;  // 200 chunks of 16 B = 3200 B
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                int q;
                if (MODE == 0) { int g = lane >> 4, l = lane & 15; int c = l + 16 * m; if (c > 49) c = 49; q = g * 50 + c; }
                else { q = lane + 64 * m; if (q > 199) q = 199; }
                base[q] = val;
            }
        }
    }
}
template <int MODE, int NARR> void run(const char* name, v2d** d, int T, int spin)
{
    const long nw = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NARR><<<256, 256>>>(d[0], d[1], d[2], d[3], T, nw, spin);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NARR><<<256, 256>>>(d[0], d[1], d[2], d[3], T, nw, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double gb = double(NARR) * T * nw * 3200 / 1e9;
    printf("%-44s spin=%4d  %.3f ms  %.2f GB  %.0f GB/s\n", name, spin, ms, gb, gb / ms * 1e3);
}
int main()
{
    const int T = 1000;
    v2d* d[4];
    for (int i = 0; i < 4; ++i) hipMalloc(&d[i], (size_t)T * 1024 * 3200);
    for (int spin : {0, 300}) {
        run<0, 1>("group-runs (current), 1 array", d, T, spin);
        run<1, 1>("wave-linear, 1 array", d, T, spin);
        run<0, 2>("group-runs (current), 2 arrays", d, T, spin);
        run<1, 2>("wave-linear, 2 arrays", d, T, spin);
        run<0, 4>("group-runs (current), 4 arrays", d, T, spin);
        run<1, 4>("wave-linear, 4 arrays", d, T, spin);
    }
    return 0;
}
