// Dev microbenchmark 2: the filter's full per-step store mix, to find which component throttles HBM writes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
// FLAGS bit0: 2 block arrays (Pp,Pf) ; bit1: 2 vector arrays (Xp,F) 80 B/model ; bit2: 2 scalar arrays (sigmas, detfs) 8 B/model
// bit3: busy VALU (independent FMAs) instead of idle
template <int FLAGS>
__global__ void k(v2d* a0, v2d* a1, double* v0, double* v1, double* s0, double* s1, int T, long nw, int work)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = lane + i;
    const double* yp = v1 + ((wave * 4 + g) * 8 + (l & 7));  // FLAGS&16: obs-like read, time-major [T][models][8]
    double ynext = (FLAGS & 16) ? yp[0] : 0.0;
    for (int t = 0; t < T; ++t) {
        if (FLAGS & 16) { acc[0] += ynext; ynext = yp[(long)(t + 1 < T ? t + 1 : t) * nw * 32]; }
        if (FLAGS & 8) { for (int s = 0; s < work; ++s) { for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], 1.0000001, 1e-9); } }
        else { for (int s = 0; s < work * 8; ++s) acc[0] = fma(acc[0], 1.0000001, 1e-9); }
        v2d val = {acc[0], acc[1]};
        if (FLAGS & 1) {
            v2d* b0 = a0 + ((long)t * nw + wave) * 200; v2d* b1 = a1 + ((long)t * nw + wave) * 200;
#pragma unroll
            for (int m = 0; m < 4; ++m) { int c = l + 16 * m; if (c > 49) c = 49; b0[g * 50 + c] = val; }
#pragma unroll
            for (int m = 0; m < 4; ++m) { int c = l + 16 * m; if (c > 49) c = 49; b1[g * 50 + c] = val; }
        }
        if (FLAGS & 2) {
            const int r = l > 9 ? 9 : l;
            v0[(((long)t * nw + wave) * 4 + g) * 10 + r] = acc[2];
            v1[(((long)t * nw + wave) * 4 + g) * 10 + r] = acc[3];
        }
        if (FLAGS & 4) {
            if (l == 0) { s0[((long)t * nw + wave) * 4 + g] = acc[4]; s1[((long)t * nw + wave) * 4 + g] = acc[5]; }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7] == 12345.678) s0[0] = 1;
}
template <int FLAGS> void run(const char* name, v2d* a0, v2d* a1, double* v0, double* v1, double* s0, double* s1, int work)
{
    const long nw = 1024; const int T = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<FLAGS><<<256, 256>>>(a0, a1, v0, v1, s0, s1, 50, nw, work); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FLAGS><<<256, 256>>>(a0, a1, v0, v1, s0, s1, T, nw, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double gb = ((FLAGS & 1) ? 2.0 * 3200 : 0) + ((FLAGS & 2) ? 2.0 * 320 : 0) + ((FLAGS & 4) ? 2.0 * 32 : 0);
    gb = gb * T * nw / 1e9;
    printf("%-46s work=%3d  %.3f ms  %.2f GB  %.0f GB/s\n", name, work, ms, gb, gb / ms * 1e3);
}
int main()
{
    const size_t T = 1000, nw = 1024;
    v2d *a0, *a1; double *v0, *v1, *s0, *s1;
    hipMalloc(&a0, T * nw * 3200); hipMalloc(&a1, T * nw * 3200);
    hipMalloc(&v0, T * nw * 320); hipMalloc(&v1, T * nw * 320); hipMalloc(&s0, T * nw * 32); hipMalloc(&s1, T * nw * 32);
    for (int work : {40}) {
        run<8>("busy VALU, no stores", a0, a1, v0, v1, s0, s1, work);
        run<8 | 1>("busy VALU + 2 block arrays", a0, a1, v0, v1, s0, s1, work);
        run<8 | 2>("busy VALU + 2 vector arrays", a0, a1, v0, v1, s0, s1, work);
        run<8 | 4>("busy VALU + 2 scalar arrays", a0, a1, v0, v1, s0, s1, work);
        run<8 | 1 | 2>("busy VALU + blocks + vectors", a0, a1, v0, v1, s0, s1, work);
        run<8 | 1 | 2 | 4>("busy VALU + blocks + vectors + scalars", a0, a1, v0, v1, s0, s1, work);
        run<8 | 16>("busy VALU + obs-like reads only", a0, a1, v0, v1, s0, s1, work);
        run<8 | 1 | 16>("busy VALU + 2 block arrays + obs-like reads", a0, a1, v0, v1, s0, s1, work);
        run<0>("idle VALU (dependent chain), no stores", a0, a1, v0, v1, s0, s1, work);
        run<1 | 2 | 4>("idle VALU + blocks + vectors + scalars", a0, a1, v0, v1, s0, s1, work);
    }
    return 0;
}
