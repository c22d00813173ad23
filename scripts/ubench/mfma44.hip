// Dev microbenchmark (MI355X): v_mfma_f64_4x4x4_4b_f64 -- rate next to the 16x16x4 form and the operand layout.
// Four independent 4x4x4 products per instruction (one per 16-lane group): the natural tile for four n <= 16 models
// per wavefront.  Layout check: for every block b, D = A B + C with random data against a host product, trying the
// candidate mappings (which of lane&3 / (lane>>2)&3 is the row and which the k / column index).
//   hipcc --offload-arch=gfx950 -O3 -o mfma44 mfma44.hip && ./mfma44
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_layout(const double *a, const double *b, const double *c, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
template <int MODE>
__global__ void __launch_bounds__(512) k_rate(double *out, int iters)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = threadIdx.x * 2e-3 + 0.5;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    v4d w0 = {0, 0, 0, 0}, w1 = w0, w2 = w0, w3 = w0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { // 8 independent chains
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
        } else if (MODE == 1) { // one dependent chain
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        } else if (MODE == 2) { // product chaining: the result is the B operand of the next product
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c1, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c2, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c3, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c4, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c5, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c6, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c7, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, c7, 0, 0, 0);
        } else { // 16x16x4 for reference (2 instructions per iteration slot count as 8 below via scaling)
            w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w0, 0, 0, 0);
            w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w1, 0, 0, 0);
            w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w2, 0, 0, 0);
            w3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w3, 0, 0, 0);
            w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w0, 0, 0, 0);
            w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w1, 0, 0, 0);
            w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w2, 0, 0, 0);
            w3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w3, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + w0.x + w1.y + w2.z + w3.w;
}
template <int MODE>
void rate(const char *name, double *d, int wpb)
{
    const int iters = 20000;
    k_rate<MODE><<<256, 64 * wpb>>>(d, 200);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k_rate<MODE><<<256, 64 * wpb>>>(d, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / (double(iters) * 8) / (wpb / 4.0);
    printf("%-44s waves/SIMD=%d %8.3f ms  %6.2f ns per instruction per SIMD (%.1f clk @2.4 GHz)\n", name, wpb / 4, ms, ns, ns * 2.4);
}
int main()
{
    std::vector<double> a(64), b(64), c(64), d(64);
    srand(1);
    for (int i = 0; i < 64; ++i) { a[i] = rand() % 17 - 8; b[i] = rand() % 13 - 6; c[i] = rand() % 11 - 5; }
    double *da, *db, *dc, *dd;
    CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dc, 512)); CK(hipMalloc(&dd, 512));
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(da, db, dc, dd);
    CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
    // brute force over the roles of the lane bit fields x0 = l & 3, x1 = (l >> 2) & 3, x2 = l >> 4 for each operand
    const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const char *names[3] = {"l&3", "(l>>2)&3", "l>>4"};
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        double A[4][4][4], B[4][4][4], C[4][4][4], D[4][4][4]; // [block][row][col]
        for (int l = 0; l < 64; ++l) {
            const int x[3] = {l & 3, (l >> 2) & 3, l >> 4};
            A[x[perms[pa][2]]][x[perms[pa][0]]][x[perms[pa][1]]] = a[l]; // (i, k, blk)
            B[x[perms[pb][2]]][x[perms[pb][0]]][x[perms[pb][1]]] = b[l]; // (k, j, blk)
            C[x[perms[pd][2]]][x[perms[pd][0]]][x[perms[pd][1]]] = c[l]; // (i, j, blk)
            D[x[perms[pd][2]]][x[perms[pd][0]]][x[perms[pd][1]]] = d[l];
        }
        int bad = 0;
        for (int blk = 0; blk < 4; ++blk)
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
                double sum = C[blk][i][j];
                for (int k = 0; k < 4; ++k) sum += A[blk][i][k] * B[blk][k][j];
                if (sum != D[blk][i][j]) ++bad;
            }
        if (bad == 0)
            printf("LAYOUT: A (i,k,blk) = (%s, %s, %s)   B (k,j,blk) = (%s, %s, %s)   D (i,j,blk) = (%s, %s, %s)\n",
                   names[perms[pa][0]], names[perms[pa][1]], names[perms[pa][2]], names[perms[pb][0]], names[perms[pb][1]],
                   names[perms[pb][2]], names[perms[pd][0]], names[perms[pd][1]], names[perms[pd][2]]);
    }
    double *out;
    CK(hipMalloc(&out, 256 * 1024 * 8));
    for (int wpb : {4, 8}) {
        rate<0>("mfma_f64_4x4x4_4b, 8 independent", out, wpb);
        rate<1>("mfma_f64_4x4x4_4b, dependent chain", out, wpb);
        rate<2>("mfma_f64_4x4x4_4b, result -> B operand chain", out, wpb);
        rate<3>("mfma_f64_16x16x4, 4 independent", out, wpb);
    }
    return 0;
}
