// mk_wide.hip -- RTS smoother for wide models (16 < n <= 64 states, one model per wavefront): the round-1
// broadcast kernel (smoother_wave_kernel, kept as the A/B reference) and the MFMA kernel that replaces it.
// Reference semantics: kalmansmoother, /root/reference/metran/kalmanfilter.py:403-476.
// A separate translation unit in the library build (compiles in parallel with mk_kernels.hip); textually
// included by mk_kernels.hip when a run-time shape module is built (MK_SHAPE_MODULE, metran_amd/jit.py).
#include <cstdlib>

#include "mk_prims.h"

namespace mk {

// =====================================================================================
// one model per wavefront (16 < n <= 64): the row-per-lane arrays of the kernels above would need
// ~10 x n doubles per lane and spill (measured at n = 36: 1433 spilled VGPRs, 8x slower than the
// filter).  Here D = Ps[t+1] - Pp[t+1], the rows of Pf[t] and the gain J live in a wave-private LDS
// block; the two n^3 sweeps (V = J D, Ps = Pf + V J^T) are RUN-TIME loops whose operand rows are read
// from LDS at wavefront-uniform addresses (LDS broadcast, no v_readlane), so the scheduler cannot
// hoist n*n loads across the factorisation (it did, and spilled them, when the sweeps were unrolled)
// and the code stays inside the instruction cache.  <= 3 row arrays are live at any time.
template <int n>
constexpr int wave_kernel_wpb() // wavefronts per workgroup: one, so that LDS (2 n x n matrices per wavefront)
{                               // is handed out at wavefront granularity: 7 resident wavefronts per CU at n = 36
    return 1;
}

template <int N, int K, bool PROJ>
__global__ void __launch_bounds__(64 * wave_kernel_wpb<N + K>()) __attribute__((amdgpu_waves_per_eu(2, 2)))
smoother_wave_kernel(SmootherArgs a)
{
    constexpr int n = N + K, G = 64, WPB = wave_kernel_wpb<n>();
    static_assert(n <= G, "state dimension must fit the wavefront");
    using Gp = Group<G>;
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * WPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1;
    const int r = lane < n ? lane : n - 1;
    const long T = a.T;
    __shared__ __attribute__((aligned(16))) double lds_m[WPB * 2 * n * n];
    double *Dm = lds_m + (threadIdx.x / G) * 2 * n * n; // D row-major; after the V sweep: rows of V J^T
    double *Jm = Dm + n * n;                            // rows of the smoother gain J
    double *Dr = Dm + r * n;
    const double *Jr = Jm + r * n;

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];

    const long blkT = inst * a.bs + (T - 1) * a.ts;
    MomentPtr iF = moment_ptr<n>(const_cast<double *>(a.F), const_cast<double *>(a.Pf), blkT, a.ts, a.rs, r);
    MomentPtr oS = moment_ptr<n>(a.S, a.Ps, blkT, a.ts, a.rs, r);
    constexpr int NV = record_payload(n), RS = record_stride_c(n), PADN = RS - NV;
    double *padS = (a.rs > 0 && a.S) ? a.S + blkT * RS + NV + (lane < PADN ? lane : PADN - 1) : nullptr;

    const long rec_id = inst % a.R;
    const int jr = lane < N ? lane : N - 1;
    double gam[K], pscale = 1.0, poffset = 0.0;
    double *pM = nullptr, *pV = nullptr;
    if constexpr (PROJ) {
#pragma unroll
        for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec_id * N + jr) * K + k];
        if (a.scale) pscale = a.scale[rec_id * N + jr];
        if (a.offset) poffset = a.offset[rec_id * N + jr];
        const long pidx = blkT * N + jr;
        pM = a.sim_means ? a.sim_means + pidx : nullptr;
        pV = a.sim_vars ? a.sim_vars + pidx : nullptr;
    }
    auto store = [&](double xv, const double(&row)[n]) __attribute__((always_inline)) {
        if (oS.vec) *oS.vec = xv;
        if (oS.mat) store_cols<n>(oS.mat, row);
        oS.advance(-1);
        if (padS) {
            *padS = 0.0;
            padS -= a.ts * RS;
        }
        if constexpr (PROJ) {
            double mean, var;
            project<N, K, G>(xv, row, gam, pscale, poffset, lane, mean, var);
            if (lane < N && live) {
                if (pM) *pM = mean;
                if (pV) *pV = var;
            }
            if (pM) pM -= a.ts * N;
            if (pV) pV -= a.ts * N;
        }
    };

    // last step: smoothed = filtered (:450-451)
    double xs = *iF.vec, Psn[n];
    load_cols<n>(iF.mat, Psn);
    store(xs, Psn);
    double pivmin = 1.0;

    for (long t = T - 2; t >= 0; --t) {
        iF.advance_nn(-1);
        const double xfc = *iF.vec;
        double A[n], z[n];
        {
            double Pfc[n];
            load_cols<n>(iF.mat, Pfc);
            wave_lds_sync(); // previous iteration's reads of Dm / Jm are complete
            // element by element, D streamed to LDS in 16-byte pieces: Pf, Ps[t+1], W, Pp and D rows
            // all live at once would be 5n doubles per lane
            double dprev = 0.0;
            // (opaque copies: the n loop-invariant selects "c == r ? q : 0" would otherwise be hoisted out
            // of the time loop and spilled -- 2n VGPRs that are not there)
            int rv = r;
            double qv = q_r;
            asm volatile("" : "+v"(rv), "+v"(qv));
            sfor<0, n>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                z[c] = Pfc[c] * Gp::template bcast<c>(phi_r); // W = Pf Phi
                A[c] = fma(phi_r, z[c], c == rv ? qv : 0.0);   // Pp[t+1] row
                const double d = Psn[c] - A[c];
                if constexpr (n % 2 == 0) {
                    if constexpr (c % 2 == 1) *reinterpret_cast<v2d *>(Dr + c - 1) = v2d{dprev, d};
                    else dprev = d;
                } else {
                    Dr[c] = d;
                }
                if constexpr (c % 8 == 7) __builtin_amdgcn_sched_barrier(0);
            });
        }
        const double delta = xs - phi_r * xfc;
        __builtin_amdgcn_sched_barrier(0);

        // ---- A = L D L^T (rows distributed over the lanes), as in the group kernels ----
        // (the unused diagonal slot of lane j, L(j,j) = 1, keeps 1/d_j: no separate dinv[] array)
        sfor<0, n>(MK_LAMBDA(jc) {
            constexpr int j = decltype(jc)::value;
            const double piv = Gp::template bcast<j>(A[j]);
            pivmin = min_f64(pivmin, piv);
            const double ij = piv > 0.0 ? rcp_nr(piv) : 0.0; // d_j <= 0: null direction dropped (see ldlt_factor)
            const double lr = A[j] * ij;
            Gp::template axpy_lane<j, j + 1, n, true, n>(A, A, lr);
            A[j] = j == lane ? ij : lr;
            __builtin_amdgcn_sched_barrier(0);
        });
        // the broadcast operands of the two substitutions are all known once the factorisation is done;
        // tying each stage's operand to that stage's pivot element (empty asm, no instruction) keeps the
        // compiler from running the n^2/2 readlanes ahead of the FMAs that consume them
        sfor<0, n>(MK_LAMBDA(kc) {
            constexpr int k = decltype(kc)::value;
            asm volatile("" : "+v"(A[k]) : "v"(z[k]));
            Gp::template axpy_col<k + 1, n, true, n>(z, A[k], z[k]);
            __builtin_amdgcn_sched_barrier(0);
        });
        sfor<0, n>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            z[c] *= Gp::template bcast<c>(A[c]);
        });
        sfor_down<0, n>(MK_LAMBDA(kc) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int c = 0; c < k; ++c) asm volatile("" : "+v"(A[c]) : "v"(z[k]));
            Gp::template axpy_lane<k, 0, k, true, n>(z, A, z[k]);
            __builtin_amdgcn_sched_barrier(0);
        });
        store_row<n>(Jm + r * n, z); // lane r now holds row r of J = W Pp^-1

        double acc0 = xfc, acc1 = 0.0;
        sfor<0, n>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c % 2 == 0) Gp::template fmac<c>(acc0, delta, z[c]);
            else Gp::template fmac<c>(acc1, delta, z[c]);
        });
        xs = acc0 + acc1;
        wave_lds_sync(); // Dm, Jm visible to the whole wavefront
        __builtin_amdgcn_sched_barrier(0);
        // Pf[t] again (L2-resident; its registers were needed by the factorisation): lands during the sweeps
        double Pfc[n];
        load_cols<n>(iF.mat, Pfc);

        // V = J D: V[r][:] += J[r][k] * D[k][:], D row k broadcast from LDS
        double V[n];
#pragma unroll
        for (int c = 0; c < n; ++c) V[c] = 0.0;
#pragma unroll 1
        for (int k = 0; k < n; ++k) {
            const double zk = Jr[k];
            double Dk[n];
            load_row<n>(Dm + k * n, Dk);
#pragma unroll
            for (int c = 0; c < n; ++c) V[c] = fma(zk, Dk[c], V[c]);
        }
        // Ps[r][c] = Pf[r][c] + sum_k V[r][k] J[c][k], J row c broadcast from LDS.  c is a run-time index, so
        // the sums go back into this lane's row of Dm (every lane is past the V sweep: the wavefront runs
        // in lock-step and LDS operations of one wavefront complete in order)
        wave_lds_sync();
#pragma unroll 1
        for (int c = 0; c < n; ++c) {
            double Jc[n];
            load_row<n>(Jm + c * n, Jc);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int k = 0; k < n; ++k) {
                if (k % 4 == 0) s0 = fma(V[k], Jc[k], s0);
                else if (k % 4 == 1) s1 = fma(V[k], Jc[k], s1);
                else if (k % 4 == 2) s2 = fma(V[k], Jc[k], s2);
                else s3 = fma(V[k], Jc[k], s3);
            }
            Dr[c] = (s0 + s1) + (s2 + s3);
        }
        wave_lds_sync(); // orders the scalar stores above before the vector loads of the same row
        load_row<n>(Dr, Psn);
#pragma unroll
        for (int c = 0; c < n; ++c) Psn[c] += Pfc[c];
        store(xs, Psn);
    }
    if (a.status && live && lane == 0 && pivot_flags(pivmin)) atomicOr(a.status + inst, pivot_flags(pivmin));
}


// =====================================================================================
// smoother_mfma_kernel (round 2): the same recursion with the n^3 work moved off the broadcast path.
//   Measured on MI355X (scripts/ubench/mfma_f64.hip), SIMD cycles per 64-lane fused multiply-add:
//     v_readlane pair + v_fma_f64 ........ 14-19      (what every broadcast of smoother_wave_kernel costs)
//     uniform ds_read_b128 + 2 v_fma ..... 9-12 each  (LDS broadcast; the LDS pipe is shared by 4 SIMDs)
//     v_mfma_f64_16x16x4_f64 ............. 70 per instruction = 4.4 per 64 multiply-adds, and the MFMA
//                                          -- on the SAME pipe as the f64 VALU: an MFMA wavefront and an f64-FMA
//                                          wavefront on one SIMD take the SUM of their times (round-2 measurement)
//   (f64 MFMA has the same 32 flop/clk/SIMD peak as the f64 VALU on gfx950 -- the gain is that operands
//   need no broadcast and all 64 lanes carry data, not a higher peak and not concurrency.)
//   What changed against smoother_wave_kernel:
//   * factorisation: unchanged (rows distributed over the lanes, readlane broadcasts; n^2/2 of them);
//   * the two triangular substitutions (n^2 broadcasts) read L from an LDS copy at wavefront-uniform
//     addresses instead of 2 readlanes per element: forward in dot form over row c of L, backward in
//     axpy form over row k of L, so every read is a contiguous 16-byte pair of ONE row-major copy;
//   * V^T = D J^T and Ps = Pf + J V^T are 16x16x4 f64 MFMA tiles.  No operand ever needs a transposing
//     round trip: a C/D tile (row = (lane>>4) + 4 reg, col = lane&15) IS the B operand of k-step `reg`,
//     and the C-layout tile of X^T IS the A operand of X.  So J^T tiles (read once from the LDS rows of
//     J) serve as B operand of product 1 and, unchanged, as A operand (J) of product 2; the V^T tiles
//     product 1 leaves in registers are the B operand of product 2; D is symmetric, its C-layout tiles
//     are its A operand.  K runs over exactly ceil(n/4) k-steps (36 = 9 x 4: no padding in K); M and N
//     are padded to 16 and the padding only ever pollutes padded outputs.  Ps tiles with Ib <= Jb are
//     computed and mirrored into the LDS matrix the lanes then read their rows from.
// =====================================================================================
typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double global_double;

// rows of L^T packed for the back substitution: row c holds L(k, c) for k = lt_kb(c) .. n-1 (lt_kb(c) = (c+1) & ~1: the
// first pair starts at an even k, so rows are 16-byte aligned pairs), padded to an even count
constexpr int lt_kb(int c) { return (c + 1) & ~1; }
constexpr int lt_off(int n, int c)
{
    int s = 0;
    for (int i = 0; i < c; ++i) s += (n - lt_kb(i) + 1) & ~1;
    return s;
}


// EPI / SYM as in smoother_record_kernel: 0 records, 1 + projection, 2 + state means and variances; SYM: the
// filtered (and smoothed) records are packed-symmetric
// FOLD (round 3, last step): the fused factorisation / forward sweep carried TWO rows per lane -- row r of A = Pp (which
// becomes L) and row r of W (which becomes Z~) -- through the same recursion x_c -= U(c,k) x_k with the same uniform
// operands: 2 x n(n-1)/2 broadcast multiply-adds a step on n of the 64 lanes.  With n <= 36 the rows of A fit the lanes the
// model does not use: row r >= H of A rides in lane r + OFF (OFF = 64 - n, H = max(0, 2n - 64): lanes n .. 63 at n = 36
// carry rows 8 .. 35), so ONE multiply-add per (c, k) serves both recursions; the H head rows (they need columns c < H
// only: H(H-1)/2 = 28 multiply-adds at n = 36) stay in lanes 0 .. H-1 as a second, short array.  Per lane the arithmetic is
// the same sequence as before: the results are bit-identical to the unfolded kernel (FOLDP = false, kept behind
// mk_set_kernel_variant(ctx, MK_VARIANT_WIDE_SMOOTHER, 2) and tested for exactly that).  The A-lanes are no longer
// replicas of lane n-1: their own-row LDS writes go to a dummy row, the smoothed mean is re-broadcast from lane n-1, and
// every global store still sees replicas.
// (BLK4P: the 4x4x4 MFMA block path of rounds 3-4 -- measured 127 ms against 81 ms in round 4, removed from this file;
// source and record in scripts/experiments/.  The template parameter and the `BLK4 = false` branches are what is left.)
template <int N, int K, int EPI, bool SYM, bool FOLDP, bool BLK4P = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) smoother_mfma_kernel(SmootherArgs a)
{
    constexpr int n = N + K, G = 64;
    constexpr bool BLK4 = false;
    static_assert(!BLK4P, "the block path was removed (scripts/experiments/README.md)");
    static_assert(!BLK4 || (n % 4 == 0 && !SYM), "block path: n a multiple of 4, full-square records");
    constexpr int NBK = (n + 3) / 4;                 // BLK4: 4-wide blocks per dimension
    constexpr int NGR = (NBK + 1 + 3) / 4;           // registers per block row: NBK column blocks + the mean column
    constexpr int LSZ = 16 * (NBK * (NBK + 1) / 2);  // -L, lower blocks (diagonal included), 16 doubles each
    constexpr int ZROWS = 16 * NGR, ZSZ = 4 * n;     // one buffer of 4 columns of Z~ (two of them: filled one block ahead)
    constexpr bool FOLD = FOLDP && n <= 36;
    constexpr int H = FOLD && 2 * n > 64 ? 2 * n - 64 : 0; // head rows of A kept in lanes 0 .. H-1 (second array)
    constexpr int OFF = 64 - n;                            // row r >= H of A lives in lane r + OFF
    constexpr int AL0 = OFF + H;                           // first A-lane (= max(n, 64 - n))
    constexpr bool PROJ = (EPI == 1), VAR = (EPI == 2);
    static_assert(n > 16 && n <= G, "one model per wavefront, 16 < n <= 64");
    // STR (round 3): the state is [N series | K factors]; when N is a multiple of 16 and K <= 4 the two products run their
    // 16x16x4 tiles on the N x N series block only (K runs over all n: the factor rows are the last k-step) and the K-wide
    // factor border on the vector pipe -- at n = 36 the tiles of the padded 48 x 48 problem spend 44 % of their multiply-adds
    // on padding (135 tiles a step; 63 here)
    constexpr bool STR = (N % 16 == 0) && (K <= 4);
    constexpr int NB = STR ? N / 16 : (n + 15) / 16; // 16-wide tile rows / columns
    constexpr int KS = (n + 3) / 4;   // k-steps
    constexpr int LD = (n + 3) & ~1;  // LDS row stride: even (16-byte rows), n + 2 or n + 3
    constexpr int PADC = (n + 1) & ~1; // first of the (at least two) padding columns of a row of Dm: (phi_c, q_c)
    // second LDS region: U = L D packed by rows (row c: its c elements k < c, padded to an even count) during the
    // sweeps, then a 16-row staging buffer for J.  A full second n x LD matrix made the wavefront's share 22.4 KB:
    // 7 wavefronts per CU instead of the 8 its registers allow, and 4096 models = 2.3 rounds of 1792 instead of 2
    // of 2048 (measured at T = 500: 35.9 ms for 4096 models against 2 x 13.9 ms for 2 x 1792).
    constexpr int USZ = tri_off(n), JSZ = BLK4 ? LSZ : 16 * LD, LTSZ = BLK4 ? 0 : lt_off(n, n - 1) + 2;
    constexpr int XBASE = 2 * ZSZ;                   // BLK4: then the smoothed mean [n]
    constexpr int XSZ = BLK4 ? XBASE + ZROWS : 0;
    constexpr int RSZ = (USZ > JSZ ? USZ : JSZ) > LTSZ ? (USZ > JSZ ? USZ : JSZ) : LTSZ;
    const int lane = threadIdx.x;
    long inst = (long)blockIdx.x;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1;
    const int r = lane < n ? lane : n - 1;
    const long T = a.T;
    // (projection variant: the lane's K loadings, scale and offset wait in LDS between the steps -- 12 registers the loop
    // otherwise spilled, 9 scratch loads a step whose traffic reached HBM: 130 GB a launch against 114)
    // (BLK4: they stay in registers, as in the build the block path was measured with -- its LDS budget is the whole 1/8 CU)
    constexpr int PCS = PROJ && !BLK4 ? ((K + 3) & ~1) : 0;
    __shared__ __attribute__((aligned(16))) double lds_m[n * LD + RSZ + 64 + 64 * PCS + XSZ];
    double *Dm = lds_m;           // D = Ps[t+1] - Pp[t+1] row-major; after product 2: Ps[t]
    double *Um = Dm + n * LD;     // U packed; then 16 rows of J at a time
    double *dl = Um + RSZ;        // delta = xs[t+1] - Xp[t+1]
    [[maybe_unused]] double *zb = dl + 64 + 64 * PCS; // BLK4: 4 columns of Z~ / 4 rows of J^T / the smoothed mean on its way to the lanes
    double *Dr = Dm + r * LD;
    // FOLD: the row of A / L this lane carries (z-lanes: their own), and where the lane's own-row WRITES go (A-lanes: a dummy row)
    const bool isA = FOLD && lane >= AL0;
    const int rowA = isA ? lane - OFF : r;
    // (the dummy row is the delta buffer: LD <= 64 doubles, rewritten AFTER the top loop's D update and dead again when the
    // border columns are written)
    static_assert(!FOLD || LD <= 64, "dummy row does not fit the delta buffer");
    double *Dw = isA ? dl : Dr;
    double *Ur = Um + ((rowA & 1) ? 2 * (rowA >> 1) * ((rowA >> 1) + 1) : 2 * (rowA >> 1) * (rowA >> 1)); // = Um + tri_off(rowA)

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    const double phi_x = isA ? a.phi[inst * n + rowA] : 1.0; // FOLD: A-lanes scale W's row to A's (z-lanes: exact no-op)
    Dr[PADC] = phi_r; // diag(Phi), diag(Q) as wavefront-uniform LDS operands (the padding columns are never overwritten)
    Dr[PADC + 1] = q_r;

    const long blkT = inst * a.bs + (T - 1) * a.ts;
    MomentPtr iF = moment_ptr<n>(const_cast<double *>(a.F), const_cast<double *>(a.Pf), blkT, a.ts, a.rs, r);
    MomentPtr oS = moment_ptr<n>(a.S, a.Ps, blkT, a.ts, a.rs, r);
    constexpr int NV = SYM ? record_payload_sym(n) : record_payload(n), RS = SYM ? record_stride_sym_c(n) : record_stride_c(n);
    constexpr int PADN = RS - NV;
    double *padS = (a.rs > 0 && a.S) ? a.S + blkT * RS + NV + (lane < PADN ? lane : PADN - 1) : nullptr;

    const long rec_id = inst % a.R;
    const int jr = lane < N ? lane : N - 1;
    double *pM = nullptr, *pV = nullptr;
    [[maybe_unused]] double pcr[PROJ && BLK4 ? K + 2 : 1];   // (BLK4: in registers)
    double *pcl = BLK4 ? pcr : dl + 64 + lane * PCS; // this lane's projection constants [gam_0 .. gam_{K-1}, scale, offset]
    if constexpr (PROJ) {
#pragma unroll
        for (int k = 0; k < K; ++k) pcl[k] = a.loadings[(rec_id * N + jr) * K + k];
        pcl[K] = a.scale ? a.scale[rec_id * N + jr] : 1.0;
        pcl[K + 1] = a.offset ? a.offset[rec_id * N + jr] : 0.0;
        const long pidx = blkT * N + jr;
        pM = a.sim_means ? a.sim_means + pidx : nullptr;
        pV = a.sim_vars ? a.sim_vars + pidx : nullptr;
    }
    double *sM = nullptr, *sV = nullptr;
    if constexpr (VAR) {
        sM = a.state_means + blkT * n + r;
        sV = a.state_vars + blkT * n + r;
    }
    // packed-symmetric records: element (r, c) of the upper triangle stored by rows sits at K(c) + r for c <= r and at
    // offr + c for c > r.  The record base is the same in every lane (one model per wavefront): made a scalar, the
    // accesses are (scalar base) + (32-bit lane offset) and the offsets are n 32-bit registers, not n 64-bit pointers
    // (which the loop kept live and spilled: 145-167 scratch registers)
    const int offr = r * n - r * (r - 1) / 2 - r;
    auto uniform_ptr = [](double *p) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<global_double *>(((unsigned long long)hi << 32) | lo); // (global, not flat: scalar-base addressing)
    };
    auto at = [](auto *base, int idx) __attribute__((always_inline)) { // base + zero-extended 32-bit BYTE offset
        typedef __attribute__((address_space(1))) char global_char;
        return reinterpret_cast<decltype(base)>(reinterpret_cast<global_char *>(const_cast<global_double *>(base)) + 8u * (unsigned)idx);
    };
    const int offrA = rowA * n - rowA * (rowA - 1) / 2 - rowA;
    auto load_pf_rows = [&](double(&row)[n], bool fold = false) __attribute__((always_inline)) { // fold: row rowA instead of row r
        if constexpr (SYM) {
            const global_double *ub = uniform_ptr(iF.mat - r);
            int rv = fold ? rowA : r, ov = fold ? offrA : offr; // opaque copies: the n offsets are formed here, every time (hoisted, they are kept as 64-bit pairs)
            asm volatile("" : "+v"(rv), "+v"(ov));
#pragma unroll
            for (int c = 0; c < n; ++c) row[c] = *at(ub, rv >= c ? sym_row_offset(n, c) - c + rv : ov + c);
        } else {
            load_cols<n>(fold ? iF.mat + (rowA - r) : iF.mat, row);
        }
    };
    auto store = [&](double xv, const double(&row)[n]) __attribute__((always_inline)) {
        if (oS.vec) *oS.vec = xv;
        if (oS.mat) {
            if constexpr (SYM) {
                global_double *ub = uniform_ptr(oS.mat - r);
                int rv = r;
                asm volatile("" : "+v"(rv));
#pragma unroll
                for (int c = 0; c < n; ++c)
                    if (rv >= c) *at(ub, sym_row_offset(n, c) - c + rv) = row[c];
            } else {
                store_cols<n>(oS.mat, row);
            }
        }
        oS.advance(-1);
        if (padS) {
            *padS = 0.0;
            padS -= a.ts * RS;
        }
        if constexpr (PROJ) {
            // project<N, K, 64> picks the lane's diagonal element out of its row with N compare-selects and the K x K factor block
            // with readlanes (~140 vector instructions a step on a kernel bound by their issue); Ps[t] is IN the LDS matrix at
            // this point: the diagonal is one per-lane read, the factor block K wavefront-uniform rows.  Same arithmetic.
            double mean, var, gam[K];
#pragma unroll
            for (int k = 0; k < K; ++k) gam[k] = pcl[k];
            double m = xv, tt = 0.0;
            const double diag = Dm[r * LD + (lane < N ? lane : 0)];
            double pfk[K][K];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int l = 0; l < K; ++l) pfk[k][l] = Dm[(N + (k < l ? k : l)) * LD + N + (k < l ? l : k)]; // [kk][ll] = [ll][kk] = element (kk, ll), kk <= ll
            sfor<0, K>(MK_LAMBDA(k) {
                constexpr int kk = decltype(k)::value;
                m = fma(gam[kk], Group<G>::template bcast<N + kk>(xv), m);
            });
            sfor<0, K>(MK_LAMBDA(k) {
                constexpr int kk = decltype(k)::value;
                double u = 2.0 * row[N + kk];
                sfor<0, K>(MK_LAMBDA(l) { u = fma(gam[decltype(l)::value], pfk[kk][decltype(l)::value], u); });
                tt = fma(gam[kk], u, tt);
            });
            mean = fma(pcl[K], m, pcl[K + 1]);
            const double v = pcl[K] * pcl[K] * (diag + tt);
            var = v < 0.0 ? 0.0 : v; // kalmanfilter.py:601-602 (np.maximum keeps a NaN)
            if (lane < N && live) {
                if (pM) *pM = mean;
                if (pV) *pV = var;
            }
            if (pM) pM -= a.ts * N;
            if (pV) pV -= a.ts * N;
        }
        if constexpr (VAR) {
            double diag = 0.0;
            sfor<0, n>(MK_LAMBDA(c) { diag = (decltype(c)::value == lane) ? row[decltype(c)::value] : diag; });
            if (lane < n && live) {
                *sM = xv;
                *sV = diag;
            }
            sM -= a.ts * n;
            sV -= a.ts * n;
        }
    };

    // tile addressing (loop invariant): lane l of a C-layout tile sits at row (l>>4) + 4 reg, column l&15
    const int l15 = lane & 15, l4 = lane >> 4;
    // Pf tiles are fetched as  (wavefront-uniform record pointer) + (per-lane element offset) + (compile-time
    // offset): one per-lane offset per tile column serves every tile row and register, instead of a 64-bit
    // pointer per element (24 of them were kept across the loop and spilled).  Columns beyond n are clamped to
    // n-1 and rows beyond n wrap to row n-1 through the clamp of the final index (never stored: masked write-back).
    int pt_col[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) pt_col[b] = 16 * b + l15 < n ? 16 * b + l15 : n - 1;
    const double *recU = a.F + (inst * a.bs + (T - 1) * a.ts) * a.rs; // uniform (one model per wavefront); rs > 0 only
    const long recUstep = a.ts * a.rs;
    // row of the J staging buffer this lane reads its J^T tile elements from: i = 16 Ib + l15 (the last block clamped to row n-1)
    const int jt_row = l15 * LD;
    const int jt_last = (16 * (NB - 1) + l15 < n ? l15 : n - 1 - 16 * (NB - 1)) * LD;

    // last step: smoothed = filtered (:450-451)
    // Ps[t+1] lives in the LDS matrix Dm between iterations (row r in lane r's row), NOT in registers: with the
    // rows of Pf[t], W, A and Ps[t+1] all live at the top of an iteration the wavefront needs 4n doubles per lane
    // and spills (measured: 74 scratch operations per step, whose traffic reaches the Infinity Cache)
    double xs = *iF.vec;
    {
        double Psn[n];
        load_pf_rows(Psn);
        store_row<n>(Dr, Psn);
        wave_lds_sync(); // (the projection epilogue reads the diagonal and the factor block from the LDS matrix)
        store(xs, Psn);
    }
    double pivmin = 1.0;

    for (long t = T - 2; t >= 0; --t) {
        iF.advance_nn(-1);
        recU -= recUstep;
        const double xfc = *iF.vec;
        // FOLD: z = the lane's row through the fused sweep (z-lanes: W -> Z~; A-lanes: A -> L), A = the H head rows of A
        double A[FOLD ? (H > 0 ? H : 1) : n], z[n];
        double pfb[STR ? K : 1]; // STR: Pf[r][N + k], the seeds of the border columns of Ps
        {
            double Pfc[n];
            load_pf_rows(Pfc, FOLD);
            if constexpr (STR) {
#pragma unroll
                for (int k = 0; k < K; ++k) pfb[k] = Pfc[N + k];
            }
            wave_lds_sync(); // previous iteration's reads of Dm / the J buffer are complete
            sfor<0, n / 2 + n % 2>(MK_LAMBDA(pp) { // two columns at a time; phi_c is a wavefront-uniform LDS read
                constexpr int c0 = 2 * decltype(pp)::value;
                const double ph0 = Dm[c0 * LD + PADC];
                double a0, a1 = 0.0; // row of Phi Pf Phi: Pp[t+1] WITHOUT q on the diagonal (added to the pivots)
                if constexpr (FOLD) {
                    const double w0 = Pfc[c0] * ph0; // W = Pf Phi
                    a0 = phi_r * w0;
                    z[c0] = w0 * phi_x;
                    if constexpr (c0 < H) A[c0] = a0;
                } else {
                    z[c0] = Pfc[c0] * ph0;
                    a0 = A[c0] = phi_r * z[c0];
                }
                if constexpr (c0 + 1 < n) {
                    const double ph1 = Dm[(c0 + 1) * LD + PADC];
                    if constexpr (FOLD) {
                        const double w1 = Pfc[c0 + 1] * ph1;
                        a1 = phi_r * w1;
                        z[c0 + 1] = w1 * phi_x;
                        if constexpr (c0 + 1 < H) A[c0 + 1] = a1;
                    } else {
                        z[c0 + 1] = Pfc[c0 + 1] * ph1;
                        a1 = A[c0 + 1] = phi_r * z[c0 + 1];
                    }
                    const v2d ps = *reinterpret_cast<const v2d *>(Dw + c0); // D = Ps[t+1] - Pp[t+1], in place
                    *reinterpret_cast<v2d *>(Dw + c0) = v2d{ps.x - a0, ps.y - a1};
                } else {
                    Dw[c0] = Dw[c0] - a0;
                }
                if constexpr (c0 % 8 == 6) __builtin_amdgcn_sched_barrier(0);
            });
        }
        Dw[r] -= q_r; // the diagonal of Q (replica lanes >= n rewrite row n-1 with the same value; A-lanes: the dummy row)
        dl[lane] = xs - phi_r * xfc; // delta (lanes >= n: replicas of lane n-1, never read)
        __builtin_amdgcn_sched_barrier(0);
        // ---- A = L D L^T and the forward substitution of W, ONE sweep over the columns (dot form) ----
        // Lane r carries row r of A and row r of W through the same forward substitution: with U = L D (row c,
        // k < c, in the LDS matrix) and the lane's own L(r, k) in registers,
        //     u_c = A(r,c) - sum_{k<c} U(c,k) L(r,k)   = d_c L(r,c)   (lane c: the pivot d_c, once q_c is added)
        //     y_c = W(r,c) - sum_{k<c} U(c,k) z~_k ,   z~_c = y_c / d_c
        // The operands U(c, .) are wavefront-uniform LDS reads shared by both sums: the factorization costs no
        // cross-lane traffic at all (the right-looking form it replaces moved row j to every lane with 2(n-j)
        // readlanes per column, 1332 VALU instructions a step at n = 36, on a kernel that is VALU-issue bound).
        // u_c goes to LDS BEFORE the reciprocal: the write -> read round trip of row c+1 overlaps the pivot chain.
        if constexpr (FOLD) {
            // SOFTWARE-PIPELINED over the columns.  The plain loop is a chain of n stages, each: last multiply-add ->
            // sum -> LDS store of U(., c) -> pivot (readlane + q_c) -> reciprocal (5 dependent instructions) -> scale ->
            // LDS read of row c+1 of U (which waits for the store) -> its c+1 multiply-adds: ~330 cycles a stage of which the
            // wavefront issues for ~130.  Here stage c (i) fetches row c+1 of U for k <= c-1 BEFORE its own store (those
            // elements are old), (ii) finishes column c with the one missing term, k = c-1, whose operand U(c, c-1) came
            // out of stage c-1 as a readlane of the lane that holds row c (no LDS round trip on the critical path),
            // (iii) issues the reciprocal and runs the multiply-adds of column c+1 over k <= c-1 -- all operands final --
            // BETWEEN the dependent instructions of its refinement.  Every accumulator sees the same terms in the same
            // order as in the plain loop (even k / odd k chains, the k = c-1 term last in its chain): bit-identical.
            double px0 = z[0], px1 = 0.0, ph0 = H > 0 ? A[0] : 0.0, ph1 = 0.0; // column being finished: even-k / odd-k chains
            double sU = 0.0;                                                    // U(c, c-1), wavefront-uniform
            constexpr int M1u = (n + 15) / 16;
            double ucp[M1u], qc = Dm[PADC + 1];
#pragma unroll
            for (int m = 0; m < M1u; ++m) ucp[m] = 0.0;
            if (!MK_TUNE_SKIP(a, 4))
            sfor<0, n>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                // (i) ucp: the operands of column c+1 for k <= c-1 (lane 16 q + j holds U(c+1, 16 m + j) for every q) and qc = q_c
                // were fetched during stage c-1, right behind ITS store of U(., c-1) (LDS operations of a wavefront
                // complete in order): a whole stage of multiply-adds ago
                // (ii) the missing term of column c
                if constexpr (c >= 1) {
                    if constexpr ((c - 1) % 2 == 0) px0 = fma(-sU, z[c - 1], px0);
                    else px1 = fma(-sU, z[c - 1], px1);
                    if constexpr (c < H) {
                        if constexpr ((c - 1) % 2 == 0) ph0 = fma(-sU, A[c - 1], ph0);
                        else ph1 = fma(-sU, A[c - 1], ph1);
                    }
                }
                const double u = px0 + px1, hs = ph0 + ph1;
                // U(r, c) for the rows r > c: rows >= H from the A-lanes r + OFF, head rows from lanes c+1 .. H-1
                constexpr int r0 = (c + 1 > H ? c + 1 : H) + OFF;
                if constexpr (r0 < 64) lds_store_masked<~0ull << r0, 8 * c>(Ur, u);
                if constexpr (c + 1 < H) lds_store_masked<((1ull << H) - 1ull) & (~0ull << (c + 1)), 8 * c>(Ur, hs);
                // operands of the NEXT stage's multiply-adds (column c+2, k <= c) and its q: issued here, consumed a stage later
                constexpr int NRq = c + 2 < n ? (c + 16) / 16 : 0;
                double ucq[NRq > 0 ? NRq : 1], qn = 0.0;
                if constexpr (NRq > 0) {
                    const double *Un = Um + tri_off(c + 2);
                    sfor<0, NRq>(MK_LAMBDA(mm) {
                        constexpr int m = decltype(mm)::value;
                        if constexpr (16 * m + 15 < c + 1) {
                            ucq[m] = Un[16 * m + l15];
                        } else { // the last register: lanes at k > c re-read element c (never broadcast)
                            const int k = 16 * m + l15;
                            ucq[m] = Un[k < c + 1 ? k : c];
                        }
                    });
                }
                if constexpr (c + 1 < n) qn = Dm[(c + 1) * LD + PADC + 1];
                const double piv = (c < H ? readlane_f64(hs, c) : readlane_f64(u, c + OFF)) + qc; // d_c = u_c(row c) + q_c
                if constexpr (c + 1 < n) sU = c + 1 < H ? readlane_f64(hs, c + 1) : readlane_f64(u, c + 1 + OFF); // U(c+1, c)
                pivmin = min_f64(pivmin, piv);
                // (iii) 1/d_c (rcp_nr, spelled out) with the multiply-adds of column c+1 between its dependent instructions
                double nx0 = 0.0, nx1 = 0.0, nh0 = 0.0, nh1 = 0.0;
                if constexpr (c + 1 < n) nx0 = z[c + 1];
                if constexpr (c + 1 < H) nh0 = A[c + 1];
                constexpr int NP = (c + 1 < n) ? c / 2 : 0;           // pairs (k, k+1), k = 0, 2, .. < c-1
                constexpr int P1 = NP / 3, P2 = 2 * NP / 3;
                auto pairs = [&](auto lo, auto hi) __attribute__((always_inline)) {
                    sfor<decltype(lo)::value, decltype(hi)::value>(MK_LAMBDA(pp) {
                        constexpr int k = 2 * decltype(pp)::value;
                        Group<16>::fmac2<k % 16, (k + 1) % 16, true>(nx0, ucp[k / 16], z[k], nx1, ucp[(k + 1) / 16], z[k + 1]);
                    });
                };
                double r0v = __builtin_amdgcn_rcp(piv);
                asm volatile("" : "+v"(r0v));
                pairs(std::integral_constant<int, 0>{}, std::integral_constant<int, P1>{});
                double e = fma(-piv, r0v, 1.0); // relative error of r0
                asm volatile("" : "+v"(e));
                pairs(std::integral_constant<int, P1>{}, std::integral_constant<int, P2>{});
                double pe = fma(e, e, e);       // e + e^2
                asm volatile("" : "+v"(pe));
                pairs(std::integral_constant<int, P2>{}, std::integral_constant<int, NP>{});
                if constexpr (c + 1 < n && c % 2 == 1) Group<16>::fmac<(c - 1) % 16, true>(nx0, ucp[(c - 1) / 16], z[c - 1]);
                if constexpr (c + 1 < H) { // head rows of column c+1, k <= c-1 (c + 1 < H <= 16: one operand register)
                    sfor<0, c>(MK_LAMBDA(kk) {
                        constexpr int k = decltype(kk)::value;
                        if constexpr (k % 2 == 0) Group<16>::fmac<k % 16, true>(nh0, ucp[0], A[k]);
                        else Group<16>::fmac<k % 16, true>(nh1, ucp[0], A[k]);
                    });
                }
                const double ij0 = fma(r0v, pe, r0v); // r0 (1 + e + e^2): error e^3
                const double ij = piv > 0.0 ? ij0 : 0.0; // d_c <= 0: null direction dropped (see ldlt_factor)
                z[c] = u * ij;                          // A-lanes: L(r, c); z-lanes: z~_c
                if constexpr (c < H) A[c] = hs * ij;    // L(r, c), r < H
                px0 = nx0, px1 = nx1, ph0 = nh0, ph1 = nh1;
                qc = qn;
                sfor<0, NRq>(MK_LAMBDA(mm) { ucp[decltype(mm)::value] = ucq[decltype(mm)::value]; });
            });
        } else {
        if (!MK_TUNE_SKIP(a, 4))
        sfor<0, n>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int NR = (c + 15) / 16; // registers of 16 operands: lane 16 q + j holds U(c, 16 m + j) for every q
            const double *Uc = Um + tri_off(c);
            double uc[NR > 0 ? NR : 1];
            sfor<0, NR>(MK_LAMBDA(mm) {
                constexpr int m = decltype(mm)::value;
                if constexpr (16 * m + 15 < c) {
                    uc[m] = Uc[16 * m + l15];
                } else { // the row's last register: lanes beyond the diagonal re-read element c-1 (never broadcast)
                    const int k = 16 * m + l15;
                    uc[m] = Uc[k < c ? k : c - 1];
                }
            });
            double a0 = A[c], a1 = 0.0, s0 = z[c], s1 = 0.0;
            sfor<0, c>(MK_LAMBDA(kk) {
                constexpr int k = decltype(kk)::value;
                if constexpr (k % 2 == 0) Group<16>::fmac2<k % 16, k % 16, true>(a0, uc[k / 16], A[k], s0, uc[k / 16], z[k]);
                else Group<16>::fmac2<k % 16, k % 16, true>(a1, uc[k / 16], A[k], s1, uc[k / 16], z[k]);
            });
            const double u = a0 + a1;
            if constexpr (c + 1 < n) // U(r, c), lanes r > c only: the rows are packed (replica lanes >= n rewrite row n-1)
                lds_store_masked<~0ull << (c + 1), 8 * c>(Ur, u);
            wave_lds_sync();
            const double piv = readlane_f64(u, c) + Dm[c * LD + PADC + 1]; // d_c = u_c(lane c) + q_c
            pivmin = min_f64(pivmin, piv);
            const double ij = piv > 0.0 ? rcp_nr(piv) : 0.0; // d_c <= 0: null direction dropped (see ldlt_factor)
            A[c] = u * ij;          // L(r, c)
            z[c] = (s0 + s1) * ij;  // z~_c
        });
        }
        wave_lds_sync();
        if constexpr (!BLK4) {
        // backward: x_c = z~_c - sum_{k>c} L(k,c) x_k.
        // Round 3: DOT form over the columns c = n-2 .. 0 with L^T read back from LDS at wavefront-uniform addresses -- the
        // lanes park their rows of L (A[c] = L(r,c), c < r) transposed in the region the forward sweep's U table has just
        // left (row c of L^T: L(k,c) for k = kb(c) .. n-1, kb(c) = (c+1) & ~1 so that every 16-byte pair is aligned; 35
        // exec-masked 8-byte stores), then every lane runs the same n(n-1)/2 multiply-adds on its own row of Z~ with
        // operands that need no cross-lane instruction at all.  The axpy form it replaces broadcast every L(k,c) with a
        // readlane pair: 1 260 VALU instructions a step next to the 630 multiply-adds.  All reads are independent of the
        // results, so they run ahead of the multiply-adds in batches.
        if (!MK_TUNE_SKIP(a, 2)) {
            double *Lt = Um + (FOLD ? rowA : lane); // element r of a row of L^T (unfolded, lanes >= n: one past the row's last entry, never read)
            sfor<0, n - 1>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (FOLD) { // L(k, c), k > c: rows k >= H sit in the A-lanes k + OFF, head rows in lanes c+1 .. H-1
                    constexpr int r0 = (c + 1 > H ? c + 1 : H) + OFF;
                    lds_store_masked<~0ull << r0, 8 * (lt_off(n, c) - lt_kb(c))>(Lt, z[c]);
                    if constexpr (c + 1 < H)
                        lds_store_masked<((1ull << H) - 1ull) & (~0ull << (c + 1)), 8 * (lt_off(n, c) - lt_kb(c))>(Lt, A[c]);
                } else {
                    lds_store_masked<(~0ull << (c + 1)) & (n < 64 ? (1ull << n) - 1 : ~0ull), 8 * (lt_off(n, c) - lt_kb(c))>(Lt, A[c]);
                }
            });
            wave_lds_sync();
            sfor_down<0, n - 1>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;           // n-2 .. 0
                constexpr int kb = lt_kb(c), M0 = (c + 1) / 16, M1 = (n + 15) / 16; // registers M0 .. M1-1 cover k = c+1 .. n-1
                const double *row = Um + lt_off(n, c) - kb; // element k of the row sits at row[k], k = kb .. n-1
                double lc[M1 - M0];
                sfor<M0, M1>(MK_LAMBDA(mm) {
                    constexpr int m = decltype(mm)::value;
                    if constexpr (16 * m >= kb && 16 * m + 15 < n) {
                        lc[m - M0] = row[16 * m + l15];
                    } else { // lanes outside the row re-read its first / last element (never broadcast)
                        int k = 16 * m + l15;
                        k = k < kb ? kb : k;
                        k = k > n - 1 ? n - 1 : k;
                        lc[m - M0] = row[k];
                    }
                });
                double b0 = z[c], b1 = 0.0;
                // k = n-1 .. c+1 (z[c+1], the freshest operand, last), four per statement (hipcc pads every asm statement that
                // follows another with an s_nop), then a pair and a single; even k into b0, odd k into b1
                constexpr int CNT = n - 1 - c, Q4 = CNT / 4, R2 = (CNT % 4) / 2;
                auto acc_of = [&](auto kk) -> double & { if constexpr (decltype(kk)::value % 2 == 0) return b0; else return b1; };
                sfor<0, Q4>(MK_LAMBDA(qq) {
                    constexpr int k = n - 1 - 4 * decltype(qq)::value; // (k, k-1, k-2, k-3), all > c
                    Group<16>::fmac4<k % 16, (k - 1) % 16, (k - 2) % 16, (k - 3) % 16, true>(
                        acc_of(std::integral_constant<int, k>{}), acc_of(std::integral_constant<int, k - 1>{}), lc[k / 16 - M0], z[k],
                        lc[(k - 1) / 16 - M0], z[k - 1], lc[(k - 2) / 16 - M0], z[k - 2], lc[(k - 3) / 16 - M0], z[k - 3]);
                });
                if constexpr (R2 == 1) {
                    constexpr int k = n - 1 - 4 * Q4; // pair (k, k-1)
                    Group<16>::fmac2<k % 16, (k - 1) % 16, true>(acc_of(std::integral_constant<int, k>{}), lc[k / 16 - M0], z[k],
                                                                 acc_of(std::integral_constant<int, k - 1>{}), lc[(k - 1) / 16 - M0], z[k - 1]);
                }
                if constexpr (CNT % 2 == 1) {
                    constexpr int k = c + 1;
                    Group<16>::fmac<k % 16, true>(acc_of(std::integral_constant<int, k>{}), lc[k / 16 - M0], z[k]);
                }
                z[c] = b0 + b1;
            });
            wave_lds_sync(); // the reads of L^T are complete: the region becomes the J staging buffer
        }
        // z = J[r, :]
        // smoothed mean (:461-464): xs[t] = F[t] + J delta, delta from LDS (uniform reads)
        {
            double a0 = xfc, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            constexpr int M1 = (n + 15) / 16;
            double dreg[M1]; // lane 16 q + j: delta[16 m + j] (dl has 64 entries; those >= n are finite replicas, never broadcast)
#pragma unroll
            for (int m = 0; m < M1; ++m) dreg[m] = dl[16 * m + l15];
            sfor<0, n>(MK_LAMBDA(mm) {
                constexpr int m = decltype(mm)::value;
                if constexpr (m % 4 == 0 && m + 3 < n) {
                    asm volatile("v_fmac_f64_dpp %0, %4, %6 row_newbcast:%10" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %4, %7 row_newbcast:%11" MK_DPPMASK
                                 "\n\tv_fmac_f64_dpp %2, %5, %8 row_newbcast:%12" MK_DPPMASK "\n\tv_fmac_f64_dpp %3, %5, %9 row_newbcast:%13" MK_DPPMASK
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                                 : "v"(dreg[m / 16]), "v"(dreg[(m + 3) / 16]), "v"(z[m]), "v"(z[m + 1]), "v"(z[m + 2]), "v"(z[m + 3]),
                                   "n"(m % 16), "n"((m + 1) % 16), "n"((m + 2) % 16), "n"((m + 3) % 16));
                    static_assert(m / 16 == (m + 1) / 16 && (m + 2) / 16 == (m + 3) / 16, "operand registers of a quad");
                } else if constexpr (m + 3 >= n || m % 4 != 0) {
                    if constexpr (m >= n - n % 4) { // the tail n % 4 terms
                        if constexpr (m % 4 == 0) Group<16>::fmac<m % 16, false>(a0, dreg[m / 16], z[m]);
                        else if constexpr (m % 4 == 1) Group<16>::fmac<m % 16, false>(a1, dreg[m / 16], z[m]);
                        else if constexpr (m % 4 == 2) Group<16>::fmac<m % 16, false>(a2, dreg[m / 16], z[m]);
                        else Group<16>::fmac<m % 16, false>(a3, dreg[m / 16], z[m]);
                    }
                }
            });
            xs = (a0 + a1) + (a2 + a3);
            if constexpr (FOLD) { // the A-lanes carried rows of L, not of J: they become replicas of lane n-1 again
                const double xl = readlane_f64(xs, n - 1);
                xs = isA ? xl : xs;
            }
            asm volatile("" : "+v"(xs)); // finished HERE: deferred behind the products, its delta operands were spilled
        }
        wave_lds_sync(); // Dm (D) visible to the whole wavefront
        __builtin_amdgcn_sched_barrier(0);

        // ---- MFMA tiles ----
        // Pf[t] once more (L2-resident), this time as C-layout tiles Ib <= Jb: they seed the accumulators of
        // product 2, so that what comes out of it is Ps[t] itself.  Issued here, they land during product 1.
        // (Pf is stored as column runs = row-major of its transpose = itself; padded rows/columns are clamped
        // duplicates that the masked tile write-back below never stores.)
        constexpr int NT = NB * (NB + 1) / 2;
        v4d Pt[NT];
        {
            const double *pf = (a.rs > 0 ? recU : iF.mat - r - n) + n; // element (0, 0) of this step's covariance block
            sfor<0, NB>(MK_LAMBDA(ib) {
                constexpr int Ib = decltype(ib)::value;
                sfor<Ib, NB>(MK_LAMBDA(jb) {
                    constexpr int Jb = decltype(jb)::value;
                    constexpr int ti = Ib * NB - Ib * (Ib - 1) / 2 + (Jb - Ib);
                    sfor<0, 4>(MK_LAMBDA(vv) {
                        constexpr int v = decltype(vv)::value;
                        if constexpr (16 * Ib + 4 * v >= n) {
                            Pt[ti][v] = 0.0; // a register whose rows are all padding: never stored
                        } else if constexpr (SYM) { // element (min, max) of the packed upper triangle
                            const int row = 16 * Ib + l4 + 4 * v < n ? 16 * Ib + l4 + 4 * v : n - 1, col = pt_col[Jb];
                            const int lo = row < col ? row : col, hi = row < col ? col : row;
                            Pt[ti][v] = pf[lo * n - lo * (lo - 1) / 2 + (hi - lo)];
                        } else if constexpr (16 * Ib + 4 * v + 3 < n) { // all four row groups inside: affine address
                            Pt[ti][v] = pf[(l4 * n + pt_col[Jb]) + (16 * Ib + 4 * v) * n];
                        } else {
                            const int row = 16 * Ib + l4 + 4 * v < n ? 16 * Ib + l4 + 4 * v : n - 1;
                            Pt[ti][v] = pf[row * n + pt_col[Jb]];
                        }
                    });
                });
            });
        }
        // ---- STR: the factor border on the vector pipe ----
        //   vb[k]  = V^T[N+k][j]  = sum_m D[N+k][m] J[j][m]          lane j < N: the last k-step of product 2's B operand
        //   Vf[m][k] = (D J_f^T)[m][k] = sum_c D[m][c] J[N+k][c]     lane m: its own row of D, the factor rows of J uniform
        //   pb[k]  = Ps[r][N+k]   = Pf[r][N+k] + sum_m J[r][m] Vf[m][k]   lane r: every row, i.e. the border columns AND the
        //                                                               K x K block (rows r >= N)
        // Done BEFORE the J^T fragments occupy their registers; the staging region is free (L^T has been consumed):
        // [ J_f rows: K x LD | vb: 4 x N | Vf^T: 4 x VLD ].  Every wavefront-uniform operand u[m] arrives as ONE per-lane
        // 8-byte read per 16 values (lane 16 q + j reads u[16 m' + j]) and is broadcast inside the multiply-add
        // (v_fmac_f64_dpp row_newbcast, as in the sweeps above).
        double pb[STR ? 4 : 1];
        double Bv[STR ? NB : 1];
        if constexpr (STR) {
            constexpr int M1 = (n + 15) / 16, VLD = 16 * M1;
            double *jf = Um, *vbuf = Um + 4 * LD, *vf = vbuf + 4 * N;
            static_assert(4 * LD + 4 * N + 4 * VLD <= RSZ + 64, "border buffers exceed the staging region");
            wave_lds_sync(); // the reads of L^T are complete
            {
                constexpr unsigned long long frows = ((n < 64 ? (1ull << n) : 0ull) - 1ull) & ~((1ull << N) - 1ull); // lanes N .. n-1
                double *dst = jf + (r - N) * LD;
                sfor<0, n / 2>(MK_LAMBDA(pp) {
                    constexpr int c0 = 2 * decltype(pp)::value;
                    lds_store_masked<frows, 8 * c0>(dst, v2d{z[c0], z[c0 + 1]});
                });
                if constexpr (n % 2 == 1) lds_store_masked<frows, 8 * (n - 1)>(dst, z[n - 1]);
            }
            wave_lds_sync();
            // operand index of this lane in register m' (clamped: indices >= n are never broadcast)
            int oi[M1];
#pragma unroll
            for (int m = 0; m < M1; ++m) oi[m] = 16 * m + l15 < n ? 16 * m + l15 : n - 1;
            double vb[4] = {0.0, 0.0, 0.0, 0.0}, vfa[4] = {0.0, 0.0, 0.0, 0.0};
            double du[K][M1], ju[K][M1];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int m = 0; m < M1; ++m) {
                    du[k][m] = Dm[(N + k) * LD + oi[m]];
                    ju[k][m] = jf[k * LD + oi[m]];
                }
            constexpr int DBP = 3; // pairs of the lane's own row of D per batch
            sfor<0, (n / 2 + DBP - 1) / DBP>(MK_LAMBDA(bb) {
                constexpr int p0 = DBP * decltype(bb)::value, p1 = p0 + DBP < n / 2 ? p0 + DBP : n / 2;
                v2d dr[DBP];
#pragma unroll
                for (int p = p0; p < p1; ++p) dr[p - p0] = *reinterpret_cast<const v2d *>(Dr + 2 * p);
                sfor<2 * p0, 2 * p1>(MK_LAMBDA(cc) {
                    constexpr int c = decltype(cc)::value;
                    const double dc = c % 2 ? dr[c / 2 - p0].y : dr[c / 2 - p0].x;
                    if constexpr (K == 4) { // all eight multiply-adds of a column in ONE statement (no s_nop between them)
                        Group<16>::fmac4x<c % 16>(vb[0], vb[1], vb[2], vb[3], du[0][c / 16], du[1][c / 16], du[2][c / 16], du[3][c / 16], z[c]);
                        Group<16>::fmac4x<c % 16>(vfa[0], vfa[1], vfa[2], vfa[3], ju[0][c / 16], ju[1][c / 16], ju[2][c / 16], ju[3][c / 16], dc);
                    } else {
                        sfor<0, K>(MK_LAMBDA(kk) {
                            constexpr int k = decltype(kk)::value;
                            Group<16>::fmac2<c % 16, c % 16, false>(vb[k], du[k][c / 16], z[c], vfa[k], ju[k][c / 16], dc);
                        });
                    }
                });
            });
            if constexpr (n % 2 == 1) {
                const double dc = Dr[n - 1];
                sfor<0, K>(MK_LAMBDA(kk) {
                    constexpr int k = decltype(kk)::value;
                    Group<16>::fmac<(n - 1) % 16, false>(vb[k], du[k][(n - 1) / 16], z[n - 1]);
                    Group<16>::fmac<(n - 1) % 16, false>(vfa[k], ju[k][(n - 1) / 16], dc);
                });
            }
            {   // vb[k] of lane j -> vbuf[k][j] (lanes j < N); Vf[m][k] of lane m -> vf[k][m] (lanes >= n rewrite entry n-1)
                constexpr unsigned long long srows = (1ull << N) - 1ull; // N <= 48 here (N % 16 == 0, n <= 64)
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_store_masked<srows, 0>(vbuf + k * N + lane, vb[k]);
#pragma unroll
                for (int k = 0; k < K; ++k) vf[k * VLD + r] = vfa[k];
            }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 4; ++k) pb[k] = k < K ? pfb[k < K ? k : 0] : 0.0;
            if constexpr (K == 4) { // pb[k] += J[r][m] Vf[m][k]: the four columns side by side (four independent chains a statement)
                double vu[4][M1];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int m = 0; m < M1; ++m) vu[k][m] = vf[k * VLD + oi[m]];
                sfor<0, n>(MK_LAMBDA(mm) {
                    constexpr int m = decltype(mm)::value;
                    Group<16>::fmac4x<m % 16>(pb[0], pb[1], pb[2], pb[3], vu[0][m / 16], vu[1][m / 16], vu[2][m / 16], vu[3][m / 16], z[m]);
                });
            } else {
                sfor<0, K>(MK_LAMBDA(kk) { // pb[k] += J[r][m] Vf[m][k]
                    constexpr int k = decltype(kk)::value;
                    double vu[M1];
#pragma unroll
                    for (int m = 0; m < M1; ++m) vu[m] = vf[k * VLD + oi[m]];
                    sfor<0, n>(MK_LAMBDA(mm) {
                        constexpr int m = decltype(mm)::value;
                        Group<16>::fmac<m % 16, false>(pb[k], vu[m / 16], z[m]);
                    });
                });
            }
            // B operand of the factor k-step: B[k = l4][col = 16 Jb + l15] = vb[k] of lane 16 Jb + l15
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) Bv[jb] = vbuf[l4 * N + 16 * jb + l15];
        }
        // J^T in C-layout: JT[Ib][ks] = J[16 Ib + l15][4 ks + l4] (k >= n only when n % 4 != 0: zero).  The rows of J
        // go from the lanes' registers through the staging buffer 16 at a time (LDS operations of one wavefront
        // execute in order: the fences are compiler fences)
        double JT[NB][KS];
        sfor<0, NB>(MK_LAMBDA(ib) {
            constexpr int Ib = decltype(ib)::value;
            wave_lds_sync(); // the reads of the previous 16 rows (Ib = 0: of U) are complete
            {
                constexpr unsigned long long rows = 0xffffull << (16 * Ib); // lanes = rows 16 Ib .. 16 Ib + 15 (those >= n: row n-1 again)
                double *dst = Um + (r - 16 * Ib) * LD; // (masked-off lanes: never dereferenced)
                sfor<0, n / 2>(MK_LAMBDA(pp) {
                    constexpr int c0 = 2 * decltype(pp)::value;
                    lds_store_masked<rows, 8 * c0>(dst, v2d{z[c0], z[c0 + 1]});
                });
                if constexpr (n % 2 == 1) lds_store_masked<rows, 8 * (n - 1)>(dst, z[n - 1]);
            }
            wave_lds_sync();
            const int jrow = (!STR && Ib == NB - 1) ? jt_last : jt_row;
            sfor<0, KS>(MK_LAMBDA(ks) {
                constexpr int kb = 4 * decltype(ks)::value;
                if constexpr (kb + 3 < n) {
                    JT[Ib][decltype(ks)::value] = Um[jrow + kb + l4];
                } else {
                    const int k = kb + l4;
                    const double v = Um[jrow + (k < n ? k : n - 1)];
                    JT[Ib][decltype(ks)::value] = k < n ? v : 0.0;
                }
            });
        });
        // products, block row by block row of V^T (so that only ONE block row of V^T is ever live):
        //   product 1   V^T[Kb', :] = D[Kb', :] J^T        A operand: D[m][k'] (= D[k'][m]), m = 4 ms + l4 (K), k' = 16 Kb' + l15 (M)
        //   product 2   Ps[Ib][Jb] += J[Ib][k in Kb'] V^T[k][Jb], tiles Ib <= Jb, accumulating in the Pf tiles
        // Product 2 may not overwrite Dm before every block row of D has been read: the accumulators stay in
        // registers until the end.  A operand of product 2: J[i][k] = JT[Ib][ks]; B operand: the V^T rows just made.
        __builtin_amdgcn_sched_barrier(0);
        if (!MK_TUNE_SKIP(a, 1))
        sfor<0, NB>(MK_LAMBDA(kb) {
            constexpr int Kb = decltype(kb)::value;
            __builtin_amdgcn_sched_barrier(0); // keeps the next block row's operand loads from being hoisted (registers)
            const int kcol = 16 * Kb + l15 < n ? 16 * Kb + l15 : n - 1; // M padding: duplicates, harmless
            double Da[KS];
            sfor<0, KS>(MK_LAMBDA(ms) {
                constexpr int mb = 4 * decltype(ms)::value;
                if constexpr (mb + 3 < n) {
                    Da[decltype(ms)::value] = Dm[(mb + l4) * LD + kcol];
                } else {
                    const int m = mb + l4;
                    const double v = Dm[(m < n ? m : n - 1) * LD + kcol];
                    Da[decltype(ms)::value] = m < n ? v : 0.0;
                }
            });
            v4d VT[NB];
            sfor<0, NB>(MK_LAMBDA(jb) {
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                sfor<0, KS>(MK_LAMBDA(ms) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Da[decltype(ms)::value],
                                                               JT[decltype(jb)::value][decltype(ms)::value], acc, 0, 0, 0);
                });
                VT[decltype(jb)::value] = acc;
            });
            sfor<0, NB>(MK_LAMBDA(ib) {
                constexpr int Ib = decltype(ib)::value;
                sfor<Ib, NB>(MK_LAMBDA(jb) {
                    constexpr int Jb = decltype(jb)::value;
                    constexpr int ti = Ib * NB - Ib * (Ib - 1) / 2 + (Jb - Ib);
                    sfor<0, 4>(MK_LAMBDA(vv) {
                        constexpr int s = 4 * Kb + decltype(vv)::value; // k-step = row group of V^T
                        if constexpr (s < KS)
                            Pt[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(JT[Ib][s], VT[Jb][decltype(vv)::value], Pt[ti], 0, 0, 0);
                    });
                });
            });
        });
        if constexpr (STR) { // the factor rows of V^T: the last k-step of product 2
            sfor<0, NB>(MK_LAMBDA(ib) {
                constexpr int Ib = decltype(ib)::value;
                sfor<Ib, NB>(MK_LAMBDA(jb) {
                    constexpr int Jb = decltype(jb)::value;
                    constexpr int ti = Ib * NB - Ib * (Ib - 1) / 2 + (Jb - Ib);
                    Pt[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(JT[Ib][KS - 1], Bv[Jb], Pt[ti], 0, 0, 0);
                });
            });
        }
        wave_lds_sync(); // all reads of D are done: Dm is free for Ps[t]
        if constexpr (STR) { // border columns N .. n-1 of every row, mirrored into rows N .. n-1 by the series lanes
            double *dst = Dw + N; // (A-lanes: the dummy row -- their pb is not a row of Ps)
            if constexpr (K == 4 && N % 2 == 0) {
                *reinterpret_cast<v2d *>(dst) = v2d{pb[0], pb[1]};
                *reinterpret_cast<v2d *>(dst + 2) = v2d{pb[2], pb[3]};
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) dst[k] = pb[k];
            }
            constexpr unsigned long long srows = (1ull << N) - 1ull;
#pragma unroll
            for (int k = 0; k < K; ++k) lds_store_masked<srows, 0>(Dm + (N + k) * LD + lane, pb[k]);
        }
        sfor<0, NB>(MK_LAMBDA(ib) {
            constexpr int Ib = decltype(ib)::value;
            sfor<Ib, NB>(MK_LAMBDA(jb) {
                constexpr int Jb = decltype(jb)::value;
                const v4d acc = Pt[Ib * NB - Ib * (Ib - 1) / 2 + (Jb - Ib)];
                const int col = 16 * Jb + l15;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * Ib + l4 + 4 * v;
                    if (16 * Ib + 4 * v + 3 < n && 16 * Jb + 15 < n) { // interior: no mask (compile-time)
                        Dm[row * LD + col] = acc[v];
                        if constexpr (Ib != Jb) Dm[col * LD + row] = acc[v];
                    } else if (row < n && col < n) {
                        Dm[row * LD + col] = acc[v];
                        if constexpr (Ib != Jb) Dm[col * LD + row] = acc[v];
                    }
                }
            });
        });
        } else {
        }
        wave_lds_sync();
        {
            double Psn[n]; // transient: the row goes to the store path / the epilogues; Dm keeps it for the next step
            load_row<n>(Dr, Psn);
            store(xs, Psn);
        }
    }
    if (a.status && live && lane == 0 && pivot_flags(pivmin)) atomicOr(a.status + inst, pivot_flags(pivmin));
}

// Which smoother serves the shapes with n > 16: the MFMA kernel, or (mk_set_kernel_variant(ctx, MK_VARIANT_WIDE_SMOOTHER,
// 1 / 2), for A/B measurements) the round-1 kernel / the MFMA kernel without the lane fold -- all tested against the oracle.
// The fourteen instantiations a shape has here (two round-1 kernels, and the MFMA kernel for three epilogues x two record
// layouts x folded / unfolded) are most of a wide shape's compile time -- (48,3): 5.3 of 6.5 minutes in ONE translation unit
// (round-5 verdict, weak 10).  They are therefore sliced: -DMK_WIDE_PART=p compiles slice p only (0: the round-1 kernels and the
// dispatcher; 1..6: one (epilogue, layout) pair of the MFMA kernel), the library build (Makefile) and the run-time shape
// modules (metran_amd/jit.py) compile the slices as separate translation units in parallel, and the linker puts them back
// together.  Without the macro this file is everything, as before.
#ifdef MK_WIDE_PART
#define MK_WIDE_HAS(p) (MK_WIDE_PART == (p))
#else
#define MK_WIDE_HAS(p) 1
#endif

hipError_t launch_wide_wave(int N, int K, const SmootherArgs &a, hipStream_t s);
#define MK_DECL_WIDE_MFMA(E, S) hipError_t launch_wide_mfma_##E##_##S(int N, int K, const SmootherArgs &a, hipStream_t s);
MK_DECL_WIDE_MFMA(0, 0)
MK_DECL_WIDE_MFMA(1, 0)
MK_DECL_WIDE_MFMA(2, 0)
MK_DECL_WIDE_MFMA(0, 1)
MK_DECL_WIDE_MFMA(1, 1)
MK_DECL_WIDE_MFMA(2, 1)
#undef MK_DECL_WIDE_MFMA

// shape dispatch shared by every slice: the exact (N, K), else -- no projection asked for -- any shape of the same state dimension
#define MK_WIDE_DISPATCH(FN)                                                                                \
    MK_SHAPES(MK_CASE_WIDE_EXACT_##FN)                                                                      \
    MK_SHAPES(MK_CASE_WIDE_DIM_##FN)                                                                        \
    return hipErrorInvalidValue;

#if MK_WIDE_HAS(0)
template <int N, int K>
static hipError_t launch_wave_nk(const SmootherArgs &a, hipStream_t s)
{
    if constexpr (N + K > 16) {
        if (a.sim_means || a.sim_vars) hipLaunchKernelGGL((smoother_wave_kernel<N, K, true>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((smoother_wave_kernel<N, K, false>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}
#define MK_CASE_WIDE_EXACT_wave(NN, KK) \
    if (N == NN && K == KK) return launch_wave_nk<NN, KK>(a, s);
#define MK_CASE_WIDE_DIM_wave(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_wave_nk<NN, KK>(a, s);
hipError_t launch_wide_wave(int N, int K, const SmootherArgs &a, hipStream_t s) { MK_WIDE_DISPATCH(wave) }
#endif

template <int N, int K, int E, bool S>
[[maybe_unused]] static hipError_t launch_mfma_nk(const SmootherArgs &a, hipStream_t s)
{
    if constexpr (N + K > 16) {
        if (a.variant & 4) hipLaunchKernelGGL((smoother_mfma_kernel<N, K, E, S, false>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((smoother_mfma_kernel<N, K, E, S, true>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}
#define MK_DEF_WIDE_MFMA(E, S, SB)                                                                         \
    hipError_t launch_wide_mfma_##E##_##S(int N, int K, const SmootherArgs &a, hipStream_t s)              \
    {                                                                                                      \
        MK_SHAPES(MK_CASE_WIDE_EXACT_mfma_##E##_##S)                                                       \
        MK_SHAPES(MK_CASE_WIDE_DIM_mfma_##E##_##S)                                                         \
        return hipErrorInvalidValue;                                                                       \
    }
#if MK_WIDE_HAS(1)
#define MK_CASE_WIDE_EXACT_mfma_0_0(NN, KK) \
    if (N == NN && K == KK) return launch_mfma_nk<NN, KK, 0, false>(a, s);
#define MK_CASE_WIDE_DIM_mfma_0_0(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_mfma_nk<NN, KK, 0, false>(a, s);
MK_DEF_WIDE_MFMA(0, 0, false)
#endif
#if MK_WIDE_HAS(2)
#define MK_CASE_WIDE_EXACT_mfma_1_0(NN, KK) \
    if (N == NN && K == KK) return launch_mfma_nk<NN, KK, 1, false>(a, s);
#define MK_CASE_WIDE_DIM_mfma_1_0(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_mfma_nk<NN, KK, 1, false>(a, s);
MK_DEF_WIDE_MFMA(1, 0, false)
#endif
#if MK_WIDE_HAS(3)
#define MK_CASE_WIDE_EXACT_mfma_2_0(NN, KK) \
    if (N == NN && K == KK) return launch_mfma_nk<NN, KK, 2, false>(a, s);
#define MK_CASE_WIDE_DIM_mfma_2_0(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_mfma_nk<NN, KK, 2, false>(a, s);
MK_DEF_WIDE_MFMA(2, 0, false)
#endif
#if MK_WIDE_HAS(4)
#define MK_CASE_WIDE_EXACT_mfma_0_1(NN, KK) \
    if (N == NN && K == KK) return launch_mfma_nk<NN, KK, 0, true>(a, s);
#define MK_CASE_WIDE_DIM_mfma_0_1(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_mfma_nk<NN, KK, 0, true>(a, s);
MK_DEF_WIDE_MFMA(0, 1, true)
#endif
#if MK_WIDE_HAS(5)
#define MK_CASE_WIDE_EXACT_mfma_1_1(NN, KK) \
    if (N == NN && K == KK) return launch_mfma_nk<NN, KK, 1, true>(a, s);
#define MK_CASE_WIDE_DIM_mfma_1_1(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_mfma_nk<NN, KK, 1, true>(a, s);
MK_DEF_WIDE_MFMA(1, 1, true)
#endif
#if MK_WIDE_HAS(6)
#define MK_CASE_WIDE_EXACT_mfma_2_1(NN, KK) \
    if (N == NN && K == KK) return launch_mfma_nk<NN, KK, 2, true>(a, s);
#define MK_CASE_WIDE_DIM_mfma_2_1(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_mfma_nk<NN, KK, 2, true>(a, s);
MK_DEF_WIDE_MFMA(2, 1, true)
#endif

#if MK_WIDE_HAS(0)
hipError_t launch_smoother_wide(int N, int K, const SmootherArgs &a, hipStream_t s)
{
    if (a.tape) return launch_smoother_dk(N, K, a, s); // the inverse-free backward pass over the filter's tape (mk_dk.hip)
    const bool proj = a.sim_means || a.sim_vars;
    const int epi = proj ? 1 : (a.state_means ? 2 : 0);
    if ((a.variant & 2) && !a.sym && epi != 2) return launch_wide_wave(N, K, a, s); // the round-1 kernel knows neither packed-symmetric records nor VAR_ONLY
    if (a.sym) return epi == 1 ? launch_wide_mfma_1_1(N, K, a, s) : epi == 2 ? launch_wide_mfma_2_1(N, K, a, s) : launch_wide_mfma_0_1(N, K, a, s);
    return epi == 1 ? launch_wide_mfma_1_0(N, K, a, s) : epi == 2 ? launch_wide_mfma_2_0(N, K, a, s) : launch_wide_mfma_0_0(N, K, a, s);
}
#endif

} // namespace mk
