// tower_x3.h — kernel C of the pipelined DeepFM step with the Dense tower on SPLIT-bf16 matrix cores ("bf16 x 3").
// Included by deepfm.hip inside namespace dt, after MlpParams / Part3 / DcnArgs / the k_mlp_fwd3 helpers.
//
// Why: gfx950 has no reduced-precision fp32 MFMA (no xf32); v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, and the
// four GEMMs of the tile kernel (Dense128, Dense64, dH1 = dH2 W2^T, dXn = dH1 W1^T: 33 K cycles of fp32-MFMA time per
// 32-row tile, deepnets.py:401-427) were half of k_mlp_fwd3's 70 K cycles.  Here every fp32 operand is split into bf16
// parts and the products run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (each bf16 x bf16 product is exact in fp32):
//   FORWARD (Dense128, Dense64): a = a1 + a2 + a3 EXACTLY (8 + 8 + 8 mantissa bits), b likewise, and
//       a b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)      six MFMAs, the dropped terms are 2^-24 of the product:
//     fp32-class pre-activations, so logits match the oracle like the exact kernel's and a relu input lands on the other
//     side of its kink no more often than with fp32 arithmetic.  (The two-part form — 2^-17 per product — was measured
//     first: logits 2e-5, but ~10 relu units per batch of 8192 x 192 had |input| below the rounding and took the other
//     derivative: 2e-3 of the weight gradients' largest entry, 0.1 of a row gradient's.)
//   BACKWARD (dH1, dXn): a = a_hi + a_lo (16 bits), a b ~ a_hi b_hi + (a_hi b_lo + a_lo b_hi): three MFMAs, 2^-17 relative
//     per product, gradients within ~6e-6 of the tensor max of the float64 oracle (no kinks: relu' is decided by the forward).
// Small terms are accumulated separately and added last.  6/16 resp. 3/16 of the fp32-MFMA time.
//
// Shape: 512 threads = 8 waves (two per SIMD: one wave's LDS / L2 latencies hide behind the other's MFMAs), a 32-row
// tile per block as in k_mlp_fwd3.  Every wave owns 16-column output tiles and BOTH 16-row halves of the tile, so a B
// operand (weights, from L2) is fetched by exactly one wave:
//   GEMM1  H1 = relu(Xn W1 + b1)    wave w: hidden units [16w, 16w+16); K = CP in steps of 32; A = Xn tile, three parts in LDS
//   GEMM2  H2 = relu(H1 W2 + b2)    wave w: row half w & 1, columns [16 (w >> 1), +16); K = 128; A split on the fly from fp32
//   dH1    = relu'(H1) (dH2 W2^T)   wave w: hidden units [16w, 16w+16); K = 64; B = rows of W2 (k contiguous)
//   dXn    = dH1 W1^T               wave w: column tiles w, w + 8, ..; K = 128; B = rows of W1 (k contiguous)
// A 16x16x32 operand is 8 consecutive k per lane (lane (i = l % 16, g = l / 16) <-> k = 32 step + 8 g + j): 16-byte LDS /
// global reads throughout; the weight parts come from lane-major (GEMM1, GEMM2) or row-major (dH1, dXn) bf16 copies the
// prep launch writes once per step (X3Weights).  The Xn parts' LDS rows are padded so that the 16 lanes of a ds_read_b128
// group hit 16 distinct 16-byte bank groups (row stride / 16 B = 2 or 10 mod 16).
#pragma once

typedef __bf16 x3_b8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_b4 __attribute__((ext_vector_type(4)));

struct X3Weights {
    // bf16 parts of the weights, part p of an array `lo` elements after part p - 1 (written by k_prep once per step):
    const __bf16* W1B; int64_t w1b_lo;      // 3 parts, lane-major: element j of lane (n, g), step s, wave w = W1[32 s + 8 g + j][16 w + n]
    const __bf16* W1R; int64_t w1r_lo;      // 2 parts, [CP rows][128] row-major copy of W1 (rows >= C zero): dXn's B operand
    const __bf16* W2B; int64_t w2b_lo;      // 3 parts, lane-major: element j of lane (n, g), step s, column tile t = W2[32 s + 8 g + j][16 t + n]
    const __bf16* W2R; int64_t w2r_lo;      // 2 parts, [128][64] row-major copy of W2: dH1's B operand
    const float* cwp;                       // DCN: [2 L + 1][CP] fp32, zero beyond C: cross kernels | cross biases | w3c
};

// a = h + l (16 mantissa bits) / a = h + m + l (all 24: exact)
__device__ __forceinline__ void x3_split2(const float (&v)[8], x3_b8& h, x3_b8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 a = (__bf16)v[e];
        h[e] = a;
        l[e] = (__bf16)(v[e] - (float)a);
    }
}
__device__ __forceinline__ void x3_split3(const float (&v)[8], x3_b8& h, x3_b8& m, x3_b8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 a = (__bf16)v[e];
        const float r1 = v[e] - (float)a;
        const __bf16 b = (__bf16)r1;
        h[e] = a; m[e] = b;
        l[e] = (__bf16)(r1 - (float)b);
    }
}
__device__ __forceinline__ x3_b8 x3_ld8(const __bf16* p) { return *reinterpret_cast<const x3_b8*>(p); }
__device__ __forceinline__ void x3_ld8f(const float* p, float (&v)[8]) {
    const floatx4 a = ld4(p), b = ld4(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
#define X3_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)
// the lower-order products of a split operand pair: left out in plain-bf16 mode (template ONE, DT_STEP_TOWER_BF16)
#define X3_LO(acc, a, b)        \
    do {                        \
        if constexpr (!ONE) X3_MFMA(acc, a, b); \
    } while (0)

// LDS plan (bytes), CP = 64 NCH:
//   xreg  3 * 32 * (CP + 16) * 2      Xn tile as three bf16 parts (GEMM1); afterwards: xhat / dXn fp32 [32][CP + 4], then
//                                      slp [8][CP] (the waves' d w_lin partial sums) | zp [4][32] | dzs [32] | cs [2][2][64]
//   bnp   4 * CP * 4                  mean | gamma rstd | beta | rstd
//   h1f   32 * (128 + 4) * 4          H1 fp32, then dH1
//   d2f   32 * (64 + 4) * 4           dH2 fp32
// DCN: + pb [2][48][16] (the two K halves of P) | crP [48][16] | crA [32][16] | crF [32][16] | zcs [32]
__host__ __device__ constexpr size_t x3_lds_bytes(int CP, bool dcn = false) {
    return (size_t)3 * kTM * (CP + 16) * 2 + (size_t)4 * CP * 4 + (size_t)kTM * (kH1 + 4) * 4 + (size_t)kTM * (kH2 + 4) * 4 +
           (dcn ? (size_t)(2 * 48 * 16 + 48 * 16 + 2 * kTM * 16 + kTM) * 4 : 0);
}
// the fp32 tile + the small arrays must fit behind each other inside xreg
__host__ __device__ constexpr bool x3_fits(int CP) {
    return (size_t)kTM * (CP + 4) * 4 + (size_t)8 * CP * 4 + (size_t)(5 * kTM + 4 * kH2) * 4 <= (size_t)3 * kTM * (CP + 16) * 2 &&
           x3_lds_bytes(CP, true) <= 160 * 1024;
}

// ONE: the plain-bf16 mode of north_star ("1e-2 bf16", DT_STEP_TOWER_BF16): only the leading product of every operand pair
// (the same layouts, the lower parts unused) — results within 1e-2 of the oracle instead of 1e-4; the Cross network's scalars
// (logits up to +-160) keep their six products.
template <int NCH, int LC = 0, bool ONE = false>   // LC = kCrossMax: DCN (the Cross network's closed form of k_mlp_fwd3 on the same tile, see there)
__global__ __launch_bounds__(512) void k_tower_x3(const float* __restrict__ X, MlpParams p, X3Weights xw, DeepFmDims dm,
                                                  const float* __restrict__ lin, const float* __restrict__ fm,
                                                  const float* __restrict__ y, float* __restrict__ H1,
                                                  float* __restrict__ dH1, float* __restrict__ dH2,
                                                  float* __restrict__ z_out, float* __restrict__ logit_out,
                                                  float* __restrict__ dlogit, float* __restrict__ dz_out,
                                                  double* __restrict__ part, unsigned long long* stamps, DcnArgs dc,
                                                  float* __restrict__ dxn_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    DT_STAMP(stamps, 0);
    constexpr int CP = 64 * NCH, NST = CP / 32, XSB = CP + 16, XS = CP + 4, HF = kH1 + 4, DF = kH2 + 4;
    constexpr int XP = kTM * XSB;                                 // elements of one bf16 part of the Xn tile
    char* base = reinterpret_cast<char*>(lds);
    __bf16* xb = reinterpret_cast<__bf16*>(base);                 // [3][32][XSB]
    float* xs = reinterpret_cast<float*>(base);                   // [32][XS]   (after GEMM1)
    float* slp = xs + kTM * XS;                                   // [8][CP]
    float* zp = slp + 8 * CP;                                     // [4][32]
    float* dzs = zp + 4 * kTM;                                    // [32]
    float* cs = dzs + kTM;                                        // [2][2][64]
    float* bnp = reinterpret_cast<float*>(base + (size_t)3 * XP * 2);   // [4][CP]
    float* h1f = bnp + 4 * CP;                                    // [32][HF]
    float* d2f = h1f + kTM * HF;                                  // [32][DF]
    float* pb = d2f + kTM * DF;                                   // DCN: [2][48][16] K halves of P
    float* crP = pb + 2 * 48 * 16;                                // DCN: [48][16] P[r][l] = Xn[r] . Wc_l (rows 0..31), b_j . Wc_l (rows 32 + j); Wc_L = w3c
    float* crA = crP + 48 * 16;                                   // DCN: [32][16] a_l of every row, later A_{l+1} and dz (crS)
    float* crF = crA + kTM * 16;                                  // DCN: [32][16] coeff[r][l] = d loss / d P[r][l]
    float* zcs = crF + kTM * 16;                                  // DCN: [32] w3c . cross(Xn) per row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.x * kTM;
    const Part3 pl = part3_layout(dm.CP, LC ? dc.L : 0, 1);
    double* prec = part + (int64_t)((int)blockIdx.x & (kRecShards - 1)) * pl.stride;     // this tile's record entries are ADDED (racc)

    // ---- prologue.  Request order = the order of use: this thread's column of kernel A's batch sums (needed first, behind
    //      one L2 round trip), the X tile (one row, NCH x 4 columns per thread), the first GEMM1 weights, then the small vectors ----
    const int srow = tid >> 4, qcol = 4 * (tid & 15);
    const int bcol = min(tid, dm.C - 1);
    double bsx[kBnShards], bsq[kBnShards];
#pragma unroll
    for (int w = 0; w < kBnShards; ++w) {
        bsx[w] = p.bnacc[(int64_t)w * 2 * dm.CP + bcol];
        bsq[w] = p.bnacc[(int64_t)w * 2 * dm.CP + dm.CP + bcol];
    }
    const float gam = p.gamma[bcol], bet = p.beta[bcol];
    floatx4 xv[NCH];                                              // raw X, kept to the end (xhat, d w_lin)
#pragma unroll
    for (int j = 0; j < NCH; ++j) xv[j] = ld4(X + (int64_t)(m0 + srow) * CP + 64 * j + qcol);   // rows beyond B are zero (host memset)
    constexpr int kPF = 5;                                        // GEMM1 B operands in flight (steps ahead: ~1 K cycles of MFMAs, an L2 round trip under load; 8: 222 VGPRs, the launch 29.8 us against 28.8)
    const __bf16* w1b = xw.W1B + ((int64_t)wave * 64 + lane) * 8; // + step * 8 * 64 * 8
    x3_b8 bq[kPF][3];
#pragma unroll
    for (int s = 0; s < kPF; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) bq[s][q] = x3_ld8(w1b + q * xw.w1b_lo + (int64_t)(s < NST ? s : 0) * 4096);
    const int mt2 = wave & 1, nt2 = wave >> 1;
    const float bias1 = p.b1[16 * wave + n16];
    const float b2v = p.b2[16 * nt2 + n16], w3v = p.w3[16 * nt2 + n16];
    float linv = 0.f, fmv = 0.f, yv = 0.f, wov = 0.f, bov = 0.f, swv = 1.f;
    const float invB = 1.0f / (float)dm.B;                         // (formed here: off the logits phase's dependent chain)
    const double invBd = 1.0 / (double)dm.B;
    if (wave == 0) {
        wov = p.wo[0];
        bov = p.bo ? p.bo[0] : 0.f;
        if (lane < 32 && m0 + lane < dm.B) {
            if (LC == 0) { linv = lin[m0 + lane]; fmv = fm[m0 + lane]; }
            yv = y[m0 + lane];
            if (dc.sw) swv = dc.sw[m0 + lane];
        }
    }
    DT_STAMP(stamps, 6);
    // BN statistics: mean / rstd of this thread's column from the batch sums (every block for itself; double: see bnacc)
    {
        const int col = tid;
        float mean = 0.f, sc = 0.f, be = 0.f, rstd = 0.f, var = 0.f;
        if (col < dm.C) {
            double sx = 0.0, sq = 0.0;
#pragma unroll
            for (int w = 0; w < kBnShards; ++w) { sx += bsx[w]; sq += bsq[w]; }
            const double md = sx * invBd, vd = sq * invBd - md * md;
            mean = (float)md;
            var = vd > 0.0 ? (float)vd : 0.f;
            rstd = 1.0f / sqrtf(var + p.eps);
            sc = rstd * gam;
            be = bet;
        }
        if (col < CP) {
            bnp[col] = mean; bnp[CP + col] = sc; bnp[2 * CP + col] = be; bnp[3 * CP + col] = rstd;
            if (blockIdx.x == 0) {        // published for the launches after this one + the moving statistics
                p.mean_w[col] = mean; p.rstd_w[col] = rstd; p.sc_w[col] = sc; p.betap_w[col] = be;
                p.gammap_w[col] = col < dm.C ? gam : 0.f;
                if (col < dm.C) {
                    if (p.moving_mean) p.moving_mean[col] = p.moving_mean[col] * p.momentum + mean * (1.f - p.momentum);
                    if (p.moving_var) p.moving_var[col] = p.moving_var[col] * p.momentum + var * (1.f - p.momentum);
                }
            }
        }
    }
    static_assert(CP <= 512, "one column per thread in the BN prologue");
    lds_barrier();
    DT_STAMP(stamps, 7);
    // Xn = BN(X), split into its three bf16 parts on the way into LDS
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const floatx4 mu = ld4(bnp + 64 * j + qcol), sc = ld4(bnp + CP + 64 * j + qcol), be = ld4(bnp + 2 * CP + 64 * j + qcol);
        const floatx4 xn = (xv[j] - mu) * sc + be;
        x3_b4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 a = (__bf16)xn[e];
            const float r1 = xn[e] - (float)a;
            const __bf16 b = (__bf16)r1;
            h[e] = a; m[e] = b; l[e] = (__bf16)(r1 - (float)b);
        }
        __bf16* dst = xb + srow * XSB + 64 * j + qcol;
        *reinterpret_cast<x3_b4*>(dst) = h;
        *reinterpret_cast<x3_b4*>(dst + XP) = m;
        *reinterpret_cast<x3_b4*>(dst + 2 * XP) = l;
    }
    lds_barrier();
    DT_STAMP(stamps, 1);

    // ---- GEMM1 (three-part operands, six products: fp32-class): wave w owns hidden units [16w, 16w+16), both row halves ----
    x3_b8 w2q[4][3];                       // GEMM2's B operand, requested in GEMM1's last steps
    float h1r[2][4];
    {
        floatx4 c1[2], c2[2], c3[2];       // a1 b1 | a1 b2 + a2 b1 | a1 b3 + a2 b2 + a3 b1
#pragma unroll
        for (int t = 0; t < 2; ++t) { c1[t] = floatx4{0.f, 0.f, 0.f, 0.f}; c2[t] = c1[t]; c3[t] = c1[t]; }
        const __bf16* arow0 = xb + n16 * XSB + 8 * kg;
        const __bf16* arow1 = xb + (16 + n16) * XSB + 8 * kg;
        x3_b8 an[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { an[0][q] = x3_ld8(arow0 + q * XP); an[1][q] = x3_ld8(arow1 + q * XP); }
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            x3_b8 a[2][3], b[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { a[0][q] = an[0][q]; a[1][q] = an[1][q]; b[q] = bq[s % kPF][q]; }
            if (s + 1 < NST) {                                     // next step's A parts: six 16-byte LDS reads
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    an[0][q] = x3_ld8(arow0 + q * XP + 32 * (s + 1));
                    an[1][q] = x3_ld8(arow1 + q * XP + 32 * (s + 1));
                }
            }
            if (s + kPF < NST) {                                   // the B parts kPF steps ahead
#pragma unroll
                for (int q = 0; q < 3; ++q) bq[s % kPF][q] = x3_ld8(w1b + q * xw.w1b_lo + (int64_t)(s + kPF) * 4096);
            }
            if (NST >= 4 && s >= NST - 4) {                        // the last four steps also request GEMM2's B operand
                const int g = s - (NST - 4);
                const __bf16* w2b = xw.W2B + (((int64_t)g * 4 + nt2) * 64 + lane) * 8;
#pragma unroll
                for (int q = 0; q < 3; ++q) w2q[g][q] = x3_ld8(w2b + q * xw.w2b_lo);
            }
            // the two row halves alternate: no MFMA waits for the one before it
            X3_MFMA(c1[0], a[0][0], b[0]); X3_MFMA(c1[1], a[1][0], b[0]);
            X3_LO(c2[0], a[0][0], b[1]); X3_LO(c2[1], a[1][0], b[1]);
            X3_LO(c3[0], a[0][0], b[2]); X3_LO(c3[1], a[1][0], b[2]);
            X3_LO(c2[0], a[0][1], b[0]); X3_LO(c2[1], a[1][1], b[0]);
            X3_LO(c3[0], a[0][1], b[1]); X3_LO(c3[1], a[1][1], b[1]);
            X3_LO(c3[0], a[0][2], b[0]); X3_LO(c3[1], a[1][2], b[0]);
        }
        if constexpr (NST < 4) {           // a GEMM1 too short to hide them in
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const __bf16* w2b = xw.W2B + (((int64_t)g * 4 + nt2) * 64 + lane) * 8;
#pragma unroll
                for (int q = 0; q < 3; ++q) w2q[g][q] = x3_ld8(w2b + q * xw.w2b_lo);
            }
        }
        DT_STAMP(stamps, 2);
        if constexpr (LC > 0) {
            // ---- Cross forward, part 1 (layers.py:428-436 in the closed form of k_mlp_fwd3): P = [Xn ; b_0 .. b_{L-1}] . [w_0 ..
            //      w_{L-1} w3c], 48 x 16, K = CP, six products (the logits reach +-160: fp32-class or nothing).  Waves 0..5:
            //      row tile w % 3 (0, 1: the Xn parts in LDS; 2: the bias vectors) x K half w / 3; the halves meet in LDS.
            if (wave < 6) {
                const int L = dc.L, mt = wave % 3, kh = wave / 3;
                const int s0 = kh * (NST / 2), s1 = kh ? NST : NST / 2;
                const float* brow = xw.cwp + (int64_t)(n16 < L ? n16 : 2 * L) * CP + 8 * kg;       // column n16: w_l, w3c, then zero
                const float bmask = n16 <= L ? 1.f : 0.f;
                const float* arow2 = xw.cwp + (int64_t)(L + min(n16, L - 1)) * CP + 8 * kg;        // third row tile: b_j
                const float amask2 = n16 < L ? 1.f : 0.f;
                floatx4 q1 = {0.f, 0.f, 0.f, 0.f}, q2 = q1, q3 = q1;
                for (int st = s0; st < s1; ++st) {
                    float bv[8];
                    x3_ld8f(brow + 32 * st, bv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] *= bmask;
                    x3_b8 b1, b2, b3, a1, a2, a3;
                    x3_split3(bv, b1, b2, b3);
                    if (mt < 2) {
                        const __bf16* ap = xb + (16 * mt + n16) * XSB + 32 * st + 8 * kg;
                        a1 = x3_ld8(ap); a2 = x3_ld8(ap + XP); a3 = x3_ld8(ap + 2 * XP);
                    } else {
                        float av[8];
                        x3_ld8f(arow2 + 32 * st, av);
#pragma unroll
                        for (int e = 0; e < 8; ++e) av[e] *= amask2;
                        x3_split3(av, a1, a2, a3);
                    }
                    X3_MFMA(q3, a1, b3); X3_MFMA(q3, a2, b2); X3_MFMA(q3, a3, b1);
                    X3_MFMA(q2, a1, b2); X3_MFMA(q2, a2, b1);
                    X3_MFMA(q1, a1, b1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[(kh * 48 + 16 * mt + 4 * kg + r) * 16 + n16] = (q3[r] + q2[r]) + q1[r];
            }
        }
        // H1 (C layout: column 16w + n16, rows 16t + 4kg + r) -> fp32 in LDS; kept in registers for relu'
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = fmaxf(((c3[t][r] + c2[t][r]) + c1[t][r]) + bias1, 0.f);
                h1r[t][r] = h;
                h1f[(16 * t + 4 * kg + r) * HF + 16 * wave + n16] = h;
            }
    }
    // operands needed later, requested now: dH1's B (rows [16w, 16w+16) of W2, this lane's 8 k of every 32), small vectors
    x3_b8 w2r[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const __bf16* src = xw.W2R + (int64_t)(16 * wave + n16) * kH2 + 32 * g + 8 * kg;
        w2r[g][0] = x3_ld8(src); w2r[g][1] = x3_ld8(src + xw.w2r_lo);
    }
    lds_barrier();
    DT_STAMP(stamps, 3);

    // ---- GEMM2 (six products; the A parts are split from the fp32 H1 tile on the fly): row half mt2, columns [16 nt2, +16) ----
    float h2r[4];
    {
        floatx4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = c1, c3 = c1;
        const float* arow = h1f + (16 * mt2 + n16) * HF + 8 * kg;
        // H1 leaves for HBM (the weight-gradient GEMMs read it) as whole rows, from LDS
        floatx4 hrow[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) hrow[u] = ld4(h1f + ((tid >> 5) + 16 * u) * HF + 4 * (tid & 31));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[8];
            x3_ld8f(arow + 32 * g, v);
            x3_b8 a1, a2, a3;
            x3_split3(v, a1, a2, a3);
            X3_MFMA(c1, a1, w2q[g][0]);
            X3_LO(c2, a1, w2q[g][1]);
            X3_LO(c3, a1, w2q[g][2]);
            X3_LO(c2, a2, w2q[g][0]);
            X3_LO(c3, a2, w2q[g][1]);
            X3_LO(c3, a3, w2q[g][0]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = m0 + (tid >> 5) + 16 * u;
            if (m < dm.B) st4_sel(H1 + (int64_t)m * kH1 + 4 * (tid & 31), hrow[u], dc.wt);
        }
        float zr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h2r[r] = fmaxf(((c3[r] + c2[r]) + c1[r]) + b2v, 0.f);
            zr[r] = group_sum<16>(h2r[r] * w3v);
        }
        // (every wave left GEMM1 before the barrier above: xreg now holds xhat | slp | zp | dzs | cs)
        if (n16 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) zp[nt2 * kTM + 16 * mt2 + 4 * kg + r] = zr[r];
        }
    }
    // xhat = (X - mean) rstd, fp32, for the dXn epilogue
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const floatx4 mu = ld4(bnp + 64 * j + qcol), rs = ld4(bnp + 3 * CP + 64 * j + qcol);
        st4(xs + srow * XS + 64 * j + qcol, (xv[j] - mu) * rs);
    }
    if constexpr (LC > 0) {
        for (int e = tid; e < 48 * 16; e += 512) crP[e] = pb[e] + pb[48 * 16 + e];
    }
    lds_barrier();
    DT_STAMP(stamps, 4);
    // dXn's B operand (rows of W1) for this wave's first two column tiles: requested here, three phases ahead of its use
    // (issued in the dH1 phase it stood in that phase's way: 16 loads per lane in front of a barrier)
    const int NT = (dm.C + 15) >> 4;                               // 16-column tiles of dXn (all C columns: dgamma / dbeta need them)
    auto w1r = [&](int nt) {
        const int col = min(16 * nt + n16, CP - 1);
        return xw.W1R + (int64_t)col * kH1 + 8 * kg;
    };
    x3_b8 bW[2][4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (wave + 8 * i < NT) {
            const __bf16* wr = w1r(wave + 8 * i);
#pragma unroll
            for (int g = 0; g < 4; ++g) { bW[i][g][0] = x3_ld8(wr + 32 * g); bW[i][g][1] = x3_ld8(wr + xw.w1r_lo + 32 * g); }
        }
    }

    // ---- logits, loss, dlogit, dz (wave 0; lanes 32..63 mirror rows 0..31 with zero inputs) ----
    if (wave == 0) {
        const int c = lane & 31, s = lane >> 5;
        const int m = m0 + c;
        float zc = 0.f;
        if constexpr (LC > 0) {
            // Cross forward, part 2: the L scalar steps of row c, everything in registers (x_l = a_l x0 + c_l):
            //   s_l = a_l p_l + q_l,  a_{l+1} = a_l + s_l,  q_l = (b_0 + .. + b_{l-1}) . Wc_l,  z_c = a_L (x0 . w3c) + c_L . w3c
            const int L = dc.L;
            static_assert(LC + 1 <= 12, "three float4 per row of crP");
            float pr[12], gq[LC][12];
            {
                const floatx4 t0 = ld4(crP + c * 16), t1 = ld4(crP + c * 16 + 4), t2 = ld4(crP + c * 16 + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) { pr[e] = t0[e]; pr[4 + e] = t1[e]; pr[8 + e] = t2[e]; }
            }
#pragma unroll
            for (int j = 0; j < LC; ++j) {                     // Gram rows b_j . Wc_l (the same address in every lane: broadcast)
                const floatx4 t0 = ld4(crP + (32 + j) * 16), t1 = ld4(crP + (32 + j) * 16 + 4), t2 = ld4(crP + (32 + j) * 16 + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gq[j][e] = t0[e]; gq[j][4 + e] = t1[e]; gq[j][8 + e] = t2[e]; }
            }
            float a = 1.f;
#pragma unroll
            for (int l = 0; l <= LC; ++l) {
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < LC; ++j)
                    if (j < l) q += gq[j][l];
                if (l < L) {
                    if (s == 0) crA[c * 16 + l] = a;
                    a += a * pr[l] + q;
                } else if (l == L) {
                    if (s == 0) crA[c * 16 + l] = a;
                    zc = a * pr[l] + q;
                }
            }
        }
        float loss = 0.f, dl = 0.f, zz = 0.f;
        if (m < dm.B) {
            const float pt = (zp[c] + zp[kTM + c]) + (zp[2 * kTM + c] + zp[3 * kTM + c]);
            zz = LC ? zc + pt                            // Dense(1)(Concatenate([cross, dnn])) (deepnets.py:194-207)
                    : (linv + fmv) + pt;                 // Add([linear, fm, dnn]) order
            const float lg = zz * wov + bov;
            if (dc.mse) {                                // regression task: 'mse' on the linear task_output
                const float df = lg - yv;
                loss = df * df;
                dl = 2.0f * df * invB;
            } else {
                // The seven other waves of the block wait at the barrier below for this chain (6.6 K of the tile's 50 K cycles
                // with libm's expf / log1pf / IEEE divisions: profiles/r04_deepfm_phase_stamps.txt), so it runs on the
                // hardware transcendentals: e = exp(-|lg|) <= 1 (v_exp_f32), sigmoid = 1 / (1 + e) or e / (1 + e) (v_rcp_f32),
                // log1p(e) = log(1 + e) (v_log_f32; absolute error <= 6e-8: 1 + e is at least 1).  ~1 ulp each: 3e-7 of
                // the largest dlogit, against the 2e-4 gradient bar.
                const float e = __expf(-fabsf(lg));
                const float r = __frcp_rn(1.0f + e);
                const float pr = lg >= 0.f ? r : e * r;
                loss = fmaxf(lg, 0.f) - lg * yv + __logf(1.0f + e);
                dl = (pr - yv) * invB;
            }
            loss *= swv;
            dl *= swv;
            if (s == 0) {
                z_out[m] = zz;
                logit_out[m] = lg;
                dlogit[m] = dl;
                dz_out[m] = dl * wov;
            }
        }
        if (s == 0) dzs[c] = dl * wov;
        if (s == 1) { loss = 0.f; dl = 0.f; }
        float aw = dl * zz, ab = dl;                     // d task_output kernel / bias
        loss = wave_sum(loss); aw = wave_sum(aw); ab = wave_sum(ab);
        if (lane == 0) { radd(prec + pl.loss, loss * invB); radd(prec + pl.dwo, aw); radd(prec + pl.dbo, ab); }
    }
    lds_barrier();
    DT_STAMP(stamps, 5);

    // ---- top of the backward: dH2 in GEMM2's C layout, its column sums; the d w_lin partials ----
    {
        float sb = 0.f, sw = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt2 + 4 * kg + r;
            const float dzv = dzs[row];
            const float g = h2r[r] > 0.f ? dzv * w3v : 0.f;
            d2f[row * DF + 16 * nt2 + n16] = g;
            sb += g;
            sw += dzv * h2r[r];
        }
        sb = row_pair16(sb, false); sw = row_pair16(sw, false);          // over the four row groups (lanes with the same n16)
        if (kg == 0) { cs[(mt2 * 2 + 0) * kH2 + 16 * nt2 + n16] = sb; cs[(mt2 * 2 + 1) * kH2 + 16 * nt2 + n16] = sw; }
        if constexpr (LC == 0) {
            // d linear_logit kernel: sum_rows dz X (raw): the wave's four rows meet through permlane swaps, the eight waves in LDS
            const float dzr = dzs[srow];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                floatx4 v = xv[j] * dzr;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = row_pair16(v[e], false);
                if (kg == 0) st4(slp + wave * CP + 64 * j + qcol, v);
            }
        } else if (wave == 7 && lane < kTM) {
            // ---- Cross backward in the closed form of the forward (x_L = a_L x0 + c_L, g = dz w3c), per row, scalars only:
            //   A_L = g . x0 = dz P[r][L];  l = L-1 .. 0:  coeff_l = A_{l+1} a_l,  A_l = A_{l+1} (1 + p_l);  coeff_L = dz a_L
            //   (k_mlp_fwd3's block of the same name; crS = A_{l+1} | dz replaces the row's a_l once they are in registers)
            const int L = dc.L, r = lane;
            float pr[12], av[12];
            {
                const floatx4 t0 = ld4(crP + r * 16), t1 = ld4(crP + r * 16 + 4), t2 = ld4(crP + r * 16 + 8);
                const floatx4 u0 = ld4(crA + r * 16), u1 = ld4(crA + r * 16 + 4), u2 = ld4(crA + r * 16 + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pr[e] = t0[e]; pr[4 + e] = t1[e]; pr[8 + e] = t2[e];
                    av[e] = u0[e]; av[4 + e] = u1[e]; av[8 + e] = u2[e];
                }
            }
            const float dzv = dzs[r];
            float pL = 0.f, aL = 0.f;
#pragma unroll
            for (int l = 0; l <= LC; ++l) { pL = l == L ? pr[l] : pL; aL = l == L ? av[l] : aL; }
            float cf[16], sa[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) { cf[l] = 0.f; sa[l] = 0.f; }
            float A = dzv * pL;
#pragma unroll
            for (int l = LC - 1; l >= 0; --l) {
                if (l < L) {
                    cf[l] = A * av[l];
                    sa[l] = A;
                    A *= 1.f + pr[l];
                }
            }
#pragma unroll
            for (int l = 0; l <= LC; ++l) cf[l] = l == L ? dzv * aL : cf[l];
            sa[15] = dzv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st4(crF + r * 16 + 4 * e, floatx4{cf[4 * e], cf[4 * e + 1], cf[4 * e + 2], cf[4 * e + 3]});
                st4(crA + r * 16 + 4 * e, floatx4{sa[4 * e], sa[4 * e + 1], sa[4 * e + 2], sa[4 * e + 3]});
            }
        }
    }
    lds_barrier();
    DT_STAMP(stamps, 9);

    // ---- dH1 = relu'(H1) (dH2 W2^T), two-part operands: wave w owns hidden units [16w, 16w+16), both row halves; K = 64 ----
    {
        floatx4 dH[2], dM[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { dH[t] = floatx4{0.f, 0.f, 0.f, 0.f}; dM[t] = dH[t]; }
        // dH2 leaves for HBM as whole rows; the small sums of this tile's record
        const floatx4 drow = ld4(d2f + (tid >> 4) * DF + 4 * (tid & 15));
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[8];
                x3_ld8f(d2f + (16 * t + n16) * DF + 32 * g + 8 * kg, v);
                x3_b8 ah, al;
                x3_split2(v, ah, al);
                X3_MFMA(dH[t], ah, w2r[g][0]);
                X3_LO(dM[t], ah, w2r[g][1]);
                X3_LO(dM[t], al, w2r[g][0]);
            }
        if (tid < kH2) {
            radd(prec + pl.db2 + tid, cs[tid] + cs[2 * kH2 + tid]);
            radd(prec + pl.dw3 + tid, cs[kH2 + tid] + cs[3 * kH2 + tid]);
        }
        {
            const int m = m0 + (tid >> 4);
            if (m < dm.B) st4_sel(dH2 + (int64_t)m * kH2 + 4 * (tid & 15), drow, dc.wt);
        }
        if constexpr (LC == 0) {
            for (int col = tid; col < CP; col += 512) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += slp[w * CP + col];
                radd(prec + pl.slin + col, v);
            }
        } else {
            // the tile's cross record (k_finish_step turns it into d w_l, d b_j, d w3c and the cross path's BN-backward sums):
            //   G_l = sum_r coeff[r][l] xhat[r]  (fp32 MFMA, K = the 32 rows),  Sco_l, SA_l, Sdz = ones^T . [coeff | A, dz]
            const int L = dc.L;
            double* rec = prec + pl.cross;
            const float* crS = crA;
            if (wave == 7) {
                floatx4 d1 = {0.f, 0.f, 0.f, 0.f}, d2 = d1;
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, crF[(4 * st + kg) * 16 + n16], d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, crS[(4 * st + kg) * 16 + n16], d2, 0, 0, 0);
                }
                if (kg == 0) { radd(rec + (L + 1) * CP + n16, d1[0]); radd(rec + (L + 1) * CP + 16 + n16, d2[0]); }
            }
            float bop[8];
#pragma unroll
            for (int st = 0; st < 8; ++st) bop[st] = crF[(4 * st + kg) * 16 + n16];
            for (int ct = wave; ct < 4 * NCH; ct += 8) {          // 16-column tiles of xhat (zero beyond C)
                floatx4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[(4 * st + kg) * XS + 16 * ct + n16], bop[st], d, 0, 0, 0);
                if (n16 <= L) {                                   // C layout: xhat column 16 ct + 4 kg + i, coefficient n16
                    st4(dc.gpart + (int64_t)blockIdx.x * dc.gstride + n16 * CP + 16 * ct + 4 * kg, d);
                }
            }
        }
        float colsum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = h1r[t][r] > 0.f ? dH[t][r] + dM[t][r] : 0.f;
                h1f[(16 * t + 4 * kg + r) * HF + 16 * wave + n16] = g;     // H1's LDS copy is dead: every read sits before two barriers
                colsum += g;
            }
        colsum = row_pair16(colsum, false);
        if (kg == 0) radd(prec + pl.db1 + 16 * wave + n16, colsum);
    }
    lds_barrier();
    DT_STAMP(stamps, 8);

    // ---- dXn = dH1 W1^T, two-part operands: wave w owns column tiles w, w + 8, ..; the two BN-backward partial sums in the epilogue ----
    {
        x3_b8 ah[2][4], al[2][4];          // the dH1 tile's parts, split once from the fp32 tile
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[8];
                x3_ld8f(h1f + (16 * t + n16) * HF + 32 * g + 8 * kg, v);
                x3_split2(v, ah[t][g], al[t][g]);
            }
        // dH1 leaves for HBM as whole rows
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = (tid >> 5) + 16 * u;
            if (m0 + r < dm.B) st4_sel(dH1 + (int64_t)(m0 + r) * kH1 + 4 * (tid & 31), ld4(h1f + r * HF + 4 * (tid & 31)), dc.wt);
        }
        DT_STAMP(stamps, 13);
        auto tile = [&](int buf, int nt) {
            floatx4 gH[2], gM[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { gH[t] = floatx4{0.f, 0.f, 0.f, 0.f}; gM[t] = gH[t]; }
            const int col = 16 * nt + n16;
            float xh_[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) xh_[t][r] = xs[(16 * t + 4 * kg + r) * XS + col];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                X3_MFMA(gH[0], ah[0][g], bW[buf][g][0]); X3_MFMA(gH[1], ah[1][g], bW[buf][g][0]);
                X3_LO(gM[0], ah[0][g], bW[buf][g][1]); X3_LO(gM[1], ah[1][g], bW[buf][g][1]);
                X3_LO(gM[0], al[0][g], bW[buf][g][0]); X3_LO(gM[1], al[1][g], bW[buf][g][0]);
            }
            if constexpr (LC > 0) {
                // + sum_l coeff[r][l] Wc_l[col] (exact fp32 MFMA, K = 16 layer slots): A = the coefficient tile, B = the layer
                // vectors of this column (padded copy: zero beyond C), Wc_L = w3c, nothing beyond L
                const int L = dc.L;
#pragma unroll
                for (int g2 = 0; g2 < 4; ++g2) {
                    const int l = 4 * g2 + kg;
                    const float bw = l <= L ? xw.cwp[(int64_t)(l < L ? l : 2 * L) * CP + col] : 0.f;
                    gH[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(crF[n16 * 16 + l], bw, gH[0], 0, 0, 0);
                    gH[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(crF[(16 + n16) * 16 + l], bw, gH[1], 0, 0, 0);
                }
            }
            if (nt + 16 < NT) {              // the tile after next reuses this buffer: its W1 rows are requested now
                const __bf16* wn = w1r(nt + 16);
#pragma unroll
                for (int g = 0; g < 4; ++g) { bW[buf][g][0] = x3_ld8(wn + 32 * g); bW[buf][g][1] = x3_ld8(wn + xw.w1r_lo + 32 * g); }
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gx = gH[t][r] + gM[t][r];
                    s1 += gx;
                    s2 += gx * xh_[t][r];
                    xs[(16 * t + 4 * kg + r) * XS + col] = gx;      // dXn replaces xhat in place (this lane's own entries)
                }
            s1 = row_pair16(s1, false); s2 = row_pair16(s2, false);
            if (kg == 0 && col < dm.C) { radd(prec + pl.sdx + col, s1); radd(prec + pl.sdxx + col, s2); }
        };
        // buffer ids are literals: the operand arrays stay in registers (C <= 512: at most 32 tiles, 4 per wave)
        if (wave < NT) tile(0, wave);
        if (wave + 8 < NT) tile(1, wave + 8);
        if (wave + 16 < NT) tile(0, wave + 16);
        if (wave + 24 < NT) tile(1, wave + 24);
    }
    lds_barrier();
    DT_STAMP(stamps, 14);
    // the tile's dXn rows leave as whole rows (the F*D embedding columns: the dense inputs need no gradient)
    {
        const int fq = (dm.F * dm.D) >> 2;                 // 16-byte pieces per row (<= 128)
        const int rgs = 512 / fq, rg = tid / fq, q = tid - rg * fq;
        if (rg < rgs) {
            for (int row = rg; row < kTM; row += rgs)
                if (m0 + row < dm.B) st4_sel(dxn_out + (int64_t)(m0 + row) * CP + 4 * q, ld4(xs + row * XS + 4 * q), dc.wt);
        }
    }
    DT_STAMP(stamps, 15);
}
