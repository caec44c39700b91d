// tower_x3.h — kernel C of the pipelined DeepFM step with the Dense tower on SPLIT-bf16 matrix cores ("bf16 x 3").
// Included by deepfm.hip inside namespace dt, after MlpParams / Part3 / DcnArgs / the k_mlp_fwd3 helpers.
//
// Why: gfx950 has no reduced-precision fp32 MFMA (no xf32); v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, and the
// four GEMMs of the tile kernel (Dense128, Dense64, dH1 = dH2 W2^T, dXn = dH1 W1^T: 33 K cycles of fp32-MFMA time per
// 32-row tile, deepnets.py:401-427) were half of k_mlp_fwd3's 70 K cycles.  Every fp32 operand is split once into two bf16
// halves  a = a_hi + a_lo  (a_hi = bf16(a) round-to-nearest, a_lo = bf16(a - a_hi): 16 mantissa bits together) and
//     a b  ~  a_hi b_hi + (a_hi b_lo + a_lo b_hi)          (the dropped a_lo b_lo term is 2^-16 of the product)
// runs as THREE v_mfma_f32_16x16x32_bf16 with fp32 accumulation — the small terms in their own accumulator, added last.
// Measured against the float64 oracle at B = 8192: DESIGN.md §3.5 (logits ~3e-5 at |logit| <= 5, gradients ~5e-6 of the
// tensor max; the fp32 path: 1e-6 / 3e-7).
//
// Shape: 512 threads = 8 waves (two per SIMD: one wave's LDS / L2 latencies hide behind the other's MFMAs), a 32-row
// tile per block as in k_mlp_fwd3.  Every wave owns 16-column output tiles and BOTH 16-row halves of the tile, so a B
// operand (weights, from L2) is fetched by exactly one wave:
//   GEMM1  H1 = relu(Xn W1 + b1)    wave w: hidden units [16w, 16w+16); K = CP in steps of 32; A = Xn tile (hi | lo) in LDS
//   GEMM2  H2 = relu(H1 W2 + b2)    wave w: row half w & 1, columns [16 (w >> 1), +16); K = 128
//   dH1    = relu'(H1) (dH2 W2^T)   wave w: hidden units [16w, 16w+16); K = 64; B = rows of W2 (k contiguous)
//   dXn    = dH1 W1^T               wave w: column tiles w, w + 8, ..; K = 128; B = rows of W1 (k contiguous)
// A 16x16x32 operand is 8 consecutive k per lane (lane (i = l % 16, g = l / 16) <-> k = 32 step + 8 g + j): 16-byte LDS /
// global reads throughout; the weight halves come from lane-major (GEMM1, GEMM2) or row-major (dH1, dXn) bf16 copies the
// prep launch writes once per step (X3Weights).  LDS rows are padded so that the 16 lanes of a ds_read_b128 group hit 16
// distinct 16-byte bank groups (row stride / 16 B = 2 or 10 mod 16).
#pragma once

typedef __bf16 x3_b8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_b4 __attribute__((ext_vector_type(4)));

struct X3Weights {
    // lane-major B operands: [step][wave or column tile][lane][8], hi half then lo half `lo_off` elements later
    const __bf16* W1B; int64_t w1b_lo;      // GEMM1: element j of lane (n, g), step s, wave w = W1[32 s + 8 g + j][16 w + n]
    const __bf16* W1R; int64_t w1r_lo;      // dXn:   [CP rows][128] row-major split copy of W1 (rows >= C zero)
    const __bf16* W2B; int64_t w2b_lo;      // GEMM2: element j of lane (n, g), step s, column tile t = W2[32 s + 8 g + j][16 t + n]
    const __bf16* W2R; int64_t w2r_lo;      // dH1:   [128][64] row-major split copy of W2
};

__device__ __forceinline__ void x3_split(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}
__device__ __forceinline__ void x3_split4(floatx4 v, x3_b4& hi, x3_b4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 h, l;
        x3_split(v[e], h, l);
        hi[e] = h; lo[e] = l;
    }
}
__device__ __forceinline__ x3_b8 x3_ld8(const __bf16* p) { return *reinterpret_cast<const x3_b8*>(p); }
// acc_hi += a_hi b_hi;  acc_mix += a_hi b_lo + a_lo b_hi
#define X3_MMA(accH, accM, ah, al, bh, bl)                                               \
    do {                                                                                 \
        accH = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, accH, 0, 0, 0);           \
        accM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, accM, 0, 0, 0);           \
        accM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, accM, 0, 0, 0);           \
    } while (0)

constexpr int kX3H1S = kH1 + 16;      // bf16 row stride of the H1 / dH1 halves (288 B: 18 = 2 mod 16)
constexpr int kX3H2S = kH2 + 16;      // bf16 row stride of the dH2 halves (160 B: 10 mod 16)

// LDS plan (bytes), CP = 64 NCH:
//   xreg  max(2 * 32 * (CP + 16) * 2, 32 * (CP + 4) * 4)   Xn tile as bf16 hi | lo (GEMM1), then xhat / dXn fp32 [32][CP + 4]
//   bnp   4 * CP * 4                                       mean | gamma rstd | beta | rstd
//   h1f   32 * (128 + 4) * 4                               H1 fp32 (row stores), then dH1
//   h1b   2 * 32 * kX3H1S * 2                              H1 hi | lo (GEMM2's A), then dH1 hi | lo (dXn's A)
//   d2f   32 * (64 + 4) * 4                                dH2 fp32
//   d2b   2 * 32 * kX3H2S * 2                              dH2 hi | lo
//   slp   8 * CP * 4                                       the waves' d w_lin partial sums
//   zp [4][32] | dzs [32] | cs [2][2][64]
__host__ __device__ constexpr size_t x3_lds_bytes(int CP) {
    const size_t a = (size_t)2 * kTM * (CP + 16) * 2, b = (size_t)kTM * (CP + 4) * 4;
    return (a > b ? a : b) + (size_t)4 * CP * 4 + (size_t)kTM * (kH1 + 4) * 4 + (size_t)2 * kTM * kX3H1S * 2 +
           (size_t)kTM * (kH2 + 4) * 4 + (size_t)2 * kTM * kX3H2S * 2 + (size_t)8 * CP * 4 + (4 * kTM + kTM + 4 * kH2) * 4;
}

template <int NCH>
__global__ __launch_bounds__(512) void k_tower_x3(const float* __restrict__ X, MlpParams p, X3Weights xw, DeepFmDims dm,
                                                  const float* __restrict__ lin, const float* __restrict__ fm,
                                                  const float* __restrict__ y, float* __restrict__ H1,
                                                  float* __restrict__ dH1, float* __restrict__ dH2,
                                                  float* __restrict__ z_out, float* __restrict__ logit_out,
                                                  float* __restrict__ dlogit, float* __restrict__ dz_out,
                                                  float* __restrict__ part, unsigned long long* stamps, DcnArgs dc,
                                                  float* __restrict__ dxn_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    DT_STAMP(stamps, 0);
    constexpr int CP = 64 * NCH, NST = CP / 32, XSB = CP + 16, XS = CP + 4, HF = kH1 + 4, DF = kH2 + 4;
    constexpr size_t kXreg = ((size_t)2 * kTM * XSB * 2 > (size_t)kTM * XS * 4) ? (size_t)2 * kTM * XSB * 2 : (size_t)kTM * XS * 4;
    char* base = reinterpret_cast<char*>(lds);
    __bf16* xh = reinterpret_cast<__bf16*>(base);                 // [32][XSB]
    __bf16* xl = xh + kTM * XSB;                                  // [32][XSB]
    float* xs = reinterpret_cast<float*>(base);                   // [32][XS]   (after GEMM1)
    float* bnp = reinterpret_cast<float*>(base + kXreg);          // [4][CP]
    float* h1f = bnp + 4 * CP;                                    // [32][HF]
    __bf16* h1bh = reinterpret_cast<__bf16*>(h1f + kTM * HF);     // [32][kX3H1S]
    __bf16* h1bl = h1bh + kTM * kX3H1S;
    float* d2f = reinterpret_cast<float*>(h1bl + kTM * kX3H1S);   // [32][DF]
    __bf16* d2bh = reinterpret_cast<__bf16*>(d2f + kTM * DF);     // [32][kX3H2S]
    __bf16* d2bl = d2bh + kTM * kX3H2S;
    float* slp = reinterpret_cast<float*>(d2bl + kTM * kX3H2S);   // [8][CP]
    float* zp = slp + 8 * CP;                                     // [4][32]
    float* dzs = zp + 4 * kTM;                                    // [32]
    float* cs = dzs + kTM;                                        // [2][2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.x * kTM;
    const Part3 pl = part3_layout(dm.CP, 0, 1);
    float* prec = part + (int64_t)blockIdx.x * pl.stride;

    // ---- prologue: the X tile (one row, NCH x 4 columns per thread) and the first GEMM1 weights are requested first ----
    const int srow = tid >> 4, qcol = 4 * (tid & 15);
    floatx4 xv[NCH];                                              // raw X, kept to the end (xhat, d w_lin)
#pragma unroll
    for (int j = 0; j < NCH; ++j) xv[j] = ld4(X + (int64_t)(m0 + srow) * CP + 64 * j + qcol);   // rows beyond B are zero (host memset)
    constexpr int kPF = 4;                                        // GEMM1 B operands in flight (steps ahead)
    const __bf16* w1b = xw.W1B + ((int64_t)wave * 64 + lane) * 8; // + step * 8 * 64 * 8
    x3_b8 bqh[kPF], bql[kPF];
#pragma unroll
    for (int s = 0; s < kPF; ++s) {
        bqh[s] = x3_ld8(w1b + (int64_t)(s < NST ? s : 0) * 4096);
        bql[s] = x3_ld8(w1b + xw.w1b_lo + (int64_t)(s < NST ? s : 0) * 4096);
    }
    DT_STAMP(stamps, 6);
    // BN level 2 behind the loads: mean / rstd of this thread's column from the level-1 slices (every block for itself)
    {
        const int col = tid;
        float mean = 0.f, sc = 0.f, be = 0.f, rstd = 0.f, var = 0.f;
        if (col < dm.C) {
            bn_merge_slices(p.bn2, dm.C, col, mean, var);
            rstd = 1.0f / sqrtf(var + p.eps);
            sc = rstd * p.gamma[col];
            be = p.beta[col];
        }
        if (col < CP) {
            bnp[col] = mean; bnp[CP + col] = sc; bnp[2 * CP + col] = be; bnp[3 * CP + col] = rstd;
            if (blockIdx.x == 0) {        // published for the launches after this one + the moving statistics
                p.mean_w[col] = mean; p.rstd_w[col] = rstd; p.sc_w[col] = sc; p.betap_w[col] = be;
                p.gammap_w[col] = col < dm.C ? p.gamma[col] : 0.f;
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
    // Xn = BN(X), split into its bf16 halves on the way into LDS
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const floatx4 mu = ld4(bnp + 64 * j + qcol), sc = ld4(bnp + CP + 64 * j + qcol), be = ld4(bnp + 2 * CP + 64 * j + qcol);
        x3_b4 hi, lo;
        x3_split4((xv[j] - mu) * sc + be, hi, lo);
        *reinterpret_cast<x3_b4*>(xh + srow * XSB + 64 * j + qcol) = hi;
        *reinterpret_cast<x3_b4*>(xl + srow * XSB + 64 * j + qcol) = lo;
    }
    lds_barrier();
    DT_STAMP(stamps, 1);

    // ---- GEMM1: wave w owns hidden units [16w, 16w+16), both row halves ----
    floatx4 aH[2], aM[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { aH[t] = floatx4{0.f, 0.f, 0.f, 0.f}; aM[t] = aH[t]; }
    // operands needed after GEMM1, requested under it: GEMM2's B (4 steps), dH1's B (2 steps), biases
    const int mt2 = wave & 1, nt2 = wave >> 1;
    x3_b8 w2h[4], w2l[4], w2rh[2], w2rl[2];
    {
        const __bf16* arow0 = xh + n16 * XSB + 8 * kg;
        const __bf16* arow1 = xh + (16 + n16) * XSB + 8 * kg;
        constexpr int LO = kTM * XSB;                              // xl = xh + LO
        x3_b8 a0h = x3_ld8(arow0), a1h = x3_ld8(arow1), a0l = x3_ld8(arow0 + LO), a1l = x3_ld8(arow1 + LO);
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            const x3_b8 bh = bqh[s % kPF], bl = bql[s % kPF];
            const x3_b8 c0h = a0h, c1h = a1h, c0l = a0l, c1l = a1l;
            if (s + 1 < NST) {                                     // next step's A halves: four 16-byte LDS reads
                a0h = x3_ld8(arow0 + 32 * (s + 1)); a1h = x3_ld8(arow1 + 32 * (s + 1));
                a0l = x3_ld8(arow0 + LO + 32 * (s + 1)); a1l = x3_ld8(arow1 + LO + 32 * (s + 1));
            }
            if (s + kPF < NST) {                                   // the B halves kPF steps ahead
                bqh[s % kPF] = x3_ld8(w1b + (int64_t)(s + kPF) * 4096);
                bql[s % kPF] = x3_ld8(w1b + xw.w1b_lo + (int64_t)(s + kPF) * 4096);
            } else if (s + kPF < NST + 4) {                        // tail slots: GEMM2's B operand
                const int g = s + kPF - NST;
                const __bf16* w2b = xw.W2B + (((int64_t)g * 4 + nt2) * 64 + lane) * 8;
                w2h[g] = x3_ld8(w2b); w2l[g] = x3_ld8(w2b + xw.w2b_lo);
            }
            X3_MMA(aH[0], aM[0], c0h, c0l, bh, bl);
            X3_MMA(aH[1], aM[1], c1h, c1l, bh, bl);
        }
    }
    DT_STAMP(stamps, 2);
#pragma unroll
    for (int g = 0; g < 2; ++g) {        // dH1's B: rows [16w, 16w+16) of W2, this lane's 8 k of every 32
        const __bf16* w2r = xw.W2R + (int64_t)(16 * wave + n16) * kH2 + 32 * g + 8 * kg;
        w2rh[g] = x3_ld8(w2r); w2rl[g] = x3_ld8(w2r + xw.w2r_lo);
    }
    const float bias1 = p.b1[16 * wave + n16];
    const float b2v = p.b2[16 * nt2 + n16], w3v = p.w3[16 * nt2 + n16];
    float linv = 0.f, fmv = 0.f, yv = 0.f, wov = 0.f, bov = 0.f, swv = 1.f;
    if (wave == 0) {
        wov = p.wo[0];
        bov = p.bo ? p.bo[0] : 0.f;
        if (lane < 32 && m0 + lane < dm.B) {
            linv = lin[m0 + lane]; fmv = fm[m0 + lane];
            yv = y[m0 + lane];
            if (dc.sw) swv = dc.sw[m0 + lane];
        }
    }
    // H1 (C layout: column 16w + n16, rows 16t + 4kg + r) -> fp32 rows + bf16 halves in LDS; kept in registers for relu'
    float h1r[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * kg + r;
            const float h = fmaxf((aH[t][r] + aM[t][r]) + bias1, 0.f);
            h1r[t][r] = h;
            __bf16 hi, lo;
            x3_split(h, hi, lo);
            h1f[row * HF + 16 * wave + n16] = h;
            h1bh[row * kX3H1S + 16 * wave + n16] = hi;
            h1bl[row * kX3H1S + 16 * wave + n16] = lo;
        }
    lds_barrier();
    DT_STAMP(stamps, 3);

    // ---- GEMM2: wave w owns row half mt2, columns [16 nt2, +16) ----
    float h2r[4];
    {
        floatx4 cH = {0.f, 0.f, 0.f, 0.f}, cM = cH;
        const __bf16* arow = h1bh + (16 * mt2 + n16) * kX3H1S + 8 * kg;
        x3_b8 ah[4], al[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { ah[g] = x3_ld8(arow + 32 * g); al[g] = x3_ld8(arow + kTM * kX3H1S + 32 * g); }
        // H1 leaves for HBM (the weight-gradient GEMMs read it) as whole rows, from LDS
        floatx4 hrow[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) hrow[u] = ld4(h1f + ((tid >> 5) + 16 * u) * HF + 4 * (tid & 31));
#pragma unroll
        for (int g = 0; g < 4; ++g) X3_MMA(cH, cM, ah[g], al[g], w2h[g], w2l[g]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = m0 + (tid >> 5) + 16 * u;
            if (m < dm.B) st4_sel(H1 + (int64_t)m * kH1 + 4 * (tid & 31), hrow[u], dc.wt);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h2r[r] = fmaxf((cH[r] + cM[r]) + b2v, 0.f);
            const float v = group_sum<16>(h2r[r] * w3v);
            if (n16 == 0) zp[nt2 * kTM + 16 * mt2 + 4 * kg + r] = v;
        }
    }
    // xhat = (X - mean) rstd replaces the bf16 halves of Xn (dead since the barrier above) — fp32, for the dXn epilogue
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const floatx4 mu = ld4(bnp + 64 * j + qcol), rs = ld4(bnp + 3 * CP + 64 * j + qcol);
        st4(xs + srow * XS + 64 * j + qcol, (xv[j] - mu) * rs);
    }
    lds_barrier();
    DT_STAMP(stamps, 4);

    // ---- logits, loss, dlogit, dz (wave 0; lanes 32..63 mirror rows 0..31 with zero inputs) ----
    if (wave == 0) {
        const int c = lane & 31, s = lane >> 5;
        const int m = m0 + c;
        float loss = 0.f, dl = 0.f, zz = 0.f;
        if (m < dm.B) {
            const float pt = (zp[c] + zp[kTM + c]) + (zp[2 * kTM + c] + zp[3 * kTM + c]);
            zz = (linv + fmv) + pt;                      // Add([linear, fm, dnn]) order
            const float lg = zz * wov + bov;
            if (dc.mse) {                                // regression task: 'mse' on the linear task_output
                const float df = lg - yv;
                loss = df * df;
                dl = 2.0f * df / (float)dm.B;
            } else {
                const float pr = 1.0f / (1.0f + expf(-lg));
                loss = fmaxf(lg, 0.f) - lg * yv + log1pf(expf(-fabsf(lg)));
                dl = (pr - yv) / (float)dm.B;
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
        if (lane == 0) { prec[pl.loss] = loss / (float)dm.B; prec[pl.dwo] = aw; prec[pl.dbo] = ab; }
    }
    lds_barrier();
    DT_STAMP(stamps, 5);

    // ---- top of the backward: dH2 in GEMM2's C layout (fp32 rows + bf16 halves), its column sums; d w_lin partials ----
    {
        float sb = 0.f, sw = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt2 + 4 * kg + r;
            const float dzv = dzs[row];
            const float g = h2r[r] > 0.f ? dzv * w3v : 0.f;
            __bf16 hi, lo;
            x3_split(g, hi, lo);
            d2f[row * DF + 16 * nt2 + n16] = g;
            d2bh[row * kX3H2S + 16 * nt2 + n16] = hi;
            d2bl[row * kX3H2S + 16 * nt2 + n16] = lo;
            sb += g;
            sw += dzv * h2r[r];
        }
        sb += __shfl_xor(sb, 16, 64); sw += __shfl_xor(sw, 16, 64);
        sb += __shfl_xor(sb, 32, 64); sw += __shfl_xor(sw, 32, 64);
        if (kg == 0) { cs[(mt2 * 2 + 0) * kH2 + 16 * nt2 + n16] = sb; cs[(mt2 * 2 + 1) * kH2 + 16 * nt2 + n16] = sw; }
        // d linear_logit kernel: sum_rows dz X (raw): the wave's four rows meet by shuffles, the eight waves in LDS
        const float dzr = dzs[srow];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            floatx4 v = xv[j] * dzr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                v[e] = t;
            }
            if (kg == 0) st4(slp + wave * CP + 64 * j + qcol, v);
        }
    }
    lds_barrier();
    DT_STAMP(stamps, 9);

    // ---- dH1 = relu'(H1) (dH2 W2^T): wave w owns hidden units [16w, 16w+16), both row halves; K = 64 ----
    {
        floatx4 dH[2], dM[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { dH[t] = floatx4{0.f, 0.f, 0.f, 0.f}; dM[t] = dH[t]; }
        x3_b8 ah[2][2], al[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const __bf16* a = d2bh + (16 * t + n16) * kX3H2S + 32 * g + 8 * kg;
                ah[t][g] = x3_ld8(a); al[t][g] = x3_ld8(a + kTM * kX3H2S);
            }
        // dH2 leaves for HBM as whole rows; the small sums of this tile's record
        const floatx4 drow = ld4(d2f + (tid >> 4) * DF + 4 * (tid & 15));
        if (tid < kH2) {
            prec[pl.db2 + tid] = cs[tid] + cs[2 * kH2 + tid];
            prec[pl.dw3 + tid] = cs[kH2 + tid] + cs[3 * kH2 + tid];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            X3_MMA(dH[0], dM[0], ah[0][g], al[0][g], w2rh[g], w2rl[g]);
            X3_MMA(dH[1], dM[1], ah[1][g], al[1][g], w2rh[g], w2rl[g]);
        }
        {
            const int m = m0 + (tid >> 4);
            if (m < dm.B) st4_sel(dH2 + (int64_t)m * kH2 + 4 * (tid & 15), drow, dc.wt);
        }
        for (int col = tid; col < CP; col += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += slp[w * CP + col];
            prec[pl.slin + col] = v;
        }
        float colsum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * kg + r;
                const float g = h1r[t][r] > 0.f ? dH[t][r] + dM[t][r] : 0.f;
                __bf16 hi, lo;
                x3_split(g, hi, lo);
                h1f[row * HF + 16 * wave + n16] = g;       // H1's LDS copies are dead: every read sits before two barriers
                h1bh[row * kX3H1S + 16 * wave + n16] = hi;
                h1bl[row * kX3H1S + 16 * wave + n16] = lo;
                colsum += g;
            }
        colsum += __shfl_xor(colsum, 16, 64);
        colsum += __shfl_xor(colsum, 32, 64);
        if (kg == 0) prec[pl.db1 + 16 * wave + n16] = colsum;
    }
    // the first column tile's W1 rows are on their way while the dH1 tile settles
    const int NT = (dm.C + 15) >> 4;                               // 16-column tiles of dXn (all C columns: dgamma / dbeta need them)
    auto w1r = [&](int nt) {
        const int col = min(16 * nt + n16, CP - 1);
        return xw.W1R + (int64_t)col * kH1 + 8 * kg;
    };
    x3_b8 bWh[2][4], bWl[2][4];
    if (wave < NT) {
        const __bf16* wr = w1r(wave);
#pragma unroll
        for (int g = 0; g < 4; ++g) { bWh[0][g] = x3_ld8(wr + 32 * g); bWl[0][g] = x3_ld8(wr + xw.w1r_lo + 32 * g); }
    }
    lds_barrier();
    DT_STAMP(stamps, 8);

    // ---- dXn = dH1 W1^T: wave w owns column tiles w, w + 8, ..; the tile's two BN-backward partial sums in the epilogue ----
    {
        x3_b8 ah[2][4], al[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const __bf16* a = h1bh + (16 * t + n16) * kX3H1S + 32 * g + 8 * kg;
                ah[t][g] = x3_ld8(a); al[t][g] = x3_ld8(a + kTM * kX3H1S);
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
            const bool more = nt + 8 < NT;
            const __bf16* wn = w1r(nt + 8);
            const int col = 16 * nt + n16;
            float xh_[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) xh_[t][r] = xs[(16 * t + 4 * kg + r) * XS + col];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                X3_MMA(gH[0], gM[0], ah[0][g], al[0][g], bWh[buf][g], bWl[buf][g]);
                X3_MMA(gH[1], gM[1], ah[1][g], al[1][g], bWh[buf][g], bWl[buf][g]);
                if (more) { bWh[buf ^ 1][g] = x3_ld8(wn + 32 * g); bWl[buf ^ 1][g] = x3_ld8(wn + xw.w1r_lo + 32 * g); }
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
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (kg == 0 && col < dm.C) { prec[pl.sdx + col] = s1; prec[pl.sdxx + col] = s2; }
        };
        // buffer ids are literals: the operand arrays stay in registers (C <= 544: at most 34 tiles, 5 per wave)
        if (wave < NT) tile(0, wave);
        if (wave + 8 < NT) tile(1, wave + 8);
        if (wave + 16 < NT) tile(0, wave + 16);
        if (wave + 24 < NT) tile(1, wave + 24);
        if (wave + 32 < NT) tile(0, wave + 32);
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
