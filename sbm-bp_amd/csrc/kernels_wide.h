// Kernels for label counts above 16 (17 .. 64), gfx950. No reference counterpart beyond the equations: the reference loops
// over Q_ with no cap (belief_propagation.cpp:991-1049).
//
// Up to Q = 16 a lane owns a whole directed edge and keeps its Q-vectors in registers (kernels.h). Above that the vectors do
// not fit, so an edge is spread over FOUR lanes: a wave works on a tile of 16 edges, lane l holds the labels (l >> 4) + 4 s,
// s = 0 .. QS-1, of edge (l & 15) ("layout L"; QP = 16 QT = 4 QS labels, padded with zeros above Q). This is exactly the
// B-operand and the C/D layout of v_mfma_f64_16x16x4_f64 when the matrix product is written as
//     b^T (labels x edges)  =  W^T (labels x labels)  .  m^T (labels x edges):
// A operand: lane l holds W^T[row q = 16 rt + (l & 15)][k = t = 4 s + (l >> 4)] = W[t][q]   (one double per lane and tile)
// B operand: lane l holds m[edge l & 15][t = 4 s + (l >> 4)]                                 (register s of layout L)
// C/D:       lane l, register r: row (l >> 4) + 4 r, column l & 15  ->  label 16 rt + (l >> 4) + 4 r of edge l & 15
//            = register r + 4 rt of layout L
// so the result of one product is the operand of the next without any lane movement, and b = W^T m costs QT * QS matrix
// instructions per 16 edges instead of Q * Q FMAs per edge. Sums over the labels of an edge are a loop over the registers
// plus two shuffles (xor 16, xor 32). The A tiles of W are read from a tile-ordered copy in HBM (32 KB at Q = 64: L1 / L2
// resident). Rows of any degree are handled inside the one sweep kernel: segments of up to WCAP edges keep their edge fields
// in LDS; a longer row is walked twice in chunks of WCAP edges (products, then cavities with the fields recomputed).
// The sweep kernel numbers the labels differently ("layout V", wide_label_v: two consecutive labels per lane and register pair,
// so that records move in 16-byte accesses) - the product does not care as long as the A tiles use the same numbering.
//
// Message-gather form only (every sweep reports the reference's 1-step difference), full Q-component message records, every cab
// entry > 0, deg_corr_flag 0 or 1. What the sweep kernel's time is made of (round 3): a workgroup is a chain
// of dependent round trips (segment -> rev -> records -> product -> row phase -> cavities), so waves in flight decide: register
// targets of 4 / 3 / 3 waves per SIMD for Q <= 32 / 48 / 64, every independent load requested before the first wait, 16-byte
// record accesses, reciprocals instead of divisions. Measured and dropped: workgroups looping over segments with the W tiles
// in registers or in LDS (fewer waves in flight cost more than the L2 traffic of the tiles saves).
#ifndef SBMBP_KERNELS_WIDE_H
#define SBMBP_KERNELS_WIDE_H

#include "kernels.h"

namespace sbmbp {

#ifndef SBMBP_WIDE_PAD
#define SBMBP_WIDE_PAD 8          // doubles added to the LDS row stride of k_wsweep
#endif
#ifndef SBMBP_WIDE_WAVES2
#define SBMBP_WIDE_WAVES2 4       // waves per SIMD the register allocation of k_wsweep<QT> aims for, QT = 2 / 3 / 4
#endif
#ifndef SBMBP_WIDE_WAVES3
#define SBMBP_WIDE_WAVES3 3
#endif
#ifndef SBMBP_WIDE_WAVES4
#define SBMBP_WIDE_WAVES4 3
#endif
constexpr int WQ = 64;     // largest label count
constexpr int WCAP = 64;   // directed edges per segment (4 tiles of 16, one per wave)
constexpr int WRCAP = 16;  // rows per segment
constexpr int WTPB = 256;
typedef double d4 __attribute__((ext_vector_type(4)));

// parameters of the wide path (arrays packed with stride Q); the scalars and the convergence state stay in dev_params
struct dev_wide {
    double W[WQ * WQ];       // cab^beta (dc 0) or cab (dc 1)
    double cab[WQ * WQ];
    double logcab[WQ * WQ];
    double eta[WQ], logeta[WQ], hN[WQ], S[WQ];
    double arS1[WQ], arS2[WQ];  // raw field sums of the last two sweeps (adaptive relaxation, signature F)
    // tile-ordered copies for the A operand (wide_tiles): W, cab, cab * log cab
    double tW[WQ * WQ], tC[WQ * WQ], tCL[WQ * WQ];
    double tWv[WQ * WQ];     // W in the tile order of layout V (k_wsweep)
};

// T[(rt * QS + s) * 64 + l] = Wm[(4 s + (l >> 4)) * Q + 16 rt + (l & 15)], zero outside the Q x Q matrix
__host__ inline void wide_tiles(const double *Wm, int Q, double *T) {
    const int QT = (Q + 15) / 16, QS = 4 * QT;
    for (int rt = 0; rt < QT; ++rt)
        for (int s = 0; s < QS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int t = 4 * s + (l >> 4), q = 16 * rt + (l & 15);
                T[(rt * QS + s) * 64 + l] = (t < Q && q < Q) ? Wm[t * Q + q] : 0.0;
            }
}

// Layout V (k_wsweep only): lane group g, register s holds label 8 (s >> 1) + 2 g + (s & 1) - the four lanes of an edge cover 64
// consecutive bytes of its record with one 16-byte access each. The matrix product does not care how the labels are numbered as
// long as the A tiles use the same numbering for their columns (operand registers) and rows (result registers):
__host__ __device__ inline int wide_label_v(int g, int s) { return 8 * (s >> 1) + 2 * g + (s & 1); }
__host__ inline void wide_tiles_v(const double *Wm, int Q, double *T) {
    const int QT = (Q + 15) / 16, QS = 4 * QT;
    for (int rt = 0; rt < QT; ++rt)
        for (int s = 0; s < QS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, k = l >> 4;                         // A[row i][column k] of the (rt, s) tile
                const int q = wide_label_v(i & 3, (i >> 2) + 4 * rt);     // result row i = g + 4 r lands in register r + 4 rt of lane group g
                const int t = wide_label_v(k, s);
                T[(rt * QS + s) * 64 + l] = (t < Q && q < Q) ? Wm[t * Q + q] : 0.0;
            }
}

// out[s'] (label g + 4 s' of edge lane & 15) = sum_t Wm[t][label] v[t]; all 64 lanes take part (EXEC all ones)
template <int QT>
__device__ __forceinline__ void wide_matvec(const double *__restrict__ T, const double (&v)[4 * QT], double (&out)[4 * QT]) {
    constexpr int QS = 4 * QT;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int rt = 0; rt < QT; ++rt) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < QS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(rt * QS + s) * 64 + lane], v[s], acc, 0, 0, 0);
        out[4 * rt + 0] = acc.x;
        out[4 * rt + 1] = acc.y;
        out[4 * rt + 2] = acc.z;
        out[4 * rt + 3] = acc.w;
    }
}
// sum / maximum over the labels of the lane's edge: the registers, then the three other lanes of the edge
template <int QS> __device__ __forceinline__ double edge_sum(const double (&v)[QS]) {
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < QS; ++s) t += v[s];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    return t;
}
// load the labels of layout L of record k (stride Q), zero above Q or when the lane's edge slot is empty
template <int QS> __device__ __forceinline__ void wide_load(const double *__restrict__ M, size_t k, int Q, bool valid, double (&v)[QS]) {
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int s = 0; s < QS; ++s) { const int t = g + 4 * s; v[s] = (valid && t < Q) ? M[k * size_t(Q) + t] : 0.0; }
}

// layout V: 16-byte loads when the records are 16-byte aligned (Q even)
template <int QS> __device__ __forceinline__ void wide_load_v(const double *__restrict__ M, size_t k, int Q, bool valid, double (&v)[QS]) {
    const int g = (threadIdx.x & 63) >> 4;
    if ((Q & 1) == 0) {
#pragma unroll
        for (int p = 0; p < QS / 2; ++p) {
            const int t = 8 * p + 2 * g;
            double2 x = make_double2(0.0, 0.0);
            if (valid && t < Q) x = *reinterpret_cast<const double2 *>(M + k * size_t(Q) + t);
            v[2 * p] = x.x;
            v[2 * p + 1] = x.y;
        }
    } else {
#pragma unroll
        for (int s = 0; s < QS; ++s) { const int t = wide_label_v(g, s); v[s] = (valid && t < Q) ? M[k * size_t(Q) + t] : 0.0; }
    }
}

// ------------------------------------------------------------------------------------------------
// One synchronous sweep (message-gather form), Q in 17 .. 64. Same equations, partial records and convergence bookkeeping as
// k_sweep: partials[b (Q+1) + q] = sum_rows g_i psi_i[q], slot Q = max |m_new - m_old| (1-step; the adaptive relaxation's
// probe: 2-step against the slot being overwritten).
// ------------------------------------------------------------------------------------------------
template <int QT>
__global__ void __launch_bounds__(WTPB) __attribute__((amdgpu_waves_per_eu(QT == 2 ? SBMBP_WIDE_WAVES2 : QT == 3 ? SBMBP_WIDE_WAVES3 : SBMBP_WIDE_WAVES4)))
k_wsweep(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rev, const double *__restrict__ Mold, double *__restrict__ Mnew,
         const double *__restrict__ psi_old, double *__restrict__ psi_new, const int32_t *__restrict__ clamp,
         const uint32_t *__restrict__ blk_row, const uint32_t *__restrict__ blk_e0, const dev_params *__restrict__ P,
         const dev_wide *__restrict__ Pw, int Q, int dc, double damp, double *__restrict__ partials) {
    constexpr int QP = 16 * QT, QS = 4 * QT;
    constexpr int QPP = QP + SBMBP_WIDE_PAD;  // LDS row stride: the 16 edges of a tile land 64 bytes apart modulo the banks
    __shared__ double sb[WCAP * QPP];     // edge fields of the segment / chunk
    __shared__ double sA[WRCAP * QPP];    // per row: log weights, then the normalised marginal
    __shared__ uint32_t srp[WRCAP + 1];
    __shared__ uint16_t srow[WCAP];
    __shared__ uint8_t sfl[WRCAP];
    __shared__ double smd[WTPB / 64];
    __shared__ double sle[WQ], shn[WQ];  // log eta, h / N
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, le_t = lane & 15, g = lane >> 4;
    const uint32_t bid = blockIdx.x;
    // everything that does not depend on other loads is requested before the first wait: a workgroup is a chain of dependent
    // round trips (segment -> rev -> message records), and each one it saves is a tenth of its lifetime
    const int stop = P->stop, probe2 = P->ar_probe2;
    const double damp_auto = P->damp_auto, beta = P->beta;
    const uint32_t r0 = blk_row[bid], r1 = blk_row[bid + 1], e0 = blk_e0[bid], e1 = blk_e0[bid + 1];
    if (tid < Q) { sle[tid] = Pw->logeta[tid]; shn[tid] = Pw->hN[tid]; }
    if (stop) return;
    damp *= damp_auto;
    const double *__restrict__ Atiles = Pw->tWv;
    auto lab = [&](int s) { return wide_label_v(g, s); };  // the label in register s of this lane
    const int nrows = int(r1 - r0), ne = int(e1 - e0);
    double md = 0.0;

    auto matvec = [&](const double (&v)[QS], double (&out)[QS]) {
#pragma unroll
        for (int rt = 0; rt < QT; ++rt) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < QS; ++s)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Atiles[(rt * QS + s) * 64 + lane], v[s], acc, 0, 0, 0);
            out[4 * rt + 0] = acc.x;
            out[4 * rt + 1] = acc.y;
            out[4 * rt + 2] = acc.z;
            out[4 * rt + 3] = acc.w;
        }
    };
    // edge fields of the 16 edges [base, base + 16) of this wave's tile: b = W^T m_in; mo = the edge's own old message
    auto tile_fields = [&](int base, double (&mo)[QS], double (&b)[QS], bool &valid, uint32_t &k) {
        const int le = base + le_t;
        valid = le < ne;
        k = e0 + uint32_t(valid ? le : 0);
        const uint32_t rk = (ne > 0) ? rev[k] : 0u;
        double mi[QS];
        wide_load_v<QS>(Mold, rk, Q, valid, mi);
        wide_load_v<QS>(Mold, k, Q, valid, mo);
        matvec(mi, b);
    };
    // new message of one edge from the row's normalised marginal A (LDS) and its field b; stores it, returns nothing
    auto cavity = [&](const double *Arow, const double (&mo)[QS], const double (&b)[QS], bool valid, uint32_t k, bool clamped) {
        double cav[QS];
#pragma unroll
        for (int s = 0; s < QS; ++s) {  // psi / b with a refined hardware reciprocal (QS divisions per lane were a tenth of the kernel's vector instructions)
            const int t = lab(s);
            double rb = __builtin_amdgcn_rcp(b[s]);
            rb = fma(fma(-b[s], rb, 1.0), rb, rb);
            rb = fma(fma(-b[s], rb, 1.0), rb, rb);
            cav[s] = (valid && t < Q) ? Arow[t] * rb : 0.0;
        }
        const double tot = edge_sum<QS>(cav);
        const double inv = 1.0 / tot;
        if (valid) {
            double outv[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                const int t = lab(s);
                outv[s] = mo[s];
                if (t < Q && !clamped) {
                    const double nv = cav[s] * inv;
                    outv[s] = damp * nv + (1.0 - damp) * mo[s];
                    const double ref = probe2 ? Mnew[size_t(k) * Q + t] : mo[s];
                    md = nanmax(md, probe2 ? fabs(ref - outv[s]) / damp : fabs(ref - nv));
                }
            }
            if ((Q & 1) == 0) {
#pragma unroll
                for (int p = 0; p < QS / 2; ++p) {
                    const int t = 8 * p + 2 * g;
                    if (t < Q) *reinterpret_cast<double2 *>(Mnew + size_t(k) * Q + t) = make_double2(outv[2 * p], outv[2 * p + 1]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < QS; ++s) { const int t = lab(s); if (t < Q) Mnew[size_t(k) * Q + t] = outv[s]; }
            }
        }
    };
    // row r of the segment: log weights in sA[r] -> normalised marginal in sA[r] and in psi_new (16 lanes per row: j = lane & 15)
    auto normalise_row = [&](int r, uint32_t row, int j) {
        double mx = -1.0e300;
        for (int q = j; q < Q; q += 16) mx = fmax(mx, sA[r * QPP + q]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 16));
        double sum = 0.0;
        for (int q = j; q < Q; q += 16) sum += exp(sA[r * QPP + q] - mx);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const double inv = 1.0 / sum;
        for (int q = j; q < Q; q += 16) {
            const double pv = exp(sA[r * QPP + q] - mx) * inv;
            sA[r * QPP + q] = pv;
            psi_new[size_t(row) * Q + q] = pv;
        }
    };

    if (ne <= WCAP) {
        // ---- a segment of whole rows
        for (int r = tid; r <= nrows; r += WTPB) srp[r] = row_ptr[r0 + r] - e0;
        double mo[QS], b[QS];
        bool valid;
        uint32_t k;
        tile_fields(wave * 16, mo, b, valid, k);
        if (valid) {
#pragma unroll
            for (int s = 0; s < QS; ++s) sb[(wave * 16 + le_t) * QPP + lab(s)] = b[s];
        }
        __syncthreads();
        for (int r = tid; r < nrows; r += WTPB) {
            for (int e = int(srp[r]); e < int(srp[r + 1]); ++e) srow[e] = uint16_t(r);
            sfl[r] = (clamp != nullptr && clamp[r0 + r] != -1) ? 1 : 0;
        }
        // row products, one (row, label) pair per thread and trip: log of prod_e b_e[q] eta_q F_i[q]
        for (int x = tid; x < nrows * QP; x += WTPB) {
            const int r = x / QP, q = x - r * QP;
            if (q >= Q) continue;
            const int es = int(srp[r]), ee = int(srp[r + 1]);
            double a = 1.0;
            int ex = 0;
            for (int e = es; e < ee; ++e) {
                a *= sb[e * QPP + q];
                if (((e - es) & 3) == 3) { int kx; a = frexp(a, &kx); ex += kx; }
            }
            const double fld = dc ? double(ee - es) : beta;
            sA[r * QPP + q] = log(a) + double(ex) * 0.6931471805599453 + sle[q] - fld * shn[q];
        }
        __syncthreads();
        {   // 16 lanes per row: 256 threads cover the WRCAP rows in one pass
            const int r = tid >> 4, j = tid & 15;
            if (r < nrows) {
                if (sfl[r]) {  // clamped row: marginal (and out-messages, below) stay as initialised (bp.cpp:1115-1124)
                    for (int q = j; q < Q; q += 16) {
                        const double pv = psi_old[size_t(r0 + r) * Q + q];
                        sA[r * QPP + q] = pv;
                        psi_new[size_t(r0 + r) * Q + q] = pv;
                    }
                } else {
                    normalise_row(r, r0 + r, j);
                }
            }
        }
        __syncthreads();
        if (tid < Q) {  // field sums of the segment, rows in order
            double S = 0.0;
            for (int r = 0; r < nrows; ++r) S += (dc ? double(srp[r + 1] - srp[r]) : 1.0) * sA[r * QPP + tid];
            partials[size_t(bid) * (Q + 1) + tid] = S;
        }
        const int rr = valid ? int(srow[wave * 16 + le_t]) : 0;
        cavity(&sA[rr * QPP], mo, b, valid, k, valid && sfl[rr] != 0);
    } else {
        // ---- one long row (nrows == 1), walked twice in chunks of WCAP edges
        const bool clamped = clamp != nullptr && clamp[r0] != -1;
        double a = 1.0;  // thread q < Q: running product of label q
        int ex = 0;
        if (!clamped) {
            for (int c0 = 0; c0 < ne; c0 += WCAP) {
                double mo[QS], b[QS];
                bool valid;
                uint32_t k;
                tile_fields(c0 + wave * 16, mo, b, valid, k);
                __syncthreads();  // the chunk before has been multiplied in
                if (valid) {
#pragma unroll
                    for (int s = 0; s < QS; ++s) sb[(wave * 16 + le_t) * QPP + lab(s)] = b[s];
                }
                __syncthreads();
                if (tid < Q) {
                    const int cnt = min(WCAP, ne - c0);
                    for (int e = 0; e < cnt; ++e) {
                        a *= sb[e * QPP + tid];
                        if ((e & 3) == 3) { int kx; a = frexp(a, &kx); ex += kx; }
                    }
                }
            }
            if (tid < Q) sA[tid] = log(a) + double(ex) * 0.6931471805599453 + sle[tid] - (dc ? double(ne) : beta) * shn[tid];
            __syncthreads();
            if (tid < 16) normalise_row(0, r0, tid);
        } else if (tid < 16) {
            for (int q = tid; q < Q; q += 16) {
                const double pv = psi_old[size_t(r0) * Q + q];
                sA[q] = pv;
                psi_new[size_t(r0) * Q + q] = pv;
            }
        }
        __syncthreads();
        if (tid < Q) partials[size_t(bid) * (Q + 1) + tid] = (dc ? double(ne) : 1.0) * sA[tid];
        for (int c0 = 0; c0 < ne; c0 += WCAP) {  // second walk: the fields again (not kept: a row may have any length), then the cavities
            double mo[QS], b[QS];
            bool valid;
            uint32_t k;
            tile_fields(c0 + wave * 16, mo, b, valid, k);
            cavity(sA, mo, b, valid, k, clamped);
        }
    }
    md = wave_nanmax(md);
    if (lane == 0) smd[wave] = md;
    __syncthreads();
    if (tid == 0) {
        double m = smd[0];
#pragma unroll
        for (int w = 1; w < WTPB / 64; ++w) m = nanmax(m, smd[w]);
        partials[size_t(bid) * (Q + 1) + Q] = m;
    }
}

// sum_i g_i psi_i[q] over row chunks (init_h, bp.cpp:320-332) for a run-time Q: partials[c (Q+1) + q], slot Q = 0
__global__ void __launch_bounds__(BLOCK)
k_wpsi_sum(const uint32_t *__restrict__ row_ptr, const double *__restrict__ psi, uint32_t n_rows, uint32_t rows_per_blk, int Q, int dc,
           double *__restrict__ partials) {
    __shared__ double sacc[BLOCK];
    const int tid = threadIdx.x, q = tid & 63, sub = tid >> 6;
    const uint32_t lo = blockIdx.x * rows_per_blk, hi = min(n_rows, lo + rows_per_blk);
    double acc = 0.0;
    if (q < Q)
        for (uint32_t i = lo + uint32_t(sub); i < hi; i += BLOCK / 64)
            acc += (dc ? double(row_ptr[i + 1] - row_ptr[i]) : 1.0) * psi[size_t(i) * Q + q];
    sacc[tid] = acc;
    __syncthreads();
    if (tid < Q) {
        double s = sacc[tid];
        for (int u = 1; u < BLOCK / 64; ++u) s += sacc[u * 64 + tid];
        partials[size_t(blockIdx.x) * (Q + 1) + tid] = s;
    }
    if (tid == 0) partials[size_t(blockIdx.x) * (Q + 1) + Q] = 0.0;
}

// K2 for a run-time Q (one workgroup): fold n_part records, relax the field sums, h / N, and - after a sweep - the convergence
// state machine with the adaptive relaxation (the same rules as finalize_update, kernels.h, written over the parameter block
// instead of registers: one lane, a handful of scalars per sweep). mode 0: after a sweep; 1: field initialisation; 2: exact
// field refresh.
__global__ void __launch_bounds__(BLOCK)
k_wfinalize(const double *__restrict__ partials, uint32_t n_part, int mode, dev_params *__restrict__ P, dev_wide *__restrict__ Pw, int Q,
            double *__restrict__ diff_hist, uint32_t hist_cap) {
    if (mode == 0 && P->stop) return;
    __shared__ double ssum[WQ + 1];
    __shared__ double sS[WQ];
    __shared__ double smix;
    __shared__ int shp;
    const int tid = threadIdx.x;
    if (tid == 0) shp = P->have_prev;
    if (tid <= Q) {  // column tid of the records, rows in order (the host folds long tables to a few hundred rows first)
        double s = 0.0;
        for (uint32_t r = 0; r < n_part; ++r) {
            const double v = partials[size_t(r) * (Q + 1) + tid];
            s = tid < Q ? s + v : nanmax(s, v);
        }
        ssum[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double mix = P->field_mix;
        if (mode == 0) {
            const double md = ssum[Q], crit = P->crit;
            const int it = P->sweep_idx, probe2 = P->ar_probe2, ar_on = P->ar_on;
            const int kind = probe2 ? 2 : 1;  // the wide path always runs the message-gather form
            int fl = P->ar_fl, gl = P->ar_gl, probing = P->ar_probing, stall = P->ar_stall, hold = P->ar_hold, holdS = P->ar_holdS,
                sigc = P->ar_sigc, nS = P->ar_nS, wn = P->ar_wn;
            const double base_mix = P->ar_base_mix;
            double v1 = P->ar_v1, v2 = P->ar_v2, wmin = P->ar_wmin, pmin = P->ar_pmin, d1p = P->ar_d1p;
            bool conv = false, esc = false;
            double rf = 0.0;  // the field gate of finalize_update
            if (shp && mix < 1.0)
                for (int q1 = 0; q1 < Q; ++q1) {
                    double acc = 0.0;
                    for (int q2 = 0; q2 < Q; ++q2) acc += Pw->cab[q2 * Q + q1] * (ssum[q2] - Pw->S[q2]);
                    rf = fmax(rf, fabs(acc) * P->invN * P->beta);
                }
            const bool field_ok = rf < crit;
            auto reset_after = [&]() {
                hold = 6; stall = 0; v1 = v2 = -1.0; probing = 0;
                wn = 0; wmin = 1e300; pmin = -1.0; nS = 0; sigc = 0; d1p = -1.0; holdS = 4;
            };
            auto cur_mix = [&]() { return fmin(fmin(base_mix, ar_field_cap(fl)), ar_gen_mix(gl)); };
            auto esc_gen = [&]() {
                const double m0 = cur_mix(), d0 = ar_gen_damp(gl);
                while (gl + 1 < AR_NG) {
                    ++gl;
                    if (cur_mix() < m0 || ar_gen_damp(gl) < d0) {
                        if (d0 == 1.0 && ar_gen_damp(gl) < 1.0) fl = 0;  // (finalize_update: the first damped level forgets the field level)
                        reset_after();
                        return;
                    }
                }
                hold = 1 << 30;
            };
            auto esc_field = [&]() {
                if (gl >= 0) { if (gl < 1) gl = 1; esc_gen(); return; }  // (finalize_update: a message-driven swing)
                int nf = fl;
                while (nf + 1 < AR_NF && !(ar_field_cap(nf) < cur_mix())) ++nf;
                if (ar_field_cap(nf) < cur_mix()) { fl = nf; reset_after(); }
                else esc_gen();
            };
            if (!ar_on) {
                conv = kind == 1 && md < crit && field_ok;
            } else if (probing) {
                probing = 0;
                if (kind == 1 && md < crit && field_ok) conv = true;
                else if (v1 >= 0.0) {
                    const double one = kind == 1 ? md : v1, two = kind == 1 ? v1 : md;
                    if (two < 0.5 * one) { if (gl < 1) gl = 1; esc_gen(); esc = true; }
                    else { hold = 8; stall = 0; }
                }
            } else {
                if (kind == 1 && md < crit && field_ok) conv = true;
                if (!conv) {
                    if (hold > 0) --hold;
                    else {
                        if (v2 >= 0.0 && md >= 0.98 * v2) ++stall; else stall = 0;
                        if (stall >= 4) { probing = 1; stall = 0; }
                    }
                    v2 = v1; v1 = md;
                    wmin = fmin(wmin, md); ++wn;
                    if (wn >= AR_WIN * (1 + (gl > 0 ? (gl < 3 ? gl : 3) : 0))) {
                        if (pmin >= 0.0 && wmin >= 0.9 * pmin && hold < (1 << 29)) { esc_gen(); esc = true; }
                        else { pmin = wmin; wmin = 1e300; wn = 0; }
                    }
                }
            }
            if (ar_on && !conv && !esc) {
                bool fe = false;
                if (holdS > 0) --holdS;
                else if (nS >= 2) {
                    double d1 = 0.0, d2 = 0.0, tot = 0.0;
                    for (int q = 0; q < Q; ++q) { d1 = fmax(d1, fabs(ssum[q] - Pw->arS1[q])); d2 = fmax(d2, fabs(ssum[q] - Pw->arS2[q])); tot += fabs(ssum[q]); }
                    const bool sig = d2 < 0.5 * d1 && d1 > 1e-9 * tot;
                    if (sig && d1 > 0.05 * tot && fl == 0) fe = true;
                    else if (sig && (d1p < 0.0 || d1 >= 0.98 * d1p)) { if (++sigc >= 6) fe = true; }
                    else sigc = 0;
                    d1p = d1;
                }
                for (int q = 0; q < Q; ++q) { Pw->arS2[q] = Pw->arS1[q]; Pw->arS1[q] = ssum[q]; }
                nS = nS < 2 ? nS + 1 : 2;
                if (fe) esc_field();
            }
            mix = ar_on ? cur_mix() : mix;
            const int k_next = probing ? 2 : 1;
            P->maxdiff = md;
            if (diff_hist != nullptr && uint32_t(it) < hist_cap) diff_hist[it] = md;
            P->last_exact = kind == 1 ? 1 : 0;
            P->exact = 1;
            P->ar_probe2 = k_next == 2 ? 1 : 0;
            P->field_mix = mix;
            P->damp_auto = ar_gen_damp(gl);
            P->ar_fl = fl; P->ar_gl = gl; P->ar_armed = 0; P->ar_probing = probing; P->ar_stall = stall; P->ar_hold = hold;
            P->ar_holdS = holdS; P->ar_sigc = sigc; P->ar_nS = nS; P->ar_wn = wn;
            P->ar_v1 = v1; P->ar_v2 = v2; P->ar_wmin = wmin; P->ar_pmin = pmin; P->ar_d1p = d1p;
            if (conv && P->conv_iter < 0) { P->conv_iter = it; P->stop = 1; }
            P->sweep_idx = it + 1;
        }
        smix = mix;
    }
    __syncthreads();
    const int have_prev = shp;
    if (tid < Q) {
        double s = ssum[tid];
        if (mode == 0 && have_prev && smix < 1.0) s = (1.0 - smix) * Pw->S[tid] + smix * s;
        sS[tid] = s;
    }
    __syncthreads();
    if (tid < Q) {
        double h = 0.0;
        for (int q2 = 0; q2 < Q; ++q2) h += Pw->cab[q2 * Q + tid] * sS[q2];  // h[q1] = sum_q2 cab[q2][q1] S[q2]   (bp.cpp:341-355)
        Pw->S[tid] = sS[tid];
        Pw->hN[tid] = h * P->invN;
    }
    if (tid == 0) P->have_prev = 1;
}

// every out-message of a row starts as the row's marginal (device initial state), full Q-component records
__global__ void __launch_bounds__(BLOCK)
k_winit_msgs_from_psi(const uint32_t *__restrict__ row_ptr, const double *__restrict__ psi, uint32_t n_rows, int Q,
                      double *__restrict__ M0, double *__restrict__ M1) {
    const uint32_t i = blockIdx.x;
    if (i >= n_rows) return;
    const uint32_t es = row_ptr[i], ee = row_ptr[i + 1];
    for (uint64_t x = threadIdx.x; x < uint64_t(ee - es) * Q; x += BLOCK) {
        const double v = psi[size_t(i) * Q + x % Q];
        M0[size_t(es) * Q + x] = v;
        M1[size_t(es) * Q + x] = v;
    }
}

// max |a - b| over n doubles (the reference's 1-step criterion on two full-record message buffers): partials[b * 2 + 1]
__global__ void __launch_bounds__(BLOCK)
k_wmsg_diff(const double *__restrict__ a, const double *__restrict__ b, uint64_t n, double *__restrict__ partials) {
    __shared__ double sred[4 * 2];
    double md = 0.0;
    for (uint64_t x = uint64_t(blockIdx.x) * BLOCK + threadIdx.x; x < n; x += uint64_t(gridDim.x) * BLOCK) md = nanmax(md, fabs(a[x] - b[x]));
    double dummy[1] = {0.0};
    block_reduce_store<1, 4>(dummy, md, sred, partials + size_t(blockIdx.x) * 2);
}

// na_expect[q] = sum_i psi_i[q], nna_expect[q] = sum_i d_i psi_i[q], confusion C[a][q] = sum_{i: true_i = a} psi_i[q] for a
// run-time Q: thread t owns the output entries t, t + 256, ... and walks the staged rows of its chunk in order (deterministic)
__global__ void __launch_bounds__(BLOCK)
k_wrow_sums(const uint32_t *__restrict__ row_ptr, const double *__restrict__ psi, const uint32_t *__restrict__ true_conf,
            uint32_t n_rows, uint32_t rows_per_blk, int Q, double *__restrict__ partials /* [grid][2Q + Q*Q] */) {
    constexpr int STAGE = 64;  // rows per stage
    __shared__ double sp[STAGE * WQ];
    __shared__ uint32_t sdeg[STAGE], scl[STAGE];
    const int T = 2 * Q + Q * Q, tid = threadIdx.x;
    constexpr int PER = (2 * WQ + WQ * WQ + BLOCK - 1) / BLOCK;
    double acc[PER];
    for (int j = 0; j < PER; ++j) acc[j] = 0.0;
    const uint32_t lo = blockIdx.x * rows_per_blk, hi = min(n_rows, lo + rows_per_blk);
    for (uint32_t base = lo; base < hi; base += STAGE) {
        const uint32_t cnt = min(uint32_t(STAGE), hi - base);
        __syncthreads();
        for (uint32_t x = tid; x < cnt * uint32_t(Q); x += BLOCK) sp[x] = psi[size_t(base) * Q + x];
        if (uint32_t(tid) < cnt) {
            sdeg[tid] = row_ptr[base + tid + 1] - row_ptr[base + tid];
            scl[tid] = true_conf ? true_conf[base + tid] : 0u;
        }
        __syncthreads();
        int j = 0;
        for (int t = tid; t < T; t += BLOCK, ++j) {
            double s = 0.0;
            if (t < Q) { for (uint32_t r = 0; r < cnt; ++r) s += sp[r * Q + t]; }
            else if (t < 2 * Q) { for (uint32_t r = 0; r < cnt; ++r) s += double(sdeg[r]) * sp[r * Q + (t - Q)]; }
            else {
                const uint32_t a = uint32_t(t - 2 * Q) / uint32_t(Q), b = uint32_t(t - 2 * Q) % uint32_t(Q);
                for (uint32_t r = 0; r < cnt; ++r) s += (scl[r] == a) ? sp[r * Q + b] : 0.0;
            }
            acc[j] += s;
        }
    }
    int j = 0;
    for (int t = tid; t < T; t += BLOCK, ++j) partials[size_t(blockIdx.x) * T + t] = acc[j];
}

// ------------------------------------------------------------------------------------------------
// The reductions of -m infer for Q in 17 .. 64, one pass (message-gather: the incoming message is M[rev[k]]):
//   [0] sum_i log Z_i   [1] sum_e log(m_in^T W m_out)   [2] e_site sum   [3] e_edge sum          (k_fe_frame / k_fe_hub)
//   [4] adjacent pairs of the non-edge term   [5] ... of its entropy part                       (k_nonedge_*_adj; adj_mode != 0)
// Edge terms through the matrix cores: m_in^T W m_out = (W^T m_in) . m_out, and likewise with cab and cab log cab. A long row
// is walked once (its per-label products carried in registers of the first Q threads).
// ------------------------------------------------------------------------------------------------
constexpr int WR_NP = FE_NP + NE_NP;
// ENT: the entropy terms too (-m infer, compute_entropy). Without them (the free energy of an EM step) a third of the LDS and two of
// the three matrix products per tile are not there: two workgroups per CU at Q = 64 instead of one.
template <int QT, bool ENT>
__global__ void __launch_bounds__(WTPB) __attribute__((amdgpu_waves_per_eu((!ENT && QT == 4) ? 2 : 1)))
k_wreduce(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rev, const uint32_t *__restrict__ nbr,
          const double *__restrict__ M, const double *__restrict__ psi, const uint32_t *__restrict__ blk_row,
          const uint32_t *__restrict__ blk_e0, const dev_params *__restrict__ P, const dev_wide *__restrict__ Pw, int Q, int dc,
          int adj_mode /* 0 none, 1 series weights, 2 exact */, const double *__restrict__ wmat, double *__restrict__ partials) {
    constexpr int QP = 16 * QT, QS = 4 * QT;
    __shared__ double sb[WCAP * QP];
    __shared__ double sc[ENT ? WCAP * QP : 1];
    __shared__ double sV[(ENT ? 3 : 1) * WRCAP * QP];  // per row: w^T psi_i, cab^T psi_i, (cab log cab)^T psi_i
    __shared__ double sL[(ENT ? 2 : 1) * WRCAP * QP];  // per row: log weights of the site term / of the entropy site term
    __shared__ uint32_t srp[WRCAP + 1];
    __shared__ uint16_t srow[WCAP];
    __shared__ double sred[(WTPB / 64) * (WR_NP + 1)];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, le_t = lane & 15, g = lane >> 4;
    const uint32_t bid = blockIdx.x;
    const uint32_t r0 = blk_row[bid], r1 = blk_row[bid + 1], e0 = blk_e0[bid];
    const int nrows = int(r1 - r0), ne = int(blk_e0[bid + 1] - e0);
    const double invN = P->invN, beta = P->beta;
    double acc[WR_NP];
#pragma unroll
    for (int x = 0; x < WR_NP; ++x) acc[x] = 0.0;

    // per row r (of nr): the three vectors the adjacent pairs need, (row, label) pairs over the threads
    auto row_vectors = [&](int nr) {
        for (int x = tid; x < nr * QP; x += WTPB) {
            const int r = x / QP, q = x - r * QP;
            double v = 0.0, vc = 0.0, vu = 0.0;
            if (q < Q) {
                const double *pi = psi + size_t(r0 + r) * Q;
                for (int a = 0; a < Q; ++a) {
                    const double p = pi[a];
                    v += wmat[a * Q + q] * p;
                    if (ENT) { vc += Pw->cab[a * Q + q] * p; vu += Pw->cab[a * Q + q] * Pw->logcab[a * Q + q] * p; }
                }
            }
            sV[r * QP + q] = v;
            if (ENT) {
                sV[(ENT ? WRCAP + r : 0) * QP + q] = vc;
                sV[(ENT ? 2 * WRCAP + r : 0) * QP + q] = vu;
            }
        }
    };
    // the 16 edges [base, base + 16) of this wave: fields to LDS slot `slot0 + le_t`, edge and adjacent-pair terms to acc
    auto tile_terms = [&](int base, int slot0, int row_of_tile /* -1: from srow */) {
        const int le = base + le_t;
        const bool valid = le < ne;
        const uint32_t k = e0 + uint32_t(valid ? le : 0);
        const uint32_t rk = (ne > 0) ? rev[k] : 0u;
        double mi[QS], mo[QS], b[QS], c[QS], cl[QS];
        wide_load<QS>(M, rk, Q, valid, mi);
        wide_load<QS>(M, k, Q, valid, mo);
        wide_matvec<QT>(Pw->tW, mi, b);
        double t1[QS];
#pragma unroll
        for (int s = 0; s < QS; ++s) t1[s] = b[s] * mo[s];
        const double nl = edge_sum<QS>(t1);
        if (valid && g == 0) acc[1] += log(nl);
        if (ENT) {  // uniform
            wide_matvec<QT>(Pw->tC, mi, c);
            wide_matvec<QT>(Pw->tCL, mi, cl);
#pragma unroll
            for (int s = 0; s < QS; ++s) t1[s] = c[s] * mo[s];
            const double den = edge_sum<QS>(t1);
#pragma unroll
            for (int s = 0; s < QS; ++s) t1[s] = cl[s] * mo[s];
            const double num = edge_sum<QS>(t1);
            if (valid && g == 0) acc[3] += num / den;
        }
        if (valid) {
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                sb[(slot0 + le_t) * QP + g + 4 * s] = b[s];
                if (ENT) sc[ENT ? (slot0 + le_t) * QP + g + 4 * s : 0] = c[s];
            }
        }
        if (adj_mode) {  // uniform
            const int r = row_of_tile >= 0 ? row_of_tile : (valid ? int(srow[le]) : 0);
            const uint32_t l = (ne > 0) ? nbr[k] : 0u;
            double pl[QS];
            wide_load<QS>(psi, l, Q, valid, pl);
#pragma unroll
            for (int s = 0; s < QS; ++s) { const int t = g + 4 * s; t1[s] = t < Q ? sV[r * QP + t] * pl[s] : 0.0; }
            const double y = edge_sum<QS>(t1);
            double yc = 0.0, u = 0.0;
            if (ENT) {
#pragma unroll
                for (int s = 0; s < QS; ++s) { const int t = g + 4 * s; t1[s] = t < Q ? sV[(ENT ? WRCAP + r : 0) * QP + t] * pl[s] : 0.0; }
                yc = edge_sum<QS>(t1);
#pragma unroll
                for (int s = 0; s < QS; ++s) { const int t = g + 4 * s; t1[s] = t < Q ? sV[(ENT ? 2 * WRCAP + r : 0) * QP + t] * pl[s] : 0.0; }
                u = edge_sum<QS>(t1);
            }
            if (valid && g == 0) {
                const double num = u * invN, den = 1.0 - yc * invN;
                if (adj_mode == 1) {
                    acc[FE_NP] += log1p(-y * invN);
                    if (ENT) acc[FE_NP + 1] += num / den;
                } else {
                    if (y != 0.0) acc[FE_NP] += log(y);
                    if (ENT && num * den != 0.0) acc[FE_NP + 1] += num / den;
                }
            }
        }
    };
    // site terms of row r from its log weights in sL (16 lanes per row; lane j == 0 adds to acc)
    auto site_terms = [&](int r, double di, int j) {
        double mx = -1.0e300;
        for (int q = j; q < Q; q += 16) mx = fmax(mx, sL[r * QP + q]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 16));
        double sum = 0.0;
        for (int q = j; q < Q; q += 16) sum += exp(sL[r * QP + q] - mx);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        if (j == 0) acc[0] += mx + log(sum);  // log Z_i  (bp.cpp:446-502)
        if (ENT) {  // e_site (bp.cpp:506-560): sum_q w_q (-h_q/N) / sum_q w_q
            double me = -1.0e300;
            for (int q = j; q < Q; q += 16) me = fmax(me, sL[(ENT ? WRCAP + r : 0) * QP + q]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) me = fmax(me, __shfl_xor(me, o, 16));
            double num = 0.0, den = 0.0;
            for (int q = j; q < Q; q += 16) { const double w = exp(sL[(ENT ? WRCAP + r : 0) * QP + q] - me); den += w; num += w * (-Pw->hN[q]); }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { num += __shfl_xor(num, o, 16); den += __shfl_xor(den, o, 16); }
            if (j == 0) acc[2] += num / den;
        }
        (void)di;
    };

    if (ne <= WCAP) {
        for (int r = tid; r <= nrows; r += WTPB) srp[r] = row_ptr[r0 + r] - e0;
        __syncthreads();
        for (int r = tid; r < nrows; r += WTPB)
            for (int e = int(srp[r]); e < int(srp[r + 1]); ++e) srow[e] = uint16_t(r);
        if (adj_mode) row_vectors(nrows);
        __syncthreads();
        tile_terms(wave * 16, wave * 16, -1);
        __syncthreads();
        for (int x = tid; x < nrows * QP; x += WTPB) {
            const int r = x / QP, q = x - r * QP;
            if (q >= Q) continue;
            const int es = int(srp[r]), ee = int(srp[r + 1]);
            double a = 1.0, cc = 1.0;
            int ex = 0, cx = 0;
            for (int e = es; e < ee; ++e) {
                a *= sb[e * QP + q];
                if (ENT) cc *= sc[ENT ? e * QP + q : 0];
                if (((e - es) & 3) == 3) { int kx; a = frexp(a, &kx); ex += kx; if (ENT) { cc = frexp(cc, &kx); cx += kx; } }
            }
            const double fld = dc ? double(ee - es) : beta;
            sL[r * QP + q] = log(a) + double(ex) * 0.6931471805599453 + Pw->logeta[q] - fld * Pw->hN[q];
            if (ENT) sL[(ENT ? WRCAP + r : 0) * QP + q] = log(cc) + double(cx) * 0.6931471805599453 + Pw->logeta[q] - Pw->hN[q];
        }
        __syncthreads();
        {
            const int r = tid >> 4, j = tid & 15;
            if (r < nrows) site_terms(r, double(srp[r + 1] - srp[r]), j);
        }
    } else {  // one long row
        if (adj_mode) row_vectors(1);
        double a = 1.0, cc = 1.0;
        int ex = 0, cx = 0;
        for (int c0 = 0; c0 < ne; c0 += WCAP) {
            __syncthreads();  // sV ready (first trip) / the chunk before has been multiplied in
            tile_terms(c0 + wave * 16, wave * 16, 0);
            __syncthreads();
            if (tid < Q) {
                const int cnt = min(WCAP, ne - c0);
                for (int e = 0; e < cnt; ++e) {
                    a *= sb[e * QP + tid];
                    if (ENT) cc *= sc[ENT ? e * QP + tid : 0];
                    if ((e & 3) == 3) { int kx; a = frexp(a, &kx); ex += kx; if (ENT) { cc = frexp(cc, &kx); cx += kx; } }
                }
            }
        }
        if (tid < Q) {
            sL[tid] = log(a) + double(ex) * 0.6931471805599453 + Pw->logeta[tid] - (dc ? double(ne) : beta) * Pw->hN[tid];
            if (ENT) sL[(ENT ? WRCAP : 0) * QP + tid] = log(cc) + double(cx) * 0.6931471805599453 + Pw->logeta[tid] - Pw->hN[tid];
        }
        __syncthreads();
        if (tid < 16) site_terms(0, double(ne), tid);
    }
    block_reduce_store<WR_NP, WTPB / 64>(acc, 0.0, sred, partials + size_t(bid) * (WR_NP + 1));
}

// all ordered pairs (i, l) of the exact non-edge term (bp.cpp:675-741) for a run-time Q, tiles of 64 x 64 vertices:
// v_i = P^T psi_i (and the two entropy vectors) staged in LDS, a pair then costs Q FMAs. partials[(by gx + bx) (NE_NP + 1)].
__global__ void __launch_bounds__(BLOCK)
k_wnonedge_exact(const double *__restrict__ psi, uint32_t n, int Q, const double *__restrict__ Pmat, const double *__restrict__ cab,
                 double invN, int want_entropy, double *__restrict__ partials) {
    constexpr int TS = 64;
    __shared__ double sv[3 * TS * WQ];
    __shared__ double sl[TS * WQ];
    __shared__ double sred[4 * (NE_NP + 1)];
    const int tid = threadIdx.x;
    const uint32_t i0 = blockIdx.x * TS, l0 = blockIdx.y * TS;
    const uint32_t ni = min(uint32_t(TS), n - i0), nl = min(uint32_t(TS), n - l0);
    for (uint32_t x = tid; x < nl * uint32_t(Q); x += BLOCK) sl[x] = psi[size_t(l0) * Q + x];
    for (uint32_t x = tid; x < ni * uint32_t(Q); x += BLOCK) {
        const uint32_t r = x / uint32_t(Q), q2 = x % uint32_t(Q);
        const double *pi = psi + size_t(i0 + r) * Q;
        double a = 0.0, an = 0.0, ad = 0.0;
        for (int q1 = 0; q1 < Q; ++q1) {
            a += Pmat[q1 * Q + q2] * pi[q1];
            if (want_entropy) {
                const double c = cab[q1 * Q + q2];
                an += (c * invN) * log(c) * pi[q1];
                ad += (1.0 - c * invN) * pi[q1];
            }
        }
        sv[r * Q + q2] = a;
        sv[(TS + r) * Q + q2] = an;
        sv[(2 * TS + r) * Q + q2] = ad;
    }
    __syncthreads();
    double acc[NE_NP] = {0.0, 0.0};
    for (uint32_t pidx = tid; pidx < ni * nl; pidx += BLOCK) {
        const uint32_t r = pidx / nl, c = pidx % nl;
        double f = 0.0, num = 0.0, den = 0.0;
        for (int q = 0; q < Q; ++q) {
            const double pl = sl[c * Q + q];
            f += sv[r * Q + q] * pl;
            if (want_entropy) { num += sv[(TS + r) * Q + q] * pl; den += sv[(2 * TS + r) * Q + q] * pl; }
        }
        if (f != 0.0) acc[0] += log(f);
        if (want_entropy && num * den != 0.0) acc[1] += num / den;
    }
    block_reduce_store<NE_NP>(acc, 0.0, sred, partials + (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * (NE_NP + 1));
}

// ------------------------------------------------------------------------------------------------
// EM numerators for Q in 17 .. 64 (compute_cab_expect, bp.cpp:892-989): G[a][b] = sum_e m_in,e[a] m_out,e[b] / (2 norm_e) with
// norm_e = m_in^T cab m_out; the host forms cab_expect[a][b] = cab[a][b] (G[a][b] + G[b][a]) (a != b), cab[a][a] G[a][a].
// Purely per edge, so the edges are taken as a flat list, 64 per trip (one tile of 16 per wave). G is a labels x labels
// matrix product over the EDGES: the trip's x = m_in and y = m_out / (2 norm) go to LDS as [edge][label] and every wave
// accumulates its share of the 16 x 16 output tiles with v_mfma_f64_16x16x4_f64 (A: x^T, lane l holds x[edge 4 step + (l >> 4)]
// [label 16 at + (l & 15)]; B: y likewise) over the 16 steps of a trip. A workgroup walks trips b, b + grid, ... and writes its
// Q x Q sums once: partials[b Q Q + a Q + b'].
// ------------------------------------------------------------------------------------------------
template <int QT>
__global__ void __launch_bounds__(WTPB)
k_wem(const uint32_t *__restrict__ rev, const double *__restrict__ M, uint64_t n_edges, const dev_wide *__restrict__ Pw, int Q,
      double *__restrict__ partials) {
    constexpr int QP = 16 * QT, QS = 4 * QT, NT = QT * QT, TPW = (NT + 3) / 4;  // output tiles, tiles per wave
    __shared__ double sx[WCAP * QP];
    __shared__ double sy[WCAP * QP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, le_t = lane & 15, g = lane >> 4;
    d4 acc[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
    const uint64_t n_trips = (n_edges + WCAP - 1) / WCAP;
    for (uint64_t trip = blockIdx.x; trip < n_trips; trip += gridDim.x) {
        const uint64_t k = trip * WCAP + uint64_t(wave * 16 + le_t);
        const bool valid = k < n_edges;
        const uint64_t kk = valid ? k : 0;
        const uint32_t rk = rev[kk];
        double mi[QS], mo[QS], c[QS], t1[QS];
        wide_load<QS>(M, rk, Q, valid, mi);
        wide_load<QS>(M, kk, Q, valid, mo);
        wide_matvec<QT>(Pw->tC, mi, c);  // cab^T m_in
#pragma unroll
        for (int s = 0; s < QS; ++s) t1[s] = c[s] * mo[s];
        const double norm = edge_sum<QS>(t1);
        const double hin = valid ? 0.5 / norm : 0.0;
        __syncthreads();  // the trip before has been consumed
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            sx[(wave * 16 + le_t) * QP + g + 4 * s] = mi[s];
            sy[(wave * 16 + le_t) * QP + g + 4 * s] = mo[s] * hin;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tile = wave + 4 * u;  // wave-uniform
            if (tile < NT) {
                const int at = tile / QT, bt = tile - at * QT;
#pragma unroll
                for (int step = 0; step < WCAP / 4; ++step) {
                    const double a = sx[(4 * step + g) * QP + 16 * at + le_t];
                    const double b = sy[(4 * step + g) * QP + 16 * bt + le_t];
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
                }
            }
        }
    }
    // C/D layout: lane l, register r: row (l >> 4) + 4 r (label a = 16 at + row), column l & 15 (label b = 16 bt + column)
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tile = wave + 4 * u;
        if (tile < NT) {
            const int at = tile / QT, bt = tile - at * QT;
            const double v[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * at + g + 4 * r, b = 16 * bt + le_t;
                if (a < Q && b < Q) partials[size_t(blockIdx.x) * Q * Q + size_t(a) * Q + b] = v[r];
            }
        }
    }
}

}  // namespace sbmbp
#endif