// HIP kernels of the BP engine, written for gfx950 (CDNA4: wave64, 256 CUs, 160 KiB LDS/CU).
//
// Work decomposition of every edge-parallel kernel ("frame"): the host cuts the CSR row sequence
// into workgroup segments [blk_row[b], blk_row[b+1]) holding at most CAP directed edges and RCAP
// rows. Inside a workgroup, phases alternate between one lane per directed edge (coalesced
// streams of rev/own messages, random gather of the incoming message, Q*Q FMAs) and one lane
// per row (product over the row's edges out of LDS). Rows with more than CAP edges ("hubs") are
// handled by a workgroup-per-row kernel. All reductions are fixed-order (block partials + one
// finalize workgroup): results are bitwise reproducible run to run.
//
// Equations: SURVEY.md Appendix A (restating belief_propagation.cpp:991-1071 for the update,
// :442-504/:562-612/:675-709 for the free energy, :892-989 for the EM expectations).
#ifndef SBMBP_KERNELS_H
#define SBMBP_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbmbp {

constexpr int BLOCK = 256;
constexpr int QMAX = 16;     // label counts of the lane-per-edge kernels and of dev_params
constexpr int QMAX_RT = 64;  // ... of the run-time-Q helper kernels (initial state, record conversion): also above 16

// Parameter/state block in HBM, read through scalar loads by every workgroup and rewritten by
// k_finalize after each sweep (arrays packed with stride Q).
struct dev_params {
    double W[QMAX * QMAX];     // sweep weights: cab^beta (dc 0), cab (dc 1), pab = cab/N (dc 2)
    double cab[QMAX * QMAX];
    double logcab[QMAX * QMAX];  // log cab (host): the entropy edge term weights cab log cab, not to be recomputed per edge
    double eta[QMAX];
    double logeta[QMAX];
    double hN[QMAX];           // h[q] / N
    double etaF[QMAX];         // eta[q] * exp(-beta h[q]/N)   (dc 0 field factor)
    double S[QMAX];            // sum_i g_i psi_i[q] after relaxation (h = cab^T S)
    double beta, invN, field_mix, crit;
    double prev_hint;          // 2-step hint of the sweep before (0: none): its ratio to the current one estimates the rate
    double maxdiff;            // of the last executed sweep
    int conv_iter;             // -1 until the first sweep with maxdiff < crit
    int sweep_idx;             // sweeps executed in the current converge call
    int stop;                  // 1: queued sweeps that follow are no-ops
    int have_prev;             // S holds a previous value (field relaxation)
    // Convergence test of the marginal-gather sweeps. Their in-kernel difference is the 2-step value |m^{t+1} - m^{t-1}|
    // (a hint: the kernel only holds its own previous message). hinted = 1 declares that; once the hint falls below
    // HINT_SCALE * crit, k_finalize sets exact = 1 and from the next sweep on the kernels read the other message buffer
    // too and report the reference's 1-step difference (bp.cpp:1059-1063), on which the stop flag then works.
    int hinted;
    int exact;                 // EFFECTIVE flag of the next marginal-gather sweep: report the 1-step difference (armed XOR probing)
    int last_exact;            // maxdiff of the last executed sweep is a 1-step difference (not a hint)
    int pause;                 // with stop: the run must continue in the message-gather form (adaptive relaxation asked for damping)
    // ---- adaptive relaxation (CPU twin: oracle/bp_oracle.cpp ar_t). Plain Jacobi sweeps can oscillate where
    // the reference's random-sequential sweep converges (it keeps h_ current inside a sweep, bp.cpp:1088-1095, and never
    // updates two neighbours at once). finalize_update watches three signatures and relaxes the schedule when one shows; the
    // fixed points do not move: (F) the raw field sums swing with period 2 -> lower field_mix; (P) the messages swing with
    // period 2 (2-step difference far below the 1-step difference: checked by ONE probe sweep of the other kind after four
    // sweeps without progress) or (W) a whole window of sweeps brought no progress -> next level of (field_mix, damping).
    int ar_fl, ar_gl;          // field level (caps 1, 0.25, 0.1, 0.05), generic level (-1 = none; (mix cap, damping) ladder)
    int ar_on;                 // host: enabled for this run (converge calls; never for fixed sweep counts)
    int ar_psi_ok;             // host: the run is in the marginal-gather form as long as damping stays 1
    int ar_armed;              // the hints have armed the exact criterion
    int ar_probing;            // the next sweep reports the OTHER kind of difference
    int ar_probe2;             // message-gather sweeps: report the 2-step difference (k_sweep reads the slot it overwrites)
    int ar_stall, ar_hold, ar_holdS, ar_sigc, ar_nS, ar_wn, ar_pad;
    double ar_base_mix;        // the caller's field_mix
    double damp_auto;          // factor on the caller's damping (message-gather sweeps)
    double ar_v1, ar_v2;       // reported differences of the last two regular sweeps (-1: none)
    double ar_wmin, ar_pmin;   // window minimum so far, minimum of the window before (-1: none)
    double ar_d1p;             // last swing of the field sums (-1: none)
    double ar_S1[QMAX], ar_S2[QMAX];  // raw field sums of the last two sweeps
    // Degree-corrected field factor of the rows of up to FT_D edges: ftab[d][q] = eta[q] exp(-d (h[q] - min h)/N), rewritten with h
    // by every finalize launch. A row then loads its Q factors (issued before its product loop) instead of evaluating Q
    // exponentials (4 % of the sweep at Q = 8, dc 1); the entries are the very expression apply_field evaluates, bit for bit.
    double ftab[(32 + 1) * QMAX];
};
constexpr int FT_D = 32;
// With linear convergence at rate r the 2-step difference of sweep t is (1 + 1/r) times its 1-step difference d_t, and the
// sweep BEFORE the first one with d_t < crit has a hint below crit (1 + 1/r) / r. The exact criterion must be armed by then, so
// it is armed at scale * crit with scale = 1.5 (1 + 1/r) / r from the measured ratio of consecutive hints, kept within
// [HINT_SCALE, HINT_SCALE_MAX]: slowly converging runs (r >= 0.5: the large graphs) pay the extra read for 3-4 sweeps only,
// fast ones arm earlier, and the returned niter equals the message-gather form's unless r < 0.15.
constexpr double HINT_SCALE = 8.0, HINT_SCALE_MAX = 64.0;

// tunables (A/B'd on MI355X, see DESIGN.md "Tuning log")
#ifndef SBMBP_EPT_LO
#define SBMBP_EPT_LO 2  // directed edges per lane, Q <= 2 (A/B on C2: 2 beats 4 and 1 by ~9 %)
#endif
#ifndef SBMBP_EPT_MID
#define SBMBP_EPT_MID 2  // Q = 3, 4
#endif
#ifndef SBMBP_EPT_HI
#define SBMBP_EPT_HI 2  // Q >= 5
#endif
#ifndef SBMBP_NT
#define SBMBP_NT 0  // bit 0: non-temporal loads, bit 1: non-temporal stores on the coalesced message streams
#endif
#ifndef SBMBP_PSI_WAVES
#define SBMBP_PSI_WAVES 0  // > 0: min waves per SIMD requested for k_sweep_psi (register cap)
#endif
#ifndef SBMBP_XCD_REMAP
#define SBMBP_XCD_REMAP 1  // contiguous eighth of the segments per XCD (A/B on MI355X: C3 2.551 -> 2.504 ms, C5 0.1258 -> 0.1226 ms)
#endif
#ifndef SBMBP_FRAME_TPB
#define SBMBP_FRAME_TPB 256  // threads per workgroup of the frame kernels (multiple of 64)
#endif
#ifndef SBMBP_FRAME_TPB_HI
#define SBMBP_FRAME_TPB_HI 128  // ... above Q = 4 (A/B on C4, Q = 8: 0.492 -> 0.472 ms; neutral on the Q <= 4 workloads, which keep 256)
#endif
constexpr int FTPB = SBMBP_FRAME_TPB;
// launch grid of the marginal-gather sweep for n segments (padded for the XCD mapping)
inline uint32_t xcd_grid(uint32_t n) { return SBMBP_XCD_REMAP ? 8u * ((n + 7u) / 8u) : n; }
template <int Q> struct frame_cfg {
    static constexpr int TPB = (Q <= 4) ? FTPB : SBMBP_FRAME_TPB_HI;  // threads per workgroup
    static constexpr int WAVES = TPB / 64;
    static constexpr int EPT = (Q <= 2) ? SBMBP_EPT_LO : (Q <= 4 ? SBMBP_EPT_MID : (Q <= 8 ? SBMBP_EPT_HI : 1));  // directed edges per lane
    static constexpr int CAP = TPB * EPT;                  // edges per workgroup segment
    static constexpr int RCAP = (CAP / 2 > 64) ? CAP / 2 : 64;  // rows per workgroup segment
};

template <int Q> __device__ __forceinline__ void load_vec(const double *__restrict__ p, double (&v)[Q]) {
    if (Q % 2 == 0) {
        const double2 *p2 = reinterpret_cast<const double2 *>(p);
#pragma unroll
        for (int j = 0; j < Q / 2; ++j) { double2 t = p2[j]; v[2 * j] = t.x; v[2 * j + 1] = t.y; }
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = p[q];
    }
}
template <int Q> __device__ __forceinline__ void store_vec(double *__restrict__ p, const double (&v)[Q]) {
    if (Q % 2 == 0) {
        double2 *p2 = reinterpret_cast<double2 *>(p);
#pragma unroll
        for (int j = 0; j < Q / 2; ++j) p2[j] = make_double2(v[2 * j], v[2 * j + 1]);
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) p[q] = v[q];
    }
}

// streaming (touched once per sweep) variants of the vector load/store: non-temporal, so the
// message streams do not evict the gathered tables from L2 / Infinity Cache
typedef double v2d __attribute__((ext_vector_type(2)));
template <int Q> __device__ __forceinline__ void load_vec_stream(const double *__restrict__ p, double (&v)[Q]) {
#if SBMBP_NT & 1
    if (Q % 2 == 0) {
        const v2d *p2 = reinterpret_cast<const v2d *>(p);
#pragma unroll
        for (int j = 0; j < Q / 2; ++j) { v2d t = __builtin_nontemporal_load(p2 + j); v[2 * j] = t.x; v[2 * j + 1] = t.y; }
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = __builtin_nontemporal_load(p + q);
    }
#else
    load_vec<Q>(p, v);
#endif
}
template <int Q> __device__ __forceinline__ void store_vec_stream(double *__restrict__ p, const double (&v)[Q]) {
#if SBMBP_NT & 2
    if (Q % 2 == 0) {
        v2d *p2 = reinterpret_cast<v2d *>(p);
#pragma unroll
        for (int j = 0; j < Q / 2; ++j) { v2d t; t.x = v[2 * j]; t.y = v[2 * j + 1]; __builtin_nontemporal_store(t, p2 + j); }
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) __builtin_nontemporal_store(v[q], p + q);
    }
#else
    store_vec<Q>(p, v);
#endif
}
// ---- message records ------------------------------------------------------------------------------
// A message is a distribution over Q groups, so only MC = Q-1 components are kept in HBM (record k at
// M + k*MC): 8 bytes less on every message read and write of the sweep. The component left out is the
// LARGEST one, restored on load as max(0, 1 - sum of the others) (a NaN stays a NaN): the small components keep
// their full relative precision - with a forbidden group pair (a zero in cab) a component of 1e-20 still decides
// between groups, and leaving out a fixed component would round it to an exact zero as soon as the others reach
// 1 - 1e-16. Which component was left out is written into the (otherwise unused) sign bits of the first NB stored
// components. The marginal table keeps all Q components: its rows are GATHERED, and a 32-byte row aligned to 32
// bytes costs one fabric request where a 24-byte row would straddle two in a quarter of the cases.
template <int Q> struct msg_rec {
    static constexpr int MC = Q - 1;
    static constexpr int NB = (Q <= 2) ? 1 : (Q <= 4) ? 2 : (Q <= 8) ? 3 : 4;  // bits of the left-out index (NB <= MC)
};
__device__ __forceinline__ unsigned sign_bit(double x) { return unsigned(static_cast<unsigned long long>(__double_as_longlong(x)) >> 63); }
__device__ __forceinline__ double with_sign(double x, unsigned bit) {
    const unsigned long long u = static_cast<unsigned long long>(__double_as_longlong(x)) & 0x7fffffffffffffffull;
    return __longlong_as_double(static_cast<long long>(u | (static_cast<unsigned long long>(bit & 1u) << 63)));
}
// w[0..MC): the stored words -> v[0..Q)
template <int Q> __device__ __forceinline__ void decode_msg(const double (&w)[Q > 1 ? Q - 1 : 1], double (&v)[Q]) {
    constexpr int MC = msg_rec<Q>::MC, NB = msg_rec<Q>::NB;
    unsigned idx = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) idx |= sign_bit(w[j]) << j;
    idx = idx < unsigned(Q) ? idx : unsigned(Q - 1);  // garbage (NaN records) stays in range
    double a[MC];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < MC; ++j) { a[j] = fabs(w[j]); s += a[j]; }
    double big = 1.0 - s;
    big = big < 0.0 ? 0.0 : big;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const double lo = a[k < MC ? k : MC - 1];      // used when k < idx
        const double hi = a[k > 0 ? k - 1 : 0];        // used when k > idx
        v[k] = unsigned(k) < idx ? lo : (unsigned(k) == idx ? big : hi);
    }
}
template <int Q> __device__ __forceinline__ void encode_msg(const double (&v)[Q], double (&w)[Q > 1 ? Q - 1 : 1]) {
    constexpr int MC = msg_rec<Q>::MC, NB = msg_rec<Q>::NB;
    unsigned idx = 0;
    double vmax = v[0];
#pragma unroll
    for (int k = 1; k < Q; ++k) if (v[k] > vmax) { vmax = v[k]; idx = unsigned(k); }  // first maximum; NaNs compare false
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        double o = unsigned(j) < idx ? v[j] : v[j + 1];
        if (j < NB) o = with_sign(o, idx >> j);
        w[j] = o;
    }
}
template <int Q> __device__ __forceinline__ void load_msg(const double *__restrict__ M, size_t k, double (&v)[Q]) {
    constexpr int MC = msg_rec<Q>::MC;
    const double *p = M + k * MC;
    double w[MC];
    if (MC % 2 == 0) {
        const double2 *p2 = reinterpret_cast<const double2 *>(p);
#pragma unroll
        for (int j = 0; j < MC / 2; ++j) { double2 t = p2[j]; w[2 * j] = t.x; w[2 * j + 1] = t.y; }
    } else {
#pragma unroll
        for (int q = 0; q < MC; ++q) w[q] = p[q];
    }
    decode_msg<Q>(w, v);
}
template <int Q> __device__ __forceinline__ void store_msg(double *__restrict__ M, size_t k, const double (&v)[Q]) {
    constexpr int MC = msg_rec<Q>::MC;
    double *p = M + k * MC;
    double w[MC];
    encode_msg<Q>(v, w);
    if (MC % 2 == 0) {
        double2 *p2 = reinterpret_cast<double2 *>(p);
#pragma unroll
        for (int j = 0; j < MC / 2; ++j) p2[j] = make_double2(w[2 * j], w[2 * j + 1]);
    } else {
#pragma unroll
        for (int q = 0; q < MC; ++q) p[q] = w[q];
    }
}
template <int Q> __device__ __forceinline__ void load_msg_stream(const double *__restrict__ M, size_t k, double (&v)[Q]) {
#if SBMBP_NT & 1
    constexpr int MC = msg_rec<Q>::MC;
    double w[MC];
#pragma unroll
    for (int q = 0; q < MC; ++q) w[q] = __builtin_nontemporal_load(M + k * MC + q);
    decode_msg<Q>(w, v);
#else
    load_msg<Q>(M, k, v);
#endif
}
template <int Q> __device__ __forceinline__ void store_msg_stream(double *__restrict__ M, size_t k, const double (&v)[Q]) {
#if SBMBP_NT & 2
    constexpr int MC = msg_rec<Q>::MC;
    double w[MC];
    encode_msg<Q>(v, w);
#pragma unroll
    for (int q = 0; q < MC; ++q) __builtin_nontemporal_store(w[q], M + k * MC + q);
#else
    store_msg<Q>(M, k, v);
#endif
}
// marginal rows shipped between shards leave out their LAST component (plain first Q-1; k_pack_rows / k_unpack_rows)
template <int Q> __device__ __forceinline__ void finish_last(double (&v)[Q]) {
    double s = v[0];
#pragma unroll
    for (int q = 1; q < Q - 1; ++q) s += v[q];
    const double r = 1.0 - s;
    v[Q - 1] = r < 0.0 ? 0.0 : r;
}
// the same encoding with a run-time Q (state conversion, initialisation, the exact-criterion kernel)
__device__ __forceinline__ void decode_msg_rt(const double *w, int Q, double *v) {
    const int mc = Q - 1, nb = (Q <= 2) ? 1 : (Q <= 4) ? 2 : (Q <= 8) ? 3 : 4;
    unsigned idx = 0;
    for (int j = 0; j < nb; ++j) idx |= sign_bit(w[j]) << j;
    idx = idx < unsigned(Q) ? idx : unsigned(Q - 1);
    double s = 0.0;
    for (int j = 0; j < mc; ++j) s += fabs(w[j]);
    double big = 1.0 - s;
    big = big < 0.0 ? 0.0 : big;
    for (int k = 0; k < Q; ++k) v[k] = unsigned(k) < idx ? fabs(w[k]) : (unsigned(k) == idx ? big : fabs(w[k - 1]));
}
__device__ __forceinline__ void encode_msg_rt(const double *v, int Q, double *w) {
    const int mc = Q - 1, nb = (Q <= 2) ? 1 : (Q <= 4) ? 2 : (Q <= 8) ? 3 : 4;
    unsigned idx = 0;
    double vmax = v[0];
    for (int k = 1; k < Q; ++k) if (v[k] > vmax) { vmax = v[k]; idx = unsigned(k); }
    for (int j = 0; j < mc; ++j) {
        double o = unsigned(j) < idx ? v[j] : v[j + 1];
        if (j < nb) o = with_sign(o, idx >> j);
        w[j] = o;
    }
}
__device__ __forceinline__ uint32_t load_idx_stream(const uint32_t *__restrict__ p) {
#if SBMBP_NT & 1
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// sticky-NaN maximum: a NaN difference must never look like convergence
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_nanmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_down(v, o, 64));
    return v;
}

// Block reduction of NS sums followed by one sticky-NaN max; thread 0 stores them to out[0..NS].
// sred: NS+1 doubles per wave (4 waves).
template <int NS, int NW = 4> __device__ __forceinline__ void block_reduce_store(double (&s)[NS], double mx, double *sred, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NS; ++q) s[q] = wave_sum(s[q]);
    mx = wave_nanmax(mx);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q) sred[wave * (NS + 1) + q] = s[q];
        sred[wave * (NS + 1) + NS] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            double a = sred[q];
#pragma unroll
            for (int w = 1; w < NW; ++w) a += sred[w * (NS + 1) + q];  // fixed order
            out[q] = a;
        }
        double m = sred[NS];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = nanmax(m, sred[w * (NS + 1) + NS]);
        out[NS] = m;
    }
}

// exact power-of-two rescale of a Q-vector so that its largest entry has exponent 0; returns the
// removed exponent (0 when no rescale was needed)
template <int Q> __device__ __forceinline__ int rescale_pow2(double (&A)[Q], double lo = 1e-100, double hi = 1e100) {
    double amax = A[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) amax = fmax(amax, A[q]);
    if (amax > lo && amax < hi) return 0;
    if (!(amax > 0.0) || amax > 1.7e308) return 0;  // zero, NaN or Inf: nothing sensible to do
    int n = ilogb(amax);
#pragma unroll
    for (int q = 0; q < Q; ++q) A[q] = ldexp(A[q], -n);
    return n;
}

// A[q] <- A[q] * eta[q] * F_i[q] up to a common factor, and their sum. F_i = exp(-beta h/N) (dc 0,
// pre-multiplied into etaF) or exp(-d_i h[q]/N) (dc 1, 2; bp.cpp:1021-1023). For large degrees the
// latter underflows in every component, so it is shifted by min_q h (cancels in both normalisations)
// and, when the spread is still extreme, combined in the log domain with a max-shift exactly as the
// reference's large-degree path does (bp.cpp:850-868).
template <int Q>
__device__ __forceinline__ double apply_field(const dev_params *__restrict__ P, int dc, double di, double (&A)[Q],
                                              const double *ft = nullptr /* the row's line of P->ftab, already in registers */) {
    double tot = 0.0;
    if (!dc) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] *= P->etaF[q]; tot += A[q]; }
        return tot;
    }
    double hmin = P->hN[0], hmax = P->hN[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) { hmin = fmin(hmin, P->hN[q]); hmax = fmax(hmax, P->hN[q]); }
    if (di * (hmax - hmin) < 300.0) {
        if (ft != nullptr) {
#pragma unroll
            for (int q = 0; q < Q; ++q) { A[q] *= ft[q]; tot += A[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) { A[q] *= P->eta[q] * exp(-di * (P->hN[q] - hmin)); tot += A[q]; }
        }
    } else {
        double lp[Q], m = -1.0e300;
#pragma unroll
        for (int q = 0; q < Q; ++q) { lp[q] = log(A[q]) + P->logeta[q] - di * P->hN[q]; m = fmax(m, lp[q]); }
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] = exp(lp[q] - m); tot += A[q]; }
    }
    return tot;
}

// ---- long rows: products with ONE EXPONENT PER COMPONENT ------------------------------------------------
// A row of several hundred edges can have partial products that are extreme in opposite directions (one half of
// the neighbours favouring group a by 1e+300, the other half group b): with a common exponent the smaller
// component of each partial product is flushed to zero and the row ends as 0 x 0. So above BIG_ROW edges (wave
// products) and for hub rows (workgroup products) every component carries its own binary exponent, and the row is
// finished in the log domain, as the reference's large-degree path does (bp.cpp:844-868).
template <int Q> __device__ __forceinline__ void x_norm(double (&A)[Q], int (&ae)[Q]) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        int k;
        A[q] = frexp(A[q], &k);  // 0 stays 0 (k = 0); NaN/Inf stay what they are
        ae[q] += k;
    }
}
// psi~[q] = A[q] 2^ae[q] eta[q] F_i[q], returned relative to its largest component (in A); the return value is the
// sum of the returned components and *lmax the log of the factor taken out (log_partition = *lmax + log(sum))
template <int Q>
__device__ __forceinline__ double apply_field_x(const dev_params *__restrict__ P, int dc, double di, double (&A)[Q], const int (&ae)[Q],
                                                double *lmax = nullptr) {
    if (lmax == nullptr) {  // sweeps: when the exponents span less than 2^900 nothing can be flushed, so skip the logs
        int emax = ae[0], emin = ae[0];
#pragma unroll
        for (int q = 1; q < Q; ++q) { emax = max(emax, ae[q]); emin = min(emin, ae[q]); }
        if (emax - emin < 900) {
#pragma unroll
            for (int q = 0; q < Q; ++q) A[q] = ldexp(A[q], ae[q] - emax);
            return apply_field<Q>(P, dc, di, A);
        }
    }
    double lp[Q], m = -1.0e300;
    const double g = dc ? di : P->beta;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        lp[q] = log(A[q]) + double(ae[q]) * 0.6931471805599453 + P->logeta[q] - g * P->hN[q];
        m = fmax(m, lp[q]);  // fmax ignores a NaN operand only if the other is a number: an all-NaN row stays NaN below
    }
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = exp(lp[q] - m); tot += A[q]; }
    if (lmax) *lmax = m;
    return tot;
}
// log( sum_q A[q] 2^ae[q] eta[q] F_i[q] )   (f_site, bp.cpp:446-502)
template <int Q>
__device__ __forceinline__ double log_partition_x(const dev_params *__restrict__ P, int dc, double di, const double (&A)[Q], const int (&ae)[Q]) {
    double t[Q], lmax;
#pragma unroll
    for (int q = 0; q < Q; ++q) t[q] = A[q];
    const double tot = apply_field_x<Q>(P, dc, di, t, ae, &lmax);
    return lmax + log(tot);
}
// e_site of one row (bp.cpp:506-560): sum_q w_q (-h_q/N) / sum_q w_q with w_q = C[q] 2^ce[q] eta_q exp(-h_q/N)
template <int Q>
__device__ __forceinline__ double entropy_site_x(const dev_params *__restrict__ P, const double (&C)[Q], const int (&ce)[Q]) {
    double lw[Q], m = -1.0e300;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        lw[q] = log(C[q]) + double(ce[q]) * 0.6931471805599453 + P->logeta[q] - P->hN[q];
        m = fmax(m, lw[q]);
    }
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const double w = exp(lw[q] - m);
        den += w;
        num += w * (-P->hN[q]);
    }
    return num / den;
}
// product of the workgroup's per-thread partial products; every thread returns with the result
template <int Q> __device__ __forceinline__ void block_product_x(double (&A)[Q], int (&ae)[Q], double *sAq, int *sEq) {
    const int tid = threadIdx.x;
    x_norm<Q>(A, ae);
#pragma unroll
    for (int q = 0; q < Q; ++q) { sAq[tid * Q + q] = A[q]; sEq[tid * Q + q] = ae[q]; }
    __syncthreads();
    for (int s = BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int q = 0; q < Q; ++q) { A[q] *= sAq[(tid + s) * Q + q]; ae[q] += sEq[(tid + s) * Q + q]; }
            x_norm<Q>(A, ae);
#pragma unroll
            for (int q = 0; q < Q; ++q) { sAq[tid * Q + q] = A[q]; sEq[tid * Q + q] = ae[q]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = sAq[q]; ae[q] = sEq[q]; }
}

// Product over a row's edges out of LDS by a whole wave: lanes take strided edges, then a shuffle
// butterfly multiplies the 64 partial products (every lane ends with the same value: a*b == b*a
// bitwise, so the result is deterministic). Used for rows above BIG_ROW edges, where the lane-per-row
// loop would serialise hundreds of LDS reads while the rest of the workgroup waits.
#ifndef SBMBP_BIG_ROW
#define SBMBP_BIG_ROW 32
#endif
constexpr int BIG_ROW = SBMBP_BIG_ROW;
template <int Q> __device__ __forceinline__ void row_product_wave(const double *sb, int es, int ee, double (&A)[Q], int (&ae)[Q]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = 1.0; ae[q] = 0; }
    for (int e = es + lane; e < ee; e += 64) {
        double b[Q];
        load_vec<Q>(&sb[e * Q], b);
#pragma unroll
        for (int q = 0; q < Q; ++q) A[q] *= b[q];
        if ((((e - es) >> 6) & 3) == 3) x_norm<Q>(A, ae);  // four factors of O(W) stay far inside the double range
    }
    x_norm<Q>(A, ae);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] *= __shfl_xor(A[q], o, 64); ae[q] += __shfl_xor(ae[q], o, 64); }
        x_norm<Q>(A, ae);  // partners hold identical values, so they renormalise identically
    }
}

#ifndef SBMBP_BATCH_RECIP
#define SBMBP_BATCH_RECIP 1  // 1: the Q reciprocals of a lane from ONE division (prefix products), marginal-gather sweep only
#endif
// r[k] = 1 / x[k] for all k from one division: prefix products p_k = x_0 .. x_k, inv = 1 / p_{Q-1}, then backwards
// r[k] = inv * p_{k-1}, inv *= x[k]. Falls back to Q divisions when the product leaves the normal range.
template <int Q> __device__ __forceinline__ void recip_all(const double (&x)[Q], double (&r)[Q]) {
    double p[Q];
    p[0] = x[0];
#pragma unroll
    for (int k = 1; k < Q; ++k) p[k] = p[k - 1] * x[k];
    double inv = 1.0 / p[Q - 1];
    if (!(p[Q - 1] > 1e-280) || !(p[Q - 1] < 1e280)) {
#pragma unroll
        for (int k = 0; k < Q; ++k) r[k] = 1.0 / x[k];
        return;
    }
#pragma unroll
    for (int k = Q - 1; k > 0; --k) { r[k] = inv * p[k - 1]; inv *= x[k]; }
    r[0] = inv;
}

#ifndef SBMBP_NODIV
#define SBMBP_NODIV 1  // 1: division-free reconstruction and cavity in the marginal-gather sweep (one division per edge instead of four)
#endif
// r[k] = prod_{j != k} x[j] from prefix and suffix products (3 (Q-1) multiplications, no division): x[k] r[k] is the same
// for every k, so r is 1 / x up to ONE common factor - and every place the marginal-gather sweep divides by a Q-vector
// normalises the result right after. Returns false when the product of all x leaves the normal range (the caller divides).
template <int Q> __device__ __forceinline__ bool excl_products(const double (&x)[Q], double (&r)[Q]) {
    double pre = x[0];
    r[0] = 1.0;
#pragma unroll
    for (int k = 1; k < Q; ++k) { r[k] = pre; pre *= x[k]; }
    double suf = x[Q - 1];
#pragma unroll
    for (int k = Q - 2; k >= 0; --k) { r[k] *= suf; suf *= x[k]; }
    return pre > 1e-250 && pre < 1e250;
}
// v <- v 2^-e with e the binary exponent of the sum of v (an exact scaling): the vector then sums to [0.5, 1)
template <int Q> __device__ __forceinline__ void pow2_normalise(double (&v)[Q]) {
    double t = v[0];
#pragma unroll
    for (int k = 1; k < Q; ++k) t += v[k];
    int e;
    (void)frexp(t, &e);  // 0, NaN, Inf: e = 0 (nothing sensible to scale)
#pragma unroll
    for (int k = 0; k < Q; ++k) v[k] = ldexp(v[k], -e);
}

// b[q] = sum_t W_il[t][q] m[t]   (belief_propagation.cpp:1000-1012)
template <int Q, bool DC2>
__device__ __forceinline__ void edge_field(const dev_params *__restrict__ P, const double (&m)[Q], double didl, double (&b)[Q]) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            double w = P->W[t * Q + q];
            if (DC2) { double x = didl * w; w = x / (1.0 + x); }
            acc += w * m[t];
        }
        b[q] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// K1: one synchronous sweep over the rows of this workgroup's segment.
// partials[b*(Q+1) + q] = sum_rows g_i psi_i[q],  partials[b*(Q+1)+Q] = max |delta message|.
// ------------------------------------------------------------------------------------------------
template <int Q> struct sweep_waves { static constexpr int N = Q == 16 ? 2 : 1; };  // register target, see k_sweep_psi (Q = 16: 258 - 265 registers otherwise)
template <int Q, bool DC2>
__global__ void __launch_bounds__(frame_cfg<Q>::TPB) __attribute__((amdgpu_waves_per_eu(sweep_waves<Q>::N)))
k_sweep(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rev, const uint32_t *__restrict__ nbr, const uint32_t *__restrict__ ndeg /* degree of every table row (DC2 only) */,
        const double *__restrict__ Mold, double *__restrict__ Mnew, const double *__restrict__ psi_old,
        double *__restrict__ psi, const int32_t *__restrict__ clamp, const uint32_t *__restrict__ blk_row,
        const uint32_t *__restrict__ blk_e0, const dev_params *__restrict__ P, int dc, double damp, double *__restrict__ partials) {
    constexpr int EPT = frame_cfg<Q>::EPT, CAP = frame_cfg<Q>::CAP, RCAP = frame_cfg<Q>::RCAP;
    __shared__ double sb[CAP * Q];     // b_e[q] of every edge of the segment
    __shared__ double sA[RCAP * Q];    // unnormalised marginal of every row
    __shared__ uint32_t srp[RCAP + 1]; // row offsets relative to the segment
    __shared__ uint16_t srow[CAP];     // row (within segment) of every edge
    __shared__ uint8_t sfl[RCAP];      // 1 = clamped row
    __shared__ double sred[frame_cfg<Q>::WAVES * (Q + 1)];
    __shared__ int sbig;               // the segment holds a row above BIG_ROW edges

    // bounds and stop flag from one level of scalar loads; streams issued before the row offsets -> LDS fill
    const int tid = threadIdx.x;
    const int stop = P->stop;
    const uint32_t r0 = blk_row[blockIdx.x], r1 = blk_row[blockIdx.x + 1];
    const uint32_t e0 = blk_e0[blockIdx.x];
    const int nrows = int(r1 - r0), ne = int(blk_e0[blockIdx.x + 1] - e0);
    if (stop || ne > CAP) return;  // stopped run, or hub row (the fragment kernels own it): uniform exit before any barrier

    // ---- phase 1: lane per directed edge: gather incoming message, b = W^T m -> LDS (branch-free loads,
    // see k_sweep_psi)
    constexpr int RPT = RCAP / frame_cfg<Q>::TPB + 1;
    double mo[EPT][Q];
    uint32_t rk[EPT], kk[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int le = j * frame_cfg<Q>::TPB + tid;
        kk[j] = (ne > 0) ? e0 + uint32_t(le < ne ? le : 0) : 0u;
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) rk[j] = load_idx_stream(rev + kk[j]);
#pragma unroll
    for (int j = 0; j < EPT; ++j) load_msg_stream<Q>(Mold, kk[j], mo[j]);
    uint32_t rpv[RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; rpv[t] = row_ptr[r0 + uint32_t(r < nrows ? r : nrows)]; }
    double mi[EPT][Q];
#pragma unroll
    for (int j = 0; j < EPT; ++j) load_msg<Q>(Mold, rk[j], mi[j]);
    if (tid == 0) sbig = 0;
#pragma unroll
    for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; if (r <= nrows) srp[r] = rpv[t] - e0; }
    __syncthreads();  // srp visible
    for (int r = tid; r < nrows; r += frame_cfg<Q>::TPB) {
        const int es = int(srp[r]), ee = int(srp[r + 1]);
        if (ee - es > BIG_ROW) sbig = 1;
        for (int e = es; e < ee; ++e) srow[e] = uint16_t(r);
        sfl[r] = (clamp != nullptr && clamp[r0 + r] != -1) ? 1 : 0;
    }
    if (DC2) __syncthreads();  // per-edge weights need the edge -> row map
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int le = j * frame_cfg<Q>::TPB + tid;
        if (le < ne) {
            double didl = 0.0;
            if (DC2) {
                const int r = srow[le];
                const uint32_t l = nbr[e0 + le];
                didl = double(srp[r + 1] - srp[r]) * double(ndeg[l]);
            }
            double b[Q];
            edge_field<Q, DC2>(P, mi[j], didl, b);
            store_vec<Q>(&sb[le * Q], b);
        }
    }
    __syncthreads();

    // ---- phase 2: lane per row (a wave per row above BIG_ROW edges): A[q] = prod_e b_e[q];
    //      psi_i = normalise(A * eta * F_i)
    double Sacc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) Sacc[q] = 0.0;
    auto finish_row = [&](int r, double di, double (&A)[Q], const int *ae /* per-component exponents of a long row, or null */,
                          const double *ft = nullptr /* the row's line of P->ftab (rows of <= FT_D edges under dc), or null */) {
        double pv[Q];
        double tot;
        if (ae) {
            int x[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) x[q] = ae[q];
            tot = apply_field_x<Q>(P, dc, di, A, x);
        } else {
            tot = apply_field<Q>(P, dc, di, A, ft);
        }
        store_vec<Q>(&sA[r * Q], A);
        const double inv = 1.0 / tot;
        const double gi = dc ? di : 1.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { pv[q] = A[q] * inv; Sacc[q] += gi * pv[q]; }
        store_vec<Q>(psi + size_t(r0 + r) * Q, pv);
    };
    for (int r = tid; r < nrows; r += frame_cfg<Q>::TPB) {
        const int es = int(srp[r]), ee = int(srp[r + 1]);
        const double di = double(ee - es);
        if (sfl[r]) {  // clamped: marginal and out-messages stay as initialised (bp.cpp:1115-1124)
            double pv[Q];
            load_vec<Q>(psi_old + size_t(r0 + r) * Q, pv);
            store_vec<Q>(psi + size_t(r0 + r) * Q, pv);
            const double gi = dc ? di : 1.0;
#pragma unroll
            for (int q = 0; q < Q; ++q) Sacc[q] += gi * pv[q];
        } else if (ee - es <= BIG_ROW) {
            double A[Q], ft[Q];
            const bool tab = dc != 0 && ee - es <= FT_D;  // the field factors of this degree: loaded while the product runs
            if (tab) load_vec<Q>(P->ftab + size_t(ee - es) * QMAX, ft);
#pragma unroll
            for (int q = 0; q < Q; ++q) A[q] = 1.0;
            for (int e = es; e < ee; ++e) {
                double b[Q];
                load_vec<Q>(&sb[e * Q], b);
#pragma unroll
                for (int q = 0; q < Q; ++q) A[q] *= b[q];
                rescale_pow2<Q>(A);
            }
            finish_row(r, di, A, nullptr, tab ? ft : nullptr);
        }
    }
    if (sbig)  // uniform: written before the barrier that ends phase 1
    for (int r = tid >> 6; r < nrows; r += frame_cfg<Q>::WAVES) {  // wave-uniform row index
        const int es = int(srp[r]), ee = int(srp[r + 1]);
        if (ee - es > BIG_ROW && !sfl[r]) {
            double A[Q];
            int ae[Q];
            row_product_wave<Q>(sb, es, ee, A, ae);
            if ((tid & 63) == 0) finish_row(r, double(ee - es), A, ae);
        }
    }
    __syncthreads();

    // ---- phase 3: lane per directed edge: cavity, normalise, damp, store
    double md = 0.0;
    const int probe2 = P->ar_probe2;  // adaptive relaxation's probe sweep: report |m^{t+1} - m^{t-1}| (m^{t-1} sits in the slot written below)
    damp *= P->damp_auto;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int le = j * frame_cfg<Q>::TPB + tid;
        if (le < ne) {
            const int r = srow[le];
            double out[Q];
            if (sfl[r]) {
#pragma unroll
                for (int q = 0; q < Q; ++q) out[q] = mo[j][q];
            } else {
                double A[Q], b[Q], cav[Q];
                load_vec<Q>(&sA[r * Q], A);
                load_vec<Q>(&sb[le * Q], b);
                bool ok = true;
                double tot = 0.0;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    cav[q] = A[q] / b[q];
                    ok = ok && (b[q] > 0.0) && (cav[q] <= 1.7e308);
                    tot += cav[q];
                }
                if (!ok) {  // exact cavity product when a division is unusable (b == 0 or overflow)
                    const int es = int(srp[r]), ee = int(srp[r + 1]);
                    const double di = double(ee - es);
                    int ce[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) { cav[q] = 1.0; ce[q] = 0; }
                    for (int e = es; e < ee; ++e) {
                        if (e == le) continue;
#pragma unroll
                        for (int q = 0; q < Q; ++q) cav[q] *= sb[e * Q + q];
                        x_norm<Q>(cav, ce);
                    }
                    tot = apply_field_x<Q>(P, dc, di, cav, ce);
                }
                const double inv = 1.0 / tot;
                double ref[Q];
                if (probe2) {  // uniform
                    load_msg<Q>(Mnew, size_t(e0 + le), ref);
                } else {
#pragma unroll
                    for (int q = 0; q < Q; ++q) ref[q] = mo[j][q];
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const double nv = cav[q] * inv;
                    out[q] = damp * nv + (1.0 - damp) * mo[j][q];
                    // 1-step: against the undamped value (bp.cpp:1059-1063); the probe compares what is stored, and a damped
                    // message moves by damp * (new - old) per sweep, so it is scaled back to compare like with like
                    md = nanmax(md, probe2 ? fabs(ref[q] - out[q]) / damp : fabs(ref[q] - nv));
                }
            }
            store_msg_stream<Q>(Mnew, size_t(e0 + le), out);
        }
    }
    block_reduce_store<Q, frame_cfg<Q>::WAVES>(Sacc, md, sred, partials + size_t(blockIdx.x) * (Q + 1));
}

// ------------------------------------------------------------------------------------------------
// K1p: the same synchronous sweep with the incoming message RECONSTRUCTED from the neighbour's
// marginal instead of gathered from the message array ("node-marginal gather"):
//     m^t_{l->i}[s]  ∝  psi^t_l[s] / ( sum_r W[r][s] m^{t-1}_{i->l}[r] )
// (psi^t_l is the product over all of l's incoming messages of sweep t-1, so dividing out the
// factor contributed by i's own message of sweep t-1 leaves the cavity message; exact in exact
// arithmetic) A lane therefore streams its OWN out-message of sweep t-1 (Mio),
// gathers psi^t_l from the N*Q table — 1/c the size of the message array, so the random reads are
// mostly served by L2 / Infinity Cache — and overwrites its own slot with m^{t+1} in place.
// Requires W > 0, no clamped rows, damping 1, and psi^t consistent with the two message buffers
// (the engine runs k_sweep first after any state/parameter change).
// partial slot Q holds a HINT: the 2-step difference max|m^{t+1}-m^{t-1}|; the exact 1-step message
// difference the reference's criterion uses is measured by k_msg_diff before convergence is declared
// (so a period-2 oscillation, whose 2-step difference vanishes, can never pass as converged).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// K2 (one lane): the folded sums S[q] and the maximum difference of a sweep -> relaxed field, h / eta exp(-beta h),
// convergence state. mode 0: after a sweep; mode 1: field initialisation (no sweep bookkeeping); mode 2: exact field
// refresh (no relaxation, no bookkeeping).
// ------------------------------------------------------------------------------------------------
// Everything is loaded before anything is stored (P may alias nothing else here, but the compiler cannot know that the
// stores to P do not change what the later loads from P return): one memory round trip instead of a chain of them. With a
// run-time Q and local arrays indexed by it the same code lived in scratch memory and took 6 (Q = 2) to 21 us (Q = 8).
// ladders of the adaptive relaxation (dev_params::ar_*; the same numbers as oracle/bp_oracle.cpp ar_t)
// field caps 1, 0.25, 0.1, 0.05; generic levels (mix cap, damping factor): (0.5,1) (0.25,1) (0.5,0.5) (0.25,0.5) (0.1,0.5) (0.25,0.25) (0.1,0.25)
constexpr int AR_NF = 4, AR_NG = 7, AR_WIN = 24;
__host__ __device__ __forceinline__ double ar_field_cap(int fl) { return fl <= 0 ? 1.0 : (fl == 1 ? 0.25 : (fl == 2 ? 0.1 : 0.05)); }
__host__ __device__ __forceinline__ double ar_gen_mix(int gl) {
    return gl < 0 ? 1.0 : (gl == 0 || gl == 2) ? 0.5 : (gl == 1 || gl == 3 || gl == 5) ? 0.25 : 0.1;
}
__host__ __device__ __forceinline__ double ar_gen_damp(int gl) { return gl <= 1 ? 1.0 : (gl <= 4 ? 0.5 : 0.25); }

template <int Q>
__device__ __forceinline__ void finalize_update(dev_params *__restrict__ P, const double *sums /* [Q] then the max */, int mode,
                                                double *__restrict__ diff_hist, uint32_t hist_cap, int md_exact, double *s_hN /* LDS [Q]: h/N for field_table */) {
    double cab[Q * Q], eta[Q], Sold[Q], S[Q];
#pragma unroll
    for (int a = 0; a < Q * Q; ++a) cab[a] = P->cab[a];
#pragma unroll
    for (int q = 0; q < Q; ++q) { eta[q] = P->eta[q]; Sold[q] = P->S[q]; }
    const double invN = P->invN, beta = P->beta, crit = P->crit;
    double mix = P->field_mix, prev_hint = P->prev_hint;
    const int have_prev = P->have_prev, hinted = P->hinted, exact = P->exact, conv_iter = P->conv_iter, it = P->sweep_idx;
    // adaptive relaxation: the whole state is loaded up front as well (one round trip)
    const int ar_on = P->ar_on, psi_ok = P->ar_psi_ok, probe2 = P->ar_probe2;
    int fl = P->ar_fl, gl = P->ar_gl, armed = P->ar_armed, probing = P->ar_probing, stall = P->ar_stall, hold = P->ar_hold,
        holdS = P->ar_holdS, sigc = P->ar_sigc, nS = P->ar_nS, wn = P->ar_wn;
    const double base_mix = P->ar_base_mix;
    double v1 = P->ar_v1, v2 = P->ar_v2, wmin = P->ar_wmin, pmin = P->ar_pmin, d1p = P->ar_d1p;
    double S1[Q], S2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { S1[q] = P->ar_S1[q]; S2[q] = P->ar_S2[q]; }

    if (mode == 0) {
        const double md = sums[Q];
        // what the sweep reported: 1 = the reference's 1-step difference (bp.cpp:1059-1063), 2 = a 2-step difference
        const int kind = md_exact ? (probe2 ? 2 : 1) : ((hinted && !exact) ? 2 : 1);
        bool conv = false, esc = false;
        // A relaxed field lags its marginals: the messages can stand still to within crit while it is still catching up. rf =
        // what a plain sweep from here would move the field term beta h/N by; the run has converged when that is below crit too.
        double rf = 0.0;
        if (have_prev && mix < 1.0) {
#pragma unroll
            for (int q1 = 0; q1 < Q; ++q1) {
                double acc = 0.0;
#pragma unroll
                for (int q2 = 0; q2 < Q; ++q2) acc += cab[q2 * Q + q1] * (sums[q2] - Sold[q2]);
                rf = fmax(rf, fabs(acc) * invN * beta);
            }
        }
        const bool field_ok = rf < crit;
        auto reset_after = [&]() {
            hold = 6; stall = 0; v1 = v2 = -1.0; prev_hint = 0.0; probing = 0; armed = 0;
            wn = 0; wmin = 1e300; pmin = -1.0; nS = 0; sigc = 0; d1p = -1.0; holdS = 4;
        };
        auto cur_mix = [&]() { return fmin(fmin(base_mix, ar_field_cap(fl)), ar_gen_mix(gl)); };
        auto esc_gen = [&]() {  // the next level that changes anything (a field level may already have the mix below a level's cap)
            const double m0 = cur_mix(), d0 = ar_gen_damp(gl);
            while (gl + 1 < AR_NG) {
                ++gl;
                if (cur_mix() < m0 || ar_gen_damp(gl) < d0) {
                    // the first damped level starts from an unrelaxed field again: damping often steadies the field by itself, and
                    // a field level inherited from the undamped sweeps can make the damped ones crawl ((F) fires again if it must)
                    if (d0 == 1.0 && ar_gen_damp(gl) < 1.0) fl = 0;
                    reset_after();
                    return;
                }
            }
            hold = 1 << 30;  // the ladder is used up: the run goes on as it is
        };
        auto esc_field = [&]() {
            // a swing of the field sums that shows although the generic ladder has already softened the field is driven by the
            // messages, not by the field's own feedback: damping answers it, a still softer field only slows everything down
            if (gl >= 0) { if (gl < 1) gl = 1; esc_gen(); return; }
            int nf = fl;
            while (nf + 1 < AR_NF && !(ar_field_cap(nf) < cur_mix())) ++nf;  // the next cap that actually lowers the mix
            if (ar_field_cap(nf) < cur_mix()) { fl = nf; reset_after(); }
            else esc_gen();
        };
        auto hint = [&](double v) {  // a 2-step value can only arm the exact criterion
            double scale = HINT_SCALE;
            if (prev_hint > 0.0 && v > 0.0 && v < prev_hint) {
                const double r = v / prev_hint;
                scale = fmin(HINT_SCALE_MAX, fmax(HINT_SCALE, 1.5 * (1.0 + 1.0 / r) / r));
            }
            prev_hint = v;
            if (v < scale * crit) armed = 1;
        };
        if (!ar_on) {
            if (kind == 1) conv = md < crit && field_ok; else hint(md);
        } else if (probing) {  // (P) the probe's answer
            probing = 0;
            if (kind == 1 && md < crit && field_ok) conv = true;
            else if (v1 >= 0.0) {
                const double one = kind == 1 ? md : v1, two = kind == 1 ? v1 : md;
                if (two < 0.5 * one) { if (gl < 1) gl = 1; esc_gen(); esc = true; }  // messages with period 2: damping answers that, a softer field does not
                else { hold = 8; stall = 0; }
            }
        } else {
            if (kind == 1) {
                if (md < crit && field_ok) conv = true;
            } else hint(md);
            if (!conv && !esc) {
                if (hold > 0) --hold;
                else {
                    if (v2 >= 0.0 && md >= 0.98 * v2) ++stall; else stall = 0;
                    if (stall >= 4) { probing = 1; stall = 0; }
                }
                v2 = v1; v1 = md;
                wmin = fmin(wmin, md); ++wn;
                if (wn >= AR_WIN * (1 + (gl > 0 ? (gl < 3 ? gl : 3) : 0))) {  // (W)
                    if (pmin >= 0.0 && wmin >= 0.9 * pmin && hold < (1 << 29)) { esc_gen(); esc = true; }
                    else { pmin = wmin; wmin = 1e300; wn = 0; }
                }
            }
        }
        if (ar_on && !conv && !esc) {  // (F) period 2 in the raw field sums
            bool fe = false;
            if (holdS > 0) --holdS;
            else if (nS >= 2) {
                double d1 = 0.0, d2 = 0.0, tot = 0.0;
#pragma unroll
                for (int q = 0; q < Q; ++q) { d1 = fmax(d1, fabs(sums[q] - S1[q])); d2 = fmax(d2, fabs(sums[q] - S2[q])); tot += fabs(sums[q]); }
                const bool sig = d2 < 0.5 * d1 && d1 > 1e-9 * tot;
                if (sig && d1 > 0.05 * tot && fl == 0) fe = true;                              // a violent swing: act at once
                else if (sig && (d1p < 0.0 || d1 >= 0.98 * d1p)) { if (++sigc >= 6) fe = true; }  // a swing that does not die out
                else sigc = 0;
                d1p = d1;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) { S2[q] = S1[q]; S1[q] = sums[q]; }
            nS = nS < 2 ? nS + 1 : 2;
            if (fe) esc_field();
        }
        mix = ar_on ? cur_mix() : mix;
        const double dampA = ar_gen_damp(gl);
        const bool psi_next = psi_ok && dampA == 1.0;
        int k_next = (!psi_next || armed) ? 1 : 2;
        if (probing) k_next = 3 - k_next;
        P->maxdiff = md;
        if (diff_hist != nullptr && uint32_t(it) < hist_cap) diff_hist[it] = md;
        P->last_exact = kind == 1 ? 1 : 0;
        P->prev_hint = prev_hint;
        P->exact = k_next == 1 ? 1 : 0;
        P->ar_probe2 = (!psi_next && k_next == 2) ? 1 : 0;
        P->field_mix = mix;
        P->damp_auto = dampA;
        P->ar_fl = fl; P->ar_gl = gl; P->ar_armed = armed; P->ar_probing = probing; P->ar_stall = stall; P->ar_hold = hold;
        P->ar_holdS = holdS; P->ar_sigc = sigc; P->ar_nS = nS; P->ar_wn = wn;
        P->ar_v1 = v1; P->ar_v2 = v2; P->ar_wmin = wmin; P->ar_pmin = pmin; P->ar_d1p = d1p;
#pragma unroll
        for (int q = 0; q < Q; ++q) { P->ar_S1[q] = S1[q]; P->ar_S2[q] = S2[q]; }
        if (conv && conv_iter < 0) {
            P->conv_iter = it;
            P->stop = 1;
        } else if (psi_ok && dampA < 1.0 && !md_exact) {  // damping needs the message-gather form: the host switches (run_sweeps)
            P->pause = 1;
            P->stop = 1;
        }
        P->sweep_idx = it + 1;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        double sv = sums[q];
        if (mode == 0 && have_prev && mix < 1.0) sv = (1.0 - mix) * Sold[q] + mix * sv;
        S[q] = sv;
    }
    double hN[Q], etaF[Q];
#pragma unroll
    for (int q1 = 0; q1 < Q; ++q1) {  // h[q1] = sum_q2 cab[q2][q1] S[q2]   (bp.cpp:341-355)
        double h = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < Q; ++q2) h += cab[q2 * Q + q1] * S[q2];
        hN[q1] = h * invN;
        etaF[q1] = eta[q1] * exp(-beta * hN[q1]);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) { P->S[q] = S[q]; P->hN[q] = hN[q]; P->etaF[q] = etaF[q]; s_hN[q] = hN[q]; }
    P->have_prev = 1;
}

// the host's answer to dev_params::pause: the run goes on in the message-gather form (1-step differences every sweep)
__global__ void k_resume(dev_params *__restrict__ P) {
    if (threadIdx.x == 0) { P->stop = 0; P->pause = 0; P->hinted = 0; P->ar_psi_ok = 0; }
}

// the whole workgroup, after finalize_update and a barrier: P->ftab from the new h (dev_params)
template <int Q>
__device__ __forceinline__ void field_table(dev_params *__restrict__ P, const double *s_hN) {
    double hmin = s_hN[0];
#pragma unroll
    for (int q = 1; q < Q; ++q) hmin = fmin(hmin, s_hN[q]);
    for (int x = threadIdx.x; x < (FT_D + 1) * Q; x += BLOCK) {
        const int d = x / Q, q = x - d * Q;
        P->ftab[d * QMAX + q] = P->eta[q] * exp(-double(d) * (s_hN[q] - hmin));
    }
}

// Shard mode of the marginal-gather sweep: halo marginals are gathered straight from the receive buffer the exchange
// filled (rows of ncomp = Q-1 or Q components; the halo is numbered in receive order, plan.py), and a freshly computed
// marginal is dropped into every send slot that ships it, so the sweep needs no pack and no unpack kernel around it.
struct shard_io {
    const uint32_t *snd_ptr;   // [n_own + 1] send slots of every own row (CSR)
    const uint32_t *snd_slot;
    double *sendbuf;           // rows of ncomp components, ordered (chunk, peer, row)
    const double *halo_stage;  // received halo rows of the table this sweep reads
    uint32_t n_own;
    int ncomp;
};
template <int Q> __device__ __forceinline__ void load_halo_row(const shard_io &io, uint32_t h, double (&v)[Q]) {
    if (io.ncomp == Q) load_vec<Q>(io.halo_stage + size_t(h) * Q, v);
    else {  // first Q-1 components; the last is max(0, 1 - sum), as k_unpack_rows restores it
        const double *p = io.halo_stage + size_t(h) * (Q - 1);
#pragma unroll
        for (int q = 0; q < Q - 1; ++q) v[q] = p[q];
        finish_last<Q>(v);
    }
}
template <int Q> __device__ __forceinline__ void send_row(const shard_io &io, uint32_t i, const double (&pv)[Q]) {
    for (uint32_t s = io.snd_ptr[i]; s < io.snd_ptr[i + 1]; ++s) {
        double *dst = io.sendbuf + size_t(io.snd_slot[s]) * io.ncomp;
#pragma unroll
        for (int q = 0; q < Q; ++q) if (q < io.ncomp) dst[q] = pv[q];
    }
}

// product of the workgroup's per-lane partial products, every lane returning the result: shuffle butterfly inside
// the waves (both partners of a step compute the same bits: a*b == b*a), then the wave results in wave order
template <int Q, int WAVES> __device__ __forceinline__ void block_product_shfl(double (&A)[Q], int (&ae)[Q], double *sAw, int *sEw) {
    x_norm<Q>(A, ae);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] *= __shfl_xor(A[q], o, 64); ae[q] += __shfl_xor(ae[q], o, 64); }
        x_norm<Q>(A, ae);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { sAw[wave * Q + q] = A[q]; sEw[wave * Q + q] = ae[q]; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = sAw[q]; ae[q] = sEw[q]; }
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] *= sAw[w * Q + q]; ae[q] += sEw[w * Q + q]; }
        x_norm<Q>(A, ae);
    }
}

// fragment tables of the hub rows (rows above one segment's edge capacity), see k_hub_frag_product below
struct hub_frags {
    const uint32_t *frag_hub;   // [n_frag] hub index of the fragment
    const uint32_t *hub_frag0;  // [n_hub + 1] first fragment of the hub
    double *b;                  // [n_frag * BLOCK][Q] edge fields between the two launches
    double *pA;                 // [n_frag][Q] fragment products: mantissas ...
    int *pE;                    // ... and binary exponents
};
// CLAMP: rows with clamp[i] != -1 (bp_conditional, bp.cpp:1100-1126) keep their marginal and out-messages. Their state is
// one-hot (-i 1 / -f), and for a one-hot neighbour the reconstruction psi_l / (W^T m) normalises to that same one-hot
// vector exactly, so clamped rows need no special case on the receiving side.
// Register targets (waves per SIMD the allocation must leave room for; 1 = none). Q = 16 needs 260 registers left alone, which
// is ONE wave per SIMD; capped at 255 it runs two (Q = 16 control workload: 1.49 -> 0.95 ms, 27 -> 42 % of the roofline; a third
// wave spills 256 B and loses again: 1.24 ms).
#ifndef SBMBP_PSI_WAVES_Q16
#define SBMBP_PSI_WAVES_Q16 2
#endif
#ifndef SBMBP_PSI_WAVES_Q9
#define SBMBP_PSI_WAVES_Q9 1   // Q = 9 .. 12
#endif
template <int Q> struct psi_waves { static constexpr int N = SBMBP_PSI_WAVES > 0 ? SBMBP_PSI_WAVES : (Q == 16 ? SBMBP_PSI_WAVES_Q16 : (Q >= 9 && Q <= 12) ? SBMBP_PSI_WAVES_Q9 : 1); };
template <int Q, bool CLAMP, bool SHARD>
__global__ void __launch_bounds__(frame_cfg<Q>::TPB) __attribute__((amdgpu_waves_per_eu(psi_waves<Q>::N)))
k_sweep_psi(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ nbr, double *__restrict__ Mio,
            const double *__restrict__ psi_old, double *__restrict__ psi_new, const uint32_t *__restrict__ blk_row,
            const uint32_t *__restrict__ blk_e0, const dev_params *__restrict__ P, int dc, double *__restrict__ partials,
            const int32_t *__restrict__ clamp, shard_io io, uint32_t n_seg /* segments = workgroups with work */, int xcd_order,
            const double *__restrict__ Mcmp /* the other message buffer (m^t): read only while P->exact, to report the
                                               exact 1-step difference instead of the 2-step one against the own previous message */,
            int first_from_psi /* sweep 0 after a device initialisation: the incoming message IS the neighbour's marginal
                                  (messages were initialised to the sender's marginal), no reconstruction */) {
    constexpr int EPT = frame_cfg<Q>::EPT, CAP = frame_cfg<Q>::CAP, RCAP = frame_cfg<Q>::RCAP;
    __shared__ double sb[CAP * Q];
    __shared__ double sA[RCAP * Q];
    __shared__ uint32_t srp[RCAP + 1];
    __shared__ uint16_t srow[CAP];
    __shared__ double sred[frame_cfg<Q>::WAVES * (Q + 1)];
    __shared__ int sbig;  // the segment holds a row above BIG_ROW edges
    __shared__ uint8_t sfl[CLAMP ? RCAP : 1];  // 1 = clamped row
    // send slots of the segment's rows kept in LDS (the rest from HBM): 2 * RCAP keeps the Q = 4 workgroup under 32 KB of LDS
    // (5 per CU; 4 * RCAP was 34 KB = 4 per CU); the 8-rank C3 plan has ~150 slots per segment. Measured neutral there.
    constexpr int SLOTS = SHARD ? 2 * RCAP : 1;
    __shared__ uint32_t ssp[SHARD ? RCAP + 1 : 1];       // send-slot offsets of the rows, relative to the segment
    __shared__ uint32_t sslot[SLOTS];
    __shared__ uint16_t ssrow[SLOTS];                    // row (within the segment) of every send slot

    // Segment bounds (row range and edge range side by side) and the stop flag come from one level of scalar
    // loads, so the streams are issued at once; the row offsets -> LDS fill, which only the later phases need,
    // overlaps them.
    const int tid = threadIdx.x;
    const int stop = P->stop;
    const int exact = P->exact;
#if SBMBP_XCD_REMAP
    // Workgroup i is dispatched to XCD i % 8. The grid is padded to 8 * per workgroups (xcd_grid) and XCD x takes the
    // contiguous segments [x * per, (x+1) * per): its L2 then sees one stretch of rows, messages and marginals instead of
    // every eighth segment of the whole graph.
    // (Only for launches of a grid padded by xcd_grid, gridDim.x != n_seg or n_seg % 8 == 0 handled by the flag: the chunk
    // launches of a shard keep the natural order - measured 3 % better there.)
    const uint32_t per = gridDim.x / 8;
    const uint32_t bid = xcd_order ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    if (bid >= n_seg) return;  // padding (uniform per workgroup)
#else
    const uint32_t bid = blockIdx.x;
#endif
    const uint32_t r0 = blk_row[bid], r1 = blk_row[bid + 1];
    const uint32_t e0 = blk_e0[bid];
    const int nrows = int(r1 - r0), ne = int(blk_e0[bid + 1] - e0);
    if (stop || ne > CAP) return;  // stopped run, or hub row (the fragment kernels own it): uniform exit before any barrier

    // ---- phase 1: lane per directed edge. Loads are branch-free (inactive lanes re-read the segment's first
    // edge) so the compiler issues them back to back: index stream, own-message stream, row offsets (kept in
    // registers), then the gathers as soon as the indices are back — no LDS write or wait in between.
    constexpr int RPT = RCAP / frame_cfg<Q>::TPB + 1;
    double mo[EPT][Q];
    uint32_t nl[EPT], kk[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int le = j * frame_cfg<Q>::TPB + tid;
        kk[j] = (ne > 0) ? e0 + uint32_t(le < ne ? le : 0) : 0u;
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) nl[j] = load_idx_stream(nbr + kk[j]);
#pragma unroll
    for (int j = 0; j < EPT; ++j) load_msg_stream<Q>(Mio, kk[j], mo[j]);
    uint32_t rpv[RPT], spv[RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; rpv[t] = row_ptr[r0 + uint32_t(r < nrows ? r : nrows)]; }
    if (SHARD) {  // the rows' send-slot offsets travel with the row offsets; the slots themselves follow after the barrier
#pragma unroll
        for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; spv[t] = io.snd_ptr[r0 + uint32_t(r < nrows ? r : nrows)]; }
    }
    double pl[EPT][Q];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        if (SHARD) {
            // one load sequence for own and halo rows: base and stride are selected, Q-1 components are loaded, the last
            // one is loaded (own rows, full-width halo rows) or restored as max(0, 1 - sum)
#ifdef SBMBP_DEBUG_NOSTAGE  // ablation build (timing only, wrong results): every gather goes to the table
            const bool halo = false;
#else
            const bool halo = nl[j] >= io.n_own;
#endif
            const bool full = !halo || io.ncomp == Q;
            const double *src = halo ? io.halo_stage + size_t(nl[j] - io.n_own) * io.ncomp : psi_old + size_t(nl[j]) * Q;
            // (Round 3 tried the same Q/2 16-byte loads for both kinds of row, a received row read one double past its end and
            // that slot restored afterwards: 0.367 against 0.363 ms per rank of the 8-rank C3 plan - the load width is not what
            // the shard variant pays for.)
#pragma unroll
            for (int q = 0; q < Q - 1; ++q) pl[j][q] = src[q];
            const double lastv = src[full ? Q - 1 : 0];
            finish_last<Q>(pl[j]);
            if (full) pl[j][Q - 1] = lastv;
        } else {
            load_vec<Q>(psi_old + size_t(nl[j]) * Q, pl[j]);
        }
    }
    if (tid == 0) sbig = 0;
#pragma unroll
    for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; if (r <= nrows) srp[r] = rpv[t] - e0; }
    uint32_t s0 = 0;
    if (SHARD) {
        s0 = io.snd_ptr[r0];
#pragma unroll
        for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; if (r <= nrows) ssp[r] = spv[t] - s0; }
    }
    __syncthreads();  // srp (and ssp) visible
    if (SHARD) {  // issue the slot loads now; they are consumed after the next barriers
        const uint32_t ns = min(ssp[nrows], uint32_t(SLOTS));
        for (uint32_t x = tid; x < ns; x += frame_cfg<Q>::TPB) sslot[x] = io.snd_slot[s0 + x];
    }
    for (int r = tid; r < nrows; r += frame_cfg<Q>::TPB) {
        const int es = int(srp[r]), ee = int(srp[r + 1]);
        if (ee - es > BIG_ROW) sbig = 1;
        for (int e = es; e < ee; ++e) srow[e] = uint16_t(r);
        if (CLAMP) sfl[r] = clamp[r0 + r] != -1 ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int le = j * frame_cfg<Q>::TPB + tid;
        if (le < ne) {
            double bo[Q], inc[Q], b[Q];
            if (first_from_psi) {  // uniform
#pragma unroll
                for (int s = 0; s < Q; ++s) inc[s] = pl[j][s];
            } else {
                edge_field<Q, false>(P, mo[j], 0.0, bo);  // what l saw of i's message at sweep t-1
#if SBMBP_NODIV
                // m^t_{l->i} up to a factor: psi_l[s] prod_{s' != s} bo[s'], scaled by an exact power of two. The factor ends
                // up in b, in the row product and in every cavity of the row alike, and cancels in their normalisations.
                double rb[Q];
                if (excl_products<Q>(bo, rb)) {
#pragma unroll
                    for (int s = 0; s < Q; ++s) inc[s] = pl[j][s] * rb[s];
                    pow2_normalise<Q>(inc);
                } else
#endif
                {
                    double tot = 0.0;
#if SBMBP_BATCH_RECIP
                    double rr[Q];
                    recip_all<Q>(bo, rr);
#pragma unroll
                    for (int s = 0; s < Q; ++s) { inc[s] = pl[j][s] * rr[s]; tot += inc[s]; }
#else
#pragma unroll
                    for (int s = 0; s < Q; ++s) { inc[s] = pl[j][s] / bo[s]; tot += inc[s]; }
#endif
                    const double inv = 1.0 / tot;
#pragma unroll
                    for (int s = 0; s < Q; ++s) inc[s] *= inv;  // m^t_{l->i}
                }
            }
            edge_field<Q, false>(P, inc, 0.0, b);
            store_vec<Q>(&sb[le * Q], b);
        }
    }
    __syncthreads();

    // ---- phase 2: lane per row (a wave per row above BIG_ROW edges)
    double Sacc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) Sacc[q] = 0.0;
    double md = 0.0;
    auto finish_row = [&](int r, double di, double (&A)[Q], const int *ae /* per-component exponents of a long row, or null */,
                          const double *ft = nullptr /* the row's line of P->ftab (rows of <= FT_D edges under dc), or null */) {
        double pv[Q];
        double tot;
        if (ae) {
            int x[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) x[q] = ae[q];
            tot = apply_field_x<Q>(P, dc, di, A, x);
        } else {
            tot = apply_field<Q>(P, dc, di, A, ft);
        }
        if (!SHARD) store_vec<Q>(&sA[r * Q], A);
        const double inv = 1.0 / tot;
        const double gi = dc ? di : 1.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { pv[q] = A[q] * inv; Sacc[q] += gi * pv[q]; }
        store_vec<Q>(psi_new + size_t(r0 + r) * Q, pv);
        if (SHARD) {
            // phase 3 only needs the row vector up to a factor, so LDS gets the normalised marginal: the send pass below
            // (one lane per send slot, after the barrier) copies it from there
            store_vec<Q>(&sA[r * Q], pv);
#ifndef SBMBP_DEBUG_NOSEND  // (defined: ablation build without the send pass, timing only)
            for (uint32_t x = ssp[r]; x < ssp[r + 1]; ++x) {
                if (x < uint32_t(SLOTS)) {
                    ssrow[x] = uint16_t(r);
                } else {  // more slots than the LDS list holds: ship this one from here
                    double *dst = io.sendbuf + size_t(io.snd_slot[s0 + x]) * io.ncomp;
#pragma unroll
                    for (int q = 0; q < Q; ++q) if (q < io.ncomp) dst[q] = pv[q];
                }
            }
#endif
        }
    };
    for (int r = tid; r < nrows; r += frame_cfg<Q>::TPB) {
        const int es = int(srp[r]), ee = int(srp[r + 1]);
        if (CLAMP && sfl[r]) {
            double pv[Q];
            load_vec<Q>(psi_old + size_t(r0 + r) * Q, pv);
            store_vec<Q>(psi_new + size_t(r0 + r) * Q, pv);
            const double gi = dc ? double(ee - es) : 1.0;
#pragma unroll
            for (int q = 0; q < Q; ++q) Sacc[q] += gi * pv[q];
        } else if (ee - es <= BIG_ROW) {
            double A[Q], ft[Q];
            const bool tab = dc != 0 && ee - es <= FT_D;  // the field factors of this degree: loaded while the product runs
            if (tab) load_vec<Q>(P->ftab + size_t(ee - es) * QMAX, ft);
#pragma unroll
            for (int q = 0; q < Q; ++q) A[q] = 1.0;
            // (Round 3 measured two variants of this loop on C4 / the Q = 8 control / C3 and dropped both: the next factor
            // loaded before the current one is multiplied in, with the rescale every fourth factor: +1.5 % kernel time
            // everywhere; BIG_ROW = 64 / 128 / 256, i.e. fewer or no wave products: C4 +0.3 / +5 / +12 %.)
            for (int e = es; e < ee; ++e) {
                double b[Q];
                load_vec<Q>(&sb[e * Q], b);
#pragma unroll
                for (int q = 0; q < Q; ++q) A[q] *= b[q];
                rescale_pow2<Q>(A);
            }
            finish_row(r, double(ee - es), A, nullptr, tab ? ft : nullptr);
        }
    }
    if (sbig)  // uniform: written before the barrier that ends phase 1
    for (int r = tid >> 6; r < nrows; r += frame_cfg<Q>::WAVES) {  // wave-uniform row index
        const int es = int(srp[r]), ee = int(srp[r + 1]);
        if (ee - es > BIG_ROW && !(CLAMP && sfl[r])) {
            double A[Q];
            int ae[Q];
            row_product_wave<Q>(sb, es, ee, A, ae);
            if ((tid & 63) == 0) finish_row(r, double(ee - es), A, ae);
        }
    }
    __syncthreads();
#ifndef SBMBP_DEBUG_NOSEND
    if (SHARD) {  // send pass: one lane per send slot, stores in flight while phase 3 runs
        const uint32_t ns = min(ssp[nrows], uint32_t(SLOTS));
        for (uint32_t x = tid; x < ns; x += frame_cfg<Q>::TPB) {
            const double *src = &sA[int(ssrow[x]) * Q];
            double *dst = io.sendbuf + size_t(sslot[x]) * io.ncomp;
#pragma unroll
            for (int q = 0; q < Q; ++q) if (q < io.ncomp) dst[q] = src[q];
        }
    }
#endif

    // ---- phase 3: lane per directed edge: cavity, normalise, overwrite own slot
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int le = j * frame_cfg<Q>::TPB + tid;
        if (le < ne && !(CLAMP && sfl[srow[le]])) {
            const int r = srow[le];
            double A[Q], b[Q], cav[Q], out[Q];
            load_vec<Q>(&sA[r * Q], A);
            load_vec<Q>(&sb[le * Q], b);
            double tot = 0.0;
            double rb[Q];
#if SBMBP_NODIV
            if (!excl_products<Q>(b, rb))  // cavity up to a factor: A[q] prod_{q' != q} b[q'] (the one division left is 1 / tot)
#endif
            {
#if SBMBP_BATCH_RECIP
                recip_all<Q>(b, rb);
#else
#pragma unroll
                for (int q = 0; q < Q; ++q) rb[q] = 1.0 / b[q];
#endif
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) { cav[q] = A[q] * rb[q]; tot += cav[q]; }
            const double inv = 1.0 / tot;
            double ref[Q];
            if (exact) {  // uniform
                load_msg<Q>(Mcmp, size_t(e0 + le), ref);
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) ref[q] = mo[j][q];
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                out[q] = cav[q] * inv;
                md = nanmax(md, fabs(ref[q] - out[q]));
            }
            store_msg_stream<Q>(Mio, size_t(e0 + le), out);
        }
    }
    block_reduce_store<Q, frame_cfg<Q>::WAVES>(Sacc, md, sred, partials + size_t(bid) * (Q + 1));
}

// K1ph: marginal-gather form of the hub-row update (rows with degree > CAP), in FRAGMENTS of BLOCK edges.
// One workgroup per hub row is a latency chain as long as the row (neighbour index -> its marginal, per 256 edges, twice),
// the sweep waits for the longest row, and a degree-1e5 hub of a large power-law graph would run for milliseconds: measured
// on C4 (1106 hub rows holding 8 % of the edges, longest 2494) that kernel took 0.14 ms alone, a third of the frame kernel
// with 92 % of the edges. So a hub row is cut into fragments of BLOCK edges and updated by two short launches over all
// fragments of all hub rows: k_hub_frag_product (the edge fields of the fragment, kept in hub_b, and their product),
// k_hub_frag_cavity (the row product from the fragment products, the new marginal, the cavities of the fragment's edges).
// Every workgroup of a row multiplies the same fragment products in the same order, so they agree bitwise on the row
// product. The maximum message difference of the fragments meets in the row's partial record through atomicMax on the bit
// pattern (differences are >= 0 or NaN, and a NaN pattern is above every number: order-independent, so reproducible).
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_hub_frag_product(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ nbr, const double *__restrict__ Mio,
                   const double *__restrict__ psi_old, const uint32_t *__restrict__ hub_row, const uint32_t *__restrict__ hub_blk,
                   hub_frags hf, uint32_t frag_first, const dev_params *__restrict__ P, double *__restrict__ partials,
                   const int32_t *__restrict__ clamp, shard_io io, int first_from_psi) {
    if (P->stop) return;
    __shared__ double sAw[(BLOCK / 64) * Q];
    __shared__ int sEw[(BLOCK / 64) * Q];
    const int tid = threadIdx.x;
    const uint32_t f = frag_first + blockIdx.x, h = hf.frag_hub[f], k = f - hf.hub_frag0[h];
    const uint32_t i = hub_row[h];
    const uint32_t e0 = row_ptr[i], d = row_ptr[i + 1] - e0;
    if (k == 0 && tid == 0) partials[size_t(hub_blk[h]) * (Q + 1) + Q] = 0.0;  // the row's difference record: the fragments max into it
    if (clamp != nullptr && clamp[i] != -1) return;  // clamped hub (uniform): nothing to multiply
    const uint32_t le = k * BLOCK + tid;
    const bool ok = le < d;
    const uint32_t lc = ok ? le : d - 1;  // branch-free loads; the slot is masked below
    const uint32_t l = nbr[e0 + lc];
    double pl[Q], mo[Q], inc[Q], b[Q];
    if (io.halo_stage != nullptr && l >= io.n_own) load_halo_row<Q>(io, l - io.n_own, pl);
    else load_vec<Q>(psi_old + size_t(l) * Q, pl);
    if (first_from_psi) {  // uniform
#pragma unroll
        for (int s = 0; s < Q; ++s) inc[s] = pl[s];
    } else {
        double bo[Q];
        load_msg<Q>(Mio, size_t(e0 + lc), mo);
        edge_field<Q, false>(P, mo, 0.0, bo);
        double tot = 0.0;
#pragma unroll
        for (int s = 0; s < Q; ++s) { inc[s] = pl[s] / bo[s]; tot += inc[s]; }
        const double inv = 1.0 / tot;
#pragma unroll
        for (int s = 0; s < Q; ++s) inc[s] *= inv;
    }
    edge_field<Q, false>(P, inc, 0.0, b);
    if (ok) store_vec<Q>(hf.b + (size_t(f) * BLOCK + tid) * Q, b);
    double A[Q];
    int ae[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = ok ? b[q] : 1.0; ae[q] = 0; }
    block_product_shfl<Q, BLOCK / 64>(A, ae, sAw, sEw);
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { hf.pA[size_t(f) * Q + q] = A[q]; hf.pE[size_t(f) * Q + q] = ae[q]; }
    }
}
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_hub_frag_cavity(const uint32_t *__restrict__ row_ptr, double *__restrict__ Mio, const double *__restrict__ psi_old,
                  double *__restrict__ psi_new, const uint32_t *__restrict__ hub_row, const uint32_t *__restrict__ hub_blk,
                  hub_frags hf, uint32_t frag_first, const dev_params *__restrict__ P, int dc, double *__restrict__ partials,
                  const int32_t *__restrict__ clamp, shard_io io, const double *__restrict__ Mcmp) {
    if (P->stop) return;
    const int exact = P->exact;
    __shared__ double sAw[(BLOCK / 64) * Q];
    __shared__ int sEw[(BLOCK / 64) * Q];
    __shared__ double smd[BLOCK / 64];
    const int tid = threadIdx.x;
    const uint32_t f = frag_first + blockIdx.x, h = hf.frag_hub[f], f0 = hf.hub_frag0[h], nf = hf.hub_frag0[h + 1] - f0;
    const uint32_t i = hub_row[h];
    const uint32_t e0 = row_ptr[i], d = row_ptr[i + 1] - e0;
    const double di = double(d);
    double *rec = partials + size_t(hub_blk[h]) * (Q + 1);
    if (clamp != nullptr && clamp[i] != -1) {  // clamped hub (uniform): marginal copied, messages stay
        if (f == f0 && tid == 0) {
            double pv[Q];
            load_vec<Q>(psi_old + size_t(i) * Q, pv);
            store_vec<Q>(psi_new + size_t(i) * Q, pv);
#pragma unroll
            for (int q = 0; q < Q; ++q) rec[q] = (dc ? di : 1.0) * pv[q];
        }
        return;
    }
    double A[Q];
    int ae[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = 1.0; ae[q] = 0; }
    for (uint32_t x = tid; x < nf; x += BLOCK) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] *= hf.pA[size_t(f0 + x) * Q + q]; ae[q] += hf.pE[size_t(f0 + x) * Q + q]; }
        x_norm<Q>(A, ae);
    }
    block_product_shfl<Q, BLOCK / 64>(A, ae, sAw, sEw);
    const double tot = apply_field_x<Q>(P, dc, di, A, ae);
    const double inv = 1.0 / tot;
    if (f == f0 && tid == 0) {
        double pv[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { pv[q] = A[q] * inv; rec[q] = (dc ? di : 1.0) * pv[q]; }
        store_vec<Q>(psi_new + size_t(i) * Q, pv);
        if (io.snd_ptr != nullptr) send_row<Q>(io, i, pv);
    }
    double md = 0.0;
    const uint32_t le = (f - f0) * BLOCK + tid;
    if (le < d) {
        double b[Q], ref[Q], out[Q], cav[Q];
        load_vec<Q>(hf.b + (size_t(f) * BLOCK + tid) * Q, b);
        load_msg<Q>(exact ? Mcmp : Mio, size_t(e0 + le), ref);  // exact 1-step difference, else the 2-step hint
        double ct = 0.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { cav[q] = A[q] / b[q]; ct += cav[q]; }
        const double ci = 1.0 / ct;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            out[q] = cav[q] * ci;
            md = nanmax(md, fabs(ref[q] - out[q]));
        }
        store_msg<Q>(Mio, size_t(e0 + le), out);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) md = nanmax(md, __shfl_xor(md, o, 64));
    if ((tid & 63) == 0) smd[tid >> 6] = md;
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) md = nanmax(md, smd[w]);
        atomicMax(reinterpret_cast<unsigned long long *>(rec + Q), (unsigned long long)__double_as_longlong(fabs(md)));
    }
}

// gather rows idx[0..n) of a [rows][Q] table into a contiguous buffer (halo send packing). With
// ncomp = Q-1 the last component is dropped: marginals sum to 1, the receiver restores it.
__global__ void __launch_bounds__(BLOCK)
k_pack_rows(const double *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t n, int Q, int ncomp,
            double *__restrict__ out) {
    const uint64_t t = uint64_t(blockIdx.x) * BLOCK + threadIdx.x;
    if (t >= uint64_t(n) * ncomp) return;
    const uint32_t r = uint32_t(t / ncomp), q = uint32_t(t % ncomp);
    out[t] = table[size_t(idx[r]) * Q + q];
}

// expand n received rows of ncomp components into table rows row0 + dst[r]: last = 1 - sum (clamped at 0)
__global__ void __launch_bounds__(BLOCK)
k_unpack_rows(const double *__restrict__ in, uint32_t n, int Q, int ncomp, double *__restrict__ table, uint32_t row0,
              const uint32_t *__restrict__ dst) {
    const uint32_t r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const size_t row = size_t(row0) + dst[r];
    double s = 0.0;
    for (int q = 0; q < ncomp; ++q) {
        const double v = in[size_t(r) * ncomp + q];
        table[row * Q + q] = v;
        s += v;
    }
    if (ncomp < Q) table[row * Q + (Q - 1)] = fmax(0.0, 1.0 - s);
}

// exact 1-step criterion of converge (bp.cpp:1059-1063): max over all message entries |a - b| (records decoded)
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_msg_diff(const double *__restrict__ a, const double *__restrict__ b, uint64_t n_msg, double *__restrict__ partials) {
    __shared__ double sred[4 * 2];
    double md = 0.0;
    for (uint64_t k = uint64_t(blockIdx.x) * BLOCK + threadIdx.x; k < n_msg; k += uint64_t(gridDim.x) * BLOCK) {
        double va[Q], vb[Q];
        load_msg<Q>(a, size_t(k), va);
        load_msg<Q>(b, size_t(k), vb);
#pragma unroll
        for (int q = 0; q < Q; ++q) md = nanmax(md, fabs(va[q] - vb[q]));
    }
    double dummy[1] = {0.0};
    block_reduce_store<1, 4>(dummy, md, sred, partials + size_t(blockIdx.x) * 2);
}

// ------------------------------------------------------------------------------------------------
// K1h: the message-gather update of the hub rows (degree > CAP), in the same fragments of BLOCK edges as the
// marginal-gather form (k_hub_frag_product / k_hub_frag_cavity above): the product launch gathers the incoming messages
// through rev, the cavity launch damps and writes the other message buffer. Differences reported are 1-step differences.
// ------------------------------------------------------------------------------------------------
template <int Q, bool DC2>
__global__ void __launch_bounds__(BLOCK)
k_hub_frag_product_msg(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rev, const uint32_t *__restrict__ nbr,
                       const uint32_t *__restrict__ ndeg /* degree of every table row (DC2 only) */, const double *__restrict__ Mold,
                       const uint32_t *__restrict__ hub_row, const uint32_t *__restrict__ hub_blk, hub_frags hf, uint32_t frag_first,
                       const dev_params *__restrict__ P, double *__restrict__ partials, const int32_t *__restrict__ clamp) {
    if (P->stop) return;
    __shared__ double sAw[(BLOCK / 64) * Q];
    __shared__ int sEw[(BLOCK / 64) * Q];
    const int tid = threadIdx.x;
    const uint32_t f = frag_first + blockIdx.x, h = hf.frag_hub[f], k = f - hf.hub_frag0[h];
    const uint32_t i = hub_row[h];
    const uint32_t e0 = row_ptr[i], d = row_ptr[i + 1] - e0;
    if (k == 0 && tid == 0) partials[size_t(hub_blk[h]) * (Q + 1) + Q] = 0.0;  // the row's difference record: the fragments max into it
    if (clamp != nullptr && clamp[i] != -1) return;  // clamped hub (uniform)
    const uint32_t le = k * BLOCK + tid;
    const bool ok = le < d;
    const uint32_t lc = ok ? le : d - 1;
    double m[Q], b[Q];
    load_msg<Q>(Mold, rev[e0 + lc], m);
    double didl = 0.0;
    if (DC2) { const uint32_t l = nbr[e0 + lc]; didl = double(d) * double(ndeg[l]); }
    edge_field<Q, DC2>(P, m, didl, b);
    if (ok) store_vec<Q>(hf.b + (size_t(f) * BLOCK + tid) * Q, b);
    double A[Q];
    int ae[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = ok ? b[q] : 1.0; ae[q] = 0; }
    block_product_shfl<Q, BLOCK / 64>(A, ae, sAw, sEw);
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { hf.pA[size_t(f) * Q + q] = A[q]; hf.pE[size_t(f) * Q + q] = ae[q]; }
    }
}
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_hub_frag_cavity_msg(const uint32_t *__restrict__ row_ptr, const double *__restrict__ Mold, double *__restrict__ Mnew,
                      const double *__restrict__ psi_old, double *__restrict__ psi, const uint32_t *__restrict__ hub_row,
                      const uint32_t *__restrict__ hub_blk, hub_frags hf, uint32_t frag_first, const dev_params *__restrict__ P, int dc,
                      double damp, double *__restrict__ partials, const int32_t *__restrict__ clamp) {
    if (P->stop) return;
    __shared__ double sAw[(BLOCK / 64) * Q];
    __shared__ int sEw[(BLOCK / 64) * Q];
    __shared__ double smd[BLOCK / 64];
    const int tid = threadIdx.x;
    const uint32_t f = frag_first + blockIdx.x, h = hf.frag_hub[f], f0 = hf.hub_frag0[h], nf = hf.hub_frag0[h + 1] - f0;
    const uint32_t i = hub_row[h];
    const uint32_t e0 = row_ptr[i], d = row_ptr[i + 1] - e0;
    const double di = double(d);
    const uint32_t le = (f - f0) * BLOCK + tid;
    double *rec = partials + size_t(hub_blk[h]) * (Q + 1);
    if (clamp != nullptr && clamp[i] != -1) {  // clamped hub (uniform): marginal and out-messages stay as initialised (bp.cpp:1115-1124)
        if (le < d) {
            double m[Q];
            load_msg<Q>(Mold, size_t(e0 + le), m);
            store_msg<Q>(Mnew, size_t(e0 + le), m);
        }
        if (f == f0 && tid == 0) {
            double pv[Q];
            load_vec<Q>(psi_old + size_t(i) * Q, pv);
            store_vec<Q>(psi + size_t(i) * Q, pv);
#pragma unroll
            for (int q = 0; q < Q; ++q) rec[q] = (dc ? di : 1.0) * pv[q];
        }
        return;
    }
    double A[Q];
    int ae[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = 1.0; ae[q] = 0; }
    for (uint32_t x = tid; x < nf; x += BLOCK) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] *= hf.pA[size_t(f0 + x) * Q + q]; ae[q] += hf.pE[size_t(f0 + x) * Q + q]; }
        x_norm<Q>(A, ae);
    }
    block_product_shfl<Q, BLOCK / 64>(A, ae, sAw, sEw);
    const double tot = apply_field_x<Q>(P, dc, di, A, ae);
    const double inv = 1.0 / tot;
    if (f == f0 && tid == 0) {
        double pv[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { pv[q] = A[q] * inv; rec[q] = (dc ? di : 1.0) * pv[q]; }
        store_vec<Q>(psi + size_t(i) * Q, pv);
    }
    double md = 0.0;
    const int probe2 = P->ar_probe2;  // as in k_sweep
    damp *= P->damp_auto;
    if (le < d) {
        double b[Q], mo[Q], out[Q], cav[Q], ref[Q];
        load_vec<Q>(hf.b + (size_t(f) * BLOCK + tid) * Q, b);
        load_msg<Q>(Mold, size_t(e0 + le), mo);
        if (probe2) {  // uniform
            load_msg<Q>(Mnew, size_t(e0 + le), ref);
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) ref[q] = mo[q];
        }
        double ct = 0.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { cav[q] = (b[q] > 0.0) ? A[q] / b[q] : 0.0; ct += cav[q]; }
        const double ci = 1.0 / ct;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const double nv = cav[q] * ci;
            out[q] = damp * nv + (1.0 - damp) * mo[q];
            md = nanmax(md, probe2 ? fabs(ref[q] - out[q]) / damp : fabs(ref[q] - nv));
        }
        store_msg<Q>(Mnew, size_t(e0 + le), out);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) md = nanmax(md, __shfl_xor(md, o, 64));
    if ((tid & 63) == 0) smd[tid >> 6] = md;
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) md = nanmax(md, smd[w]);
        atomicMax(reinterpret_cast<unsigned long long *>(rec + Q), (unsigned long long)__double_as_longlong(fabs(md)));
    }
}

// ------------------------------------------------------------------------------------------------
// sum_i g_i psi_i[q] over row chunks (init_h, belief_propagation.cpp:320-332); chunk c covers rows
// [c*rows_per_blk, ...). Writes partials[c*(Q+1) + q]; slot Q (max) = 0.
// ------------------------------------------------------------------------------------------------
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_psi_sum(const uint32_t *__restrict__ row_ptr, const double *__restrict__ psi, uint32_t n_rows, uint32_t rows_per_blk,
          int dc, double *__restrict__ partials) {
    __shared__ double sred[4 * (Q + 1)];
    double S[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) S[q] = 0.0;
    const uint32_t lo = blockIdx.x * rows_per_blk;
    const uint32_t hi = min(n_rows, lo + rows_per_blk);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += BLOCK) {
        double pv[Q];
        load_vec<Q>(psi + size_t(i) * Q, pv);
        const double gi = dc ? double(row_ptr[i + 1] - row_ptr[i]) : 1.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) S[q] += gi * pv[q];
    }
    block_reduce_store<Q>(S, 0.0, sred, partials + size_t(blockIdx.x) * (Q + 1));
}

// ------------------------------------------------------------------------------------------------
// K2: fold the workgroup partials in a fixed order, relax the field, refresh h/exph, record the
// convergence state. mode 0: after a sweep; mode 1: field initialisation (no sweep bookkeeping);
// mode 2: exact field refresh (no relaxation, no bookkeeping).
// ------------------------------------------------------------------------------------------------
// fold of rows [lo, hi) of a [.][Q + 1] record table (Q sums, then a sticky-NaN max) by the workgroup, in a fixed order;
// lane 0 leaves the result in out[0..Q] (LDS), visible to all after the closing barrier
template <int Q, bool SC1>
__device__ __forceinline__ void fold_rows(const double *rows, uint32_t lo, uint32_t hi, double *sacc /* LDS [BLOCK/64][Q+1] */, double *out) {
    const int tid = threadIdx.x;
    double acc[Q], mx = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0.0;
    for (uint32_t r = lo + tid; r < hi; r += BLOCK) {
        const double *rec = rows + size_t(r) * (Q + 1);
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] += SC1 ? __hip_atomic_load(rec + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rec[q];
        mx = nanmax(mx, SC1 ? __hip_atomic_load(rec + Q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rec[Q]);
    }
    block_reduce_store<Q, BLOCK / 64>(acc, mx, sacc, out);
    __syncthreads();
}
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_finalize(const double *__restrict__ partials, uint32_t n_part, int mode, dev_params *__restrict__ P,
           double *__restrict__ diff_hist, uint32_t hist_cap, int md_exact /* the sweep reported 1-step differences by nature */) {
    if (mode == 0 && P->stop) return;
    __shared__ double sacc[(BLOCK / 64) * (Q + 1)];
    __shared__ double sout[Q + 1];
    __shared__ double s_hN[Q];
    fold_rows<Q, false>(partials, 0, n_part, sacc, sout);
    if (threadIdx.x == 0) finalize_update<Q>(P, sout, mode, diff_hist, hist_cap, md_exact, s_hN);
    __syncthreads();
    field_table<Q>(P, s_hN);
}

// K2 with its fold in ONE launch (single engine, after a sweep): workgroup b folds the contiguous chunk b of the sweep's
// records into stage row b; the workgroup that arrives LAST at the counter folds the stage rows and runs the update. Who
// that is depends on timing, what it reads and in which order does not, so the result is bitwise reproducible. Two
// launches per sweep (k_fold_stage, k_finalize) were 5 + 6 us on C2 and 11 + 21 us on C4 plus a launch gap, next to sweep
// kernels of 50 and 450 us. (The same fold inside the SWEEP launch was tried and dropped: the record must have left
// the CU before the counter moves, so every sweep workgroup drained its message stores and waited for an atomic's return
// before it could retire: C3 2.49 -> 2.57 ms, C2 50 -> 73 us.)
// Visibility between the workgroups follows MI355X_MICROARCH.md (sc1 hand-off with one unsharded counter): stage rows
// stored sc1 by one lane, that lane's s_waitcnt vmcnt(0), then its agent-scope atomic add; the workgroup whose add came
// last reads the rows with sc1 loads behind a workgroup barrier.
// workgroup b folds the contiguous chunk b of a [rows][Q + 1] record table into out row b (the per-sweep fold of a shard:
// SBMBP_FOLD_ROWS rows per rank go into the all-gather). Compile-time Q: the generic k_fold_stage walks the columns one by
// one with a barrier tree each, 12 us where this takes 4.
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_fold_records(const double *__restrict__ partials, uint32_t rows, uint32_t chunk, double *__restrict__ out) {
    __shared__ double sacc[(BLOCK / 64) * (Q + 1)];
    __shared__ double sout[Q + 1];
    const uint32_t lo = min(rows, blockIdx.x * chunk), hi = min(rows, lo + chunk);
    fold_rows<Q, false>(partials, lo, hi, sacc, sout);
    if (threadIdx.x <= Q) out[size_t(blockIdx.x) * (Q + 1) + threadIdx.x] = sout[threadIdx.x];
}

template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_fold_finalize(const double *__restrict__ partials, uint32_t n_part, uint32_t chunk, dev_params *__restrict__ P,
                double *__restrict__ diff_hist, uint32_t hist_cap, int md_exact, double *__restrict__ stage /* [gridDim.x][Q + 1] */,
                uint32_t *__restrict__ counter /* zero between launches */) {
    if (P->stop) return;
    __shared__ double sacc[(BLOCK / 64) * (Q + 1)];
    __shared__ double sout[Q + 1];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const uint32_t lo = blockIdx.x * chunk, hi = min(n_part, lo + chunk);
    fold_rows<Q, false>(partials, lo, hi, sacc, sout);
    if (gridDim.x > 1) {
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q <= Q; ++q) __hip_atomic_store(stage + size_t(blockIdx.x) * (Q + 1) + q, sout[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row has left this CU before the counter moves
            s_last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;  // uniform
        fold_rows<Q, true>(stage, 0, gridDim.x, sacc, sout);
        if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __shared__ double s_hN[Q];
    if (tid == 0) finalize_update<Q>(P, sout, 0, diff_hist, hist_cap, md_exact, s_hN);
    __syncthreads();  // (uniform: only the last workgroup is still here)
    field_table<Q>(P, s_hN);
}

// ------------------------------------------------------------------------------------------------
// K3: site and edge terms of the Bethe free energy and of the entropy, one pass over the frame.
// partials[b*NP + ...] = { sum_i log Z_i, sum_e log(m_in^T W m_out), e_site numerator sum,
//                          e_edge sum }                      
// Hub rows are covered by k_fe_hub.
// ------------------------------------------------------------------------------------------------
constexpr int FE_NP = 4;

// edge terms shared by the frame and hub kernels. Wf = weights of f_edge: cab^beta (dc 0),
// cab (dc 1; the d_i d_l prefactor is a graph constant added on the host), x/(1+x) (dc 2).
template <int Q, bool DC2>
__device__ __forceinline__ void edge_terms(const dev_params *__restrict__ P, const double (&mi)[Q], const double (&mo)[Q],
                                           double didl, int want_entropy, double &log_norm, double &ent) {
    double nl = 0.0, num = 0.0, den = 0.0;
#pragma unroll
    for (int q1 = 0; q1 < Q; ++q1) {
#pragma unroll
        for (int q2 = 0; q2 < Q; ++q2) {
            const double pr = mi[q1] * mo[q2];
            double wf = P->W[q1 * Q + q2];
            double we = P->cab[q1 * Q + q2];  // entropy uses cab without beta (bp.cpp:628-666)
            if (DC2) { double x = didl * wf; wf = x / (1.0 + x); we = wf; }
            nl += wf * pr;
            if (want_entropy) {  // uniform
                den += we * pr;
                num += we * P->logcab[q1 * Q + q2] * pr;
            }
        }
    }
    log_norm = log(nl);
    ent = want_entropy ? num / den : 0.0;
}

template <int Q, bool DC2>
__global__ void __launch_bounds__(frame_cfg<Q>::TPB)
k_fe_frame(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rev, const uint32_t *__restrict__ nbr, const uint32_t *__restrict__ ndeg /* degree of every table row (DC2 only) */,
           const double *__restrict__ M, const double *__restrict__ Min /* incoming messages in edge order, or null: gather M[rev] */,
           const uint32_t *__restrict__ blk_row, const uint32_t *__restrict__ blk_e0, const dev_params *__restrict__ P,
           int dc, int want_entropy, double *__restrict__ partials) {
    constexpr int EPT = frame_cfg<Q>::EPT, CAP = frame_cfg<Q>::CAP, RCAP = frame_cfg<Q>::RCAP;
    __shared__ double sb[CAP * Q];
    __shared__ double sc[CAP * Q];  // entropy: b with plain cab weights (no beta)
    __shared__ uint32_t srp[RCAP + 1];
    __shared__ uint16_t srow[CAP];
    __shared__ double sred[frame_cfg<Q>::WAVES * (FE_NP + 1)];
    const int tid = threadIdx.x;
    // as in the sweep kernels: segment bounds from one level of scalar loads, then every load of the lane-per-edge phase is
    // issued branch-free and back to back (reverse index, own record, row offsets, then the gathers) before anything waits
    const uint32_t r0 = blk_row[blockIdx.x], r1 = blk_row[blockIdx.x + 1];
    const int nrows = int(r1 - r0);
    const uint32_t e0 = blk_e0[blockIdx.x];
    const int ne = int(blk_e0[blockIdx.x + 1] - e0);
    double acc[FE_NP] = {0.0, 0.0, 0.0, 0.0};
    if (ne <= CAP) {
        constexpr int RPT = RCAP / frame_cfg<Q>::TPB + 1;
        uint32_t kk[EPT], rk[EPT], rpv[RPT];
        double mo_[EPT][Q], mi_[EPT][Q];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int le = j * frame_cfg<Q>::TPB + tid;
            kk[j] = (ne > 0) ? e0 + uint32_t(le < ne ? le : 0) : 0u;
        }
        if (Min == nullptr) {  // uniform
#pragma unroll
            for (int j = 0; j < EPT; ++j) rk[j] = rev[kk[j]];
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j) load_msg<Q>(M, size_t(kk[j]), mo_[j]);
#pragma unroll
        for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; rpv[t] = row_ptr[r0 + uint32_t(r < nrows ? r : nrows)]; }
#pragma unroll
        for (int j = 0; j < EPT; ++j) load_msg<Q>(Min ? Min : M, Min ? size_t(kk[j]) : size_t(rk[j]), mi_[j]);
#pragma unroll
        for (int t = 0; t < RPT; ++t) { const int r = tid + t * frame_cfg<Q>::TPB; if (r <= nrows) srp[r] = rpv[t] - e0; }
        __syncthreads();
        if (DC2) {  // per-edge weights need the edge -> row map
            for (int r = tid; r < nrows; r += frame_cfg<Q>::TPB)
                for (int e = int(srp[r]); e < int(srp[r + 1]); ++e) srow[e] = uint16_t(r);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int le = j * frame_cfg<Q>::TPB + tid;
            if (le < ne) {
                double (&mi)[Q] = mi_[j];
                double (&mo)[Q] = mo_[j];
                double b[Q];
                double didl = 0.0;
                if (DC2) {
                    const int r = srow[le];
                    const uint32_t l = nbr[e0 + le];
                    didl = double(srp[r + 1] - srp[r]) * double(ndeg[l]);
                }
                edge_field<Q, DC2>(P, mi, didl, b);
                store_vec<Q>(&sb[le * Q], b);
                if (want_entropy) {
                    double c[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        double a = 0.0;
#pragma unroll
                        for (int t = 0; t < Q; ++t) a += P->cab[t * Q + q] * mi[t];
                        c[q] = a;
                    }
                    store_vec<Q>(&sc[le * Q], c);
                }
                double ln, en;
                edge_terms<Q, DC2>(P, mi, mo, didl, want_entropy, ln, en);
                acc[1] += ln;
                if (want_entropy) acc[3] += en;
            }
        }
        __syncthreads();
        for (int r = tid; r < nrows; r += frame_cfg<Q>::TPB) {
            const int es = int(srp[r]), ee = int(srp[r + 1]);
            const double di = double(ee - es);
            double A[Q];
            int ae[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) { A[q] = 1.0; ae[q] = 0; }
            for (int e = es; e < ee; ++e) {
#pragma unroll
                for (int q = 0; q < Q; ++q) A[q] *= sb[e * Q + q];
                if (((e - es) & 7) == 7) x_norm<Q>(A, ae);  // eight factors of O(W) stay far inside the double range
            }
            x_norm<Q>(A, ae);
            acc[0] += log_partition_x<Q>(P, dc, di, A, ae);  // log Z_i  (bp.cpp:446-502)
            if (want_entropy) {  // e_site (bp.cpp:506-560): no beta, weights exp(a + log eta - h/N)
                double C[Q];
                int ce[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) { C[q] = 1.0; ce[q] = 0; }
                for (int e = es; e < ee; ++e) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) C[q] *= sc[e * Q + q];
                    if (((e - es) & 7) == 7) x_norm<Q>(C, ce);
                }
                x_norm<Q>(C, ce);
                acc[2] += entropy_site_x<Q>(P, C, ce);
            }
        }
    }
    block_reduce_store<FE_NP, frame_cfg<Q>::WAVES>(acc, 0.0, sred, partials + size_t(blockIdx.x) * (FE_NP + 1));
}

// ------------------------------------------------------------------------------------------------
// K3p: the reductions of one inference / EM step in ONE pass, on the marginal-gather reconstruction (single engine, dc 0/1,
// every cab entry > 0, psi consistent with the two message buffers: exactly the state a converge call leaves behind).
// k_fe_frame / k_em_edges gather the incoming message M[rev[k]]: a 8(Q-1)-byte record out of the whole message array (2.4 GB
// at C3), one or two 64-byte fabric requests each, once per kernel. Here it is reconstructed as in k_sweep_psi,
//     m^T_{l->i}  ∝  psi^T_l / (W^T m^{T-1}_{i->l}),
// from the neighbour's marginal (the N*Q table, 1/c the size, cache resident with the XCD-aware numbering) and the row's own
// previous record (streamed), and everything that needs (m_in, m_out) of an edge or psi_l is formed while it is in registers:
//   FE_NP  site and edge terms of the free energy and the entropy              (k_fe_frame)
//   NE_NP  the adjacent pairs of the non-edge term, which gather psi_l too     (k_nonedge_adj / k_nonedge_exact_adj; dc 0)
//   EM     the Q(Q+1)/2 numerators of cab_expect                               (k_em_edges; template EM, Q <= 8)
// Record layout: [FE_NP | NE_NP | T] sums then one unused max slot. Hub rows (above one segment's capacity): their site and
// edge terms come from k_fe_hub (records behind the segments'); this kernel adds their adjacent-pair and EM terms.
// ------------------------------------------------------------------------------------------------
constexpr int NE_NP = 2;
template <int Q, bool EM> struct fe_psi_cfg {
    static constexpr int T = EM ? Q * (Q + 1) / 2 : 0;
    static constexpr int NP = FE_NP + NE_NP + T;
};
#ifndef SBMBP_FE_WAVES_Q8
#define SBMBP_FE_WAVES_Q8 2   // k_fe_psi<8>: 356 - 470 registers left alone; at 256 (two waves per SIMD) the C4 pass takes 1.00 instead of 1.43 ms
#endif
template <int Q> struct fe_waves { static constexpr int N = Q == 8 ? SBMBP_FE_WAVES_Q8 : 1; };  // register target (k_sweep_psi)
template <int Q, bool EM>
__global__ void __launch_bounds__(frame_cfg<Q>::TPB) __attribute__((amdgpu_waves_per_eu(fe_waves<Q>::N)))
k_fe_psi(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ nbr, const double *__restrict__ Mcur /* m^T */,
         const double *__restrict__ Mprev /* m^{T-1} */, const double *__restrict__ psi /* psi^T */,
         const uint32_t *__restrict__ blk_row, const uint32_t *__restrict__ blk_e0, const dev_params *__restrict__ P, int dc,
         int want_entropy, int adj_mode /* 0: no non-edge term (dc != 0), 1: series weights (wmat), 2: exact (pmat) */,
         const double *__restrict__ wmat /* mode 1: N(1-(1-cab/N)^beta); mode 2: (1-cab/N)^beta */, uint32_t n_seg,
         double *__restrict__ partials) {
    constexpr int EPT = frame_cfg<Q>::EPT, CAP = frame_cfg<Q>::CAP, RCAP = frame_cfg<Q>::RCAP, TPB = frame_cfg<Q>::TPB;
    constexpr int T = fe_psi_cfg<Q, EM>::T, NP = fe_psi_cfg<Q, EM>::NP;
    __shared__ double sb[CAP * Q];
    __shared__ double sc[CAP * Q];  // entropy: b with plain cab weights (no beta)
    __shared__ uint32_t srp[RCAP + 1];
    __shared__ uint16_t srow[CAP];
    __shared__ double sred[frame_cfg<Q>::WAVES * (NP + 1)];
    __shared__ double sclc[Q * Q];  // cab log cab (entropy of the adjacent pairs)
    const int tid = threadIdx.x;
#if SBMBP_XCD_REMAP
    const uint32_t per = gridDim.x / 8;  // grid padded by xcd_grid: XCD x takes the contiguous segments [x per, (x+1) per)
    const uint32_t bid = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (bid >= n_seg) return;
#else
    const uint32_t bid = blockIdx.x;
#endif
    const uint32_t r0 = blk_row[bid], r1 = blk_row[bid + 1];
    const int nrows = int(r1 - r0);
    const uint32_t e0 = blk_e0[bid];
    const int ne = int(blk_e0[bid + 1] - e0);
    const double invN = P->invN;
    double acc[NP];
#pragma unroll
    for (int x = 0; x < NP; ++x) acc[x] = 0.0;
    if (want_entropy && adj_mode)
        for (int x = tid; x < Q * Q; x += TPB) sclc[x] = P->cab[x] * P->logcab[x];

    // everything of one directed edge that needs its two messages or the neighbour's marginal; `site`: also b (and c) to LDS
    auto edge = [&](uint32_t k, int le, const double (&mo)[Q], const double (&mp)[Q], const double (&pl)[Q], const double *pi, bool site) {
        double bo[Q], mi[Q], rb[Q];
        edge_field<Q, false>(P, mp, 0.0, bo);  // what l saw of i's message at T-1
        double tot = 0.0;
        if (!excl_products<Q>(bo, rb)) {
#pragma unroll
            for (int q = 0; q < Q; ++q) rb[q] = 1.0 / bo[q];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) { mi[q] = pl[q] * rb[q]; tot += mi[q]; }
        const double inv = 1.0 / tot;
#pragma unroll
        for (int q = 0; q < Q; ++q) mi[q] *= inv;  // m^T_{l->i}
        if (site) {
            double b[Q];
            edge_field<Q, false>(P, mi, 0.0, b);
            store_vec<Q>(&sb[le * Q], b);
            if (want_entropy) {
                double c[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    double a = 0.0;
#pragma unroll
                    for (int t = 0; t < Q; ++t) a += P->cab[t * Q + q] * mi[t];
                    c[q] = a;
                }
                store_vec<Q>(&sc[le * Q], c);
            }
            double ln, en;
            edge_terms<Q, false>(P, mi, mo, 0.0, want_entropy, ln, en);
            acc[1] += ln;
            if (want_entropy) acc[3] += en;
        }
        if (adj_mode) {  // adjacent pairs of the non-edge term (bp.cpp:675-741 sums over NON-adjacent pairs: these are taken out)
            double y = 0.0, yc = 0.0, u = 0.0;
#pragma unroll
            for (int q1 = 0; q1 < Q; ++q1) {
#pragma unroll
                for (int q2 = 0; q2 < Q; ++q2) {
                    const double pp = pi[q1] * pl[q2];
                    y += wmat[q1 * Q + q2] * pp;
                    if (want_entropy) {
                        yc += P->cab[q1 * Q + q2] * pp;
                        u += sclc[q1 * Q + q2] * pp;
                    }
                }
            }
            const double num = u * invN, den = 1.0 - yc * invN;
            if (adj_mode == 1) {  // series form (k_nonedge_adj): y = sum N (1 - (1 - cab/N)^beta) pp
                acc[FE_NP + 0] += log1p(-y * invN);
                if (want_entropy) acc[FE_NP + 1] += num / den;
            } else {              // exact form (k_nonedge_exact_adj): y = sum (1 - cab/N)^beta pp, with the tiled kernel's guards
                if (y != 0.0) acc[FE_NP + 0] += log(y);
                if (want_entropy && num * den != 0.0) acc[FE_NP + 1] += num / den;
            }
        }
        if (EM) {  // numerators of cab_expect (bp.cpp:892-989)
            double term[T > 0 ? T : 1], norm_L = 0.0;
            int t = 0;
#pragma unroll
            for (int q1 = 0; q1 < Q; ++q1) {
#pragma unroll
                for (int q2 = q1; q2 < Q; ++q2, ++t) {
                    const double pr = (q1 == q2) ? (mi[q1] * mo[q2]) : (mi[q1] * mo[q2] + mi[q2] * mo[q1]);
                    term[t] = P->cab[q1 * Q + q2] * pr;
                    norm_L += term[t];
                }
            }
            const double hin = 0.5 / norm_L;
#pragma unroll
            for (int x = 0; x < T; ++x) acc[FE_NP + NE_NP + x] += term[x] * hin;
        }
        (void)k;
    };

    if (ne <= CAP) {
        constexpr int RPT = RCAP / TPB + 1;
        uint32_t kk[EPT], nl[EPT], rpv[RPT];
        double mo_[EPT][Q], mp_[EPT][Q], pl_[EPT][Q];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int le = j * TPB + tid;
            kk[j] = (ne > 0) ? e0 + uint32_t(le < ne ? le : 0) : 0u;
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j) nl[j] = nbr[kk[j]];
#pragma unroll
        for (int j = 0; j < EPT; ++j) load_msg<Q>(Mcur, size_t(kk[j]), mo_[j]);
#pragma unroll
        for (int j = 0; j < EPT; ++j) load_msg<Q>(Mprev, size_t(kk[j]), mp_[j]);
#pragma unroll
        for (int t = 0; t < RPT; ++t) { const int r = tid + t * TPB; rpv[t] = row_ptr[r0 + uint32_t(r < nrows ? r : nrows)]; }
#pragma unroll
        for (int j = 0; j < EPT; ++j) load_vec<Q>(psi + size_t(nl[j]) * Q, pl_[j]);
#pragma unroll
        for (int t = 0; t < RPT; ++t) { const int r = tid + t * TPB; if (r <= nrows) srp[r] = rpv[t] - e0; }
        __syncthreads();
        if (adj_mode) {  // the adjacent pairs need the row of every edge (its own marginal)
            for (int r = tid; r < nrows; r += TPB)
                for (int e = int(srp[r]); e < int(srp[r + 1]); ++e) srow[e] = uint16_t(r);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int le = j * TPB + tid;
            if (le < ne) {
                double pi[Q];
                if (adj_mode) load_vec<Q>(psi + size_t(r0 + srow[le]) * Q, pi);
                edge(kk[j], le, mo_[j], mp_[j], pl_[j], pi, true);
            }
        }
        __syncthreads();
        for (int r = tid; r < nrows; r += TPB) {
            const int es = int(srp[r]), ee = int(srp[r + 1]);
            const double di = double(ee - es);
            double A[Q];
            int ae[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) { A[q] = 1.0; ae[q] = 0; }
            for (int e = es; e < ee; ++e) {
#pragma unroll
                for (int q = 0; q < Q; ++q) A[q] *= sb[e * Q + q];
                if (((e - es) & 7) == 7) x_norm<Q>(A, ae);  // eight factors of O(W) stay far inside the double range
            }
            x_norm<Q>(A, ae);
            acc[0] += log_partition_x<Q>(P, dc, di, A, ae);  // log Z_i  (bp.cpp:446-502)
            if (want_entropy) {  // e_site (bp.cpp:506-560): no beta, weights exp(a + log eta - h/N)
                double C[Q];
                int ce[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) { C[q] = 1.0; ce[q] = 0; }
                for (int e = es; e < ee; ++e) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) C[q] *= sc[e * Q + q];
                    if (((e - es) & 7) == 7) x_norm<Q>(C, ce);
                }
                x_norm<Q>(C, ce);
                acc[2] += entropy_site_x<Q>(P, C, ce);
            }
        }
    } else if (adj_mode || EM) {  // a hub row (its site and edge terms: k_fe_hub): adjacent pairs and EM numerators, lanes strided over its edges
        double pi[Q];
        if (adj_mode) load_vec<Q>(psi + size_t(r0) * Q, pi);
        __syncthreads();  // sclc
        for (int le = tid; le < ne; le += TPB) {
            const uint32_t k = e0 + uint32_t(le);
            double mo[Q], mp[Q], pl[Q];
            load_msg<Q>(Mcur, size_t(k), mo);
            load_msg<Q>(Mprev, size_t(k), mp);
            load_vec<Q>(psi + size_t(nbr[k]) * Q, pl);
            edge(k, 0, mo, mp, pl, pi, false);
        }
    }
    block_reduce_store<NP, frame_cfg<Q>::WAVES>(acc, 0.0, sred, partials + size_t(bid) * (NP + 1));
}

template <int Q, bool DC2>
__global__ void __launch_bounds__(BLOCK)
k_fe_hub(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rev, const uint32_t *__restrict__ nbr, const uint32_t *__restrict__ ndeg /* degree of every table row (DC2 only) */,
         const double *__restrict__ M, const double *__restrict__ Min, const uint32_t *__restrict__ hub_row,
         const uint32_t *__restrict__ hub_blk, const dev_params *__restrict__ P, int dc, int want_entropy,
         double *__restrict__ partials, uint32_t rec_stride /* doubles per record (>= FE_NP + 1) */,
         uint32_t rec_first /* record of hub 0, or 0xffffffff: the hub's own segment record hub_blk[h] */) {
    __shared__ double sAq[BLOCK * Q];
    __shared__ double sCq[BLOCK * Q];
    __shared__ int sEq[BLOCK * Q];
    __shared__ double sred[4 * (FE_NP + 1)];
    const int tid = threadIdx.x;
    const uint32_t i = hub_row[blockIdx.x];
    const uint32_t e0 = row_ptr[i], d = row_ptr[i + 1] - e0;
    const double di = double(d);
    double acc[FE_NP] = {0.0, 0.0, 0.0, 0.0};
    double A[Q], C[Q];
    int ae[Q], ce[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = 1.0; C[q] = 1.0; ae[q] = 0; ce[q] = 0; }
    for (uint32_t le = tid; le < d; le += BLOCK) {
        double mi[Q], mo[Q], b[Q];
        load_msg<Q>(Min ? Min : M, Min ? size_t(e0 + le) : size_t(rev[e0 + le]), mi);
        load_msg<Q>(M, size_t(e0 + le), mo);
        double didl = 0.0;
        if (DC2) { const uint32_t l = nbr[e0 + le]; didl = di * double(ndeg[l]); }
        edge_field<Q, DC2>(P, mi, didl, b);
#pragma unroll
        for (int q = 0; q < Q; ++q) A[q] *= b[q];
        x_norm<Q>(A, ae);
        if (want_entropy) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                double a = 0.0;
#pragma unroll
                for (int t = 0; t < Q; ++t) a += P->cab[t * Q + q] * mi[t];
                C[q] *= a;
            }
            x_norm<Q>(C, ce);
        }
        double ln, en;
        edge_terms<Q, DC2>(P, mi, mo, didl, want_entropy, ln, en);
        acc[1] += ln;
        if (want_entropy) acc[3] += en;
    }
    block_product_x<Q>(A, ae, sAq, sEq);
    if (want_entropy) {  // uniform
        __syncthreads();
        block_product_x<Q>(C, ce, sCq, sEq);
    }
    if (tid == 0) {
        acc[0] += log_partition_x<Q>(P, dc, di, A, ae);
        if (want_entropy) acc[2] += entropy_site_x<Q>(P, C, ce);
    }
    double *rec = partials + size_t(rec_first == 0xffffffffu ? hub_blk[blockIdx.x] : rec_first + blockIdx.x) * rec_stride;
    block_reduce_store<FE_NP>(acc, 0.0, sred, rec);
    for (uint32_t x = FE_NP + 1 + tid; x < rec_stride; x += BLOCK) rec[x] = 0.0;  // the other columns of a wider record (k_fe_psi)
}

// ------------------------------------------------------------------------------------------------
// K4a: per-edge correction of the non-edge terms (adjacent ordered pairs are excluded from the
// all-pairs moment series): partial[0] = sum_dir log(1 - psi_i^T w psi_l / N),
// partial[1] = sum_dir (u/N)/(1 - y/N) with u = psi_i^T (cab∘log cab) psi_l, y = psi_i^T cab psi_l.
// One wave-friendly row loop: lane per row chunk, rows strided; gathers psi_l (N*Q table, cache
// resident for the sizes where it matters).
// ------------------------------------------------------------------------------------------------
template <int Q>
__global__ void __launch_bounds__(frame_cfg<Q>::TPB)
k_nonedge_adj(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ nbr, const double *__restrict__ psi,
              const double *__restrict__ wmat /* Q*Q: N(1-(1-cab/N)^beta) */, const double *__restrict__ cab,
              const uint32_t *__restrict__ blk_row, double invN, int want_entropy, double *__restrict__ partials) {
    __shared__ double sred[frame_cfg<Q>::WAVES * (NE_NP + 1)];
    __shared__ uint32_t srp[frame_cfg<Q>::RCAP + 1];
    const uint32_t r0 = blk_row[blockIdx.x], r1 = blk_row[blockIdx.x + 1];
    const int nrows = int(r1 - r0);
    const uint32_t e0 = row_ptr[r0];
    __shared__ double sclc[Q * Q];  // cab log cab, once per workgroup instead of Q*Q logs per edge
    for (int r = threadIdx.x; r <= nrows; r += frame_cfg<Q>::TPB) srp[r] = row_ptr[r0 + r] - e0;
    if (want_entropy)
        for (int x = threadIdx.x; x < Q * Q; x += frame_cfg<Q>::TPB) sclc[x] = cab[x] * log(cab[x]);
    __syncthreads();
    const int ne = int(srp[nrows]);
    double acc[NE_NP] = {0.0, 0.0};
    // lane per directed edge; the row of an edge is found by binary search in the LDS offsets
    for (int le = threadIdx.x; le < ne; le += frame_cfg<Q>::TPB) {
        int lo = 0, hi = nrows;  // find r with srp[r] <= le < srp[r+1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (int(srp[mid]) <= le) lo = mid; else hi = mid; }
        double pi[Q], pl[Q];
        load_vec<Q>(psi + size_t(r0 + lo) * Q, pi);
        load_vec<Q>(psi + size_t(nbr[e0 + le]) * Q, pl);
        double y = 0.0, yc = 0.0, u = 0.0;
#pragma unroll
        for (int q1 = 0; q1 < Q; ++q1) {
#pragma unroll
            for (int q2 = 0; q2 < Q; ++q2) {
                const double pp = pi[q1] * pl[q2];
                y += wmat[q1 * Q + q2] * pp;
                if (want_entropy) {
                    const double c = cab[q1 * Q + q2];
                    yc += c * pp;
                    u += sclc[q1 * Q + q2] * pp;
                }
            }
        }
        acc[0] += log1p(-y * invN);
        if (want_entropy) acc[1] += (u * invN) / (1.0 - yc * invN);
    }
    block_reduce_store<NE_NP, frame_cfg<Q>::WAVES>(acc, 0.0, sred, partials + size_t(blockIdx.x) * (NE_NP + 1));
}

// ------------------------------------------------------------------------------------------------
// K4b: moment tensors M_k[a_1..a_k] = sum_i prod_j psi_i[a_j], k = 1..K packed back to back
// (T = Q + Q^2 + ... + Q^K entries). Each workgroup stages a chunk of rows in LDS; thread t owns
// tensor entries t, t+256, ...; partial tensors per workgroup are folded by k_fold_columns.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK)
k_moments(const double *__restrict__ psi, uint32_t n_rows, int Q, int K, uint32_t rows_per_blk, int T,
          double *__restrict__ partials /* [gridDim.x][T] */) {
    __shared__ double sp[BLOCK * QMAX];
    __shared__ double scomb[BLOCK];
    const int tid = threadIdx.x;
    constexpr int MAXE = 20;  // entries per thread: T <= 256*20 = 5120 >= 8+64+512+4096 (engine.hip: max_series_order)
    double acc[MAXE];
    for (int j = 0; j < MAXE; ++j) acc[j] = 0.0;
    // few entries (T < 128, e.g. 84 for Q = 4 at order 3): several threads share one entry, each taking every nsub-th row
    const int nsub = T < BLOCK / 2 ? BLOCK / T : 1;
    const int sub = nsub > 1 ? tid / T : 0;
    const bool active = nsub == 1 || tid < nsub * T;
    const uint32_t lo = blockIdx.x * rows_per_blk, hi = min(n_rows, lo + rows_per_blk);
    const uint32_t stage_rows = Q <= QMAX ? uint32_t(BLOCK) : uint32_t(BLOCK * QMAX / Q);  // (label counts above 16: fewer rows per stage)
    for (uint32_t base = lo; base < hi; base += stage_rows) {
        const uint32_t cnt = min(stage_rows, hi - base);
        __syncthreads();
        for (uint32_t x = tid; x < cnt * Q; x += BLOCK) sp[x] = psi[size_t(base) * Q + x];
        __syncthreads();
        int j = 0;
        for (int ent = nsub > 1 ? tid % T : tid; active && ent < T; ent += BLOCK, ++j) {
            // decode (order k, multi-index) of the packed entry
            int k = 1, off = 0, sz = Q;
            while (ent >= off + sz) { off += sz; sz *= Q; ++k; }
            int idx = ent - off;
            int a[4];
            for (int t = 0; t < 4; ++t) { a[t] = idx % Q; idx /= Q; }
            double s = 0.0;
            for (uint32_t r = uint32_t(sub); r < cnt; r += uint32_t(nsub)) {
                const double *p = &sp[r * Q];
                double v = p[a[0]];
                if (k > 1) v *= p[a[1]];
                if (k > 2) v *= p[a[2]];
                if (k > 3) v *= p[a[3]];
                s += v;
            }
            acc[j] += s;
        }
    }
    if (nsub > 1) {  // combine the sub-sums of an entry in a fixed order
        __syncthreads();
        scomb[tid] = active ? acc[0] : 0.0;
        __syncthreads();
        if (tid < T) {
            double s = scomb[tid];
            for (int u = 1; u < nsub; ++u) s += scomb[u * T + tid];
            partials[size_t(blockIdx.x) * T + tid] = s;
        }
        return;
    }
    int j = 0;
    for (int ent = tid; ent < T; ent += BLOCK, ++j) partials[size_t(blockIdx.x) * T + ent] = acc[j];
}

// First stage of a two-stage fold of [rows][stride] partials: workgroup b reduces the contiguous
// row chunk [b*chunk, (b+1)*chunk) into out[b*stride + c] — sums for c < ncols_sum, sticky-NaN max
// for column ncols_sum when has_max. Fixed order: deterministic.
__global__ void __launch_bounds__(BLOCK)
k_fold_stage(const double *__restrict__ in, uint32_t rows, uint32_t chunk, int ncols_sum, int has_max, uint32_t stride,
             double *__restrict__ out) {
    __shared__ double s[BLOCK];
    const uint32_t lo = blockIdx.x * chunk, hi = min(rows, lo + chunk);
    const int ncols = ncols_sum + (has_max ? 1 : 0);
    for (int c = 0; c < ncols; ++c) {
        const bool is_max = has_max && c == ncols_sum;
        double a = 0.0;
        for (uint32_t r = lo + threadIdx.x; r < hi; r += BLOCK) {
            const double v = in[size_t(r) * stride + c];
            a = is_max ? nanmax(a, v) : a + v;
        }
        s[threadIdx.x] = a;
        __syncthreads();
        for (int k = BLOCK / 2; k > 0; k >>= 1) {
            if (int(threadIdx.x) < k) s[threadIdx.x] = is_max ? nanmax(s[threadIdx.x], s[threadIdx.x + k]) : s[threadIdx.x] + s[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[size_t(blockIdx.x) * stride + c] = s[0];
        __syncthreads();
    }
}

// column sums of a [rows][cols] partial matrix in fixed order: out[c] = sum_r in[r][c]
__global__ void __launch_bounds__(BLOCK)
k_fold_columns(const double *__restrict__ in, uint32_t rows, uint32_t cols, double *__restrict__ out) {
    const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (uint32_t r = 0; r < rows; ++r) s += in[size_t(r) * cols + c];
    out[c] = s;
}
// sticky-NaN column max companion (used for the max slot of frame partials)
__global__ void __launch_bounds__(BLOCK)
k_fold_rows_sum(const double *__restrict__ in, uint32_t rows, uint32_t cols, uint32_t stride, double *__restrict__ out) {
    // out[c] = sum over rows of in[r*stride + c], c < cols; one workgroup, fixed order
    __shared__ double s[BLOCK];
    for (uint32_t c = 0; c < cols; ++c) {
        double a = 0.0;
        for (uint32_t r = threadIdx.x; r < rows; r += BLOCK) a += in[size_t(r) * stride + c];
        s[threadIdx.x] = a;
        __syncthreads();
        for (int k = BLOCK / 2; k > 0; k >>= 1) {
            if (int(threadIdx.x) < k) s[threadIdx.x] += s[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c] = s[0];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K4x: exact non-edge sums over ALL ordered pairs (i,l) — the reference's O(N^2) loop
// (belief_propagation.cpp:675-741) as a tiled kernel; the adjacent pairs are removed afterwards
// with k_nonedge_adj. Tile = 256 rows i (one per lane) x 256 rows l staged in LDS.
// partial[0] = sum log(psi_i^T P psi_l), P = (1-cab/N)^beta;  partial[1] = sum num/den (entropy).
// ------------------------------------------------------------------------------------------------
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_nonedge_exact(const double *__restrict__ psi /* rows i: n of them (a shard: its own rows) */, uint32_t n,
                const double *__restrict__ psi_l /* rows l: n_l of them (all vertices) */, uint32_t n_l,
                const double *__restrict__ Pmat /* Q*Q */,
                const double *__restrict__ cab, double invN, int want_entropy, double *__restrict__ partials) {
    __shared__ double sl[BLOCK * Q];
    __shared__ double sred[4 * (NE_NP + 1)];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t l0 = blockIdx.y * BLOCK;
    const uint32_t cnt = min(uint32_t(BLOCK), n_l - l0);
    for (uint32_t x = threadIdx.x; x < cnt * Q; x += BLOCK) sl[x] = psi_l[size_t(l0) * Q + x];
    __syncthreads();
    double acc[NE_NP] = {0.0, 0.0};
    if (i < n) {
        double pi[Q], v[Q], vn[Q], vd[Q];
        load_vec<Q>(psi + size_t(i) * Q, pi);
#pragma unroll
        for (int q2 = 0; q2 < Q; ++q2) {  // v = P^T psi_i etc., so a pair costs Q FMAs
            double a = 0.0, an = 0.0, ad = 0.0;
#pragma unroll
            for (int q1 = 0; q1 < Q; ++q1) {
                a += Pmat[q1 * Q + q2] * pi[q1];
                if (want_entropy) {
                    const double c = cab[q1 * Q + q2];
                    an += (c * invN) * log(c) * pi[q1];
                    ad += (1.0 - c * invN) * pi[q1];
                }
            }
            v[q2] = a; vn[q2] = an; vd[q2] = ad;
        }
        for (uint32_t r = 0; r < cnt; ++r) {
            double f = 0.0, num = 0.0, den = 0.0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const double pl = sl[r * Q + q];
                f += v[q] * pl;
                if (want_entropy) { num += vn[q] * pl; den += vd[q] * pl; }
            }
            if (f != 0.0) acc[0] += log(f);
            if (want_entropy && num * den != 0.0) acc[1] += num / den;
        }
    }
    block_reduce_store<NE_NP>(acc, 0.0, sred, partials + (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * (NE_NP + 1));
}

// exact per-edge terms to subtract from the all-pairs sums of k_nonedge_exact
template <int Q>
__global__ void __launch_bounds__(frame_cfg<Q>::TPB)
k_nonedge_exact_adj(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ nbr, const double *__restrict__ psi,
                    const double *__restrict__ Pmat, const double *__restrict__ cab, const uint32_t *__restrict__ blk_row,
                    double invN, int want_entropy, double *__restrict__ partials) {
    __shared__ double sred[frame_cfg<Q>::WAVES * (NE_NP + 1)];
    __shared__ uint32_t srp[frame_cfg<Q>::RCAP + 1];
    const uint32_t r0 = blk_row[blockIdx.x], r1 = blk_row[blockIdx.x + 1];
    const int nrows = int(r1 - r0);
    const uint32_t e0 = row_ptr[r0];
    __shared__ double sclc[Q * Q];  // (cab/N) log cab, once per workgroup
    for (int r = threadIdx.x; r <= nrows; r += frame_cfg<Q>::TPB) srp[r] = row_ptr[r0 + r] - e0;
    if (want_entropy)
        for (int x = threadIdx.x; x < Q * Q; x += frame_cfg<Q>::TPB) sclc[x] = (cab[x] * invN) * log(cab[x]);
    __syncthreads();
    const int ne = int(srp[nrows]);
    double acc[NE_NP] = {0.0, 0.0};
    for (int le = threadIdx.x; le < ne; le += frame_cfg<Q>::TPB) {
        int lo = 0, hi = nrows;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (int(srp[mid]) <= le) lo = mid; else hi = mid; }
        double pi[Q], pl[Q];
        load_vec<Q>(psi + size_t(r0 + lo) * Q, pi);
        load_vec<Q>(psi + size_t(nbr[e0 + le]) * Q, pl);
        double f = 0.0, num = 0.0, den = 0.0;
#pragma unroll
        for (int q1 = 0; q1 < Q; ++q1) {
#pragma unroll
            for (int q2 = 0; q2 < Q; ++q2) {
                const double pp = pi[q1] * pl[q2];
                f += Pmat[q1 * Q + q2] * pp;
                if (want_entropy) {
                    const double c = cab[q1 * Q + q2];
                    num += sclc[q1 * Q + q2] * pp;
                    den += (1.0 - c * invN) * pp;
                }
            }
        }
        if (f != 0.0) acc[0] += log(f);
        if (want_entropy && num * den != 0.0) acc[1] += num / den;
    }
    block_reduce_store<NE_NP, frame_cfg<Q>::WAVES>(acc, 0.0, sred, partials + size_t(blockIdx.x) * (NE_NP + 1));
}

// ------------------------------------------------------------------------------------------------
// K5: EM expectation numerators (belief_propagation.cpp:892-965): per directed edge
// 0.5 * W'[q1][q2] * (m_in[q1] m_out[q2] + [q1!=q2] m_in[q2] m_out[q1]) / norm_L for q1 <= q2,
// accumulated per lane over a grid-stride loop, then folded. Output row per workgroup: Q*Q entries
// (upper triangle filled). W' = cab (dc 0, no beta), cab (dc 1: prefactor cancels), x/(1+x) (dc 2).
// ------------------------------------------------------------------------------------------------
template <int Q, bool DC2>
__global__ void __launch_bounds__(BLOCK)
k_em_edges(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rev, const uint32_t *__restrict__ nbr, const uint32_t *__restrict__ ndeg /* degree of every table row (DC2 only) */,
           const uint32_t *__restrict__ src /* row of each edge, DC2 only */, const double *__restrict__ M,
           const double *__restrict__ Min, uint32_t n_edges, const dev_params *__restrict__ P,
           double *__restrict__ partials /* [grid][Q*Q] */) {
    constexpr int T = Q * (Q + 1) / 2;
    __shared__ double sred[4 * (T + 1)];
    double acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < n_edges; k += gridDim.x * BLOCK) {
        double mi[Q], mo[Q];
        load_msg<Q>(Min ? Min : M, Min ? size_t(k) : size_t(rev[k]), mi);
        load_msg<Q>(M, size_t(k), mo);
        double didl = 0.0;
        if (DC2) {
            const uint32_t i = src[k], l = nbr[k];
            didl = double(row_ptr[i + 1] - row_ptr[i]) * double(ndeg[l]);
        }
        double term[T], norm_L = 0.0;
        int t = 0;
#pragma unroll
        for (int q1 = 0; q1 < Q; ++q1) {
#pragma unroll
            for (int q2 = q1; q2 < Q; ++q2, ++t) {
                double w = P->cab[q1 * Q + q2];
                if (DC2) { double x = didl * w * P->invN; w = x / (1.0 + x); }
                const double pr = (q1 == q2) ? (mi[q1] * mo[q2]) : (mi[q1] * mo[q2] + mi[q2] * mo[q1]);
                term[t] = w * pr;
                norm_L += term[t];
            }
        }
        const double inv = 0.5 / norm_L;
#pragma unroll
        for (int u = 0; u < T; ++u) acc[u] += term[u] * inv;
    }
    block_reduce_store<T>(acc, 0.0, sred, partials + size_t(blockIdx.x) * (T + 1));
}

// incoming messages in edge order, reconstructed from the marginal table and the own messages of the
// previous sweep (same identity as k_sweep_psi): Min[k] = N( psi[nbr[k]] / (W^T Mprev[k]) ). Used by the
// reductions of sharded engines, which hold no reverse-edge index across shards.
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_materialize_in(const uint32_t *__restrict__ nbr, const double *__restrict__ Mprev, const double *__restrict__ psi,
                 uint32_t n_edges, const dev_params *__restrict__ P, double *__restrict__ Min) {
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < n_edges; k += gridDim.x * BLOCK) {
        double pl[Q], mo[Q], bo[Q], inc[Q];
        load_vec<Q>(psi + size_t(nbr[k]) * Q, pl);
        load_msg<Q>(Mprev, size_t(k), mo);
        edge_field<Q, false>(P, mo, 0.0, bo);
        double tot = 0.0;
#pragma unroll
        for (int s = 0; s < Q; ++s) { inc[s] = pl[s] / bo[s]; tot += inc[s]; }
        const double inv = 1.0 / tot;
#pragma unroll
        for (int s = 0; s < Q; ++s) inc[s] *= inv;
        store_msg<Q>(Min, size_t(k), inc);
    }
}

// K5b/K6: per-row sums: na_expect, nna_expect (belief_propagation.cpp:428-440) and the confusion
// matrix C[a][b] = sum_{i: true_i = a} psi_i[b] (compute_overlap, :775-811). Row chunk per workgroup,
// LDS accumulation by (class, label) with one owner thread per output entry.
template <int Q>
__global__ void __launch_bounds__(BLOCK)
k_row_sums(const uint32_t *__restrict__ row_ptr, const double *__restrict__ psi, const uint32_t *__restrict__ true_conf,
           uint32_t n_rows, uint32_t rows_per_blk, double *__restrict__ partials /* [grid][2Q + Q*Q] */) {
    __shared__ double sp[BLOCK * Q];
    __shared__ uint32_t sdeg[BLOCK];
    __shared__ uint32_t scl[BLOCK];
    constexpr int T = 2 * Q + Q * Q;
    constexpr int PER = (T + BLOCK - 1) / BLOCK;  // output entries per thread: 1 up to Q = 15, 2 at Q = 16
    const int tid = threadIdx.x;
    double acc[PER];  // thread t owns output entries t, t + BLOCK, ...
#pragma unroll
    for (int j = 0; j < PER; ++j) acc[j] = 0.0;
    const uint32_t lo = blockIdx.x * rows_per_blk, hi = min(n_rows, lo + rows_per_blk);
    for (uint32_t base = lo; base < hi; base += BLOCK) {
        const uint32_t cnt = min(uint32_t(BLOCK), hi - base);
        __syncthreads();
        for (uint32_t x = tid; x < cnt * Q; x += BLOCK) sp[x] = psi[size_t(base) * Q + x];
        if (uint32_t(tid) < cnt) {
            sdeg[tid] = row_ptr[base + tid + 1] - row_ptr[base + tid];
            scl[tid] = true_conf ? true_conf[base + tid] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int t = tid + j * BLOCK;
            if (t < Q) {
                for (uint32_t r = 0; r < cnt; ++r) acc[j] += sp[r * Q + t];
            } else if (t < 2 * Q) {
                for (uint32_t r = 0; r < cnt; ++r) acc[j] += double(sdeg[r]) * sp[r * Q + (t - Q)];
            } else if (t < T) {
                const uint32_t a = uint32_t(t - 2 * Q) / Q, b = uint32_t(t - 2 * Q) % Q;
                for (uint32_t r = 0; r < cnt; ++r) acc[j] += (scl[r] == a) ? sp[r * Q + b] : 0.0;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (tid + j * BLOCK < T) partials[size_t(blockIdx.x) * T + tid + j * BLOCK] = acc[j];
}

// row index of every directed edge (built once; used by reductions that need d_i per edge)
__global__ void __launch_bounds__(BLOCK)
k_fill_src(const uint32_t *__restrict__ row_ptr, uint32_t n_rows, uint32_t *__restrict__ src) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_rows) return;
    for (uint32_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) src[k] = i;
}

// counter-based initial state: splitmix64 of (seed, slot) -> uniform(0,1), normalised per Q-vector
__device__ __forceinline__ double u01(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
// ncomp = Q: marginal rows; ncomp = Q-1: message records (encoded, see msg_rec)
__global__ void __launch_bounds__(BLOCK)
k_init_random(double *__restrict__ v, uint64_t n_vec, int Q, int ncomp, uint64_t seed, uint64_t salt, uint64_t index0) {
    const uint64_t i = uint64_t(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= n_vec) return;
    double t[QMAX_RT], w[QMAX_RT], norm = 0.0;
    for (int q = 0; q < Q; ++q) { t[q] = u01(seed ^ salt, (index0 + i) * uint64_t(Q) + q); norm += t[q]; }
    for (int q = 0; q < Q; ++q) t[q] /= norm;
    if (ncomp == Q) {
        for (int q = 0; q < Q; ++q) v[i * Q + q] = t[q];
    } else {
        encode_msg_rt(t, Q, w);
        for (int q = 0; q < ncomp; ++q) v[i * ncomp + q] = w[q];
    }
}

// device initialisation: every out-message of a row starts as the row's (random) marginal, in both message buffers.
// That state is consistent for the marginal-gather sweep without an explicit first sweep: the message l receives from i
// IS psi_i, which is what the first sweep gathers (k_sweep_psi: first_from_psi).
// On the sweep's segments, Q at compile time: a lane per directed edge (records written as the sweep writes them, consecutive
// lanes consecutive records), the edge's row from the segment's row table. (Round 2 had a run-time-Q kernel - a lane per ROW
// walking its edges, two Q-vectors in scratch memory -: 7.9 ms at C3 for 4.8 GB of records.)
template <int Q>
__global__ void __launch_bounds__(frame_cfg<Q>::TPB)
k_init_msgs_from_psi_seg(const uint32_t *__restrict__ row_ptr, const double *__restrict__ psi, const uint32_t *__restrict__ blk_row,
                         const uint32_t *__restrict__ blk_e0, double *__restrict__ Ma, double *__restrict__ Mb) {
    constexpr int CAP = frame_cfg<Q>::CAP, RCAP = frame_cfg<Q>::RCAP, TPB = frame_cfg<Q>::TPB;
    __shared__ uint16_t srow[CAP];
    const int tid = threadIdx.x;
    const uint32_t bid = blockIdx.x;
    const uint32_t r0 = blk_row[bid], r1 = blk_row[bid + 1], e0 = blk_e0[bid];
    const int nrows = int(r1 - r0), ne = int(blk_e0[bid + 1] - e0);
    if (ne > CAP) {  // a hub row: one marginal for all its records
        double v[Q];
        load_vec<Q>(psi + size_t(r0) * Q, v);
        for (int le = tid; le < ne; le += TPB) { store_msg<Q>(Ma, size_t(e0) + le, v); store_msg<Q>(Mb, size_t(e0) + le, v); }
        return;
    }
    for (int r = tid; r < nrows && r < RCAP; r += TPB)
        for (uint32_t e = row_ptr[r0 + r] - e0, ee = row_ptr[r0 + r + 1] - e0; e < ee; ++e) srow[e] = uint16_t(r);
    __syncthreads();
    for (int le = tid; le < ne; le += TPB) {
        double v[Q];
        load_vec<Q>(psi + size_t(r0 + srow[le]) * Q, v);
        store_msg<Q>(Ma, size_t(e0) + le, v);
        store_msg<Q>(Mb, size_t(e0) + le, v);
    }
}

// host layout (Q components per message, sbmbp_set_state / sbmbp_get_state) <-> message records (Q-1 words)
__global__ void __launch_bounds__(BLOCK)
k_msgs_to_records(const double *__restrict__ full, uint64_t n_msg, int Q, double *__restrict__ rec) {
    const uint64_t k = uint64_t(blockIdx.x) * BLOCK + threadIdx.x;
    if (k >= n_msg) return;
    double v[QMAX_RT], w[QMAX_RT];
    for (int q = 0; q < Q; ++q) v[q] = full[k * Q + q];
    encode_msg_rt(v, Q, w);
    for (int q = 0; q < Q - 1; ++q) rec[k * (Q - 1) + q] = w[q];
}
__global__ void __launch_bounds__(BLOCK)
k_records_to_msgs(const double *__restrict__ rec, uint64_t n_msg, int Q, double *__restrict__ full) {
    const uint64_t k = uint64_t(blockIdx.x) * BLOCK + threadIdx.x;
    if (k >= n_msg) return;
    double v[QMAX_RT], w[QMAX_RT];
    for (int q = 0; q < Q - 1; ++q) w[q] = rec[k * (Q - 1) + q];
    decode_msg_rt(w, Q, v);
    for (int q = 0; q < Q; ++q) full[k * Q + q] = v[q];
}

}  // namespace sbmbp
#endif
