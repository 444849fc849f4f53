// TEST INFRASTRUCTURE — the parity oracle. Not product code, never on the product path.
//
// CPU restatement (clean-room, C++14, plain loops) of the reference's belief-propagation path
// on the NEW data layout (flat CSR, out-ordered messages). Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library.
//
// Parity status: PINNED. tests/test_oracle_golden.py checks this file against fixtures under
// tests/golden/ that oracle/make_golden.py generated from the compiled, untouched reference
// (oracle/_ref/bp_ref). The asynchronous schedule below reproduces the reference bit for bit
// (same std::mt19937 stream, same operation order): identical niter, marginals and free energy.
//
// Every function cites the reference lines (relative to /root/reference/src/) it follows.
//
// Layout (differs from the reference on purpose — see DESIGN.md):
//   row_ptr[N+1], nbr[E2] ascending per row (== std::set order, graph_utilities.cpp:60-77),
//   rev[k] = index of the reverse directed edge,
//   M[k*Q + q] = message FROM row(k) TO nbr[k]   (out-ordered).
//   The reference's mmap_[i][l][q] (message INTO i from its l-th neighbour, belief_propagation.h:65-66)
//   is M[rev[row_ptr[i]+l]*Q + q].
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <numeric>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct graph_t {
    uint32_t N = 0;
    std::vector<uint64_t> row_ptr;
    std::vector<uint32_t> nbr, rev;
    uint64_t E2() const { return nbr.size(); }
    uint32_t deg(uint32_t i) const { return uint32_t(row_ptr[i + 1] - row_ptr[i]); }
};

struct rng_t {
    std::mt19937 engine;
    std::uniform_real_distribution<> u{0, 1};  // belief_propagation.h:20,90
    explicit rng_t(unsigned seed) : engine(seed) {}
    double draw() { return u(engine); }
};

struct bp_t {
    const graph_t *g = nullptr;
    uint32_t N = 0, Q = 0, dc = 0;
    double beta = 1.0;
    std::vector<double> cab, logcab, pab, W;  // Q*Q row-major; W = cab^beta (bp.cpp:1004)
    std::vector<uint32_t> na;
    std::vector<double> eta, logeta;
    std::vector<double> psi;      // N*Q
    std::vector<double> M, Mnew;  // E2*Q out-ordered
    std::vector<double> h, exph;
    std::vector<int32_t> planted;  // conf_planted_, -1 = free (bp.cpp:284)
    std::vector<uint32_t> conf_true;
    std::vector<double> na_expect, nna_expect, cab_expect;
    std::vector<double> Sprev;  // previous sweep's sum_i g_i psi_i (field relaxation of the synchronous schedule)
    double field_mix = 1.0;
    // bookkeeping of the synchronous twin (mirrors the engine, not the reference): which sweep form the engine would run
    unsigned init_flag = 0;       // flag of the last init_messages (1/3: clamped rows are one-hot)
    bool consistent = false;      // psi holds the marginals of the message pair (an undamped sweep ran since the last change)
    bool msg_form_only = false;   // the engine was told to gather messages (sbmbp_set_gather_mode 1)
    bool auto_relax = true;       // adaptive relaxation of converge_sync
    int ar_fl = 0, ar_gl = -1;    // levels reached by the last converge_sync
    int learn_unconverged = 0;    // BP runs of the last learning call that hit the sweep limit
    const unsigned LARGE_DEGREE = 50;  // belief_propagation.h:68
    const double EPS = 1.0e-50;        // belief_propagation.h:69
};

// ---------------------------------------------------------------------------------------------
// graph_utilities.cpp:60-77 (edge_to_adj: symmetrise, dedup via std::set, auto-grow) and
// belief_propagation.cpp:246-266 (graph_neis_, graph_neis_inv_) restated as sort+unique on pairs.
graph_t *graph_from_edges(const uint32_t *pairs, uint64_t n_pairs, uint32_t N) {
    auto *g = new graph_t();
    uint32_t n = N;
    for (uint64_t e = 0; e < n_pairs; ++e) {
        n = std::max(n, pairs[2 * e] + 1);
        n = std::max(n, pairs[2 * e + 1] + 1);
    }
    std::vector<uint64_t> keys;
    keys.reserve(2 * n_pairs);
    for (uint64_t e = 0; e < n_pairs; ++e) {
        uint64_t a = pairs[2 * e], b = pairs[2 * e + 1];
        keys.push_back((a << 32) | b);
        keys.push_back((b << 32) | a);  // a self-loop yields the same key twice; unique keeps one (B15)
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    g->N = n;
    g->row_ptr.assign(size_t(n) + 1, 0);
    g->nbr.resize(keys.size());
    for (size_t k = 0; k < keys.size(); ++k) {
        g->row_ptr[(keys[k] >> 32) + 1]++;
        g->nbr[k] = uint32_t(keys[k] & 0xffffffffu);
    }
    for (uint32_t i = 0; i < n; ++i) g->row_ptr[i + 1] += g->row_ptr[i];
    g->rev.resize(keys.size());
    for (uint32_t i = 0; i < n; ++i) {
        for (uint64_t k = g->row_ptr[i]; k < g->row_ptr[i + 1]; ++k) {
            uint32_t j = g->nbr[k];
            auto b = g->nbr.begin() + g->row_ptr[j], e = g->nbr.begin() + g->row_ptr[j + 1];
            g->rev[k] = uint32_t(std::lower_bound(b, e, i) - g->nbr.begin());
        }
    }
    return g;
}

// graph_utilities.cpp:42-58 (load_edge_list), including its quirk that a blank/malformed line
// re-pushes the previous pair (node_a/node_b keep their old values; B14).
bool load_edge_pairs(const char *path, std::vector<uint32_t> &pairs) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string line;
    unsigned a = 0, b = 0;
    while (std::getline(f, line)) {
        std::stringstream ls(line);
        ls >> a;
        ls >> b;
        pairs.push_back(a);
        pairs.push_back(b);
    }
    return true;
}

// belief_propagation.cpp:290-317 (expand_bp_params) + set_beta (:417-419)
void set_params(bp_t &s, const double *cab, const uint32_t *na, double beta) {
    uint32_t Q = s.Q;
    s.beta = beta;
    s.consistent = false;
    s.cab.assign(cab, cab + Q * Q);
    s.na.assign(na, na + Q);
    s.logcab.resize(Q * Q);
    s.pab.resize(Q * Q);
    s.W.resize(Q * Q);
    s.eta.resize(Q);
    s.logeta.resize(Q);
    for (uint32_t q = 0; q < Q; ++q) {
        s.eta[q] = 1.0 * s.na[q] / s.N;
        s.logeta[q] = std::log(s.eta[q]);
        for (uint32_t j = 0; j < Q; ++j) {
            s.pab[q * Q + j] = s.cab[q * Q + j] / s.N;
            s.logcab[q * Q + j] = std::log(s.cab[q * Q + j]);
            s.W[q * Q + j] = std::pow(s.cab[q * Q + j], beta);
        }
    }
}

// belief_propagation.cpp:101-217 (init_messages). Flag 0 and 1 restated; flags 2/3 abort in the
// reference for any node planted in group 1 (assert typo, B5) and are defined sanely here:
// 2 = planted + noise (normalised), 3 = hard planted — the engine documents the same deviation.
void init_messages(bp_t &s, unsigned flag, const int32_t *conf, const uint32_t *true_conf, rng_t &r) {
    const graph_t &g = *s.g;
    uint32_t N = s.N, Q = s.Q;
    s.psi.assign(size_t(N) * Q, 0.0);
    s.M.assign(g.E2() * Q, 0.0);
    s.Mnew.assign(g.E2() * Q, 0.0);
    s.planted.assign(N, -1);
    s.conf_true.assign(true_conf, true_conf + N);
    s.h.assign(Q, 0.0);
    s.exph.assign(Q, 0.0);
    s.na_expect.assign(Q, 0.0);
    s.nna_expect.assign(Q, 0.0);
    s.cab_expect.assign(size_t(Q) * Q, 0.0);
    if (flag != 0 && conf) s.planted.assign(conf, conf + N);
    for (uint32_t i = 0; i < N; ++i) {
        int32_t p = flag == 0 ? -1 : s.planted[i];
        auto fill = [&](double *dst) {
            if (flag == 0 || (flag == 1 && p == -1)) {  // :112-119, :144-151
                double norm = 0.0;
                for (uint32_t q = 0; q < Q; ++q) { dst[q] = r.draw(); norm += dst[q]; }
                for (uint32_t q = 0; q < Q; ++q) dst[q] /= norm;
            } else if (flag == 2 && p != -1) {  // :175-193 intent, normalised
                const double noise = 0.1;
                double norm = 0.0;
                for (uint32_t q = 0; q < Q; ++q) {
                    dst[q] = (int32_t(q) == p) ? noise + (1.0 - noise) * r.draw() : r.draw() * (1.0 - noise);
                    norm += dst[q];
                }
                for (uint32_t q = 0; q < Q; ++q) dst[q] /= norm;
            } else if (p != -1) {  // :136-142, :157-163, :196-213
                for (uint32_t q = 0; q < Q; ++q) dst[q] = (int32_t(q) == p) ? 1.0 : 0.0;
            } else {
                double norm = 0.0;
                for (uint32_t q = 0; q < Q; ++q) { dst[q] = r.draw(); norm += dst[q]; }
                for (uint32_t q = 0; q < Q; ++q) dst[q] /= norm;
            }
        };
        fill(&s.psi[size_t(i) * Q]);
        // out-messages of i in ascending neighbour order == mmap_[j][idxji] fills of :120-131
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) fill(&s.M[k * Q]);
    }
    if (flag == 0) s.planted.assign(N, -1);  // B6: with -i 0 the planted vector is never stored
    s.init_flag = flag;
    s.consistent = false;
}

inline double gweight(const bp_t &s, uint32_t i) { return s.dc == 0 ? 1.0 : double(s.g->deg(i)); }

// belief_propagation.cpp:334-360 (update_h), :363-368 (update_exph_with_h), :320-332 (init_h)
void update_h(bp_t &s, uint32_t i, int mode) {
    uint32_t Q = s.Q;
    double di = double(s.g->deg(i));
    const double *p = &s.psi[size_t(i) * Q];
    for (uint32_t q1 = 0; q1 < Q; ++q1)
        for (uint32_t q2 = 0; q2 < Q; ++q2) {
            double c = s.cab[q2 * Q + q1];
            if (mode < 0) {
                if (s.dc == 0) s.h[q1] -= c * p[q2];
                else s.h[q1] -= di * c * p[q2];
            } else {
                if (s.dc == 0) s.h[q1] += c * p[q2];
                else s.h[q1] += di * c * p[q2];
            }
        }
}
void update_exph(bp_t &s) {
    for (uint32_t q = 0; q < s.Q; ++q) s.exph[q] = std::exp(-s.beta * s.h[q] / s.N);
}
void init_h(bp_t &s) {
    for (uint32_t q = 0; q < s.Q; ++q) s.h[q] = 0.0;
    for (uint32_t i = 0; i < s.N; ++i) update_h(s, i, +1);
    update_exph(s);
}

// Edge weight W_il[t][q] of Appendix A.1: bp.cpp:1003-1011
inline double weight(const bp_t &s, double di, double dl, uint32_t t, uint32_t q) {
    if (s.dc == 0) return s.W[t * s.Q + q];
    if (s.dc == 1) return di * dl * s.cab[t * s.Q + q];
    double tmp = di * dl * s.pab[t * s.Q + q];
    return tmp / (1.0 + tmp);
}

// One asynchronous node update, small-degree path:
// clean_mmap_total_at_node_i_ (:422-426) + sum_all_messages_to_i (:991-1049) + update_h(-1) +
// norm_m_at_i (:1051-1071) + update_h(+1) + update_exph_with_h  == bp_basic::bp_iter_update_psi (:1079-1098)
double node_update_small(bp_t &s, uint32_t i, double damp) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    uint64_t k0 = g.row_ptr[i];
    uint32_t d = g.deg(i);
    double di = double(d);
    std::vector<double> field(d, 0.0), total(d, 0.0), nb(size_t(Q) * d, 0.0), psiq(Q, 0.0);
    double psi_total = 0.0;
    for (uint32_t q = 0; q < Q; ++q) {
        double a = 1.0;
        for (uint32_t l = 0; l < d; ++l) {
            double b = 0.0;
            double dl = double(g.deg(g.nbr[k0 + l]));
            const double *m = &s.M[size_t(g.rev[k0 + l]) * Q];
            for (uint32_t t = 0; t < Q; ++t) {
                if (s.dc == 0) b += s.W[t * Q + q] * m[t];
                else if (s.dc == 1) b += di * dl * s.cab[t * Q + q] * m[t];
                else { double tmp = di * dl * s.pab[t * Q + q]; b += tmp / (1.0 + tmp) * m[t]; }
            }
            if (b == 0.) continue;  // "sanity check 1" (:1013-1016): field[l] keeps its stale value
            a *= b;
            field[l] = b;
        }
        if (s.dc == 0) psiq[q] = a * s.eta[q] * s.exph[q];
        else psiq[q] = a * s.eta[q] * std::exp(-1.0 * di * s.h[q] / s.N);
        psi_total += psiq[q];
        for (uint32_t l = 0; l < d; ++l) {
            if (field[l] < s.EPS) {  // :1029-1042
                double tmprob = 1.0;
                for (uint32_t lx = 0; lx < d; ++lx) {
                    if (lx == l) continue;
                    if (field[lx] != 0) tmprob *= field[lx];
                }
                nb[size_t(q) * d + l] = tmprob;
            } else {
                nb[size_t(q) * d + l] = psiq[q] / field[l];
            }
            total[l] += nb[size_t(q) * d + l];
        }
    }
    update_h(s, i, -1);
    double mymaxdiff = -100.0;
    for (uint32_t q = 0; q < Q; ++q) {
        s.psi[size_t(i) * Q + q] = psiq[q] / psi_total;
        for (uint32_t l = 0; l < d; ++l) {
            double *slot = &s.M[(k0 + l) * Q + q];  // mmap_[i2][l2][q] == our out-message k0+l
            double nv = nb[size_t(q) * d + l] / total[l];
            double mydiff = std::fabs(*slot - nv);
            if (mydiff > mymaxdiff) mymaxdiff = mydiff;
            *slot = (damp) * nb[size_t(q) * d + l] / total[l] + (1.0 - damp) * *slot;
        }
    }
    update_h(s, i, +1);
    update_exph(s);
    return mymaxdiff;
}

// bp_iter_update_psi_large_degree (:813-890): log domain, ignores beta (B4)
double node_update_large(bp_t &s, uint32_t i, double damp) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    uint64_t k0 = g.row_ptr[i];
    uint32_t d = g.deg(i);
    double di = double(d);
    std::vector<double> field(d, 0.0), total(d, 0.0), maxpom(d, -100000000.0), nb(size_t(Q) * d, 0.0), psiq(Q, 0.0);
    double psi_total = 0.0, maxpom_psi = -100000000.0;
    for (uint32_t q = 0; q < Q; ++q) {
        double a = 0.0;
        for (uint32_t l = 0; l < d; ++l) {
            double b = 0.0;
            double dl = double(g.deg(g.nbr[k0 + l]));
            const double *m = &s.M[size_t(g.rev[k0 + l]) * Q];
            for (uint32_t t = 0; t < Q; ++t) {
                if (s.dc == 0) b += s.cab[t * Q + q] * m[t];
                else if (s.dc == 1) b += di * dl * s.cab[t * Q + q] * m[t];
                else { double tmp = di * dl * s.pab[t * Q + q]; b += tmp / (1.0 + tmp) * m[t]; }
            }
            double tmp = std::log(b);
            a += tmp;
            field[l] = tmp;
        }
        if (s.dc == 0) psiq[q] = a + s.logeta[q] - s.h[q] / s.N;
        else psiq[q] = a + s.logeta[q] - 1.0 * di * s.h[q] / s.N;
        if (psiq[q] > maxpom_psi) maxpom_psi = psiq[q];
        for (uint32_t l = 0; l < d; ++l) {
            nb[size_t(q) * d + l] = psiq[q] - field[l];
            if (nb[size_t(q) * d + l] > maxpom[l]) maxpom[l] = nb[size_t(q) * d + l];
        }
    }
    for (uint32_t q = 0; q < Q; ++q) {
        psi_total += std::exp(psiq[q] - maxpom_psi);
        for (uint32_t l = 0; l < d; ++l) total[l] += std::exp(nb[size_t(q) * d + l] - maxpom[l]);
    }
    update_h(s, i, -1);
    double mymaxdiff = -100.0;
    for (uint32_t q = 0; q < Q; ++q) {
        s.psi[size_t(i) * Q + q] = std::exp(psiq[q] - maxpom_psi) / psi_total;
        for (uint32_t l = 0; l < d; ++l) {
            double *slot = &s.M[(k0 + l) * Q + q];
            double nv = std::exp(nb[size_t(q) * d + l] - maxpom[l]) / total[l];
            double mydiff = std::fabs(*slot - nv);
            if (mydiff > mymaxdiff) mymaxdiff = mydiff;
            *slot = (damp) * nv + (1 - damp) * *slot;
        }
    }
    update_h(s, i, +1);
    update_exph(s);
    return mymaxdiff;
}

// converge (:386-415): random-sequential schedule, N draws with replacement per sweep.
// conditional != 0 -> bp_conditional::bp_iter_update_psi (:1100-1126): planted nodes are skipped
// (only on the small-degree path, exactly as the reference dispatches, :397-401).
int converge_async(bp_t &s, float bp_err, unsigned max_iter, float dumping_rate, rng_t &r, int conditional) {
    init_h(s);
    for (int it = 0; it < int(max_iter); ++it) {
        double maxdiffm = -100.0;
        for (uint32_t k = 0; k < s.N; ++k) {
            auto i = unsigned(int(r.draw() * s.N));
            double diffm;
            if (s.g->deg(i) >= s.LARGE_DEGREE) diffm = node_update_large(s, i, dumping_rate);
            else if (conditional && s.planted[i] != -1) diffm = 0;
            else diffm = node_update_small(s, i, dumping_rate);
            if (diffm > maxdiffm) maxdiffm = diffm;
        }
        if (maxdiffm < bp_err) return it;
    }
    return -1;
}

// ---- synchronous (Jacobi) schedule: the algorithm the GPU engine runs ----
// All rows read M (sweep t) and write Mnew (sweep t+1); h is formed once per sweep from psi^t.
// Same equations as node_update_small; evaluated with a per-row rescaled product so any degree is
// safe (the reference's separate log-domain hub path is the same mathematics, :813-890). beta is
// applied uniformly (the reference's hub path drops it, B4 — documented deviation).
void compute_h_full(bp_t &s, double field_mix = 1.0) {
    uint32_t Q = s.Q;
    std::vector<double> S(Q, 0.0);
    for (uint32_t i = 0; i < s.N; ++i) {
        double gi = gweight(s, i);
        for (uint32_t q = 0; q < Q; ++q) S[q] += gi * s.psi[size_t(i) * Q + q];
    }
    // optional relaxation of the global field (fixed point unchanged): S <- (1-a) S_prev + a S
    if (s.Sprev.size() == Q && field_mix < 1.0)
        for (uint32_t q = 0; q < Q; ++q) S[q] = (1.0 - field_mix) * s.Sprev[q] + field_mix * S[q];
    s.Sprev = S;
    for (uint32_t q1 = 0; q1 < Q; ++q1) {
        double acc = 0.0;
        for (uint32_t q2 = 0; q2 < Q; ++q2) acc += s.cab[q2 * Q + q1] * S[q2];
        s.h[q1] = acc;
    }
    update_exph(s);
}

double sweep_sync(bp_t &s, double damp) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    compute_h_full(s, s.field_mix);
    double maxdiff = 0.0;
    std::vector<double> b, A(Q), cav(Q);
    for (uint32_t i = 0; i < s.N; ++i) {
        uint64_t k0 = g.row_ptr[i];
        uint32_t d = g.deg(i);
        double di = double(d);
        if (s.planted[i] != -1) {  // clamped rows emit constant messages (:1115-1124)
            for (uint64_t k = k0; k < k0 + d; ++k)
                for (uint32_t q = 0; q < Q; ++q) s.Mnew[k * Q + q] = s.M[k * Q + q];
            continue;
        }
        b.assign(size_t(d) * Q, 0.0);
        const bool long_row = d > 32;
        std::vector<int> Ae(Q, 0);
        for (uint32_t q = 0; q < Q; ++q) A[q] = 1.0;
        for (uint32_t l = 0; l < d; ++l) {
            double dl = double(g.deg(g.nbr[k0 + l]));
            const double *m = &s.M[size_t(g.rev[k0 + l]) * Q];
            double bmax = 0.0;
            for (uint32_t q = 0; q < Q; ++q) {
                double acc = 0.0;
                for (uint32_t t = 0; t < Q; ++t) {
                    // dc 1: the d_i d_l prefactor cancels in both normalisations (SURVEY A.2 remark 1)
                    double w = (s.dc == 0) ? s.W[t * Q + q] : (s.dc == 1 ? s.cab[t * Q + q] : weight(s, di, dl, t, q));
                    acc += w * m[t];
                }
                b[size_t(l) * Q + q] = acc;
                bmax = std::max(bmax, acc);
            }
            if (long_row) {  // one binary exponent per component (see below)
                for (uint32_t q = 0; q < Q; ++q) { int k; A[q] = std::frexp(A[q] * b[size_t(l) * Q + q], &k); Ae[q] += k; }
                continue;
            }
            double amax = 0.0;
            for (uint32_t q = 0; q < Q; ++q) { A[q] *= b[size_t(l) * Q + q]; amax = std::max(amax, A[q]); }
            if (amax > 0.0 && (amax < 1e-100 || amax > 1e100)) for (uint32_t q = 0; q < Q; ++q) A[q] /= amax;
        }
        double tot = 0.0;
        // dc != 0: exp(-d_i h/N) underflows in every component for hub rows; shifting by min_q h changes only a
        // common factor (the reference's large-degree path works in the log domain with a max-shift, :850-868)
        double hmin = *std::min_element(s.h.begin(), s.h.end());
        if (long_row) {
            // Rows above 32 edges: partial products can be extreme in opposite directions, so a common rescaling would
            // flush the smaller component of each to zero. Components keep their own exponent and the row is finished
            // in the log domain with a max-shift, as the reference's large-degree path does (:844-868).
            double m = -1e300;
            std::vector<double> lp(Q);
            for (uint32_t q = 0; q < Q; ++q) {
                double g = (s.dc == 0) ? s.beta : di;
                lp[q] = std::log(A[q]) + double(Ae[q]) * 0.6931471805599453 + std::log(s.eta[q]) - g * s.h[q] / s.N;
                m = std::max(m, lp[q]);
            }
            for (uint32_t q = 0; q < Q; ++q) { A[q] = std::exp(lp[q] - m); tot += A[q]; }
        } else
        for (uint32_t q = 0; q < Q; ++q) {
            double F = (s.dc == 0) ? s.exph[q] : std::exp(-di * (s.h[q] - hmin) / s.N);
            A[q] = A[q] * s.eta[q] * F;
            tot += A[q];
        }
        for (uint32_t q = 0; q < Q; ++q) s.psi[size_t(i) * Q + q] = A[q] / tot;  // psi^{t+1}; h already taken from psi^t
        for (uint32_t l = 0; l < d; ++l) {
            double csum = 0.0;
            bool usable = true;
            for (uint32_t q = 0; q < Q; ++q) {
                double bq = b[size_t(l) * Q + q];
                if (bq > 0.0 && A[q] / bq < std::numeric_limits<double>::infinity()) cav[q] = A[q] / bq;
                else usable = false;
            }
            if (!usable) {
                // exact cavity product when a division is not usable (b == 0 with a forbidden group pair, :1029-1042). ALL
                // components are recomputed, each with its own exponent, and finished in the log domain: mixing a quotient
                // A[q]/b[q] (A carries an arbitrary common factor) with an unscaled product would compare different scales.
                std::vector<double> pm(Q, 1.0), lpq(Q);
                std::vector<int> pe(Q, 0);
                for (uint32_t lx = 0; lx < d; ++lx) {
                    if (lx == l) continue;
                    for (uint32_t q = 0; q < Q; ++q) { int k; pm[q] = std::frexp(pm[q] * b[size_t(lx) * Q + q], &k); pe[q] += k; }
                }
                double mx = -1e300;
                for (uint32_t q = 0; q < Q; ++q) {
                    double g = (s.dc == 0) ? s.beta : di;
                    lpq[q] = std::log(pm[q]) + double(pe[q]) * 0.6931471805599453 + std::log(s.eta[q]) - g * s.h[q] / s.N;
                    mx = std::max(mx, lpq[q]);
                }
                for (uint32_t q = 0; q < Q; ++q) cav[q] = std::exp(lpq[q] - mx);
            }
            for (uint32_t q = 0; q < Q; ++q) csum += cav[q];
            for (uint32_t q = 0; q < Q; ++q) {
                double nv = cav[q] / csum;
                double old = s.M[(k0 + l) * Q + q];
                maxdiff = std::max(maxdiff, std::fabs(old - nv));
                s.Mnew[(k0 + l) * Q + q] = damp * nv + (1.0 - damp) * old;
            }
        }
    }
    s.M.swap(s.Mnew);
    {   // engine.hip run_sweeps: psi_consistent after an executed sweep
        bool clamp = false, wpos = true;
        for (int32_t p : s.planted) if (p != -1) { clamp = true; break; }
        for (double c : s.cab) if (!(c > 0.0)) wpos = false;
        s.consistent = damp == 1.0 && (!clamp || s.init_flag == 1 || s.init_flag == 3) && wpos;
    }
    return maxdiff;
}

// ---- adaptive relaxation of the synchronous schedule (no reference counterpart) ----------------
// The reference's random-sequential sweep keeps h_ current inside a sweep (:1088-1095) and never updates two neighbours at
// once; Jacobi sweeps can oscillate where it converges. converge_sync therefore watches three signatures and relaxes the
// schedule when one shows (the fixed points do not move): (F) the raw field sums swing with period 2, (P) the messages do
// (2-step difference far below the 1-step difference), (W) no progress over a window of sweeps. This restates, on the CPU,
// the state machine the engine runs on the device (kernels.h finalize_update) so that both stop on the same sweep.
struct ar_t {
    static constexpr int NF = 4, NG = 7, WIN = 24;
    const double FIELD[NF] = {1.0, 0.25, 0.1, 0.05};
    const double GMIX[NG] = {0.5, 0.25, 0.5, 0.25, 0.1, 0.25, 0.1};
    const double GDAMP[NG] = {1.0, 1.0, 0.5, 0.5, 0.5, 0.25, 0.25};
    bool on = true, psi_ok = false;
    double crit = 0, base_mix = 1.0;
    int fl = 0, gl = -1;
    bool armed = false, probing = false;
    int stall = 0, hold = 0, holdS = 0, sigc = 0, nS = 0, wn = 0;
    double v1 = -1, v2 = -1, wmin = 1e300, pmin = -1, d1p = -1, prev_hint = 0;
    std::vector<double> S1, S2;
    double mix() const { double m = std::min(base_mix, FIELD[fl]); return gl >= 0 ? std::min(m, GMIX[gl]) : m; }
    double damp() const { return gl >= 0 ? GDAMP[gl] : 1.0; }
    bool psi_form() const { return psi_ok && damp() == 1.0; }
    // what the next sweep reports: 1 = 1-step difference, 2 = 2-step difference
    int kind_next(bool explicit_first) const {
        int k = (!psi_form() || armed || explicit_first) ? 1 : 2;
        return probing ? 3 - k : k;
    }
    void reset_after() {
        hold = 6; stall = 0; v1 = v2 = -1; prev_hint = 0; probing = false; armed = false;
        wn = 0; wmin = 1e300; pmin = -1; nS = 0; sigc = 0; d1p = -1; holdS = 4;
    }
    int sweep = 0;            // (diagnostics only)
    const char *why = "";
    void trace(const char *what) const {
        if (std::getenv("ORC_AR_TRACE")) std::fprintf(stderr, "[ar] sweep %d: %s (%s) -> field level %d, generic level %d, mix %g, damping x%g\n", sweep, what, why, fl, gl, mix(), damp());
    }
    void esc_gen() {  // the next level that changes anything (a field level may already have taken the mix below a level's cap)
        const double m0 = mix(), d0 = damp();
        while (gl + 1 < NG) {
            ++gl;
            if (mix() < m0 || damp() < d0) {
                // the first damped level starts from an unrelaxed field again: damping often steadies the field by itself, and a
                // field level inherited from the undamped sweeps can make the damped ones crawl ((F) fires again if it must)
                if (d0 == 1.0 && damp() < 1.0) fl = 0;
                reset_after(); trace("generic"); return;
            }
        }
        hold = 1 << 30;  // the ladder is used up: the run goes on as it is
    }
    void esc_field() {
        // a swing of the field sums that shows although the generic ladder has already softened the field is driven by the
        // messages, not by the field's own feedback: damping answers it, a still softer field only slows everything down
        if (gl >= 0) { if (gl < 1) gl = 1; esc_gen(); return; }
        int nf = fl;
        while (nf + 1 < NF && !(FIELD[nf] < mix())) ++nf;   // the next cap that actually lowers the mix
        if (FIELD[nf] < mix()) { fl = nf; reset_after(); trace("field"); }
        else esc_gen();
    }
    // after sweep `it`: v = reported difference of kind `kind`, Sraw = unrelaxed sums of the new marginals. Returns true when
    // the run has converged. `was_psi`: the sweep ran in the marginal-gather form.
    // `rf`: what a plain sweep from here would move the field term beta h/N by (0 while the field is not relaxed): a relaxed
    // field lags its marginals, and the messages can stand still to within crit while it is still catching up.
    bool after_sweep(double v, int kind, const std::vector<double> &Sraw, bool was_psi, double rf = 0.0) {
        bool conv = false, esc = false;
        const bool field_ok = rf < crit;
        if (!on) {
            if (kind == 1) conv = v < crit && field_ok;
            else { hint(v); }
            return conv;
        }
        if (probing) {
            probing = false;
            if (kind == 1 && v < crit && field_ok) conv = true;
            else if (v1 >= 0) {
                const double one = kind == 1 ? v : v1, two = kind == 1 ? v1 : v;
                if (two < 0.5 * one) { why = "P: probe"; if (gl < 1) gl = 1; esc_gen(); esc = true; }  // messages with period 2: damping answers that, a softer field does not
                else { hold = 8; stall = 0; }
            }
        } else {
            if (kind == 1) {
                if (v < crit && field_ok) conv = true;
            } else hint(v);
            if (!conv && !esc) {
                if (hold > 0) --hold;
                else {
                    if (v2 >= 0 && v >= 0.98 * v2) ++stall; else stall = 0;
                    if (stall >= 4) { probing = true; stall = 0; }
                }
                v2 = v1; v1 = v;
                wmin = std::min(wmin, v); ++wn;
                if (wn >= WIN * (1 + std::min(3, std::max(0, gl)))) {
                    if (pmin >= 0 && wmin >= 0.9 * pmin && hold < (1 << 29)) { why = "W"; esc_gen(); esc = true; }
                    else { pmin = wmin; wmin = 1e300; wn = 0; }
                }
            }
        }
        if (conv || esc) return conv;
        // (F) period 2 in the raw field sums
        const size_t Q = Sraw.size();
        if (holdS > 0) { --holdS; S2 = S1; S1 = Sraw; nS = std::min(nS + 1, 2); return false; }
        bool fe = false;
        if (nS >= 2) {
            double d1 = 0, d2 = 0, tot = 0;
            for (size_t q = 0; q < Q; ++q) { d1 = std::max(d1, std::fabs(Sraw[q] - S1[q])); d2 = std::max(d2, std::fabs(Sraw[q] - S2[q])); tot += std::fabs(Sraw[q]); }
            const bool sig = d2 < 0.5 * d1 && d1 > 1e-9 * tot;
            if (sig && d1 > 0.05 * tot && fl == 0) fe = true;                                  // a violent swing: act at once
            else if (sig && (d1p < 0 || d1 >= 0.98 * d1p)) { if (++sigc >= 6) fe = true; }  // a swing that does not die out
            else sigc = 0;
            d1p = d1;
        }
        S2 = S1; S1 = Sraw; nS = std::min(nS + 1, 2);
        if (fe) { why = "F"; esc_field(); }
        return false;
    }
    void hint(double v) {  // a 2-step value can only arm the exact criterion (kernels.h HINT_SCALE)
        double scale = 8.0;
        if (prev_hint > 0 && v > 0 && v < prev_hint) { const double r = v / prev_hint; scale = std::min(64.0, std::max(8.0, 1.5 * (1.0 + 1.0 / r) / r)); }
        prev_hint = v;
        if (v < scale * crit) armed = true;
    }
};

// would the engine run this configuration in the marginal-gather form (engine.hip psi_form_allowed)?
bool engine_psi_form(const bp_t &s, double damp) {
    bool clamp = false;
    for (int32_t p : s.planted) if (p != -1) { clamp = true; break; }
    bool wpos = true;
    for (double c : s.cab) if (!(c > 0.0)) wpos = false;
    const bool onehot = s.init_flag == 1 || s.init_flag == 3;
    return !s.msg_form_only && damp == 1.0 && (!clamp || onehot) && s.dc != 2 && wpos && s.g->E2() > 0;
}

int converge_sync(bp_t &s, double crit, unsigned max_iter, double damp, double *last_diff) {
    ar_t ar;
    ar.on = s.auto_relax && crit >= 0;
    ar.crit = crit;
    ar.base_mix = s.field_mix;
    ar.psi_ok = engine_psi_form(s, damp);
    const bool first_explicit = !s.consistent;
    const double keep_mix = s.field_mix;
    const uint32_t Q = s.Q;
    std::vector<double> Sraw(Q);
    int result = -1;
    for (int it = 0; it < int(max_iter); ++it) {
        const bool was_psi = ar.psi_form() && !(it == 0 && first_explicit);
        const int kind = ar.kind_next(it == 0 && first_explicit && !ar.probing);
        s.field_mix = ar.mix();
        std::vector<double> Mold;
        if (kind == 2) Mold = s.Mnew;  // m^{t-1}: what the sweep before last left in the buffer this sweep overwrites
        const double d1 = sweep_sync(s, damp * ar.damp());
        double v = d1;
        if (kind == 2) {
            v = 0.0;
            if (Mold.size() == s.M.size()) {
                for (size_t k = 0; k < s.M.size(); ++k) { const double x = std::fabs(s.M[k] - Mold[k]); if (x > v || x != x) v = x; }
                v /= damp * ar.damp();  // a damped message moves by damp * (new - old) per sweep: the probe compares like with like
            } else v = d1;
        }
        if (last_diff) *last_diff = d1;
        std::fill(Sraw.begin(), Sraw.end(), 0.0);
        for (uint32_t i = 0; i < s.N; ++i) { const double gi = gweight(s, i); for (uint32_t q = 0; q < Q; ++q) Sraw[q] += gi * s.psi[size_t(i) * Q + q]; }
        double rf = 0.0;
        if (s.field_mix < 1.0 && s.Sprev.size() == Q)  // s.Sprev: the sums this sweep's field was formed from
            for (uint32_t q1 = 0; q1 < Q; ++q1) {
                double acc = 0.0;
                for (uint32_t q2 = 0; q2 < Q; ++q2) acc += s.cab[q2 * Q + q1] * (Sraw[q2] - s.Sprev[q2]);
                rf = std::max(rf, std::fabs(acc) / s.N * s.beta);
            }
        ar.sweep = it;
        if (ar.after_sweep(v, kind, Sraw, was_psi, rf)) { result = it; break; }
    }
    s.ar_fl = ar.fl; s.ar_gl = ar.gl;
    s.field_mix = keep_mix;
    compute_h_full(s);
    return result;
}

// ---- free energy (:442-504, :562-612, :675-709, :744-750) --------------------------------------
double f_site(const bp_t &s) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    double fs = 0.0;
    std::vector<double> rq(Q);
    for (uint32_t i = 0; i < s.N; ++i) {
        double di = double(g.deg(i));
        double resc = -100000.;
        for (uint32_t q = 0; q < Q; ++q) {
            double a = 0.0;
            for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
                double b = 0.0, dl = double(g.deg(g.nbr[k]));
                const double *m = &s.M[size_t(g.rev[k]) * Q];
                for (uint32_t t = 0; t < Q; ++t) b += weight(s, di, dl, t, q) * m[t];
                a += std::log(b);
            }
            if (s.dc == 0) rq[q] = a + s.logeta[q] - s.beta * s.h[q] / s.N;
            else rq[q] = a + s.logeta[q] - di * s.h[q] / s.N;
            if (rq[q] > resc) resc = rq[q];
        }
        double nrm = 0.0;
        for (uint32_t q = 0; q < Q; ++q) nrm += std::exp(rq[q] - resc);
        fs += resc + std::log(nrm);
    }
    return fs / s.N;
}

double f_edge(const bp_t &s) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    double fl = 0.0;
    for (uint32_t i = 0; i < s.N; ++i) {
        double di = double(g.deg(i));
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
            double dl = double(g.deg(g.nbr[k]));
            const double *min = &s.M[size_t(g.rev[k]) * Q];  // mmap_[i][l]
            const double *mout = &s.M[k * Q];                 // mmap_[i2][l2]
            double norm_L = 0.0;
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = q1; q2 < Q; ++q2) {
                    double w = weight(s, di, dl, q1, q2);
                    if (q1 == q2) norm_L += w * (min[q1] * mout[q2]);
                    else norm_L += w * (min[q1] * mout[q2] + min[q2] * mout[q1]);
                }
            fl += std::log(norm_L);
        }
    }
    return fl / (2. * s.N);
}

double f_nonedge_exact(const bp_t &s) {  // :675-709, O(N^2 Q^2); dc != 0 -> 0 (:687-700)
    const graph_t &g = *s.g;
    uint32_t Q = s.Q, N = s.N;
    if (s.dc != 0) return 0.0;
    std::vector<double> P(size_t(Q) * Q);
    for (uint32_t a = 0; a < Q * Q; ++a) P[a] = std::pow((1 - s.cab[a] / N), s.beta);
    std::vector<char> adj(N, 0);
    double acc = 0.0;
    for (uint32_t i = 0; i < N; ++i) {
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) adj[g.nbr[k]] = 1;
        for (uint32_t l = 0; l < N; ++l) {
            if (adj[l]) continue;
            double v = 0;
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = 0; q2 < Q; ++q2) v += P[q1 * Q + q2] * s.psi[size_t(i) * Q + q1] * s.psi[size_t(l) * Q + q2];
            if (v != 0.) acc += std::log(v);
        }
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) adj[g.nbr[k]] = 0;
    }
    return acc / (2. * N);
}

// Moment tensor M_k = sum_i psi_i^{(x)k} (Q^k entries) and the contraction
// <M_k, (m_0 (x) ... (x) m_{k-1}) M_k> = sum_{i,l} prod_j (psi_i^T m_j psi_l)   
std::vector<double> moment_tensor(const bp_t &s, unsigned k) {
    uint32_t Q = s.Q;
    size_t T = 1;
    for (unsigned j = 0; j < k; ++j) T *= Q;
    std::vector<double> Mk(T, 0.0);
    std::vector<uint32_t> idx(k);
    for (uint32_t i = 0; i < s.N; ++i) {
        const double *p = &s.psi[size_t(i) * Q];
        for (size_t e = 0; e < T; ++e) {
            size_t r = e;
            double v = 1.0;
            for (unsigned j = 0; j < k; ++j) { v *= p[r % Q]; r /= Q; }
            Mk[e] += v;
        }
    }
    return Mk;
}
double contract(const std::vector<double> &Mk, uint32_t Q, const std::vector<const double *> &mats) {
    unsigned k = unsigned(mats.size());
    size_t T = Mk.size();
    double acc = 0.0;
    for (size_t a = 0; a < T; ++a) {
        if (Mk[a] == 0.0) continue;
        for (size_t b = 0; b < T; ++b) {
            double w = 1.0;
            size_t ra = a, rb = b;
            for (unsigned j = 0; j < k; ++j) { w *= mats[j][(ra % Q) * Q + (rb % Q)]; ra /= Q; rb /= Q; }
            acc += w * Mk[a] * Mk[b];
        }
    }
    return acc;
}

double f_nonedge_series(const bp_t &s, unsigned K) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q, N = s.N;
    if (s.dc != 0) return 0.0;
    std::vector<double> w(size_t(Q) * Q);
    for (uint32_t a = 0; a < Q * Q; ++a) w[a] = double(N) * (1.0 - std::pow((1 - s.cab[a] / N), s.beta));
    double all = 0.0, Nk = 1.0;
    for (unsigned k = 1; k <= K; ++k) {
        Nk *= double(N);
        std::vector<double> Mk = moment_tensor(s, k);
        std::vector<const double *> mats(k, w.data());
        all -= contract(Mk, Q, mats) / (double(k) * Nk);
    }
    double adjsum = 0.0;
    for (uint32_t i = 0; i < N; ++i)
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
            uint32_t l = g.nbr[k];
            double y = 0.0;
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = 0; q2 < Q; ++q2) y += w[q1 * Q + q2] * s.psi[size_t(i) * Q + q1] * s.psi[size_t(l) * Q + q2];
            adjsum += std::log1p(-y / N);
        }
    return (all - adjsum) / (2. * N);
}

// ---- entropy (:506-560, :614-672, :711-741, :752-758), restated literally incl. the zero term ----
double e_site(const bp_t &s) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    if (s.dc != 0) return std::numeric_limits<double>::quiet_NaN();  // 0/0 at :556 (B11)
    double es = 0.0;
    for (uint32_t i = 0; i < s.N; ++i) {
        double num = 0., den = 0.;
        uint64_t k0 = g.row_ptr[i], k1 = g.row_ptr[i + 1];
        for (uint32_t q = 0; q < Q; ++q) {
            double a = 0.0;
            for (uint64_t k = k0; k < k1; ++k) {
                double b = 0;
                const double *m = &s.M[size_t(g.rev[k]) * Q];
                for (uint32_t t = 0; t < Q; ++t) b += s.cab[t * Q + q] * m[t];
                a += std::log(b);
            }
            double a2 = 0.;  // numerator_a_2 (:529-548): identically 0 for finite inputs, NaN if log(cab)=-inf meets 0
            for (uint64_t k = k0; k < k1; ++k) {
                double b2 = 0.;
                const double *m = &s.M[size_t(g.rev[k]) * Q];
                for (uint32_t t = 0; t < Q; ++t) b2 += std::log(s.cab[t * Q + q]) * s.cab[t * Q + q] * m[t];
                double sumlogs = 0;
                for (uint64_t kk = k0; kk < k1; ++kk) {
                    double ex = 0.;
                    const double *mm = &s.M[size_t(g.rev[kk]) * Q];
                    for (uint32_t t = 0; t < Q; ++t) if (kk != k) ex += s.cab[t * Q + q] * mm[t];
                    sumlogs += std::log(ex);
                }
                a2 += b2 * std::exp(sumlogs);
            }
            den += std::exp(a + s.logeta[q] - s.h[q] / s.N);
            num += std::exp(a + s.logeta[q] - s.h[q] / s.N) * (-s.h[q] / s.N);
            num += a2 * std::exp(s.logeta[q]) / std::exp(s.h[q] / s.N);
        }
        es += num / den;
    }
    return es / s.N;
}

double e_edge(const bp_t &s) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    double sl = 0.0;
    for (uint32_t i = 0; i < s.N; ++i) {
        double di = double(g.deg(i));
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
            double dl = double(g.deg(g.nbr[k]));
            const double *min = &s.M[size_t(g.rev[k]) * Q];
            const double *mout = &s.M[k * Q];
            double num = 0., den = 0.;
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = q1; q2 < Q; ++q2) {
                    double w;
                    if (s.dc == 0) w = s.cab[q1 * Q + q2];
                    else if (s.dc == 1) w = di * dl * s.cab[q1 * Q + q2];
                    else { double tmp = di * dl * s.pab[q1 * Q + q2]; w = tmp / (1.0 + tmp); }
                    double pr = (q1 == q2) ? (min[q1] * mout[q2]) : (min[q1] * mout[q2] + min[q2] * mout[q1]);
                    den += w * pr;
                    num += w * std::log(s.cab[q1 * Q + q2]) * pr;
                }
            sl += num / den;
        }
    }
    return sl / (2. * s.N);
}

double e_nonedge_exact(const bp_t &s) {  // :711-741
    const graph_t &g = *s.g;
    uint32_t Q = s.Q, N = s.N;
    if (s.dc != 0) return 0.0;
    std::vector<char> adj(N, 0);
    double acc = 0.0;
    for (uint32_t i = 0; i < N; ++i) {
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) adj[g.nbr[k]] = 1;
        for (uint32_t l = 0; l < N; ++l) {
            if (adj[l]) continue;
            double num = 0., den = 0.;
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = 0; q2 < Q; ++q2) {
                    double pi = s.psi[size_t(i) * Q + q1], pl = s.psi[size_t(l) * Q + q2];
                    den += (1 - s.cab[q1 * Q + q2] / N) * pi * pl;
                    num += (s.cab[q1 * Q + q2] / N) * std::log(s.cab[q1 * Q + q2]) * pi * pl;
                }
            if (num * den != 0) acc += num / den;
        }
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) adj[g.nbr[k]] = 0;
    }
    return acc / (2. * N);
}

// series form of e_nonedge: sum_{i,l} (u/N) sum_{k>=0} (y/N)^k minus the adjacent pairs
double e_nonedge_series(const bp_t &s, unsigned K) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q, N = s.N;
    if (s.dc != 0) return 0.0;
    std::vector<double> v(size_t(Q) * Q);
    for (uint32_t a = 0; a < Q * Q; ++a) v[a] = s.cab[a] * std::log(s.cab[a]);
    double all = 0.0, Nk = 1.0;
    for (unsigned k = 0; k < K; ++k) {  // term k uses M_{k+1}
        Nk *= double(N);
        std::vector<double> Mk = moment_tensor(s, k + 1);
        std::vector<const double *> mats(k + 1, s.cab.data());
        mats[0] = v.data();
        all += contract(Mk, Q, mats) / Nk;
    }
    double adjsum = 0.0;
    for (uint32_t i = 0; i < N; ++i)
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
            uint32_t l = g.nbr[k];
            double u = 0.0, y = 0.0;
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = 0; q2 < Q; ++q2) {
                    double pp = s.psi[size_t(i) * Q + q1] * s.psi[size_t(l) * Q + q2];
                    u += v[q1 * Q + q2] * pp;
                    y += s.cab[q1 * Q + q2] * pp;
                }
            adjsum += (u / N) / (1.0 - y / N);
        }
    return (all - adjsum) / (2. * N);
}

// ---- EM expectations (:428-440, :892-989) -------------------------------------------------------
void em_expect(bp_t &s) {
    const graph_t &g = *s.g;
    uint32_t Q = s.Q;
    for (uint32_t q = 0; q < Q; ++q) { s.na_expect[q] = 0.0; s.nna_expect[q] = 0.0; }
    for (uint32_t i = 0; i < s.N; ++i)
        for (uint32_t q = 0; q < Q; ++q) {
            s.na_expect[q] += s.psi[size_t(i) * Q + q];
            s.nna_expect[q] += g.deg(i) * s.psi[size_t(i) * Q + q];
        }
    std::fill(s.cab_expect.begin(), s.cab_expect.end(), 0.0);
    for (uint32_t i = 0; i < s.N; ++i) {
        double di = double(g.deg(i));
        for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
            double dl = double(g.deg(g.nbr[k]));
            const double *min = &s.M[size_t(g.rev[k]) * Q];
            const double *mout = &s.M[k * Q];
            auto wgt = [&](uint32_t q1, uint32_t q2) {
                if (s.dc == 0) return s.cab[q1 * Q + q2];
                if (s.dc == 1) return di * dl * s.cab[q1 * Q + q2];
                double tmp = di * dl * s.pab[q1 * Q + q2];
                return tmp / (1.0 + tmp);
            };
            double norm_L = 0.0;
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = q1; q2 < Q; ++q2) {
                    if (q1 == q2) norm_L += wgt(q1, q2) * (min[q1] * mout[q2]);
                    else norm_L += wgt(q1, q2) * (min[q1] * mout[q2] + min[q2] * mout[q1]);
                }
            for (uint32_t q1 = 0; q1 < Q; ++q1)
                for (uint32_t q2 = q1; q2 < Q; ++q2) {
                    if (q1 == q2) s.cab_expect[q1 * Q + q2] += 0.5 * wgt(q1, q2) * (min[q1] * mout[q2]) / norm_L;
                    else {
                        s.cab_expect[q1 * Q + q2] += 0.5 * wgt(q1, q2) * (min[q1] * mout[q2] + min[q2] * mout[q1]) / norm_L;
                        s.cab_expect[q2 * Q + q1] = s.cab_expect[q1 * Q + q2];
                    }
                }
        }
    }
    for (uint32_t q1 = 0; q1 < Q; ++q1)
        for (uint32_t q2 = q1; q2 < Q; ++q2) {
            if ((s.na_expect[q1] > s.EPS) && (s.na_expect[q2] > s.EPS)) {
                const std::vector<double> &nn = (s.dc == 0) ? s.na_expect : s.nna_expect;
                if (q1 != q2) {
                    s.cab_expect[q1 * Q + q2] *= s.N / (nn[q1] * nn[q2]);
                    s.cab_expect[q2 * Q + q1] = s.cab_expect[q1 * Q + q2];
                } else {
                    s.cab_expect[q1 * Q + q2] *= 2. * s.N / (nn[q1] * nn[q2]);
                }
            }
        }
}

// learning_step (:53-75). snap = 0 is the reference; the synchronous EM run of the engine treats a value within snap
// below an integer as that integer (include/sbmbp.h: sbmbp_set_learning_schedule).
void learning_step(bp_t &s, float learning_rate, double snap = 0.0) {
    uint32_t Q = s.Q;
    auto _N = s.N;
    for (uint32_t i = 0; i + 1 < Q; ++i) {
        s.na[i] = unsigned(int(learning_rate * s.na_expect[i] + (1.0 - learning_rate) * s.na[i] + snap));
        _N -= s.na[i];
    }
    s.na[Q - 1] = _N;
    for (uint32_t i = 0; i < Q; ++i) {
        s.eta[i] = double(s.na[i]) / s.N;
        s.logeta[i] = std::log(s.eta[i]);
        for (uint32_t j = 0; j < Q; ++j) {
            s.cab[i * Q + j] = learning_rate * s.cab_expect[i * Q + j] + (1.0 - learning_rate) * s.cab[i * Q + j];
            s.logcab[i * Q + j] = std::log(s.cab[i * Q + j]);
            s.pab[i * Q + j] = s.cab[i * Q + j] / s.N;
            s.W[i * Q + j] = std::pow(s.cab[i * Q + j], s.beta);
        }
    }
}

// compute_overlap (:775-811): max over label permutations (identity only when Q > 8)
double overlap(bp_t &s) {
    uint32_t Q = s.Q;
    std::vector<uint32_t> perm(Q);
    std::iota(perm.begin(), perm.end(), 0u);
    double max_ov = -1.0;
    do {
        double ov = 0.0;
        for (uint32_t i = 0; i < s.N; ++i) ov += s.psi[size_t(i) * Q + perm[s.conf_true[i]]];
        ov /= s.N;
        if (ov > max_ov) max_ov = ov;
        if (Q > 8) break;
    } while (std::next_permutation(perm.begin(), perm.end()));
    return max_ov;
}

double free_energy(const bp_t &s, int series_K, double *parts) {
    double fs = f_site(s), fe = f_edge(s);
    double fn = series_K > 0 ? f_nonedge_series(s, unsigned(series_K)) : f_nonedge_exact(s);
    if (parts) { parts[0] = fs; parts[1] = fe; parts[2] = fn; }
    return (-fs + fe + fn);  // :744-750
}

}  // namespace

// =================================== C interface for ctypes ======================================
extern "C" {

void *orc_graph_from_edges(const uint32_t *pairs, uint64_t n_pairs, uint32_t N) { return graph_from_edges(pairs, n_pairs, N); }
void *orc_graph_load_edgelist(const char *path, uint32_t N) {
    std::vector<uint32_t> pairs;
    load_edge_pairs(path, pairs);  // an unopened file yields an empty graph, as main.cpp:278-280 ignores the status
    return graph_from_edges(pairs.data(), pairs.size() / 2, N);
}
void orc_graph_free(void *g) { delete static_cast<graph_t *>(g); }
uint32_t orc_graph_n(void *g) { return static_cast<graph_t *>(g)->N; }
uint64_t orc_graph_e2(void *g) { return static_cast<graph_t *>(g)->E2(); }
void orc_graph_copy(void *gp, uint64_t *row_ptr, uint32_t *nbr, uint32_t *rev) {
    auto *g = static_cast<graph_t *>(gp);
    std::copy(g->row_ptr.begin(), g->row_ptr.end(), row_ptr);
    std::copy(g->nbr.begin(), g->nbr.end(), nbr);
    std::copy(g->rev.begin(), g->rev.end(), rev);
}

void *orc_rng_create(unsigned seed) { return new rng_t(seed); }
void orc_rng_free(void *r) { delete static_cast<rng_t *>(r); }
double orc_rng_draw(void *r) { return static_cast<rng_t *>(r)->draw(); }

// blockmodel.cpp:229-272 (bp_param_from_epsilon_c): every na[q] = unsigned(int(pa*N)) (B7)
void orc_param_from_epsilon_c(uint32_t N, uint32_t Q, double epsilon, double c, double *cab, uint32_t *na) {
    double cin, co;
    for (uint32_t q = 0; q < Q; ++q) na[q] = unsigned(int((1.0 / Q) * N));
    if (epsilon < 0) { cin = 0; co = c * Q / (Q - 1); }
    else { cin = c * Q / ((Q - 1) * epsilon + 1); co = epsilon * cin; }
    for (uint32_t q = 0; q < Q; ++q)
        for (uint32_t t = 0; t < Q; ++t) cab[q * Q + t] = (q == t) ? cin : co;
}
// blockmodel.cpp:274-302 (bp_param_from_direct): cab given as upper triangle, row-major
void orc_param_from_direct(uint32_t N, uint32_t Q, const double *pa, const double *cabv, double *cab, uint32_t *na) {
    for (uint32_t q = 0; q < Q; ++q) na[q] = unsigned(int(pa[q] * N));
    for (uint32_t q = 0; q < Q; ++q) {
        cab[q * Q + q] = cabv[q * Q - q * (q - 1) / 2];
        for (uint32_t t = q + 1; t < Q; ++t) {
            cab[q * Q + t] = cabv[q * Q - q * (q - 1) / 2 + t - q];
            cab[t * Q + q] = cab[q * Q + t];
        }
    }
}

void *orc_bp_create(void *g, uint32_t Q, uint32_t dc) {
    auto *s = new bp_t();
    s->g = static_cast<graph_t *>(g);
    s->N = s->g->N;
    s->Q = Q;
    s->dc = dc;
    return s;
}
void orc_bp_free(void *s) { delete static_cast<bp_t *>(s); }
void orc_bp_init_messages(void *s, unsigned flag, const int32_t *conf, const uint32_t *true_conf, void *rng) {
    init_messages(*static_cast<bp_t *>(s), flag, conf, true_conf, *static_cast<rng_t *>(rng));
}
void orc_bp_set_params(void *s, const double *cab, const uint32_t *na, double beta) { set_params(*static_cast<bp_t *>(s), cab, na, beta); }
void orc_bp_get_params(void *sp, double *cab, uint32_t *na) {
    auto &s = *static_cast<bp_t *>(sp);
    std::copy(s.cab.begin(), s.cab.end(), cab);
    std::copy(s.na.begin(), s.na.end(), na);
}
void orc_bp_get_state(void *sp, double *psi, double *msg_out) {
    auto &s = *static_cast<bp_t *>(sp);
    if (psi) std::copy(s.psi.begin(), s.psi.end(), psi);
    if (msg_out) std::copy(s.M.begin(), s.M.end(), msg_out);
}
void orc_bp_set_state(void *sp, const double *psi, const double *msg_out) {
    auto &s = *static_cast<bp_t *>(sp);
    if (psi) std::copy(psi, psi + s.psi.size(), s.psi.begin());
    if (msg_out) std::copy(msg_out, msg_out + s.M.size(), s.M.begin());
    s.consistent = false;
    s.init_flag = 0;  // an arbitrary state: clamped rows are no longer known to be one-hot
}
void orc_bp_get_h(void *sp, double *h) { auto &s = *static_cast<bp_t *>(sp); std::copy(s.h.begin(), s.h.end(), h); }
void orc_bp_init_h(void *sp) { init_h(*static_cast<bp_t *>(sp)); }
void orc_bp_compute_h(void *sp) { compute_h_full(*static_cast<bp_t *>(sp)); }
void orc_bp_set_field_mix(void *sp, double a) { static_cast<bp_t *>(sp)->field_mix = a; static_cast<bp_t *>(sp)->Sprev.clear(); }
// the same without forgetting the previous sums (a schedule that lowers the mix in the middle of a run)
void orc_bp_set_field_mix_keep(void *sp, double a) { static_cast<bp_t *>(sp)->field_mix = a; }
int orc_bp_learn_unconverged(void *sp) { return static_cast<bp_t *>(sp)->learn_unconverged; }
void orc_bp_set_auto_relax(void *sp, int on) { static_cast<bp_t *>(sp)->auto_relax = on != 0; }
void orc_bp_set_msg_form(void *sp, int on) { static_cast<bp_t *>(sp)->msg_form_only = on != 0; }
void orc_bp_ar_levels(void *sp, int *fl, int *gl) { *fl = static_cast<bp_t *>(sp)->ar_fl; *gl = static_cast<bp_t *>(sp)->ar_gl; }
double orc_bp_node_update(void *sp, uint32_t i, double damp, int large) {
    auto &s = *static_cast<bp_t *>(sp);
    return large ? node_update_large(s, i, damp) : node_update_small(s, i, damp);
}
int orc_bp_converge_async(void *sp, float crit, unsigned tmax, float damp, void *rng, int conditional) {
    return converge_async(*static_cast<bp_t *>(sp), crit, tmax, damp, *static_cast<rng_t *>(rng), conditional);
}
double orc_bp_sweep_sync(void *sp, double damp) { return sweep_sync(*static_cast<bp_t *>(sp), damp); }
int orc_bp_converge_sync(void *sp, double crit, unsigned tmax, double damp, double *last) {
    return converge_sync(*static_cast<bp_t *>(sp), crit, tmax, damp, last);
}
// parts = {f_site, f_edge, f_nonedge}; series_K = 0 -> exact O(N^2) loop of the reference
double orc_bp_free_energy(void *sp, int series_K, double *parts) { return free_energy(*static_cast<bp_t *>(sp), series_K, parts); }
double orc_bp_entropy(void *sp, int series_K, double *parts) {
    auto &s = *static_cast<bp_t *>(sp);
    double es = e_site(s), ee = e_edge(s);
    double en = series_K > 0 ? e_nonedge_series(s, unsigned(series_K)) : e_nonedge_exact(s);
    if (parts) { parts[0] = es; parts[1] = ee; parts[2] = en; }
    return (-es + ee - en);  // :752-758
}
void orc_bp_em_expect(void *sp, double *na_e, double *nna_e, double *cab_e) {
    auto &s = *static_cast<bp_t *>(sp);
    em_expect(s);
    if (na_e) std::copy(s.na_expect.begin(), s.na_expect.end(), na_e);
    if (nna_e) std::copy(s.nna_expect.begin(), s.nna_expect.end(), nna_e);
    if (cab_e) std::copy(s.cab_expect.begin(), s.cab_expect.end(), cab_e);
}
double orc_bp_overlap(void *sp) { return overlap(*static_cast<bp_t *>(sp)); }

// learning (:14-51). sync = 0: the reference's asynchronous converge; sync = 1: Jacobi converge
// (what the engine runs). series_K selects the non-edge evaluation. Returns the number of EM
// steps taken (learning_step calls); f_out = last free energy.
int orc_bp_learning(void *sp, float learning_conv_crit, unsigned learning_max_time, float learning_rate,
                    float dumping_rate, void *rng, int sync, int series_K, double *f_out) {
    auto &s = *static_cast<bp_t *>(sp);
    double fold = 0.0, fdiff = 1.0;
    int steps = 0;
    s.learn_unconverged = 0;
    // the two rules of the engine's synchronous EM loop (sbmbp_set_learning_schedule defaults): field relaxation 0.3
    // inside the BP runs, and the snap tolerance min(N * crit, 0.01) of the group-size truncation
    const double keep_mix = s.field_mix;
    if (sync) { s.field_mix = std::min(s.field_mix, 0.3); s.Sprev.clear(); }
    for (unsigned t = 0; t < learning_max_time; ++t) {
        if (fdiff < learning_conv_crit) learning_conv_crit *= 0.1;
        const int it = sync ? converge_sync(s, learning_conv_crit, learning_max_time, dumping_rate, nullptr)
                            : converge_async(s, learning_conv_crit, learning_max_time, dumping_rate, *static_cast<rng_t *>(rng), 0);
        if (it < 0) ++s.learn_unconverged;
        em_expect(s);
        double fnew = free_energy(s, series_K, nullptr);
        fdiff = std::fabs(fnew - fold);
        fold = fnew;
        if (std::isnan(fold) || std::isinf(fold)) break;
        if (fdiff < learning_conv_crit) break;
        learning_step(s, learning_rate, sync ? std::min(1.0 * double(s.N) * double(learning_conv_crit), 0.01) : 0.0);
        ++steps;
    }
    s.field_mix = keep_mix;
    if (f_out) *f_out = fold;
    return steps;
}

}  // extern "C"
