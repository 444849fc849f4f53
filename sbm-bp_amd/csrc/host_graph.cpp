// Host-side graph + parameter utilities (C++14). See include/sbmbp.h for the reference lines
// each function replaces.
#include "host_graph.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

#include "../../include/sbmbp.h"

namespace sbmbp {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const std::string &get_error() { return g_err; }

// run fn(t, lo, hi) over [0, n) split into contiguous pieces on up to `want` threads
template <typename F> static void parallel_ranges(uint64_t n, unsigned want, F fn) {
    unsigned nt = std::max(1u, std::min<unsigned>(want, unsigned(n / 65536 + 1)));
    if (nt == 1) { fn(0u, uint64_t(0), n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(fn, t, n * t / nt, n * (t + 1) / nt);
    for (auto &x : th) x.join();
}

static unsigned host_threads() {
    unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hc ? hc : 1u, 32u));
}

// Builds the symmetric, de-duplicated, sorted adjacency that edge_to_adj produces with
// std::set (graph_utilities.cpp:60-77) as flat CSR: bucket by source (counting sort), then
// sort+unique each row; reverse-edge index (belief_propagation.cpp:254-265) by binary search.
// The per-row work (sort, unique, compaction, reverse search) runs on host threads; the result does not
// depend on the thread count.
int graph_from_pairs(sbmbp_graph &g, const uint32_t *pairs, uint64_t n_pairs, uint32_t n_vertices) {
    uint32_t n = n_vertices;
    for (uint64_t e = 0; e < n_pairs; ++e) {  // the reference grows the adjacency to the largest id (:65-72)
        uint32_t a = pairs[2 * e], b = pairs[2 * e + 1];
        if (a == UINT32_MAX || b == UINT32_MAX) { set_error("vertex id 2^32-1 is reserved"); return SBMBP_ERR_ARG; }
        if (a >= n) n = a + 1;
        if (b >= n) n = b + 1;
    }
    if (2 * n_pairs >= (uint64_t(1) << 32)) { set_error("more than 2^32-1 directed edges"); return SBMBP_ERR_UNSUPPORTED; }
    const unsigned nt = host_threads();
    std::vector<uint64_t> cnt(size_t(n) + 1, 0);
    for (uint64_t e = 0; e < n_pairs; ++e) {
        uint32_t a = pairs[2 * e], b = pairs[2 * e + 1];
        cnt[a + 1]++;
        if (a != b) cnt[b + 1]++;  // a self-loop is one set entry
    }
    for (uint32_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    std::vector<uint32_t> adj(cnt[n]);
    {
        std::vector<uint64_t> pos(cnt.begin(), cnt.end() - 1);
        for (uint64_t e = 0; e < n_pairs; ++e) {
            uint32_t a = pairs[2 * e], b = pairs[2 * e + 1];
            adj[pos[a]++] = b;
            if (a != b) adj[pos[b]++] = a;
        }
    }
    // sort + unique every row in place (front of its bucket), remember the unique length
    std::vector<uint32_t> ulen(n, 0);
    parallel_ranges(n, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            uint32_t *b = adj.data() + cnt[i], *e = adj.data() + cnt[i + 1];
            std::sort(b, e);
            ulen[i] = uint32_t(std::unique(b, e) - b);
        }
    });
    g.n = n;
    g.row_ptr.assign(size_t(n) + 1, 0);
    uint32_t maxdeg = 0;
    for (uint32_t i = 0; i < n; ++i) { g.row_ptr[i + 1] = g.row_ptr[i] + ulen[i]; maxdeg = std::max(maxdeg, ulen[i]); }
    g.max_degree = maxdeg;
    g.nbr.resize(g.row_ptr[n]);
    parallel_ranges(n, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) std::copy(adj.data() + cnt[i], adj.data() + cnt[i] + ulen[i], g.nbr.data() + g.row_ptr[i]);
    });
    std::vector<uint32_t>().swap(adj);
    g.rev.resize(g.nbr.size());
    parallel_ranges(n, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i)
            for (uint64_t k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
                uint32_t j = g.nbr[k];
                const uint32_t *b = g.nbr.data() + g.row_ptr[j], *e = g.nbr.data() + g.row_ptr[j + 1];
                g.rev[k] = uint32_t(std::lower_bound(b, e, uint32_t(i)) - g.nbr.data());
            }
    });
    return SBMBP_OK;
}

int graph_from_csr(sbmbp_graph &g, uint32_t n, uint64_t e2, const uint64_t *row_ptr, const uint32_t *nbr,
                   const uint32_t *rev) {
    if (!row_ptr || (!nbr && e2)) { set_error("null CSR array"); return SBMBP_ERR_ARG; }
    if (e2 >= (uint64_t(1) << 32)) { set_error("more than 2^32-1 directed edges"); return SBMBP_ERR_UNSUPPORTED; }
    if (row_ptr[0] != 0 || row_ptr[n] != e2) { set_error("row_ptr does not span [0, E2]"); return SBMBP_ERR_ARG; }
    g.n = n;
    g.row_ptr.assign(row_ptr, row_ptr + n + 1);
    g.nbr.assign(nbr, nbr + e2);
    g.max_degree = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (row_ptr[i + 1] < row_ptr[i]) { set_error("row_ptr not monotone"); return SBMBP_ERR_ARG; }
        g.max_degree = std::max(g.max_degree, uint32_t(row_ptr[i + 1] - row_ptr[i]));
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            if (nbr[k] >= n) { set_error("neighbour id out of range"); return SBMBP_ERR_ARG; }
            if (k > row_ptr[i] && nbr[k] <= nbr[k - 1]) { set_error("neighbours must be strictly ascending per row"); return SBMBP_ERR_ARG; }
        }
    }
    g.rev.resize(e2);
    for (uint32_t i = 0; i < n; ++i)
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            uint32_t j = g.nbr[k];
            const uint32_t *b = g.nbr.data() + row_ptr[j], *e = g.nbr.data() + row_ptr[j + 1];
            const uint32_t *p = std::lower_bound(b, e, i);
            if (p == e || *p != i) { set_error("adjacency is not symmetric"); return SBMBP_ERR_ARG; }
            g.rev[k] = uint32_t(p - g.nbr.data());
            if (rev && rev[k] != g.rev[k]) { set_error("rev[] inconsistent with the adjacency"); return SBMBP_ERR_ARG; }
        }
    return SBMBP_OK;
}

// load_edge_list (graph_utilities.cpp:42-58): one "a b" pair per line, whitespace separated,
// 0-based ids. Deviations (documented): blank lines are skipped and a line without two
// non-negative integers is an error (the reference silently re-pushes the previous pair, B14).
int read_edgelist(const char *path, std::vector<uint32_t> &pairs) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open edge list: ") + path); return SBMBP_ERR_IO; }
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf(size_t(sz) + 1);
    size_t got = std::fread(buf.data(), 1, size_t(sz), f);
    std::fclose(f);
    buf[got] = '\n';
    pairs.clear();
    pairs.reserve(got / 6);
    const char *p = buf.data(), *end = buf.data() + got + 1;
    uint64_t line = 1;
    while (p < end) {
        uint64_t v[2];
        int nv = 0;
        while (p < end && *p != '\n') {
            if (*p == ' ' || *p == '\t' || *p == '\r') { ++p; continue; }
            if (*p < '0' || *p > '9') {
                if (nv < 2) { set_error("edge list line " + std::to_string(line) + ": expected two non-negative integers"); return SBMBP_ERR_IO; }
                while (p < end && *p != '\n') ++p;  // trailing columns (weights etc.) are ignored like the reference's stream reads
                break;
            }
            uint64_t x = 0;
            while (p < end && *p >= '0' && *p <= '9') { x = x * 10 + uint64_t(*p - '0'); ++p; }
            if (nv < 2) {
                if (x >= UINT32_MAX) { set_error("edge list line " + std::to_string(line) + ": id too large"); return SBMBP_ERR_IO; }
                v[nv] = x;
            }
            ++nv;
        }
        if (nv == 1) { set_error("edge list line " + std::to_string(line) + ": only one id"); return SBMBP_ERR_IO; }
        if (nv >= 2) { pairs.push_back(uint32_t(v[0])); pairs.push_back(uint32_t(v[1])); }
        ++p;
        ++line;
    }
    return SBMBP_OK;
}

// load_beliefs / load_confs (graph_utilities.cpp:8-40): one integer per line
int read_int_column(const char *path, std::vector<int64_t> &values) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open file: ") + path); return SBMBP_ERR_IO; }
    values.clear();
    char line[256];
    while (std::fgets(line, sizeof line, f)) {
        char *endp = nullptr;
        long long v = std::strtoll(line, &endp, 10);
        if (endp == line) continue;
        values.push_back(v);
    }
    std::fclose(f);
    return SBMBP_OK;
}

// bp_param_from_epsilon_c (blockmodel.cpp:229-272). The "last group gets the remainder" write at
// :243-245 is overwritten at :248, so every na[q] = unsigned(int(pa*N)).
void param_from_epsilon_c(uint32_t N, uint32_t Q, double epsilon, double c, double *cab, uint32_t *na) {
    for (uint32_t q = 0; q < Q; ++q) na[q] = unsigned(int((1.0 / Q) * N));
    double cin, co;
    if (epsilon < 0) { cin = 0; co = c * Q / (Q - 1); }
    else { cin = c * Q / ((Q - 1) * epsilon + 1); co = epsilon * cin; }
    for (uint32_t q = 0; q < Q; ++q)
        for (uint32_t t = 0; t < Q; ++t) cab[q * Q + t] = (q == t) ? cin : co;
}

// bp_param_from_direct (blockmodel.cpp:274-302): cab upper triangle row-major, mirrored
void param_from_direct(uint32_t N, uint32_t Q, const double *pa, const double *cabv, double *cab, uint32_t *na) {
    for (uint32_t q = 0; q < Q; ++q) na[q] = unsigned(int(pa[q] * N));
    for (uint32_t q = 0; q < Q; ++q) {
        uint32_t base = q * Q - q * (q - 1) / 2;
        cab[q * Q + q] = cabv[base];
        for (uint32_t t = q + 1; t < Q; ++t) {
            cab[q * Q + t] = cabv[base + t - q];
            cab[t * Q + q] = cab[q * Q + t];
        }
    }
}

// init_messages (belief_propagation.cpp:101-217). In the out-ordered layout the reference's
// per-vertex order "psi_i, then message i->j for each neighbour j ascending" (:112-130) is one
// sequential fill. flag 1: planted rows get one-hot psi/messages, others random (:132-174).
// flags 2/3 abort in the reference whenever a node is planted in group 1 (assert typo, B5);
// here 2 = planted + 0.1 noise (normalised) and 3 = hard planted on all given rows.
void init_state_host(uint32_t n, const uint32_t *row_ptr, uint64_t e2, uint32_t Q, uint32_t flag, const int32_t *conf,
                     uint32_t seed, std::vector<double> &psi, std::vector<double> &msg) {
    std::mt19937 engine(seed);
    std::uniform_real_distribution<> random_real(0, 1);
    psi.assign(size_t(n) * Q, 0.0);
    msg.assign(e2 * Q, 0.0);
    for (uint32_t i = 0; i < n; ++i) {
        int32_t p = (flag == 0 || !conf) ? -1 : conf[i];
        auto fill = [&](double *dst) {
            if (p == -1 || flag == 0) {
                double norm = 0.0;
                for (uint32_t q = 0; q < Q; ++q) { dst[q] = random_real(engine); norm += dst[q]; }
                for (uint32_t q = 0; q < Q; ++q) dst[q] /= norm;
            } else if (flag == 2) {
                const double noise = 0.1;
                double norm = 0.0;
                for (uint32_t q = 0; q < Q; ++q) {
                    dst[q] = (int32_t(q) == p) ? noise + (1.0 - noise) * random_real(engine)
                                               : random_real(engine) * (1.0 - noise);
                    norm += dst[q];
                }
                for (uint32_t q = 0; q < Q; ++q) dst[q] /= norm;
            } else {
                for (uint32_t q = 0; q < Q; ++q) dst[q] = (int32_t(q) == p) ? 1.0 : 0.0;
            }
        };
        fill(&psi[size_t(i) * Q]);
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) fill(&msg[k * Q]);
    }
}

}  // namespace sbmbp
