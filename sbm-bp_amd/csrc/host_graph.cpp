// Host-side graph + parameter utilities (C++14). See include/sbmbp.h for the reference lines
// each function replaces.
#include "host_graph.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <thread>

#include "../../include/sbmbp.h"

namespace sbmbp {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const std::string &get_error() { return g_err; }
int arg_error(const char *func, int line) {
    g_err = std::string("invalid argument (") + func + ", check at line " + std::to_string(line) + ")";
    return -1;  // SBMBP_ERR_ARG
}

// run fn(t, lo, hi) over [0, n) split into contiguous pieces on up to `want` threads
template <typename F> static void parallel_ranges(uint64_t n, unsigned want, F fn) {
    unsigned nt = std::max(1u, std::min<unsigned>(want, unsigned(n / 65536 + 1)));
    if (nt == 1) { fn(0u, uint64_t(0), n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(fn, t, n * t / nt, n * (t + 1) / nt);
    for (auto &x : th) x.join();
}

// SBMBP_HOST_TIMING=1: phase times of the host-side graph build on stderr (measurement aid)
struct phase_timer {
    bool on = std::getenv("SBMBP_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[sbmbp host] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

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
    const unsigned nt = host_threads();
    phase_timer pt;
    uint32_t n = n_vertices;
    {   // the reference grows the adjacency to the largest id (:65-72)
        std::vector<uint32_t> top(nt, 0);
        std::vector<char> bad(nt, 0);
        parallel_ranges(n_pairs, nt, [&](unsigned t, uint64_t lo, uint64_t hi) {
            uint32_t m = 0;
            for (uint64_t e = lo; e < hi; ++e) m = std::max(m, std::max(pairs[2 * e], pairs[2 * e + 1]));
            if (hi > lo && m == UINT32_MAX) bad[t] = 1;
            top[t] = hi > lo ? m + 1 : 0;
        });
        for (unsigned t = 0; t < nt; ++t) {
            if (bad[t]) { set_error("vertex id 2^32-1 is reserved"); return SBMBP_ERR_ARG; }
            n = std::max(n, top[t]);
        }
    }
    if (2 * n_pairs >= (uint64_t(1) << 32)) { set_error("more than 2^32-1 directed edges"); return SBMBP_ERR_UNSUPPORTED; }
    // bucket by source: the pair array is cut into one piece per thread; bucket sizes and slots are claimed
    // with relaxed atomic adds (the order inside a bucket is arbitrary here and fixed by the sort below)
    std::vector<uint32_t> fill(size_t(n), 0);
    parallel_ranges(n_pairs, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t e = lo; e < hi; ++e) {
            uint32_t a = pairs[2 * e], b = pairs[2 * e + 1];
            __atomic_fetch_add(&fill[a], 1u, __ATOMIC_RELAXED);
            if (a != b) __atomic_fetch_add(&fill[b], 1u, __ATOMIC_RELAXED);  // a self-loop is one set entry
        }
    });
    pt.lap("max id + bucket sizes");
    std::vector<uint64_t> cnt(size_t(n) + 1, 0);
    for (uint32_t i = 0; i < n; ++i) { cnt[i + 1] = cnt[i] + fill[i]; fill[i] = 0; }
    std::vector<uint32_t> adj(cnt[n]);
    parallel_ranges(n_pairs, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t e = lo; e < hi; ++e) {
            uint32_t a = pairs[2 * e], b = pairs[2 * e + 1];
            adj[cnt[a] + __atomic_fetch_add(&fill[a], 1u, __ATOMIC_RELAXED)] = b;
            if (a != b) adj[cnt[b] + __atomic_fetch_add(&fill[b], 1u, __ATOMIC_RELAXED)] = a;
        }
    });
    std::vector<uint32_t>().swap(fill);
    pt.lap("bucket fill");
    // sort + unique every row in place (front of its bucket), remember the unique length
    std::vector<uint32_t> ulen(n, 0);
    parallel_ranges(n, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            uint32_t *b = adj.data() + cnt[i], *e = adj.data() + cnt[i + 1];
            std::sort(b, e);
            ulen[i] = uint32_t(std::unique(b, e) - b);
        }
    });
    pt.lap("row sort + unique");
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
    pt.lap("compaction");
    // Reverse index by counting: rows are sorted, so while the edge array is read front to back (sources ascending) the
    // c-th time vertex j appears as a neighbour, the source is j's c-th smallest neighbour: rev[k] = row_ptr[j] + c.
    // Each thread owns a range of target vertices (its counters stay in cache), scans the whole array and handles the
    // entries that fall in its range; no searches, no random reads of other rows.
    g.rev.resize(g.nbr.size());
    {
        const uint64_t e2 = g.nbr.size();
        unsigned parts = std::max(1u, std::min<unsigned>(nt, unsigned(e2 / 262144 + 1)));
        std::vector<uint32_t> seen(n, 0);
        // target ranges with equal shares of the edges
        std::vector<uint32_t> cut(parts + 1, n);
        cut[0] = 0;
        for (unsigned t = 1; t < parts; ++t)
            cut[t] = uint32_t(std::lower_bound(g.row_ptr.begin(), g.row_ptr.end(), e2 * t / parts) - g.row_ptr.begin());
        for (unsigned t = 1; t <= parts; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
        cut[parts] = n;
        std::vector<std::thread> th;
        auto work = [&](unsigned t) {
            const uint32_t lo = cut[t], hi = cut[t + 1];
            if (lo >= hi) return;
            const uint32_t *nb = g.nbr.data();
            uint32_t *rv = g.rev.data();
            const uint64_t *rp = g.row_ptr.data();
            uint32_t *sn = seen.data();
            for (uint64_t k = 0; k < e2; ++k) {
                const uint32_t j = nb[k];
                if (j - lo < hi - lo) rv[k] = uint32_t(rp[j]) + sn[j]++;
            }
        };
        for (unsigned t = 1; t < parts; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    pt.lap("reverse index");
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
namespace {
struct parse_result {
    std::vector<uint32_t> pairs;
    uint64_t lines = 0;       // complete lines consumed before the error (or all of them)
    int err = 0;              // 0 ok, 1 = not two integers, 2 = id too large, 3 = only one id
};

// parse [p, end): `end` is either the end of the file or just past a '\n' (chunks are cut at line starts)
void parse_lines(const char *p, const char *end, parse_result &out) {
    out.pairs.reserve(size_t(end - p) / 7);
    while (p < end) {
        uint64_t v[2] = {0, 0};
        int nv = 0;
        while (p < end && *p != '\n') {
            if (*p == ' ' || *p == '\t' || *p == '\r') { ++p; continue; }
            if (*p < '0' || *p > '9') {
                if (nv < 2) { out.err = 1; return; }
                while (p < end && *p != '\n') ++p;  // trailing columns (weights etc.) are ignored like the reference's stream reads
                break;
            }
            uint64_t x = 0;
            while (p < end && *p >= '0' && *p <= '9') { x = x * 10 + uint64_t(*p - '0'); ++p; }
            if (nv < 2) {
                if (x >= UINT32_MAX) { out.err = 2; return; }
                v[nv] = x;
            }
            ++nv;
        }
        if (nv == 1) { out.err = 3; return; }
        if (nv >= 2) { out.pairs.push_back(uint32_t(v[0])); out.pairs.push_back(uint32_t(v[1])); }
        ++p;
        ++out.lines;
    }
}
}  // namespace

// The file is mapped and cut at line starts into one piece per host thread; the pieces are parsed
// concurrently and concatenated in file order (so the result, and the first error reported, are those
// of a sequential read).
int read_edgelist(const char *path, std::vector<uint32_t> &pairs) {
    pairs.clear();
    phase_timer pt;
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { set_error(std::string("cannot open edge list: ") + path); return SBMBP_ERR_IO; }
    struct stat sb;
    if (::fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { ::close(fd); set_error(std::string("cannot stat edge list: ") + path); return SBMBP_ERR_IO; }
    const size_t sz = size_t(sb.st_size);
    if (sz == 0) { ::close(fd); return SBMBP_OK; }
    void *map = ::mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) { set_error(std::string("cannot map edge list: ") + path); return SBMBP_ERR_IO; }
    ::madvise(map, sz, MADV_SEQUENTIAL);
    const char *base = static_cast<const char *>(map);
    const unsigned nt = unsigned(std::max<size_t>(1, std::min<size_t>(host_threads(), sz / (size_t(4) << 20) + 1)));
    std::vector<size_t> cut(nt + 1, sz);
    cut[0] = 0;
    for (unsigned t = 1; t < nt; ++t) {
        size_t c = std::max(cut[t - 1], sz * t / nt);
        while (c < sz && c > 0 && base[c - 1] != '\n') ++c;  // advance to the next line start
        cut[t] = c;
    }
    std::vector<parse_result> res(nt);
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back([&, t] { parse_lines(base + cut[t], base + cut[t + 1], res[t]); });
        parse_lines(base + cut[0], base + cut[1], res[0]);
        for (auto &x : th) x.join();
    }
    ::munmap(map, sz);
    uint64_t line = 1;
    std::vector<size_t> off(nt + 1, 0);
    for (unsigned t = 0; t < nt; ++t) {
        if (res[t].err) {
            const std::string where = "edge list line " + std::to_string(line + res[t].lines);
            set_error(where + (res[t].err == 1 ? ": expected two non-negative integers" : res[t].err == 2 ? ": id too large" : ": only one id"));
            return SBMBP_ERR_IO;
        }
        line += res[t].lines;
        off[t + 1] = off[t] + res[t].pairs.size();
    }
    pt.lap("parse (mapped, parallel)");
    if (nt == 1) { pairs.swap(res[0].pairs); return SBMBP_OK; }
    pairs.resize(off[nt]);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] { if (!res[t].pairs.empty()) std::memcpy(pairs.data() + off[t], res[t].pairs.data(), res[t].pairs.size() * sizeof(uint32_t)); });
        for (auto &x : th) x.join();
    }
    pt.lap("concatenate");
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
//
// The draws are those of std::mt19937(seed) through std::uniform_real_distribution<double>(0, 1) (two
// 32-bit words per double, generate_canonical<double, 53>), reproduced bit for bit by a block generator:
// one thread produces the raw words of the next slab of vertices while the other host threads turn the
// previous slab into normalised vectors (a vertex's position in the stream follows from a prefix sum of
// 1 + degree over the vertices that draw at all).
namespace {
#if defined(__x86_64__)
// AVX2 bodies of the two MT19937 loops (eight words per step; the scalar code below is the definition)
// (no lambdas here: a lambda body does not inherit the function's target attribute)
#define SBMBP_MT_STEP(I, FAR)                                                                                           \
    do {                                                                                                                \
        const __m256i hi_ = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + (I)));                             \
        const __m256i lo_ = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + (I) + 1));                         \
        const __m256i far_ = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + (FAR)));                          \
        const __m256i y_ = _mm256_or_si256(_mm256_and_si256(hi_, upper), _mm256_and_si256(lo_, lower));                 \
        const __m256i mag_ = _mm256_and_si256(_mm256_sub_epi32(zero, _mm256_and_si256(y_, one)), matrix);               \
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(s + (I)),                                                       \
                            _mm256_xor_si256(_mm256_xor_si256(far_, _mm256_srli_epi32(y_, 1)), mag_));                  \
    } while (0)
#define SBMBP_MT_SCALAR(I, NXT, FAR)                                                  \
    do {                                                                              \
        const uint32_t y_ = (s[(I)] & 0x80000000u) | (s[(NXT)] & 0x7fffffffu);        \
        s[(I)] = s[(FAR)] ^ (y_ >> 1) ^ ((0u - (y_ & 1u)) & 0x9908b0dfu);             \
    } while (0)
__attribute__((target("avx2"))) static void mt_twist_avx2(uint32_t *s) {
    const __m256i upper = _mm256_set1_epi32(int(0x80000000u)), lower = _mm256_set1_epi32(0x7fffffff);
    const __m256i one = _mm256_set1_epi32(1), matrix = _mm256_set1_epi32(int(0x9908b0dfu)), zero = _mm256_setzero_si256();
    unsigned i = 0;
    for (; i + 8 <= 227; i += 8) SBMBP_MT_STEP(i, i + 397);  // reads s[i+1 .. i+8] and s[i+397 .. i+404]: all still old
    for (; i < 227; ++i) SBMBP_MT_SCALAR(i, i + 1, i + 397);
    for (; i + 8 <= 623; i += 8) SBMBP_MT_STEP(i, i - 227);  // s[i-227 ..] are new values, s[i+1 .. i+8] old
    for (; i < 623; ++i) SBMBP_MT_SCALAR(i, i + 1, i - 227);
    SBMBP_MT_SCALAR(623, 0, 396);
}
#undef SBMBP_MT_STEP
#undef SBMBP_MT_SCALAR
__attribute__((target("avx2"))) static void mt_temper_avx2(const uint32_t *s, uint32_t *out, unsigned count) {
    const __m256i b = _mm256_set1_epi32(int(0x9d2c5680u)), c = _mm256_set1_epi32(int(0xefc60000u));
    unsigned i = 0;
    for (; i + 8 <= count; i += 8) {
        __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i));
        y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 11));
        y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 7), b));
        y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 15), c));
        y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 18));
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + i), y);
    }
    for (; i < count; ++i) {
        uint32_t y = s[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        out[i] = y ^ (y >> 18);
    }
}
static const bool g_have_avx2 = __builtin_cpu_supports("avx2");
#else
static const bool g_have_avx2 = false;
#endif

struct mt19937_words {  // MT19937 (Matsumoto & Nishimura 1998), same parameters and seeding as std::mt19937
    uint32_t s[624 + 8];  // + 8: the vector loop's last load may touch (not use) words past the state
    unsigned pos = 624;
    explicit mt19937_words(uint32_t seed) {
        s[0] = seed;
        for (uint32_t i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + i;
        for (uint32_t i = 624; i < 632; ++i) s[i] = 0;
    }
    void twist() {
        pos = 0;
#if defined(__x86_64__)
        if (g_have_avx2) { mt_twist_avx2(s); return; }
#endif
        auto mix = [](uint32_t hi, uint32_t lo, uint32_t far) {
            uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
            return far ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        };
        for (unsigned i = 0; i < 227; ++i) s[i] = mix(s[i], s[i + 1], s[i + 397]);
        for (unsigned i = 227; i < 623; ++i) s[i] = mix(s[i], s[i + 1], s[i - 227]);
        s[623] = mix(s[623], s[0], s[396]);
    }
    static uint32_t temper(uint32_t y) {
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        return y ^ (y >> 18);
    }
    void fill(uint32_t *out, uint64_t count) {
        while (count) {
            if (pos == 624) twist();
            const unsigned take = unsigned(std::min<uint64_t>(count, 624 - pos));
#if defined(__x86_64__)
            if (g_have_avx2) mt_temper_avx2(s + pos, out, take);
            else
#endif
            for (unsigned i = 0; i < take; ++i) out[i] = temper(s[pos + i]);
            pos += take; out += take; count -= take;
        }
    }
};

// generate_canonical<double, 53> over a 32-bit engine: (w0 + w1 * 2^32) / 2^64, rounded as libstdc++ does
inline double canonical(const uint32_t *w) {
    double sum = double(w[0]);
    sum += double(w[1]) * 4294967296.0;
    double r = sum / 18446744073709551616.0;
    return r >= 1.0 ? std::nextafter(1.0, 0.0) : r;
}
}  // namespace

void init_state_host(uint32_t n, const uint32_t *row_ptr, uint64_t e2, uint32_t Q, uint32_t flag, const int32_t *conf,
                     uint32_t seed, double *psi, double *msg, const state_sink *sink) {
    phase_timer pt;
    (void)e2;
    auto planted = [&](uint32_t i) -> int32_t { return (flag == 0 || !conf) ? -1 : conf[i]; };
    auto draws = [&](uint32_t i) { return planted(i) == -1 || flag == 2; };  // hard-planted rows draw nothing
    const uint64_t words_per_vec = 2 * uint64_t(Q);
    const uint64_t slab_words = uint64_t(32) << 20;  // 128 MB of raw words per slab
    // slabs of consecutive vertices with at most slab_words words each (a single huge row may exceed it)
    std::vector<uint32_t> slab_start{0};
    std::vector<uint64_t> slab_size;
    {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t w = draws(i) ? (1 + uint64_t(row_ptr[i + 1] - row_ptr[i])) * words_per_vec : 0;
            if (acc && acc + w > slab_words) { slab_start.push_back(i); slab_size.push_back(acc); acc = 0; }
            acc += w;
        }
        slab_start.push_back(n);
        slab_size.push_back(acc);
    }
    const size_t n_slabs = slab_size.size();
    uint64_t biggest = 0;
    for (uint64_t w : slab_size) biggest = std::max(biggest, w);
    std::vector<uint32_t> raw[2];
    raw[0].resize(biggest);
    if (n_slabs > 1) raw[1].resize(biggest);
    mt19937_words gen(seed);
    const unsigned nt = host_threads();
    // with a sink the rows of a slab go to a reusable buffer that is handed over as soon as the slab is done (the engine
    // uploads it while the generator thread is busy with the next slab); without, they go to the caller's full arrays
    std::unique_ptr<double[]> pbuf, mbuf;
    double *pslab = nullptr, *mslab = nullptr;
    if (sink) {
        uint64_t max_rows = 1, max_edges = 1;
        for (size_t sl = 0; sl < n_slabs; ++sl) {
            max_rows = std::max<uint64_t>(max_rows, slab_start[sl + 1] - slab_start[sl]);
            max_edges = std::max<uint64_t>(max_edges, uint64_t(row_ptr[slab_start[sl + 1]]) - row_ptr[slab_start[sl]]);
        }
        if (!sink->alloc || !sink->alloc(max_rows * Q, max_edges * Q, &pslab, &mslab)) {
            pbuf.reset(new double[max_rows * Q]);
            mbuf.reset(new double[max_edges * Q]);
            pslab = pbuf.get();
            mslab = mbuf.get();
        }
    }
    auto consume = [&](size_t sl, const uint32_t *words) {
        const uint32_t lo = slab_start[sl], hi = slab_start[sl + 1];
        double *const pbase = sink ? pslab - size_t(lo) * Q : psi;                 // row i at pbase + i*Q
        double *const mbase = sink ? mslab - size_t(row_ptr[lo]) * Q : msg;        // edge k at mbase + k*Q
        // word offset of every vertex of the slab: prefix sum, then the rows are independent
        std::vector<uint64_t> off(size_t(hi - lo) + 1, 0);
        for (uint32_t i = lo; i < hi; ++i)
            off[i - lo + 1] = off[i - lo] + (draws(i) ? (1 + uint64_t(row_ptr[i + 1] - row_ptr[i])) * words_per_vec : 0);
        parallel_ranges(hi - lo, nt, [&](unsigned, uint64_t a, uint64_t b) {
            for (uint64_t r = a; r < b; ++r) {
                const uint32_t i = lo + uint32_t(r);
                const int32_t p = planted(i);
                const uint32_t *w = words + off[r];
                auto fill = [&](double *dst) {
                    if (p == -1) {
                        double norm = 0.0;
                        for (uint32_t q = 0; q < Q; ++q) { dst[q] = canonical(w + 2 * q); norm += dst[q]; }
                        for (uint32_t q = 0; q < Q; ++q) dst[q] /= norm;
                        w += words_per_vec;
                    } else if (flag == 2) {
                        const double noise = 0.1;
                        double norm = 0.0;
                        for (uint32_t q = 0; q < Q; ++q) {
                            const double u = canonical(w + 2 * q);
                            dst[q] = (int32_t(q) == p) ? noise + (1.0 - noise) * u : u * (1.0 - noise);
                            norm += dst[q];
                        }
                        for (uint32_t q = 0; q < Q; ++q) dst[q] /= norm;
                        w += words_per_vec;
                    } else {
                        for (uint32_t q = 0; q < Q; ++q) dst[q] = (int32_t(q) == p) ? 1.0 : 0.0;
                    }
                };
                fill(pbase + size_t(i) * Q);
                for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) fill(mbase + k * Q);
            }
        });
    };
    gen.fill(raw[0].data(), slab_size[0]);
    for (size_t sl = 0; sl < n_slabs; ++sl) {
        std::thread producer;
        if (sl + 1 < n_slabs) producer = std::thread([&, sl] { gen.fill(raw[(sl + 1) & 1].data(), slab_size[sl + 1]); });
        consume(sl, raw[sl & 1].data());
        if (sink) sink->put(slab_start[sl], slab_start[sl + 1], pslab, mslab);
        if (producer.joinable()) producer.join();
    }
    pt.lap("initial state (mt19937)");
}

}  // namespace sbmbp
