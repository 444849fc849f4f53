// Host-side sanitizer job: the engine's host code (csrc/host_graph.cpp: edge-list parser, CSR
// builder, the std::mt19937-compatible initial state, parameter constructors) and the CPU restatement (oracle/bp_oracle.cpp)
// built with -fsanitize=address,undefined and driven through their edge cases. CPU build only; no GPU code is involved.
// Exit code 0 and no sanitizer report = pass (tests/test_capi_cpu.py::test_host_code_under_asan_ubsan).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../sbm-bp_amd/csrc/host_graph.h"

extern "C" {
void *orc_graph_from_edges(const uint32_t *pairs, uint64_t n_pairs, uint32_t N);
void *orc_graph_load_edgelist(const char *path, uint32_t N);
void orc_graph_free(void *g);
uint32_t orc_graph_n(void *g);
uint64_t orc_graph_e2(void *g);
void orc_graph_copy(void *g, uint64_t *row_ptr, uint32_t *nbr, uint32_t *rev);
void *orc_rng_create(unsigned seed);
void orc_rng_free(void *r);
void orc_param_from_epsilon_c(uint32_t N, uint32_t Q, double eps, double c, double *cab, uint32_t *na);
void *orc_bp_create(void *g, uint32_t Q, uint32_t dc);
void orc_bp_free(void *s);
void orc_bp_init_messages(void *s, unsigned flag, const int32_t *conf, const uint32_t *true_conf, void *rng);
void orc_bp_set_params(void *s, const double *cab, const uint32_t *na, double beta);
void orc_bp_get_state(void *s, double *psi, double *msg);
int orc_bp_converge_async(void *s, float crit, unsigned tmax, float damp, void *rng, int conditional);
int orc_bp_converge_sync(void *s, double crit, unsigned tmax, double damp, double *last);
double orc_bp_free_energy(void *s, int series_K, double *parts);
double orc_bp_entropy(void *s, int series_K, double *parts);
void orc_bp_em_expect(void *s, double *na_e, double *nna_e, double *cab_e);
double orc_bp_overlap(void *s);
int orc_bp_learning(void *s, float lcrit, unsigned tmax, float lr, float damp, void *rng, int sync, int series_K, double *f);
}

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "host_sanitize: %s failed at line %d\n", #c, __LINE__); return 1; } } while (0)

static int run(const char *dataset) {
    using namespace sbmbp;
    // ---- parser: the shipped data set, a ragged file (blank lines, trailing spaces, no final newline), a missing file
    std::vector<uint32_t> pairs;
    REQUIRE(read_edgelist(dataset, pairs) == 0 && pairs.size() == 2 * 1498);
    const std::string tmp = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/host_sanitize_ragged.txt";
    { std::ofstream f(tmp); f << "0 1\n\n  2   3  \n3 4\r\n7 7\n5 6"; }
    std::vector<uint32_t> ragged;
    REQUIRE(read_edgelist(tmp.c_str(), ragged) == 0 && ragged.size() >= 10);
    std::vector<uint32_t> none;
    REQUIRE(read_edgelist("/nonexistent/edge/list", none) != 0);
    { std::ofstream f(tmp); }  // an empty file
    std::vector<uint32_t> empty;
    (void)read_edgelist(tmp.c_str(), empty);
    std::remove(tmp.c_str());
    // ---- CSR builder: duplicates, self-loops, isolated vertices, ids at the upper bound, an empty graph, one hub row
    sbmbp_graph g;
    REQUIRE(graph_from_pairs(g, pairs.data(), pairs.size() / 2, 1000) == 0 && g.n == 1000 && g.e2() == 2996);
    for (uint64_t k = 0; k < g.e2(); ++k) REQUIRE(g.rev[g.rev[k]] == k);
    sbmbp_graph g0;
    REQUIRE(graph_from_pairs(g0, nullptr, 0, 5) == 0 && g0.e2() == 0 && g0.n == 5);
    const uint32_t odd[] = {0, 0, 1, 2, 2, 1, 1, 2, 4, 4, 3, 0};
    sbmbp_graph g1;
    REQUIRE(graph_from_pairs(g1, odd, 6, 5) == 0 && g1.n == 5);
    std::vector<uint32_t> star;
    for (uint32_t v = 1; v < 700; ++v) { star.push_back(0); star.push_back(v); }
    sbmbp_graph gs;
    REQUIRE(graph_from_pairs(gs, star.data(), star.size() / 2, 700) == 0 && gs.max_degree == 699);
    sbmbp_graph gc;
    REQUIRE(graph_from_csr(gc, g.n, g.e2(), g.row_ptr.data(), g.nbr.data(), g.rev.data()) == 0 && gc.e2() == g.e2());
    const uint32_t beyond[] = {0, 9};
    sbmbp_graph gb;
    (void)graph_from_pairs(gb, beyond, 1, 5);  // an id >= n: grows or is rejected, must not write out of bounds
    // ---- parameters, incl. the truncation quirks and epsilon < 0
    double cab[9];
    uint32_t na[3];
    param_from_epsilon_c(1001, 3, 0.1, 3.0, cab, na);
    REQUIRE(na[0] == 333 && na[1] == 333 && na[2] == 333);
    param_from_epsilon_c(1000, 2, -1.0, 3.0, cab, na);
    const double pa[2] = {0.5, 0.5}, cu[3] = {3.63, 2.36, 3.63};
    param_from_direct(1000, 2, pa, cu, cab, na);
    REQUIRE(cab[1] == 2.36 && cab[2] == 2.36);
    // ---- the reference-compatible initial state: every init flag, with and without a sink, equal to the oracle's bit for bit
    std::vector<uint32_t> rp32(g.row_ptr.begin(), g.row_ptr.end());
    std::vector<uint32_t> tc(1000);
    for (uint32_t i = 0; i < 1000; ++i) tc[i] = i / 500;
    std::vector<int32_t> conf(1000, -1);
    for (uint32_t i = 0; i < 1000; i += 7) conf[i] = int32_t(tc[i]);
    void *og = orc_graph_from_edges(pairs.data(), pairs.size() / 2, 1000);
    REQUIRE(orc_graph_n(og) == 1000 && orc_graph_e2(og) == g.e2());
    for (unsigned flag = 0; flag < 4; ++flag) {
        std::vector<double> psi(size_t(1000) * 2), msg(g.e2() * 2), opsi(psi.size()), omsg(msg.size());
        init_state_host(1000, rp32.data(), g.e2(), 2, flag, flag ? conf.data() : nullptr, 11 + flag, psi.data(), msg.data());
        void *bp = orc_bp_create(og, 2, 0);
        void *rng = orc_rng_create(11 + flag);
        orc_bp_init_messages(bp, flag, flag ? conf.data() : nullptr, tc.data(), rng);
        orc_bp_get_state(bp, opsi.data(), omsg.data());
        REQUIRE(std::memcmp(psi.data(), opsi.data(), psi.size() * 8) == 0 && std::memcmp(msg.data(), omsg.data(), msg.size() * 8) == 0);
        orc_rng_free(rng);
        orc_bp_free(bp);
        // the streamed form: slabs handed to a sink
        std::vector<double> spsi(psi.size()), smsg(msg.size());
        state_sink sink;
        sink.put = [&](uint32_t lo, uint32_t hi, const double *pr, const double *mr) {
            std::memcpy(spsi.data() + size_t(lo) * 2, pr, size_t(hi - lo) * 2 * 8);
            std::memcpy(smsg.data() + size_t(rp32[lo]) * 2, mr, size_t(rp32[hi] - rp32[lo]) * 2 * 8);
        };
        init_state_host(1000, rp32.data(), g.e2(), 2, flag, flag ? conf.data() : nullptr, 11 + flag, nullptr, nullptr, &sink);
        REQUIRE(std::memcmp(psi.data(), spsi.data(), psi.size() * 8) == 0 && std::memcmp(msg.data(), smsg.data(), msg.size() * 8) == 0);
    }
    // ---- the oracle: both schedules (the synchronous one with its adaptive relaxation), reductions, learning, a hub graph
    double pcab[4];
    uint32_t pna[2];
    orc_param_from_epsilon_c(1000, 2, 0.1, 3.0, pcab, pna);
    for (uint32_t dc = 0; dc < 3; ++dc) {
        double c2[4] = {pcab[0], pcab[1], pcab[2], pcab[3]};
        if (dc) for (double &x : c2) x /= 9.0;
        void *bp = orc_bp_create(og, 2, dc);
        void *rng = orc_rng_create(0);
        orc_bp_init_messages(bp, 0, nullptr, tc.data(), rng);
        orc_bp_set_params(bp, c2, pna, 1.0);
        REQUIRE(orc_bp_converge_async(bp, 5e-6f, 200, 1.0f, rng, 1) >= 0);
        double last = 0, parts[3];
        REQUIRE(orc_bp_converge_sync(bp, 1e-9, 500, 1.0, &last) >= 0 && last < 1e-9);
        REQUIRE(std::isfinite(orc_bp_free_energy(bp, dc ? 2 : 0, parts)));
        (void)orc_bp_entropy(bp, 2, parts);
        double nae[2], nnae[2], cabe[4];
        orc_bp_em_expect(bp, nae, nnae, cabe);
        REQUIRE(orc_bp_overlap(bp) > 0.5);
        orc_rng_free(rng);
        orc_bp_free(bp);
    }
    {
        void *bp = orc_bp_create(og, 2, 0);
        void *rng = orc_rng_create(0);
        orc_bp_init_messages(bp, 0, nullptr, tc.data(), rng);
        const double c5[4] = {5, 1, 1, 5};
        orc_bp_set_params(bp, c5, pna, 1.0);
        double f = 0;
        REQUIRE(orc_bp_learning(bp, 1e-4f, 40, 0.2f, 1.0f, nullptr, 1, 2, &f) > 0 && std::isfinite(f));
        orc_rng_free(rng);
        orc_bp_free(bp);
    }
    {
        void *sg = orc_graph_from_edges(star.data(), star.size() / 2, 700);  // one row of 699 edges: the long-row products
        std::vector<uint32_t> t3(700);
        for (uint32_t i = 0; i < 700; ++i) t3[i] = i % 3;
        const double c3[9] = {9, 1.5, 1.5, 1.5, 9, 1.5, 1.5, 1.5, 9};
        const uint32_t n3[3] = {234, 233, 233};
        void *bp = orc_bp_create(sg, 3, 0);
        void *rng = orc_rng_create(3);
        orc_bp_init_messages(bp, 0, nullptr, t3.data(), rng);
        orc_bp_set_params(bp, c3, n3, 1.0);
        double last = 0;
        (void)orc_bp_converge_sync(bp, 1e-9, 300, 1.0, &last);
        REQUIRE(last == last);
        orc_rng_free(rng);
        orc_bp_free(bp);
        orc_graph_free(sg);
    }
    orc_graph_free(og);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: host_sanitize <edge list of the shipped data set>\n"); return 2; }
    const int rc = run(argv[1]);
    if (rc == 0) std::printf("host_sanitize ok\n");
    return rc;
}
