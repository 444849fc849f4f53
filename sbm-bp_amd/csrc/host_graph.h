// Host-side graph + parameter utilities of the engine (C++14, no device code).
#ifndef SBMBP_HOST_GRAPH_H
#define SBMBP_HOST_GRAPH_H

#include <cstdint>
#include <string>
#include <functional>
#include <vector>

struct sbmbp_graph {
    uint32_t n = 0;                 // vertices
    std::vector<uint64_t> row_ptr;  // n+1
    std::vector<uint32_t> nbr;      // E2, ascending per row
    std::vector<uint32_t> rev;      // E2, index of the reverse directed edge
    uint32_t max_degree = 0;
    uint64_t e2() const { return nbr.size(); }
    uint32_t deg(uint32_t i) const { return uint32_t(row_ptr[i + 1] - row_ptr[i]); }
};

namespace sbmbp {

void set_error(const std::string &msg);
const std::string &get_error();
int arg_error(const char *func, int line);  // SBMBP_ERR_ARG with a message that names the check (never a stale one)

// returns 0 or an SBMBP_ERR_* code
int graph_from_pairs(sbmbp_graph &g, const uint32_t *pairs, uint64_t n_pairs, uint32_t n_vertices);
int graph_from_csr(sbmbp_graph &g, uint32_t n, uint64_t e2, const uint64_t *row_ptr, const uint32_t *nbr,
                   const uint32_t *rev);
int read_edgelist(const char *path, std::vector<uint32_t> &pairs);
int read_int_column(const char *path, std::vector<int64_t> &values);  // load_beliefs/load_confs format

void param_from_epsilon_c(uint32_t N, uint32_t Q, double epsilon, double c, double *cab, uint32_t *na);
void param_from_direct(uint32_t N, uint32_t Q, const double *pa, const double *cab_upper, double *cab, uint32_t *na);

// init_messages (belief_propagation.cpp:101-217) on the out-ordered layout; fills psi (N*Q) and
// msg (E2*Q), both caller-allocated (need not be initialised), from std::mt19937(seed) in the reference's
// draw order.
// With a sink, psi/msg may be null: the rows are produced in slabs of consecutive vertices [lo, hi) and each slab is
// handed over when complete (psi_rows: (hi-lo)*Q doubles, msg_rows: (row_ptr[hi]-row_ptr[lo])*Q doubles, valid during the
// call) while the generator already works on the next one.
struct state_sink {
    // optional: provide the two slab buffers (e.g. page-locked memory); default new[]
    std::function<bool(uint64_t psi_doubles, uint64_t msg_doubles, double **psi_buf, double **msg_buf)> alloc;
    std::function<void(uint32_t lo, uint32_t hi, const double *psi_rows, const double *msg_rows)> put;
};
void init_state_host(uint32_t n, const uint32_t *row_ptr, uint64_t e2, uint32_t Q, uint32_t flag, const int32_t *conf,
                     uint32_t seed, double *psi, double *msg, const state_sink *sink = nullptr);

}  // namespace sbmbp
#endif
