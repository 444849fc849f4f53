/*
 * sbmbp.h — C ABI of the MI355X-native belief-propagation engine for the (degree-corrected)
 * stochastic block model. This is the drop-in boundary for the reference's BP path: every entry
 * point names the reference interface (file:line under junipertcy/sbm-bp `src/`) it replaces.
 *
 * The reference has no FFI layer; its de-facto source-level boundary is the public interface of
 * `class belief_propagation` as `main.cpp:318-365` drives it (belief_propagation.h:96-142), plus
 * `blockmodel.h:105-107` (parameter constructors) and `graph_utilities.h:14-22` (graph input).
 * A maintainer binds these functions from `main.cpp` as shown in INTEGRATION.md.
 *
 * Conventions: plain C types only; 0 = SBMBP_OK, negative = error (sbmbp_strerror);
 * the caller owns every host array it passes (the library copies); the library owns all device
 * memory; one engine handle = one GPU = one host thread; no exceptions cross the boundary.
 * All floating-point state is IEEE double, as in the reference (types.h:38-41).
 *
 * Data layout (HBM): CSR rows = vertices, neighbours ascending (== std::set order of
 * types.h:14-15); rev[k] = index of the reverse directed edge; messages are OUT-ordered AoS:
 * msg[k*Q+q] = message from row(k) to nbr[k]. The reference's in-ordered mmap_[i][l][q]
 * (belief_propagation.h:65-66) is msg[rev[row_ptr[i]+l]*Q+q]. That is the layout AT THIS BOUNDARY
 * (sbmbp_set_state / sbmbp_get_state / sbmbp_host_init_state). On the device a message is kept as Q-1 of
 * its components (messages are distributions: every row of msg must sum to 1, as every message the
 * reference produces does): all but the largest, which is restored as max(0, 1 - sum of the others);
 * sbmbp_get_state therefore returns that ONE component per message within a few ulp of what
 * sbmbp_set_state was given, the others bit for bit.
 */
#ifndef SBMBP_H
#define SBMBP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBMBP_OK 0
#define SBMBP_ERR_ARG (-1)         /* invalid argument / shape mismatch */
#define SBMBP_ERR_HIP (-2)         /* a HIP runtime call failed (detail in sbmbp_last_error) */
#define SBMBP_ERR_NODEVICE (-3)    /* no usable GPU: the engine never falls back to the CPU */
#define SBMBP_ERR_STATE (-4)       /* call order violated (e.g. converge before set_params) */
#define SBMBP_ERR_IO (-5)          /* file could not be read */
#define SBMBP_ERR_UNSUPPORTED (-6) /* e.g. Q above SBMBP_MAX_Q */
#define SBMBP_ERR_NOMEM (-7)
#define SBMBP_ERR_COMM (-8)        /* a collective (RCCL or the caller's transport) failed */

#define SBMBP_MAX_Q 64 /* label counts 2 .. 16: lane-per-edge kernels (every mode, sharded too); 17 .. 64: matrix-core kernels
                          (single engine, deg_corr_flag 0 / 1, cab > 0; kernels_wide.h) */

typedef struct sbmbp_graph sbmbp_graph_t;   /* host-side CSR graph */
typedef struct sbmbp_engine sbmbp_engine_t; /* device engine (one GPU) */

const char *sbmbp_strerror(int code);
const char *sbmbp_last_error(void); /* thread-local detail string of the last failure */
const char *sbmbp_version(void);
int sbmbp_device_count(void); /* GPUs visible to this process (0 if none) */

/* ---------------------------------------------------------------------------------------------
 * Graph input. Replaces load_edge_list + edge_to_adj (graph_utilities.cpp:42-77: text "a b" per
 * line, symmetrised, de-duplicated, neighbours sorted, vertex count grown to the largest id) and
 * the graph_neis_/graph_neis_inv_ build of bp_allocate (belief_propagation.cpp:246-266).
 * Deviation (documented): a missing file is an error here, not an empty graph.
 * ------------------------------------------------------------------------------------------- */
int sbmbp_graph_load_edgelist(sbmbp_graph_t **out, const char *path, uint32_t n_vertices);
int sbmbp_graph_from_edges(sbmbp_graph_t **out, const uint32_t *pairs /* [2*n_pairs] */, uint64_t n_pairs,
                           uint32_t n_vertices);
/* adopt an existing CSR (rev may be NULL: it is then computed). Arrays are copied and validated. */
int sbmbp_graph_from_csr(sbmbp_graph_t **out, uint32_t n_vertices, uint64_t n_directed,
                         const uint64_t *row_ptr, const uint32_t *nbr, const uint32_t *rev);
uint32_t sbmbp_graph_num_vertices(const sbmbp_graph_t *g);
uint64_t sbmbp_graph_num_directed_edges(const sbmbp_graph_t *g); /* E2 = 2|E| */
uint32_t sbmbp_graph_max_degree(const sbmbp_graph_t *g);         /* blockmodel_t::get_graph_max_degree, blockmodel.cpp:51 */
int sbmbp_graph_copy_csr(const sbmbp_graph_t *g, uint64_t *row_ptr, uint32_t *nbr, uint32_t *rev);
void sbmbp_graph_destroy(sbmbp_graph_t *g);

/* ---------------------------------------------------------------------------------------------
 * Model parameters. Replace bp_param_from_epsilon_c (blockmodel.cpp:229-272) and
 * bp_param_from_direct (blockmodel.cpp:274-302) incl. the na[q]=unsigned(int(pa[q]*N)) truncation.
 * cab is Q*Q row-major; cab_upper is the upper triangle, row-major, as `--cab` passes it.
 * ------------------------------------------------------------------------------------------- */
int sbmbp_param_from_epsilon_c(uint32_t n_vertices, uint32_t Q, double epsilon, double c, double *cab, uint32_t *na);
int sbmbp_param_from_direct(uint32_t n_vertices, uint32_t Q, const double *pa, const double *cab_upper,
                            double *cab, uint32_t *na);

/* ---------------------------------------------------------------------------------------------
 * Engine life cycle. sbmbp_create replaces bp_allocate (belief_propagation.cpp:223-288): it
 * uploads the CSR and allocates messages/marginals in HBM. device < 0 selects the current device.
 * Q = 2 .. SBMBP_MAX_Q (the reference has no cap, main.cpp:271). Up to 16 labels a lane owns a directed edge; above, an
 * edge is spread over four lanes and W^T m runs on the matrix cores (csrc/kernels_wide.h): every entry point below works
 * there too (converge / sweep / free energy / entropy / EM expectations / overlap / inference / learning, damping, beta,
 * clamped rows, deg_corr_flag 0 and 1, the adaptive relaxation), except deg_corr_flag 2 and cab entries <= 0, which
 * return SBMBP_ERR_UNSUPPORTED; the shard entry points and the multi-GPU driver take Q <= 16.
 * ------------------------------------------------------------------------------------------- */
int sbmbp_create(sbmbp_engine_t **out, const sbmbp_graph_t *g, uint32_t Q, uint32_t deg_corr_flag, int device);
void sbmbp_destroy(sbmbp_engine_t *e);
/* run all work of this engine on an existing HIP stream (hipStream_t passed as void*; 0 = the default
 * stream). Engines start on a private non-blocking stream. */
int sbmbp_set_stream(sbmbp_engine_t *e, void *hip_stream);

/* init_messages (belief_propagation.cpp:101-217): same std::mt19937(seed) stream and fill order as
 * the reference for flag 0/1 (per vertex: psi, then its out-messages in ascending neighbour
 * order). conf: N entries, -1 = unknown, may be NULL for flag 0. true_conf: N entries.
 * conditional != 0 selects bp_conditional semantics (rows with conf != -1 are clamped,
 * belief_propagation.cpp:1100-1126); 0 selects bp_basic. Flags 2/3: see DESIGN.md (reference
 * asserts). */
int sbmbp_init_messages(sbmbp_engine_t *e, uint32_t flag, const int32_t *conf, const uint32_t *true_conf,
                        uint32_t seed, int conditional);
/* the same initial state on the host only (no device, no engine): psi N*Q and msg_out E2*Q doubles in the
 * layout of sbmbp_set_state; what sbmbp_init_messages uploads. */
int sbmbp_host_init_state(const sbmbp_graph_t *g, uint32_t Q, uint32_t flag, const int32_t *conf, uint32_t seed,
                          double *psi, double *msg_out);
/* device-side random initialisation (counter-based generator) for large synthetic runs where the
 * sequential mt19937 fill would dominate; not stream-compatible with the reference. */
int sbmbp_init_messages_device(sbmbp_engine_t *e, uint64_t seed, const uint32_t *true_conf);

/* expand_bp_params + set_beta (belief_propagation.cpp:290-317, 417-419) */
int sbmbp_set_params(sbmbp_engine_t *e, const double *cab, const uint32_t *na, double beta);
int sbmbp_get_params(sbmbp_engine_t *e, double *cab, uint32_t *na);

/* direct state access (the reference keeps real_psi_/mmap_ protected, belief_propagation.h:45,66) */
int sbmbp_set_state(sbmbp_engine_t *e, const double *psi /* N*Q or NULL */, const double *msg_out /* E2*Q or NULL */);
int sbmbp_get_state(sbmbp_engine_t *e, double *psi /* or NULL */, double *msg_out /* or NULL */);
int sbmbp_get_field(sbmbp_engine_t *e, double *h /* Q */); /* h_ of belief_propagation.cpp:320-360 */

/* Schedule of the synchronous sweep (no reference counterpart: the reference is random-sequential,
 * belief_propagation.cpp:394-401). field_mix in (0,1]: relaxation of the global field between
 * sweeps (1 = plain Jacobi). check_every >= 1: sweeps enqueued between two host reads of the
 * convergence flag (the returned niter is exact regardless). */
int sbmbp_set_schedule(sbmbp_engine_t *e, double field_mix, uint32_t check_every);

/* Adaptive relaxation of converge / inference / learning (on by default; no reference counterpart). The reference sweeps
 * random-sequentially and keeps h_ current inside a sweep (belief_propagation.cpp:394-401, 1088-1095); synchronous sweeps can
 * oscillate where it converges. With this on, the device-side convergence logic watches for (F) a period-2 swing of the
 * field sums, (P) a period-2 swing of the messages, (W) a window of sweeps without progress, and lowers field_mix
 * (1 -> 0.25 -> 0.1 -> 0.05) or moves down a (field_mix cap, damping factor) ladder ((0.5,1), (0.25,1), (0.5,0.5),
 * (0.25,0.5), (0.1,0.5), (0.25,0.25), (0.1,0.25)); the fixed points do not move, and runs that never oscillate are untouched. Fixed sweep
 * counts (sbmbp_sweep) are never relaxed. sbmbp_get_relaxation reports where the last converge call ended
 * (field level 0 and generic level -1: it never relaxed). */
int sbmbp_set_auto_relax(sbmbp_engine_t *e, int on);
int sbmbp_get_relaxation(const sbmbp_engine_t *e, int *field_level, int *generic_level, double *field_mix, double *damping_factor);

/* Which form of the sweep kernel runs. 0 = automatic: incoming messages are reconstructed from the
 * neighbours' marginals (same iterates, cache-friendly gather) whenever that is exact — damping 1,
 * every cab entry > 0, no clamped rows, deg_corr_flag != 2 — with the reference's 1-step message
 * criterion verified on the message buffers before convergence is declared; otherwise, and always
 * with mode 1, incoming messages are gathered from the message array. */
int sbmbp_set_gather_mode(sbmbp_engine_t *e, int mode);

/* converge (belief_propagation.cpp:386-415): returns in *niter the 0-based index of the first sweep
 * whose max |delta message| < crit, or -1 after max_sweeps. damping == dumping_rate. */
int sbmbp_converge(sbmbp_engine_t *e, double crit, uint32_t max_sweeps, double damping, int *niter,
                   double *last_maxdiff);
/* exactly n_sweeps synchronous sweeps, no convergence test (benchmarking / stepping) */
int sbmbp_sweep(sbmbp_engine_t *e, double damping, uint32_t n_sweeps, double *last_maxdiff);

/* compute_free_energy (belief_propagation.cpp:744-750) = -f_site + f_edge + f_nonedge;
 * parts = {f_site, f_edge, f_nonedge} (:442-504, :562-612, :675-709). The non-edge term is exact
 * (tiled N^2 kernel) for N <= the exact limit, else the moment series of DESIGN.md. */
int sbmbp_free_energy(sbmbp_engine_t *e, double *f, double *parts /* 3 or NULL */);
/* compute_entropy (belief_propagation.cpp:752-758); NaN for deg_corr_flag != 0 as the reference */
int sbmbp_entropy(sbmbp_engine_t *e, double *entropy, double *parts /* 3 or NULL */);
/* nonedge_mode: 0 = automatic, 1 = force exact N^2, 2 = force series; series_order 0 = automatic */
int sbmbp_set_nonedge_mode(sbmbp_engine_t *e, int nonedge_mode, int series_order);

/* compute_na_expect + compute_cab_expect (belief_propagation.cpp:428-440, 892-989) */
int sbmbp_em_expectations(sbmbp_engine_t *e, double *na_expect, double *nna_expect, double *cab_expect);
/* confusion matrix C[a*Q+b] = sum_{i: true_i = a} psi_i[b] and compute_overlap (:775-811) */
int sbmbp_confusion(sbmbp_engine_t *e, double *C);
int sbmbp_overlap(sbmbp_engine_t *e, double *overlap);

/* inference (belief_propagation.cpp:77-99): converge, free energy, entropy, overlap */
typedef struct sbmbp_infer_result {
    double entropy, free_energy, overlap;
    int niter;
    double last_maxdiff;
} sbmbp_infer_result;
int sbmbp_inference(sbmbp_engine_t *e, float conv_crit, uint32_t time_conv, float dumping_rate,
                    sbmbp_infer_result *out);

/* learning + learning_step (belief_propagation.cpp:14-75): EM loop with the reference's stopping
 * rule (criterion shrinks by 0.1 whenever fdiff falls below it; -e is not used).
 * On return eta/cab hold what the reference prints (:48-49). status: 0 = ran out of steps,
 * 1 = "fdiff < learning_conv_crit", 2 = free energy became NaN/Inf. */
typedef struct sbmbp_learn_result {
    int em_steps;
    int status;
    double free_energy;
    double overlap;
    uint64_t total_sweeps;
} sbmbp_learn_result;
int sbmbp_learning(sbmbp_engine_t *e, float learning_conv_crit, uint32_t learning_max_time, float learning_rate,
                   float dumping_rate, sbmbp_learn_result *out);
/* Two schedule rules of the EM loop (no reference counterpart). field_mix: relaxation of the global
 * field inside the BP runs of the EM loop (default 0.3; the smaller of this and sbmbp_set_schedule's value is used; 1 =
 * plain Jacobi). snap: the integer truncation of the group sizes (belief_propagation.cpp:58-63) treats a value within
 * min(snap * N * learning_conv_crit, 0.01) BELOW an integer as that integer (default snap = 1; 0 = truncate exactly). */
int sbmbp_set_learning_schedule(sbmbp_engine_t *e, double field_mix, double snap);

/* counters for the metric "BP edge-message updates per second" */
typedef struct sbmbp_stats {
    uint64_t sweeps;             /* synchronous sweeps executed so far */
    uint64_t edge_msg_updates;   /* sweeps * E2 */
    double sweep_kernel_ms;      /* HIP-event time of the sweep kernels (sum) on the engine's stream */
    uint64_t sweep_launches;     /* number of sweep-kernel launches timed */
    double bytes_per_sweep;      /* algorithmic bytes of one sweep */
    uint64_t device_bytes;       /* HBM held by this engine */
    uint32_t n_blocks;           /* workgroups of one sweep launch */
    uint32_t n_hub_rows;         /* rows above one segment's edge capacity: updated in fragments of 256 edges by launches of their own, not by the frame kernel */
    uint64_t psi_form_sweeps;    /* of `sweeps`, how many ran the marginal-gather form */
    uint64_t hub_edges;          /* directed edges of those rows (sweep_kernel_ms times the frame kernel: it moves the other edges) */
} sbmbp_stats;
int sbmbp_get_stats(sbmbp_engine_t *e, sbmbp_stats *out);
int sbmbp_reset_stats(sbmbp_engine_t *e);
/* when on, every sweep launch is bracketed by HIP events on the engine's stream (bench.py) */
int sbmbp_set_timing(sbmbp_engine_t *e, int on);

/* ---------------------------------------------------------------------------------------------
 * Vertex-range sharding: the per-shard STEPS (one engine per GPU). No reference counterpart: the
 * reference is single-process. A shard owns a contiguous range of rows, their out-messages and
 * marginals; the marginals of remote neighbours ("halo") live behind the owned rows in the same
 * table, so nbr_local indexes one array, and the message records received for the cut edges sit
 * behind the own records of each message buffer, so rev_local does too. The caller of these steps is
 * the C++ multi-GPU driver further down (sbmbp_dist_*, csrc/dist.hip), which moves the data between
 * the shards itself over its RCCL communicators (sbmbp_comm_*): boundary marginals (marginal-gather
 * sweep) or cut-edge messages (message-gather sweep) once per sweep, the Q+1 reduction values by one
 * all-gather. Every sweep form of the single engine runs sharded: damping, clamped rows,
 * deg_corr_flag 2, zeros in cab take the message-gather sweep. The device buffers named in the
 * descriptor are owned by the caller (the driver) so that its collectives can address them.
 * ------------------------------------------------------------------------------------------- */
typedef struct sbmbp_shard_desc {
    uint32_t n_global;          /* vertices of the whole graph (field scaling, eta) */
    uint32_t n_own;             /* rows owned by this shard */
    uint32_t n_halo;            /* remote vertices whose marginals this shard reads */
    uint32_t row0;              /* global id of the first owned row */
    uint64_t n_edges;           /* directed edges leaving the owned rows */
    uint64_t edge0;             /* global index of the first owned directed edge (device RNG stream) */
    const uint64_t *row_ptr;    /* host [n_own+1], local offsets */
    const uint32_t *nbr_local;  /* host [n_edges]: index into the marginal table: own < n_own <= halo */
    void *psi_buf0, *psi_buf1;  /* device, (n_own+n_halo)*Q doubles each */
    void *red_buf;              /* device, >= SBMBP_RED_GATHER_OFFSET + n_ranks * SBMBP_FOLD_ROWS * 17 doubles (and >= 8192):
                                   reduction hand-off buffer */
    uint32_t n_chunks;          /* row chunks for overlapping the halo exchange with the sweep (0 or 1 = none) */
    const uint32_t *chunk_row;  /* host [n_chunks+1] local row boundaries, chunk_row[0] = 0, last = n_own */
    /* optional: what the message-gather sweep needs on a shard (any damping, clamped rows, deg_corr_flag 2, zeros in cab) */
    const uint32_t *rev_local;  /* host [n_edges] or NULL: record holding the reverse message of every edge: < n_edges for an own
                                   neighbour, n_edges + r for the r-th record received from the peers */
    uint64_t n_halo_msgs;       /* records received from the peers, kept behind the own records of each message buffer */
    const uint32_t *table_deg;  /* host [n_own + n_halo] or NULL: degree of every vertex of the marginal table (deg_corr_flag 2) */
} sbmbp_shard_desc;

typedef struct sbmbp_conv_state {
    double maxdiff; /* of the last executed sweep: the 1-step message difference if last_exact, else the 2-step hint */
    int conv_iter;  /* first sweep whose 1-step difference fell below the criterion, or -1 */
    int sweep_idx;  /* sweeps executed since sbmbp_shard_begin */
    int stop;       /* queued sweeps after the trigger were skipped */
    int last_exact;
    int pause;      /* with stop: adaptive relaxation asked for damping; answer with sbmbp_shard_resume and go on in the
                       message-gather form from sweep_idx */
    int ar_field_level, ar_generic_level; /* levels of the adaptive relaxation (sbmbp_get_relaxation) */
} sbmbp_conv_state;

int sbmbp_shard_create(sbmbp_engine_t **out, const sbmbp_shard_desc *desc, uint32_t Q, uint32_t deg_corr_flag, int device);
/* start a run of sweeps: uploads parameters and the convergence criterion (< 0: never converges). The sweeps report
 * 2-step hints until one falls below 8 * crit; from then on they report the reference's 1-step difference and the
 * device-side stop flag works on it (all on the device: no host round trip) */
int sbmbp_shard_begin(sbmbp_engine_t *e, double crit, int hinted /* 1: the run uses the marginal-gather sweep */);
/* planted configuration and true labels of the owned rows (init_messages, belief_propagation.cpp:106-108, 132-216);
 * any_clamp_global: some vertex of the WHOLE graph is clamped (every shard must take the same kernel variants) */
int sbmbp_shard_set_labels(sbmbp_engine_t *e, const int32_t *conf_local, const uint32_t *true_conf_local, uint32_t flag,
                           int conditional, int any_clamp_global);
/* 0: every cab entry > 0; 1: clamped rows exist; 2: they hold the one-hot state of init flag 1/3; 3: the state is the device
 * initialisation (message = sender's marginal); 4: the shard has a reverse index */
int sbmbp_shard_query(sbmbp_engine_t *e, int what);
/* message-gather form of sweep j over all owned rows (k_sweep: belief_propagation.cpp:991-1071 as the single engine runs
 * it). The incoming messages of the cut edges are the records behind the own ones of the buffer sweep j reads:
 * sbmbp_shard_msg_halo(e, j) is where the caller receives them (n_halo_msgs records of Q-1 doubles, in rev_local's
 * numbering) and sbmbp_shard_pack_msgs gathers the records it has to send from the same buffer. */
int sbmbp_shard_sweep_explicit(sbmbp_engine_t *e, uint32_t j, double damping);
void *sbmbp_shard_msg_halo(sbmbp_engine_t *e, uint32_t j);
int sbmbp_shard_pack_msgs(sbmbp_engine_t *e, uint32_t j, const uint32_t *d_edge_idx, uint32_t n, double *d_out);
/* where the reductions below take the incoming messages from: 0 = materialised from the marginal table (needs marginals
 * consistent with the messages and cab > 0), 1 = gathered through rev_local (the caller received the halo records of the
 * current buffer, sbmbp_shard_msg_halo(e, 0)) */
int sbmbp_shard_set_incoming(sbmbp_engine_t *e, int source);
/* gather rows idx[0..n) of the marginal table that sweep j READS into out (device pointers), ncomp
 * components per row: Q, or Q-1 to ship the marginals without their last component (they sum to 1) */
int sbmbp_shard_pack(sbmbp_engine_t *e, uint32_t j, const uint32_t *d_idx, uint32_t n, double *d_out, uint32_t ncomp);
/* expand n received (staged) rows of ncomp components into the halo rows d_halo_row[0..n) (each < n_halo) of the
 * table sweep j reads */
int sbmbp_shard_unpack(sbmbp_engine_t *e, uint32_t j, const double *d_in, const uint32_t *d_halo_row, uint32_t n, uint32_t ncomp);
/* Fused exchange buffers (optional; without this call every sweep needs sbmbp_shard_pack before and sbmbp_shard_unpack
 * after the exchange). snd_ptr[n_own+1] / snd_slot[snd_ptr[n_own]] (host, copied): the rows of d_sendbuf (ncomp
 * components each) that ship the marginal of every own row. d_stage0/1: receive buffers of halo table 0/1, n_halo rows of
 * ncomp components in halo order. With them sbmbp_shard_sweep_chunk writes each new marginal of the chunk into its send
 * slots and gathers halo marginals of the table it reads from the matching receive buffer; the halo ROWS of the marginal
 * tables are then only refreshed by sbmbp_shard_unpack (needed before the reductions, not between sweeps). */
int sbmbp_shard_set_io(sbmbp_engine_t *e, const uint32_t *snd_ptr, const uint32_t *snd_slot, double *d_sendbuf,
                       const double *d_stage0, const double *d_stage1, uint32_t ncomp);
/* which of the two marginal buffers sweep j reads (0/1); the other one is written */
int sbmbp_shard_read_buffer(sbmbp_engine_t *e, uint32_t j);
/* red[0..Q) = sum over owned rows of g_i psi_i of the buffer sweep j reads (field initialisation) */
int sbmbp_shard_field_partial(sbmbp_engine_t *e, uint32_t j);
/* sweep j over the owned rows; red[0..Q) = partial sums of the new marginals, red[Q] = hint */
int sbmbp_shard_sweep_partial(sbmbp_engine_t *e, uint32_t j);
/* the same in pieces: sweep j over row chunk c only (the caller ships chunk c's new marginals while chunk c+1 runs), then
 * ONE fold launch of all chunks' partials into SBMBP_FOLD_ROWS rows at red[0..): the caller all-gathers those rows of every
 * rank behind red + SBMBP_RED_GATHER_OFFSET and sbmbp_shard_finalize folds them (n_rows = n_ranks * SBMBP_FOLD_ROWS) */
int sbmbp_shard_sweep_chunk(sbmbp_engine_t *e, uint32_t j, uint32_t c);
/* the same on another HIP stream of the engine's device (the caller orders it against the engine's stream with events):
 * consecutive chunks on alternating streams overlap each other's tail */
int sbmbp_shard_sweep_chunk_on(sbmbp_engine_t *e, uint32_t j, uint32_t c, void *hip_stream);
int sbmbp_shard_sweep_fold(sbmbp_engine_t *e);
#define SBMBP_FOLD_ROWS 64          /* rows of (Q+1) doubles sbmbp_shard_sweep_fold leaves at red[0..) */
#define SBMBP_RED_GATHER_OFFSET 2048 /* > SBMBP_FOLD_ROWS * 17 (sharded engines: Q <= 16): gathered rows never overlap a shard's own rows */
/* consume the reduction values: n_rows rows of (Q+1) doubles starting at red + SBMBP_RED_GATHER_OFFSET (the caller
 * all-gathers every shard's red[0..Q] there; n_rows = number of shards). Rows are folded in order —
 * sums for the Q field entries, max for the hint. mode 0 after a sweep, 1 field initialisation */
int sbmbp_shard_finalize(sbmbp_engine_t *e, int mode, uint32_t n_rows, int md_exact /* the sweep was a message-gather sweep */);
/* red[0] = max |m_a - m_b| over the two message buffers of this shard (exact 1-step criterion) */
int sbmbp_shard_msgdiff_partial(sbmbp_engine_t *e);
/* red[0..2Q+Q*Q) = na_expect, nna_expect, confusion sums over the owned rows (current marginals) */
int sbmbp_shard_rowsums_partial(sbmbp_engine_t *e);
/* Reductions at compute_* time (belief_propagation.cpp:744-758, 428-440, 892-989) on shards: each
 * *_partial leaves this shard's sums in red, the caller all-reduces (SUM) the stated number of doubles,
 * *_finish turns them into the reference's quantities on the host. Incoming messages are first
 * materialised from the marginal table (whose halo must be current) and the previous own messages.
 * The non-edge term uses the moment series (the exact all-pairs kernel would need every marginal). */
int sbmbp_shard_fe_partial(sbmbp_engine_t *e, int want_entropy);                 /* all-reduce red[0..5) */
int sbmbp_shard_fe_finish(sbmbp_engine_t *e, double *out /* f_site, f_edge, e_site, e_edge */);
int sbmbp_shard_nonedge_partial(sbmbp_engine_t *e, int want_entropy, uint32_t *n_values, int *order);
int sbmbp_shard_nonedge_finish(sbmbp_engine_t *e, int want_entropy, int order, double *out /* f_nonedge, e_nonedge */);
/* exact non-edge term for small graphs: d_psi_all = marginals of ALL vertices, global row order (device); leaves 4 doubles
 * in red (all-pairs f/e over (own i, every l), adjacent f/e): all-reduce, then f = (red[0]-red[2])/2N, e = (red[1]-red[3])/2N */
int sbmbp_shard_nonedge_exact_partial(sbmbp_engine_t *e, const double *d_psi_all, int want_entropy);
int sbmbp_shard_em_partial(sbmbp_engine_t *e, uint32_t *n_values);
int sbmbp_shard_em_finish(sbmbp_engine_t *e, double *na_expect, double *nna_expect, double *cab_expect);

/* wait for the stream and read the convergence state */
int sbmbp_shard_poll(sbmbp_engine_t *e, sbmbp_conv_state *out);
/* the same without idling the GPU: record queues a copy of the state into page-locked slot 0/1 behind the work queued so far,
 * wait blocks until that copy has landed (the caller may queue the next batch in between) */
int sbmbp_shard_state_record(sbmbp_engine_t *e, int slot);
int sbmbp_shard_state_wait(sbmbp_engine_t *e, int slot, sbmbp_conv_state *out);
int sbmbp_shard_resume(sbmbp_engine_t *e);
/* after a poll: `executed` sweeps of the queued batch really ran; flips the buffer parities */
int sbmbp_shard_commit(sbmbp_engine_t *e, uint32_t executed);
/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: communicators and the per-rank driver (csrc/dist.hip). No reference counterpart; this is what replaces the
 * call sequence of main.cpp:318-365 when the graph is sharded over the GPUs of one node. One sbmbp_dist_t = one rank = one
 * GPU = one host thread. The ranks of a run are processes (one per GPU; they share a communicator id) or threads of one
 * process (bin/bp --gpus N). All ranks hold the whole host graph and make the same calls in the same order; every
 * convergence decision is taken on the device from identical all-gathered values, so no rank ever needs to be told.
 * Per sweep: the boundary marginals travel by grouped ncclSend/ncclRecv over xGMI while the next row chunk is swept, then
 * ONE all-gather of Q+1 doubles per rank (field sums + message difference). The message-gather sweep (damping, clamped
 * rows, deg_corr_flag 2, zeros in cab, first sweep after a state or parameter change) ships the messages of the cut edges
 * instead. Results are partition invariant up to the summation order of the Q field sums.
 * ------------------------------------------------------------------------------------------- */
typedef struct sbmbp_comm sbmbp_comm_t;
typedef struct sbmbp_dist sbmbp_dist_t;
#define SBMBP_COMM_ID_BYTES 256
/* RCCL: rank 0 draws an id (two ncclUniqueId: one communicator for the halo exchange, one for the reductions) and hands it
 * to the other ranks by whatever means the launcher has (torch.distributed in bench.py, shared memory between threads) */
int sbmbp_comm_unique_id(void *id_out /* SBMBP_COMM_ID_BYTES */);
int sbmbp_comm_init_rank(sbmbp_comm_t **out, const void *id, int n_ranks, int rank, int device);
/* the ranks are n threads of this process (possibly sharing one GPU, where RCCL refuses duplicate devices): fills out[0..n),
 * one handle per rank thread. Tests and rehearsals. */
int sbmbp_comm_init_local(sbmbp_comm_t **out, int n_ranks);
/* the ranks are processes with the caller's own transport between them (e.g. gloo): buffers are staged through host memory.
 * Rows for / from peer p are contiguous, in rank order; every function returns 0 on success. Rehearsals on a 1-GPU box. */
typedef struct sbmbp_comm_callbacks {
    void *user;
    int (*exchange)(void *user, const double *send, const uint64_t *send_rows, double *recv, const uint64_t *recv_rows, int row_doubles);
    int (*allgather)(void *user, const double *in, uint64_t n, double *out /* n_ranks * n */);
    int (*allreduce)(void *user, double *buf, uint64_t n, int op /* 0 sum, 1 max */);
} sbmbp_comm_callbacks;
int sbmbp_comm_init_callbacks(sbmbp_comm_t **out, int n_ranks, int rank, const sbmbp_comm_callbacks *cb);
/* measurement only: rank `rank` of an n-rank plan alone on the GPU; exchanges deliver nothing, every peer "reports" this
 * rank's reduction values (tools/shard_budget.py: what one rank of the 8-GPU run executes per sweep) */
int sbmbp_comm_init_null(sbmbp_comm_t **out, int n_ranks, int rank);
void sbmbp_comm_destroy(sbmbp_comm_t *c);
/* a rank that gives up (an error outside a collective) tells its peers, so that none of them blocks waiting for it: the
 * in-process transport fails every later collective of the group (it also times out after SBMBP_LOCAL_TIMEOUT_S, default
 * 600 s), RCCL communicators are aborted (ncclCommAbort) */
void sbmbp_comm_abort(sbmbp_comm_t *c);
int sbmbp_comm_rank(const sbmbp_comm_t *c);
int sbmbp_comm_size(const sbmbp_comm_t *c);
const char *sbmbp_comm_transport(const sbmbp_comm_t *c); /* "rccl" | "local" | "callbacks" */

typedef struct sbmbp_dist_info_t {
    int rank, world;
    uint32_t n_global, row0, n_own, n_halo, n_chunks, halo_components;
    uint64_t n_edges, e2_global, n_halo_msgs;
    uint64_t sent_rows_per_sweep, busiest_peer_rows; /* marginal rows shipped per marginal-gather sweep */
} sbmbp_dist_info_t;

/* the shard of rank sbmbp_comm_rank(comm): plan (row range balanced on sum(deg+2), halo, send lists, cut edges), device
 * buffers, shard engine. g is borrowed and must outlive the engine (as the adjacency must in the reference,
 * belief_propagation.h:27). n_chunks = 0: default (4, or SBMBP_SHARD_CHUNKS; 1 for a single rank). */
int sbmbp_dist_create(sbmbp_dist_t **out, sbmbp_comm_t *comm, const sbmbp_graph_t *g, uint32_t Q, uint32_t deg_corr_flag, int device,
                      uint32_t n_chunks);
void sbmbp_dist_destroy(sbmbp_dist_t *d);
int sbmbp_dist_info(const sbmbp_dist_t *d, sbmbp_dist_info_t *out);
int sbmbp_dist_peer_rows(const sbmbp_dist_t *d, uint64_t *send_rows /* [world] or NULL */, uint64_t *recv_rows);
/* the calls below mirror the single-engine ones above (same reference citations); conf / true_conf are the GLOBAL vectors */
int sbmbp_dist_init_messages(sbmbp_dist_t *d, uint32_t flag, const int32_t *conf, const uint32_t *true_conf, uint32_t seed,
                             int conditional);
int sbmbp_dist_init_messages_device(sbmbp_dist_t *d, uint64_t seed, const uint32_t *true_conf);
int sbmbp_dist_set_state(sbmbp_dist_t *d, const double *psi_own /* n_own*Q */, const double *msg_own /* n_edges*Q */);
int sbmbp_dist_get_state(sbmbp_dist_t *d, double *psi_own, double *msg_own);
int sbmbp_dist_gather_marginals(sbmbp_dist_t *d, double *psi_all /* n_global*Q, filled on every rank */);
int sbmbp_dist_set_params(sbmbp_dist_t *d, const double *cab, const uint32_t *na, double beta);
int sbmbp_dist_get_params(sbmbp_dist_t *d, double *cab, uint32_t *na);
int sbmbp_dist_set_schedule(sbmbp_dist_t *d, double field_mix, uint32_t check_every);
int sbmbp_dist_set_learning_schedule(sbmbp_dist_t *d, double field_mix, double snap);
int sbmbp_dist_set_gather_mode(sbmbp_dist_t *d, int mode);
int sbmbp_dist_set_auto_relax(sbmbp_dist_t *d, int on);
int sbmbp_dist_get_relaxation(const sbmbp_dist_t *d, int *field_level, int *generic_level);
int sbmbp_dist_converge(sbmbp_dist_t *d, double crit, uint32_t max_sweeps, double damping, int *niter, double *last_maxdiff);
int sbmbp_dist_sweep(sbmbp_dist_t *d, double damping, uint32_t n_sweeps, double *last_maxdiff);
int sbmbp_dist_free_energy(sbmbp_dist_t *d, double *f, double *parts);
int sbmbp_dist_entropy(sbmbp_dist_t *d, double *entropy, double *parts);
int sbmbp_dist_em_expectations(sbmbp_dist_t *d, double *na_expect, double *nna_expect, double *cab_expect);
int sbmbp_dist_confusion(sbmbp_dist_t *d, double *C);
int sbmbp_dist_overlap(sbmbp_dist_t *d, double *overlap);
int sbmbp_dist_inference(sbmbp_dist_t *d, float conv_crit, uint32_t time_conv, float dumping_rate, sbmbp_infer_result *out);
int sbmbp_dist_learning(sbmbp_dist_t *d, float learning_conv_crit, uint32_t learning_max_time, float learning_rate, float dumping_rate,
                        sbmbp_learn_result *out);
int sbmbp_dist_get_stats(sbmbp_dist_t *d, sbmbp_stats *out); /* kernel times and bytes are this rank's; sweeps are the run's */
int sbmbp_dist_reset_stats(sbmbp_dist_t *d);
int sbmbp_dist_set_timing(sbmbp_dist_t *d, int on);
/* while timing is on: mean ms per marginal-gather sweep on this rank's compute stream spent in {chunk kernels (with whatever
 * exchange time they could not hide), fold + all-gather + finalize, waiting for exchanges still in flight} */
int sbmbp_dist_phase_times(sbmbp_dist_t *d, double ms_per_sweep[3], uint64_t *n_sweeps);
/* the plan without a device (tests, dry runs) */
int sbmbp_plan_summary(const sbmbp_graph_t *g, int world, int rank, uint32_t n_chunks, sbmbp_dist_info_t *info, uint64_t *send_rows,
                       uint64_t *recv_rows, uint64_t *msg_rows);
int sbmbp_plan_arrays(const sbmbp_graph_t *g, int world, int rank, uint32_t n_chunks, uint32_t *nbr_local, uint32_t *halo_global,
                      uint32_t *chunk_row, uint64_t *send_counts_cp, uint64_t *recv_counts_cp, uint32_t *send_idx_chunked,
                      uint32_t *snd_ptr, uint32_t *snd_slot, uint32_t *rev_local, uint32_t *msg_send_edge);

#ifdef __cplusplus
}
#endif
#endif /* SBMBP_H */
