// Multi-GPU belief propagation behind the C ABI (include/sbmbp.h, section "Multi-GPU"): vertex-range shard plan,
// communicators and the per-rank driver. Host code only (C++14 + HIP runtime + RCCL); every device step is one of the
// sbmbp_shard_* entry points of engine.hip. One sbmbp_dist_t = one rank = one GPU = one host thread; the ranks of a run are
// either processes (bench.py, one per GPU, communicator from an id they share) or threads of one process (bin/bp --gpus N).
//
// No reference counterpart: junipertcy/sbm-bp is single-process. What is sharded is the path of
// belief_propagation.cpp:386-415 (converge) and the reductions :428-440, :744-758, :892-989.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/sbmbp.h"
#include "host_graph.h"

using sbmbp::set_error;
using sbmbp::arg_error;
typedef uint32_t u32;
typedef uint64_t u64;

#define HIPCHK(call)                                                              \
    do {                                                                          \
        hipError_t _e = (call);                                                   \
        if (_e != hipSuccess) {                                                   \
            set_error(std::string(#call) + ": " + hipGetErrorString(_e));         \
            return SBMBP_ERR_HIP;                                                 \
        }                                                                         \
    } while (0)
#define NCCLCHK(call)                                                             \
    do {                                                                          \
        ncclResult_t _r = (call);                                                 \
        if (_r != ncclSuccess) {                                                  \
            set_error(std::string(#call) + ": " + ncclGetErrorString(_r));        \
            return SBMBP_ERR_COMM;                                                \
        }                                                                         \
    } while (0)
#define CHK(call)                          \
    do {                                   \
        int _r = (call);                   \
        if (_r != SBMBP_OK) return _r;     \
    } while (0)

// =====================================================================================================
// Shard plan. Each shard owns a contiguous row range chosen so that sum(deg + 2) is balanced; its marginal table is
// [owned rows | halo vertices]. Because the graph is symmetric, "vertices of mine that peer p reads" equals "my vertices
// with a neighbour in p's range", so every rank derives its send lists from its own rows alone and they line up with the
// receivers' halo order without negotiation. The owned rows are cut into chunks so that the new marginals of chunk c can
// travel while chunk c+1 is swept; the halo is numbered in RECEIVE order (chunk, peer, global id): what arrives for one
// chunk is one contiguous slice and received row r IS halo row r (the sweep kernel gathers halo marginals straight from the
// receive buffer). For the message-gather sweep the plan also holds the cut EDGES: rev_local points own-own edges at their
// local reverse record and cut edges at the record received from the neighbour's owner (kept behind the own records).
// =====================================================================================================
namespace {

struct shard_plan {
    int rank = 0, world = 1;
    u32 n_global = 0, row0 = 0, n_own = 0, n_halo = 0, n_chunks = 1;
    u64 edge0 = 0, n_edges = 0, e2_global = 0;
    std::vector<u64> bounds;           // world + 1
    std::vector<u64> row_ptr;          // n_own + 1, local offsets
    std::vector<u32> nbr_local;        // n_edges: index into the marginal table
    std::vector<u32> halo_global;      // n_halo: global id of every halo row (receive order)
    std::vector<u32> chunk_row;        // n_chunks + 1 local row boundaries
    std::vector<u64> send_counts, recv_counts;          // [world] rows per peer and sweep
    std::vector<u64> send_counts_cp, recv_counts_cp;    // [n_chunks][world]
    std::vector<u64> send_off_c, stage_off_c;           // [n_chunks + 1] row offsets of a chunk's slice in sendbuf / recvbuf
    std::vector<u32> send_idx_chunked;                  // local row shipped by every send slot, ordered (chunk, peer, row)
    std::vector<u32> snd_ptr, snd_slot;                 // CSR: send slots of every own row
    // message-gather form
    std::vector<u32> rev_local;                         // n_edges
    std::vector<u32> msg_send_edge;                     // local edge of every record to send, ordered (peer, receiver's edge order)
    std::vector<u64> msg_counts;                        // [world] records per peer (sent == received, by symmetry)
    u64 n_halo_msgs = 0;
    std::vector<u32> table_deg;                         // n_own + n_halo (dc 2)
};

// c(i) = sum_{k <= i} (deg_k + 2) over rows [lo, lo + n) of the global CSR
inline u64 cum_weight(const u64 *rp, u64 lo, u64 i) { return (rp[lo + i + 1] - rp[lo]) + 2 * (i + 1); }
// first i in [0, n) with c(i) >= x, or n
u64 search_weight(const u64 *rp, u64 lo, u64 n, double x) {
    u64 a = 0, b = n;
    while (a < b) {
        const u64 m = (a + b) / 2;
        if (double(cum_weight(rp, lo, m)) >= x) b = m; else a = m + 1;
    }
    return a;
}

std::vector<u64> partition_rows(const u64 *rp, u64 n, int world) {
    std::vector<u64> bounds(1, 0);
    const double total = n ? double(cum_weight(rp, 0, n - 1)) : 0.0;
    for (int r = 1; r < world; ++r) {
        u64 b = search_weight(rp, 0, n, total * r / world) + 1;
        b = std::min(std::max(b, bounds.back() + 1), n - u64(world - r));  // every shard keeps at least one row
        bounds.push_back(b);
    }
    bounds.push_back(n);
    return bounds;
}

std::vector<u64> chunk_rows(const u64 *rp, u64 lo, u64 n, u32 n_chunks) {  // local boundaries of rows [lo, lo + n)
    std::vector<u64> cuts(1, 0);
    const double total = n ? double(cum_weight(rp, lo, n - 1)) : 0.0;
    for (u32 k = 1; k < n_chunks; ++k) {
        const u64 b = n ? search_weight(rp, lo, n, total * k / n_chunks) + 1 : 0;
        cuts.push_back(std::min(std::max(b, cuts.back()), n));
    }
    cuts.push_back(n);
    return cuts;
}

// SBMBP_HOST_TIMING=1: phases of the plan on stderr (rank 0 only; measurement aid)
struct plan_timer {
    bool on;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit plan_timer(int rank) : on(rank == 0 && std::getenv("SBMBP_HOST_TIMING") != nullptr) {}
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[sbmbp plan] %-26s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int build_plan(const sbmbp_graph &g, int world, int rank, u32 n_chunks, shard_plan &P) {
    const u64 N = g.n;
    plan_timer tm(rank);
    if (world < 1 || rank < 0 || rank >= world) { set_error("rank outside the communicator"); return SBMBP_ERR_ARG; }
    if (N < u64(world)) { set_error("fewer vertices than shards"); return SBMBP_ERR_ARG; }
    const u64 *rp = g.row_ptr.data();
    P.rank = rank;
    P.world = world;
    P.n_global = g.n;
    P.e2_global = g.e2();
    P.n_chunks = std::max<u32>(1, n_chunks);
    P.bounds = partition_rows(rp, N, world);
    const u64 lo = P.bounds[rank], hi = P.bounds[rank + 1];
    P.row0 = u32(lo);
    P.n_own = u32(hi - lo);
    P.edge0 = rp[lo];
    P.n_edges = rp[hi] - rp[lo];
    if (P.n_edges >= (u64(1) << 32)) { set_error("a shard holds 2^32 or more directed edges"); return SBMBP_ERR_UNSUPPORTED; }
    P.row_ptr.resize(size_t(P.n_own) + 1);
    for (u64 i = 0; i <= P.n_own; ++i) P.row_ptr[i] = rp[lo + i] - P.edge0;
    const u32 W = u32(world), Cn = P.n_chunks;
    auto owner_of = [&](u64 v) { return int(std::upper_bound(P.bounds.begin(), P.bounds.end(), v) - P.bounds.begin()) - 1; };
    // chunk boundaries of EVERY shard (global rows): the receive side cuts a peer's halo slice by the peer's chunks
    std::vector<std::vector<u64>> all_chunks(W);
    for (u32 p = 0; p < W; ++p) {
        all_chunks[p] = chunk_rows(rp, P.bounds[p], P.bounds[p + 1] - P.bounds[p], Cn);
        for (auto &x : all_chunks[p]) x += P.bounds[p];
    }
    P.chunk_row.resize(Cn + 1);
    for (u32 c = 0; c <= Cn; ++c) P.chunk_row[c] = u32(all_chunks[rank][c] - lo);
    // halo vertices: ascending global id == grouped by owner
    const u32 *nb = g.nbr.data() + P.edge0;
    std::vector<u32> hmap(N, 0xffffffffu);
    for (u64 k = 0; k < P.n_edges; ++k) { const u64 l = nb[k]; if (l < lo || l >= hi) hmap[l] = 0; }
    std::vector<u32> remote;
    for (u64 v = 0; v < N; ++v) if (hmap[v] == 0) { hmap[v] = u32(remote.size()); remote.push_back(u32(v)); }
    P.n_halo = u32(remote.size());
    tm.lap("halo vertices");
    P.recv_counts.assign(W, 0);
    std::vector<u64> halo_off(W + 1, 0);
    for (u32 x = 0; x < P.n_halo; ++x) P.recv_counts[owner_of(remote[x])]++;
    for (u32 p = 0; p < W; ++p) halo_off[p + 1] = halo_off[p] + P.recv_counts[p];
    P.recv_counts_cp.assign(size_t(Cn) * W, 0);
    std::vector<u64> pm_off(size_t(Cn) * W, 0);  // peer-major index of the first halo vertex of (chunk, peer)
    for (u32 p = 0; p < W; ++p) {
        const u32 *h = remote.data() + halo_off[p];
        const u64 nh = P.recv_counts[p];
        std::vector<u64> cut(Cn + 1);
        for (u32 c = 0; c <= Cn; ++c) cut[c] = u64(std::lower_bound(h, h + nh, all_chunks[p][c], [](u32 a, u64 b) { return u64(a) < b; }) - h);
        for (u32 c = 0; c < Cn; ++c) {
            P.recv_counts_cp[size_t(c) * W + p] = cut[c + 1] - cut[c];
            pm_off[size_t(c) * W + p] = halo_off[p] + cut[c];
        }
    }
    // receive order (chunk, peer, id)
    std::vector<u32> renamed(P.n_halo);
    P.halo_global.resize(P.n_halo);
    P.stage_off_c.assign(Cn + 1, 0);
    u32 r = 0;
    for (u32 c = 0; c < Cn; ++c) {
        for (u32 p = 0; p < W; ++p)
            for (u64 x = pm_off[size_t(c) * W + p], xe = x + P.recv_counts_cp[size_t(c) * W + p]; x < xe; ++x) {
                renamed[x] = r;
                P.halo_global[r] = remote[x];
                ++r;
            }
        P.stage_off_c[c + 1] = r;
    }
    P.nbr_local.resize(P.n_edges);
    for (u64 k = 0; k < P.n_edges; ++k) {
        const u64 l = nb[k];
        P.nbr_local[k] = (l >= lo && l < hi) ? u32(l - lo) : P.n_own + renamed[hmap[l]];
    }
    tm.lap("local neighbour ids");
    // send lists: my rows with a neighbour owned by p, ascending, per peer
    std::vector<std::vector<u32>> rows_p(W);
    std::vector<std::vector<u32>> cut_edges(W);  // my cut edges per peer, in my edge order (row, neighbour)
    for (u32 i = 0; i < P.n_own; ++i) {
        int last = -1, o = 0;
        for (u64 k = P.row_ptr[i]; k < P.row_ptr[i + 1]; ++k) {
            const u64 l = nb[k];
            if (l >= lo && l < hi) continue;
            while (l >= P.bounds[size_t(o) + 1]) ++o;  // neighbours ascend within a row, so owners do: no search per edge
            cut_edges[o].push_back(u32(k));
            if (o != last) { rows_p[o].push_back(i); last = o; }
        }
    }
    tm.lap("cut edges per peer");
    P.send_counts.assign(W, 0);
    P.send_counts_cp.assign(size_t(Cn) * W, 0);
    std::vector<std::vector<u64>> cutp(W, std::vector<u64>(Cn + 1, 0));
    for (u32 p = 0; p < W; ++p) {
        P.send_counts[p] = rows_p[p].size();
        for (u32 c = 0; c <= Cn; ++c) cutp[p][c] = u64(std::lower_bound(rows_p[p].begin(), rows_p[p].end(), P.chunk_row[c]) - rows_p[p].begin());
        for (u32 c = 0; c < Cn; ++c) P.send_counts_cp[size_t(c) * W + p] = cutp[p][c + 1] - cutp[p][c];
    }
    P.send_off_c.assign(Cn + 1, 0);
    P.send_idx_chunked.clear();
    for (u32 c = 0; c < Cn; ++c) {
        for (u32 p = 0; p < W; ++p)
            P.send_idx_chunked.insert(P.send_idx_chunked.end(), rows_p[p].begin() + cutp[p][c], rows_p[p].begin() + cutp[p][c + 1]);
        P.send_off_c[c + 1] = P.send_idx_chunked.size();
    }
    tm.lap("send lists");
    // send slots of every own row (where the sweep kernel drops a fresh marginal for its readers)
    P.snd_ptr.assign(size_t(P.n_own) + 1, 0);
    for (u32 row : P.send_idx_chunked) P.snd_ptr[row + 1]++;
    for (u32 i = 0; i < P.n_own; ++i) P.snd_ptr[i + 1] += P.snd_ptr[i];
    P.snd_slot.resize(P.send_idx_chunked.size());
    {
        std::vector<u32> fill(P.snd_ptr.begin(), P.snd_ptr.end() - 1);
        for (u32 s = 0; s < P.send_idx_chunked.size(); ++s) P.snd_slot[fill[P.send_idx_chunked[s]]++] = s;
    }
    tm.lap("send slots");
    // cut edges: what I receive for peer p sits behind my own records in MY edge order; what I send to p is ordered as p's
    // edge order, i.e. by (neighbour, row)
    P.msg_counts.assign(W, 0);
    P.rev_local.resize(P.n_edges);
    for (u64 k = 0; k < P.n_edges; ++k) {
        const u64 l = nb[k];
        if (l >= lo && l < hi) P.rev_local[k] = u32(g.rev[P.edge0 + k] - P.edge0);
    }
    u64 off = 0;
    P.msg_send_edge.clear();
    for (u32 p = 0; p < W; ++p) {
        P.msg_counts[p] = cut_edges[p].size();
        for (u64 x = 0; x < cut_edges[p].size(); ++x) P.rev_local[cut_edges[p][x]] = u32(P.n_edges + off + x);
        off += cut_edges[p].size();
        // the peer's edge order is (neighbour, row): a stable counting sort of my (row, neighbour)-ordered cut edges by the
        // neighbour's offset in the peer's row range (a comparison sort through nb[] took most of the plan's time)
        const std::vector<u32> &ce = cut_edges[p];
        if (!ce.empty()) {
            const u64 plo = P.bounds[p], pn = P.bounds[p + 1] - plo;
            std::vector<u32> cnt(size_t(pn) + 1, 0);
            for (u32 a : ce) cnt[size_t(nb[a] - plo) + 1]++;
            for (u64 v = 0; v < pn; ++v) cnt[v + 1] += cnt[v];
            const size_t base = P.msg_send_edge.size();
            P.msg_send_edge.resize(base + ce.size());
            for (u32 a : ce) P.msg_send_edge[base + cnt[size_t(nb[a] - plo)]++] = a;
        }
    }
    tm.lap("cut-edge records");
    P.n_halo_msgs = off;
    if (P.n_edges + P.n_halo_msgs >= (u64(1) << 32)) { set_error("a shard's message buffer exceeds 2^32 records"); return SBMBP_ERR_UNSUPPORTED; }
    P.table_deg.resize(size_t(P.n_own) + P.n_halo);
    for (u32 i = 0; i < P.n_own; ++i) P.table_deg[i] = u32(rp[lo + i + 1] - rp[lo + i]);
    for (u32 x = 0; x < P.n_halo; ++x) { const u64 v = P.halo_global[x]; P.table_deg[P.n_own + x] = u32(rp[v + 1] - rp[v]); }
    tm.lap("table degrees");
    return SBMBP_OK;
}

// =====================================================================================================
// Communicators. Three transports behind one interface (all buffers are device memory, counts are rows of `width` doubles):
//   rccl      — production: two RCCL communicators over xGMI, `halo` for the per-chunk all-to-all-v (grouped
//               ncclSend/ncclRecv on the driver's exchange stream) and `red` for the small all-gather / all-reduce on the
//               compute stream, so the two kinds of traffic never wait for each other inside one communicator;
//   local     — the ranks are threads of this process (tests and rehearsals on one GPU, where RCCL refuses duplicate
//               devices): device-to-device copies between the ranks' buffers, host barriers between the phases;
//   callbacks — the ranks are processes without a device collective library between them (rehearsal over gloo on a
//               one-GPU box): buffers are staged through page-locked host memory and handed to the caller's functions.
// =====================================================================================================
struct local_group {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    u64 generation = 0;
    bool failed = false;  // a rank gave up (error or time-out): every barrier returns false from then on, nobody blocks
    double timeout_s = 600.0;
    struct slot {
        const double *send = nullptr;
        const u64 *send_counts = nullptr;
        int width = 0;
        hipEvent_t ready = nullptr, done = nullptr;
        std::vector<double> host;  // all-reduce staging
    };
    std::vector<slot> slots;
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const u64 gen = generation;
        if (++waiting == n) { waiting = 0; ++generation; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || failed; })) {
            failed = true;  // a peer never arrived (it failed outside a collective, or the ranks' call sequences differ)
            cv.notify_all();
        }
        return !failed;
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};
#define LOCAL_BARRIER(g)                                                                    \
    do {                                                                                    \
        if (!(g).barrier()) { set_error("a peer rank failed or timed out"); return SBMBP_ERR_COMM; } \
    } while (0)
// a HIP error between two barriers must not leave the peers waiting
#define LOCAL_HIP(g, call)                                                                  \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess) { (g).fail(); set_error(std::string(#call) + ": " + hipGetErrorString(_e)); return SBMBP_ERR_HIP; } \
    } while (0)

}  // namespace

struct sbmbp_comm {
    int kind = 0;  // 0 rccl, 1 local, 2 callbacks, 3 null (measurement: one rank of a W-rank plan alone, peers never answer)
    int rank = 0, world = 1, device = -1;
    ncclComm_t halo = nullptr, red = nullptr;
    bool one_comm = false;             // SBMBP_SHARD_ONE_COMM=1: red IS halo (both kinds of traffic serialised on one communicator)
    std::atomic<bool> aborted{false};  // sbmbp_comm_abort may come from ANOTHER thread (a rank that failed outside a collective)
    std::shared_ptr<local_group> grp;
    sbmbp_comm_callbacks cb{};
    double *h_a = nullptr, *h_b = nullptr;  // page-locked staging (callbacks)
    size_t h_cap = 0;
};

namespace {

int ensure_host(sbmbp_comm *c, size_t doubles) {
    if (doubles <= c->h_cap) return SBMBP_OK;
    if (c->h_a) (void)hipHostFree(c->h_a);
    if (c->h_b) (void)hipHostFree(c->h_b);
    c->h_a = c->h_b = nullptr;
    c->h_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_a), doubles * 8, hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_b), doubles * 8, hipHostMallocDefault));
    c->h_cap = doubles;
    return SBMBP_OK;
}

u64 sum_counts(const u64 *c, int n) { u64 s = 0; for (int i = 0; i < n; ++i) s += c[i]; return s; }

// all-to-all-v of rows: my rows for peer p start at row sum(send_counts[0..p)) of `send`; what p sends me lands at row
// sum(recv_counts[0..p)) of `recv`
int comm_exchange(sbmbp_comm *c, const double *send, const u64 *send_counts, double *recv, const u64 *recv_counts, int width,
                  hipStream_t stream) {
    const int W = c->world;
    if (W == 1 || c->kind == 3) return SBMBP_OK;
    if (c->kind == 0) {
        // The group is ALWAYS closed: returning between ncclGroupStart and ncclGroupEnd would leave this thread's group open
        // and queue every later RCCL call of the thread - abort and destroy included - into it. The first error is kept.
        // (Two communicators run side by side on one device: `halo` here on the exchange stream, `red` for the all-gather /
        // all-reduce on the compute stream, so that the two kinds of traffic never queue behind each other inside one
        // communicator. Their kernels must be co-resident for that; SBMBP_SHARD_ONE_COMM=1 puts both on ONE communicator and
        // serialises them by stream order, as a fallback for a node where they are not - sbmbp_comm_init_rank.)
        NCCLCHK(ncclGroupStart());
        ncclResult_t first = ncclSuccess;
        u64 so = 0, ro = 0;
        for (int p = 0; p < W && first == ncclSuccess; ++p) {
            if (send_counts[p]) first = ncclSend(send + so * width, send_counts[p] * width, ncclDouble, p, c->halo, stream);
            if (first == ncclSuccess && recv_counts[p]) first = ncclRecv(recv + ro * width, recv_counts[p] * width, ncclDouble, p, c->halo, stream);
            so += send_counts[p];
            ro += recv_counts[p];
        }
        const ncclResult_t end = ncclGroupEnd();
        if (first != ncclSuccess || end != ncclSuccess) {
            set_error(std::string("halo exchange (grouped ncclSend/ncclRecv): ") + ncclGetErrorString(first != ncclSuccess ? first : end));
            return SBMBP_ERR_COMM;
        }
        return SBMBP_OK;
    }
    if (c->kind == 1) {
        local_group &g = *c->grp;
        local_group::slot &me = g.slots[c->rank];
        me.send = send;
        me.send_counts = send_counts;
        me.width = width;
        LOCAL_HIP(g, hipEventRecord(me.ready, stream));  // my send rows are final behind this point of my stream
        LOCAL_BARRIER(g);
        u64 ro = 0;
        for (int p = 0; p < W; ++p) {  // pull what every peer has for me
            if (recv_counts[p]) {
                const local_group::slot &pe = g.slots[p];
                u64 so = 0;
                for (int q = 0; q < c->rank; ++q) so += pe.send_counts[q];
                if (pe.send_counts[c->rank] != recv_counts[p] || pe.width != width) {
                    g.fail();
                    set_error("exchange counts of two ranks disagree");
                    return SBMBP_ERR_COMM;
                }
                LOCAL_HIP(g, hipStreamWaitEvent(stream, pe.ready, 0));
                LOCAL_HIP(g, hipMemcpyAsync(recv + ro * width, pe.send + so * width, recv_counts[p] * width * 8, hipMemcpyDeviceToDevice, stream));
            }
            ro += recv_counts[p];
        }
        LOCAL_HIP(g, hipEventRecord(me.done, stream));  // my reads of the peers' send rows end here
        LOCAL_BARRIER(g);
        for (int p = 0; p < W; ++p)  // a peer that read my rows must be through before later work on my stream rewrites them
            if (send_counts[p]) LOCAL_HIP(g, hipStreamWaitEvent(stream, g.slots[p].done, 0));
        LOCAL_BARRIER(g);  // nobody re-records an event a peer has not consumed yet
        return SBMBP_OK;
    }
    const u64 ns = sum_counts(send_counts, W), nr = sum_counts(recv_counts, W);
    CHK(ensure_host(c, std::max<u64>(1, std::max(ns, nr) * width)));
    if (ns) HIPCHK(hipMemcpyAsync(c->h_a, send, ns * width * 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (c->cb.exchange(c->cb.user, c->h_a, send_counts, c->h_b, recv_counts, width) != 0) { set_error("exchange callback failed"); return SBMBP_ERR_COMM; }
    if (nr) HIPCHK(hipMemcpyAsync(recv, c->h_b, nr * width * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return SBMBP_OK;
}

// every rank's in[0..n) -> out[r * n ..] on all ranks (in and out must not overlap)
int comm_allgather(sbmbp_comm *c, const double *in, size_t n, double *out, hipStream_t stream) {
    const int W = c->world;
    if (W == 1) { HIPCHK(hipMemcpyAsync(out, in, n * 8, hipMemcpyDeviceToDevice, stream)); return SBMBP_OK; }
    if (c->kind == 3) {  // every peer "reports" what this rank does
        for (int p = 0; p < W; ++p) HIPCHK(hipMemcpyAsync(out + size_t(p) * n, in, n * 8, hipMemcpyDeviceToDevice, stream));
        return SBMBP_OK;
    }
    if (c->kind == 0) { NCCLCHK(ncclAllGather(in, out, n, ncclDouble, c->red, stream)); return SBMBP_OK; }
    if (c->kind == 1) {
        local_group &g = *c->grp;
        local_group::slot &me = g.slots[c->rank];
        me.send = in;
        LOCAL_HIP(g, hipEventRecord(me.ready, stream));
        LOCAL_BARRIER(g);
        for (int p = 0; p < W; ++p) {
            if (p != c->rank) LOCAL_HIP(g, hipStreamWaitEvent(stream, g.slots[p].ready, 0));
            LOCAL_HIP(g, hipMemcpyAsync(out + size_t(p) * n, g.slots[p].send, n * 8, hipMemcpyDeviceToDevice, stream));
        }
        LOCAL_HIP(g, hipEventRecord(me.done, stream));
        LOCAL_BARRIER(g);
        for (int p = 0; p < W; ++p) if (p != c->rank) LOCAL_HIP(g, hipStreamWaitEvent(stream, g.slots[p].done, 0));
        LOCAL_BARRIER(g);
        return SBMBP_OK;
    }
    CHK(ensure_host(c, std::max<size_t>(1, n * W)));
    HIPCHK(hipMemcpyAsync(c->h_a, in, n * 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (c->cb.allgather(c->cb.user, c->h_a, n, c->h_b) != 0) { set_error("allgather callback failed"); return SBMBP_ERR_COMM; }
    HIPCHK(hipMemcpyAsync(out, c->h_b, n * W * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return SBMBP_OK;
}

// in-place all-reduce (op 0 sum, 1 max). Sums are taken in rank order on the non-RCCL transports.
int comm_allreduce(sbmbp_comm *c, double *buf, size_t n, int op, hipStream_t stream) {
    const int W = c->world;
    if (W == 1 || n == 0 || c->kind == 3) return SBMBP_OK;
    if (c->kind == 0) { NCCLCHK(ncclAllReduce(buf, buf, n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->red, stream)); return SBMBP_OK; }
    if (c->kind == 1) {
        local_group &g = *c->grp;
        local_group::slot &me = g.slots[c->rank];
        me.host.resize(n);
        LOCAL_HIP(g, hipMemcpyAsync(me.host.data(), buf, n * 8, hipMemcpyDeviceToHost, stream));
        LOCAL_HIP(g, hipStreamSynchronize(stream));
        LOCAL_BARRIER(g);
        std::vector<double> acc(g.slots[0].host);
        for (int p = 1; p < W; ++p) {
            if (g.slots[p].host.size() != n) { g.fail(); set_error("all-reduce sizes of two ranks disagree"); return SBMBP_ERR_COMM; }
            for (size_t i = 0; i < n; ++i) {
                const double v = g.slots[p].host[i];
                acc[i] = op == 0 ? acc[i] + v : ((v > acc[i] || v != v) ? v : acc[i]);
            }
        }
        LOCAL_BARRIER(g);  // everybody has read the staged vectors
        LOCAL_HIP(g, hipMemcpyAsync(buf, acc.data(), n * 8, hipMemcpyHostToDevice, stream));
        LOCAL_HIP(g, hipStreamSynchronize(stream));  // acc is a local
        return SBMBP_OK;
    }
    CHK(ensure_host(c, n));
    HIPCHK(hipMemcpyAsync(c->h_a, buf, n * 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (c->cb.allreduce(c->cb.user, c->h_a, n, op) != 0) { set_error("allreduce callback failed"); return SBMBP_ERR_COMM; }
    HIPCHK(hipMemcpyAsync(buf, c->h_a, n * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return SBMBP_OK;
}

}  // namespace

extern "C" {

int sbmbp_comm_unique_id(void *id_out) {
    if (!id_out) return arg_error(__func__, __LINE__);
    static_assert(2 * sizeof(ncclUniqueId) <= SBMBP_COMM_ID_BYTES, "SBMBP_COMM_ID_BYTES too small");
    ncclUniqueId ids[2];
    NCCLCHK(ncclGetUniqueId(&ids[0]));
    NCCLCHK(ncclGetUniqueId(&ids[1]));
    std::memset(id_out, 0, SBMBP_COMM_ID_BYTES);
    std::memcpy(id_out, ids, sizeof ids);
    return SBMBP_OK;
}

int sbmbp_comm_init_rank(sbmbp_comm_t **out, const void *id, int n_ranks, int rank, int device) {
    if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return arg_error(__func__, __LINE__);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device visible"); return SBMBP_ERR_NODEVICE; }
    if (device >= 0) HIPCHK(hipSetDevice(device));
    auto *c = new sbmbp_comm();
    c->kind = 0;
    c->rank = rank;
    c->world = n_ranks;
    hipGetDevice(&c->device);
    ncclUniqueId ids[2];
    std::memcpy(ids, id, sizeof ids);
    ncclResult_t r = ncclCommInitRank(&c->halo, n_ranks, ids[0], rank);
    const char *one = std::getenv("SBMBP_SHARD_ONE_COMM");
    c->one_comm = one && std::atoi(one) != 0;
    if (c->one_comm) c->red = c->halo;  // fallback for a node where two RCCL kernels of one device are not co-resident
    else if (r == ncclSuccess) r = ncclCommInitRank(&c->red, n_ranks, ids[1], rank);
    if (r != ncclSuccess) {
        set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
        if (c->halo) ncclCommDestroy(c->halo);
        delete c;
        return SBMBP_ERR_COMM;
    }
    *out = c;
    return SBMBP_OK;
}

int sbmbp_comm_init_local(sbmbp_comm_t **out, int n_ranks) {
    if (!out || n_ranks < 1) return arg_error(__func__, __LINE__);
    auto g = std::make_shared<local_group>();
    g->n = n_ranks;
    if (const char *t = std::getenv("SBMBP_LOCAL_TIMEOUT_S")) g->timeout_s = std::max(1.0, std::atof(t));
    g->slots.resize(n_ranks);
    for (int r = 0; r < n_ranks; ++r) {
        auto *c = new sbmbp_comm();
        c->kind = 1;
        c->rank = r;
        c->world = n_ranks;
        c->grp = g;
        out[r] = c;
    }
    return SBMBP_OK;
}

int sbmbp_comm_init_callbacks(sbmbp_comm_t **out, int n_ranks, int rank, const sbmbp_comm_callbacks *cb) {
    if (!out || !cb || !cb->exchange || !cb->allgather || !cb->allreduce || n_ranks < 1 || rank < 0 || rank >= n_ranks) return arg_error(__func__, __LINE__);
    auto *c = new sbmbp_comm();
    c->kind = 2;
    c->rank = rank;
    c->world = n_ranks;
    c->cb = *cb;
    *out = c;
    return SBMBP_OK;
}

int sbmbp_comm_init_null(sbmbp_comm_t **out, int n_ranks, int rank) {
    if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return arg_error(__func__, __LINE__);
    auto *c = new sbmbp_comm();
    c->kind = 3;
    c->rank = rank;
    c->world = n_ranks;
    *out = c;
    return SBMBP_OK;
}

void sbmbp_comm_destroy(sbmbp_comm_t *c) {
    if (!c) return;
    if (!c->aborted.load()) {  // ncclCommAbort has already freed an aborted communicator
        if (c->halo) ncclCommDestroy(c->halo);
        if (c->red && !c->one_comm) ncclCommDestroy(c->red);
    }
    if (c->kind == 1 && c->grp) {
        local_group::slot &me = c->grp->slots[c->rank];
        if (me.ready) (void)hipEventDestroy(me.ready);
        if (me.done) (void)hipEventDestroy(me.done);
        me.ready = me.done = nullptr;
    }
    if (c->h_a) (void)hipHostFree(c->h_a);
    if (c->h_b) (void)hipHostFree(c->h_b);
    delete c;
}

// a rank that gives up tells the others, so that none of them blocks in a collective waiting for it
void sbmbp_comm_abort(sbmbp_comm_t *c) {
    if (!c) return;
    if (c->kind == 1 && c->grp) c->grp->fail();
    if (c->kind == 0 && !c->aborted.exchange(true)) {
        // ncclCommAbort is safe from a thread other than the one driving the communicator: kernels spinning on a dead peer
        // end, the owner's queued calls return an error. The handles stay where they are (the owner may be reading them);
        // sbmbp_comm_destroy skips what was aborted.
        if (c->halo) ncclCommAbort(c->halo);
        if (c->red && !c->one_comm) ncclCommAbort(c->red);
    }
}

int sbmbp_comm_rank(const sbmbp_comm_t *c) { return c ? c->rank : -1; }
int sbmbp_comm_size(const sbmbp_comm_t *c) { return c ? c->world : 0; }
const char *sbmbp_comm_transport(const sbmbp_comm_t *c) { return !c ? "" : (c->kind == 0 ? "rccl" : (c->kind == 1 ? "local" : (c->kind == 2 ? "callbacks" : "null"))); }

}  // extern "C"

// =====================================================================================================
// The per-rank driver
// =====================================================================================================
struct sbmbp_dist {
    sbmbp_comm *comm = nullptr;
    const sbmbp_graph_t *graph = nullptr;  // borrowed: must outlive the engine, as the adjacency must in the reference (belief_propagation.h:27)
    int rank = 0, world = 1, device = 0;
    u32 Q = 0, dc = 0, ncomp = 0;
    shard_plan plan;
    sbmbp_engine_t *eng = nullptr;
    hipStream_t s_compute = nullptr, s_comm = nullptr;
    hipStream_t s_aux[2] = {nullptr, nullptr};  // the row chunks of a sweep alternate between two streams: a chunk's tail overlaps the next chunk
    hipEvent_t ev_start = nullptr;
    bool two_streams = true;
    std::vector<hipEvent_t> ev_chunk, ev_xchg;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    double *d_psi[2] = {nullptr, nullptr};
    double *d_red = nullptr, *d_sendbuf = nullptr, *d_recvbuf[2] = {nullptr, nullptr}, *d_msg_send = nullptr, *d_psi_all = nullptr;
    u32 *d_send_idx = nullptr, *d_stage_to_halo = nullptr, *d_msg_send_edge = nullptr;
    std::vector<double> cab;
    std::vector<u32> na;
    std::vector<u32> true_conf_own;
    double beta = 1.0, field_mix = 1.0, learn_field_mix = 0.3, learn_snap = 1.0;
    u32 check_every = 8;
    int gather_mode = 0;
    bool have_params = false, have_state = false;
    bool consistent = false;  // psi == marginals of the message pair: the marginal-gather sweep may start at once
    u64 total_sweeps = 0, psi_sweeps = 0;
    bool timing = false;
    std::vector<hipEvent_t> phase_ev;  // 4 per timed sweep
    size_t phase_used = 0;
    double phase_ms[3] = {0, 0, 0};
    u64 phase_n = 0;
};

namespace {

struct device_guard {  // the current device is a per-thread setting: every entry point selects the rank's GPU
    explicit device_guard(int dev) { (void)hipSetDevice(dev); }
};

template <typename T> int dalloc(T **p, size_t count) {
    hipError_t r = hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T));
    if (r != hipSuccess) { set_error(std::string("hipMalloc: ") + hipGetErrorString(r)); return r == hipErrorOutOfMemory ? SBMBP_ERR_NOMEM : SBMBP_ERR_HIP; }
    return SBMBP_OK;
}

inline const u64 *cp_row(const std::vector<u64> &v, u32 c, int W) { return v.data() + size_t(c) * W; }

// ship rows over the exchange stream: the send rows are final at `ready` (recorded on the compute stream by the caller),
// `done` is recorded on the exchange stream behind the transfer
int ship(sbmbp_dist *d, const double *send, const u64 *sc, double *recv, const u64 *rc, int width, hipEvent_t ready, hipEvent_t done,
         hipStream_t producer = nullptr) {
    HIPCHK(hipEventRecord(ready, producer ? producer : d->s_compute));
    HIPCHK(hipStreamWaitEvent(d->s_comm, ready, 0));
    CHK(comm_exchange(d->comm, send, sc, recv, rc, width, d->s_comm));
    HIPCHK(hipEventRecord(done, d->s_comm));
    return SBMBP_OK;
}

// boundary marginals of the table sweep j reads -> the receive buffer of that table on every peer (all chunks), then
// expanded into the halo rows of the table (reductions and the field read them there)
int refresh_marginal_halo(sbmbp_dist *d, u32 j) {
    const shard_plan &P = d->plan;
    if (d->world == 1) return SBMBP_OK;
    const int rb = sbmbp_shard_read_buffer(d->eng, j);
    for (u32 c = 0; c < P.n_chunks; ++c) {
        const u64 off = P.send_off_c[c], n = P.send_off_c[c + 1] - off;
        if (n) CHK(sbmbp_shard_pack(d->eng, j, d->d_send_idx + off, u32(n), d->d_sendbuf + off * d->ncomp, d->ncomp));
        CHK(ship(d, d->d_sendbuf + off * d->ncomp, cp_row(P.send_counts_cp, c, d->world), d->d_recvbuf[rb] + P.stage_off_c[c] * d->ncomp,
                 cp_row(P.recv_counts_cp, c, d->world), int(d->ncomp), d->ev_chunk[c], d->ev_xchg[c]));
    }
    for (u32 c = 0; c < P.n_chunks; ++c) HIPCHK(hipStreamWaitEvent(d->s_compute, d->ev_xchg[c], 0));
    if (P.n_halo) CHK(sbmbp_shard_unpack(d->eng, j, d->d_recvbuf[rb], d->d_stage_to_halo, P.n_halo, d->ncomp));
    return SBMBP_OK;
}

// records of the cut edges of the message buffer sweep j reads -> behind the own records of that buffer on the peers
int refresh_message_halo(sbmbp_dist *d, u32 j) {
    const shard_plan &P = d->plan;
    if (d->world == 1) return SBMBP_OK;
    // (a rank without cut edges still takes part: a collective is a collective for every rank of the communicator)
    if (P.n_halo_msgs) CHK(sbmbp_shard_pack_msgs(d->eng, j, d->d_msg_send_edge, u32(P.n_halo_msgs), d->d_msg_send));
    CHK(ship(d, d->d_msg_send, P.msg_counts.data(), static_cast<double *>(sbmbp_shard_msg_halo(d->eng, j)), P.msg_counts.data(), int(d->Q) - 1,
             d->ev_a, d->ev_b));
    HIPCHK(hipStreamWaitEvent(d->s_compute, d->ev_b, 0));
    return SBMBP_OK;
}

// every rank's first `rows` rows of (Q+1) reduction values -> behind red + SBMBP_RED_GATHER_OFFSET on all ranks, in rank
// order: ONE collective per sweep; k_finalize folds the gathered rows identically everywhere
int gather_red(sbmbp_dist *d, u32 rows) {
    return comm_allgather(d->comm, d->d_red, size_t(rows) * (d->Q + 1), d->d_red + SBMBP_RED_GATHER_OFFSET, d->s_compute);
}

int reduce_red(sbmbp_dist *d, size_t n_sum, size_t n_max) {
    if (n_sum) CHK(comm_allreduce(d->comm, d->d_red, n_sum, 0, d->s_compute));
    if (n_max) CHK(comm_allreduce(d->comm, d->d_red + n_sum, n_max, 1, d->s_compute));
    return SBMBP_OK;
}

// the marginal-gather form on shards: as on the single engine (damping 1, cab > 0, dc != 2), and no clamped rows at all
// (their marginals would have to be re-shipped unchanged every sweep; clamped runs take the message-gather form)
bool psi_form_allowed(const sbmbp_dist *d, double damping) {
    return d->gather_mode == 0 && damping == 1.0 && d->dc != 2 && d->plan.e2_global > 0 && sbmbp_shard_query(d->eng, 0) == 1 &&
           sbmbp_shard_query(d->eng, 1) == 0;
}

hipEvent_t *phase_quad(sbmbp_dist *d) {
    if (!d->timing) return nullptr;
    if (d->phase_used + 4 > d->phase_ev.size()) {
        const size_t old = d->phase_ev.size();
        d->phase_ev.resize(old + 256);
        for (size_t i = old; i < d->phase_ev.size(); ++i) if (hipEventCreate(&d->phase_ev[i]) != hipSuccess) return nullptr;
    }
    hipEvent_t *q = &d->phase_ev[d->phase_used];
    d->phase_used += 4;
    return q;
}

// marginal-gather sweep j: the table it reads has its halo in the receive buffer (shipped during sweep j-1, or by begin);
// the new marginals of chunk c travel (exchange stream) while chunk c+1 is swept (compute stream)
int queue_sweep_psi(sbmbp_dist *d, u32 j) {
    const shard_plan &P = d->plan;
    const int rb_next = sbmbp_shard_read_buffer(d->eng, j + 1);
    hipEvent_t *ev = phase_quad(d);
    if (ev) HIPCHK(hipEventRecord(ev[0], d->s_compute));
    const bool split = d->two_streams && P.n_chunks > 1;
    if (split) HIPCHK(hipEventRecord(d->ev_start, d->s_compute));  // finalize of the last sweep and its exchanges are behind this point
    for (u32 c = 0; c < P.n_chunks; ++c) {
        hipStream_t sc = split ? d->s_aux[c & 1] : d->s_compute;
        if (split) HIPCHK(hipStreamWaitEvent(sc, d->ev_start, 0));
        CHK(sbmbp_shard_sweep_chunk_on(d->eng, j, c, sc));
        if (d->world > 1) {
            const u64 off = P.send_off_c[c];
            CHK(ship(d, d->d_sendbuf + off * d->ncomp, cp_row(P.send_counts_cp, c, d->world), d->d_recvbuf[rb_next] + P.stage_off_c[c] * d->ncomp,
                     cp_row(P.recv_counts_cp, c, d->world), int(d->ncomp), d->ev_chunk[c], d->ev_xchg[c], sc));
        } else if (split) {
            HIPCHK(hipEventRecord(d->ev_chunk[c], sc));
        }
    }
    if (split)  // the folds read every chunk's partials
        for (u32 c = 0; c < P.n_chunks; ++c) HIPCHK(hipStreamWaitEvent(d->s_compute, d->ev_chunk[c], 0));
    if (ev) HIPCHK(hipEventRecord(ev[1], d->s_compute));
    CHK(sbmbp_shard_sweep_fold(d->eng));  // the local fold overlaps with the last chunk's exchange (it does not touch the halo)
    CHK(gather_red(d, SBMBP_FOLD_ROWS));
    CHK(sbmbp_shard_finalize(d->eng, 0, u32(d->world) * SBMBP_FOLD_ROWS, 0));
    if (ev) HIPCHK(hipEventRecord(ev[2], d->s_compute));
    if (d->world > 1)  // the next sweep reads the receive buffers: its kernels wait for the exchanges here
        for (u32 c = 0; c < P.n_chunks; ++c) HIPCHK(hipStreamWaitEvent(d->s_compute, d->ev_xchg[c], 0));
    if (ev) HIPCHK(hipEventRecord(ev[3], d->s_compute));
    return SBMBP_OK;
}

// message-gather sweep j (any damping, clamped rows, dc 2, zeros in cab; also the first sweep after a state or parameter
// change): the incoming messages of the cut edges are shipped first; when a marginal-gather sweep may follow, the boundary
// marginals it will read are shipped afterwards
int queue_sweep_explicit(sbmbp_dist *d, u32 j, double damping, bool ship_marginals) {
    const shard_plan &P = d->plan;
    CHK(refresh_message_halo(d, j));
    CHK(sbmbp_shard_sweep_explicit(d->eng, j, damping));
    if (ship_marginals && d->world > 1) {
        const int rb = sbmbp_shard_read_buffer(d->eng, j + 1);
        for (u32 c = 0; c < P.n_chunks; ++c) {
            const u64 off = P.send_off_c[c], n = P.send_off_c[c + 1] - off;
            if (n) CHK(sbmbp_shard_pack(d->eng, j + 1, d->d_send_idx + off, u32(n), d->d_sendbuf + off * d->ncomp, d->ncomp));
            CHK(ship(d, d->d_sendbuf + off * d->ncomp, cp_row(P.send_counts_cp, c, d->world), d->d_recvbuf[rb] + P.stage_off_c[c] * d->ncomp,
                     cp_row(P.recv_counts_cp, c, d->world), int(d->ncomp), d->ev_chunk[c], d->ev_xchg[c]));
        }
    }
    CHK(sbmbp_shard_sweep_fold(d->eng));
    CHK(gather_red(d, SBMBP_FOLD_ROWS));
    CHK(sbmbp_shard_finalize(d->eng, 0, u32(d->world) * SBMBP_FOLD_ROWS, 1));
    if (ship_marginals && d->world > 1)
        for (u32 c = 0; c < P.n_chunks; ++c) HIPCHK(hipStreamWaitEvent(d->s_compute, d->ev_xchg[c], 0));
    return SBMBP_OK;
}

int collect_phases(sbmbp_dist *d) {
    for (size_t i = 0; i + 3 < d->phase_used; i += 4) {
        float a = 0, b = 0, c = 0;
        HIPCHK(hipEventElapsedTime(&a, d->phase_ev[i], d->phase_ev[i + 1]));
        HIPCHK(hipEventElapsedTime(&b, d->phase_ev[i + 1], d->phase_ev[i + 2]));
        HIPCHK(hipEventElapsedTime(&c, d->phase_ev[i + 2], d->phase_ev[i + 3]));
        d->phase_ms[0] += a;
        d->phase_ms[1] += b;
        d->phase_ms[2] += c;
        d->phase_n++;
    }
    d->phase_used = 0;
    return SBMBP_OK;
}

// field of the current marginals (init_h, bp.cpp:320-332) on every rank; mode 1 of k_finalize
int begin_run(sbmbp_dist *d, double crit, bool hinted, bool marginal_halo) {
    if (marginal_halo) CHK(refresh_marginal_halo(d, 0));
    CHK(sbmbp_shard_begin(d->eng, crit, hinted ? 1 : 0));
    CHK(sbmbp_shard_field_partial(d->eng, 0));
    CHK(gather_red(d, 1));
    CHK(sbmbp_shard_finalize(d->eng, 1, u32(d->world), 0));
    return SBMBP_OK;
}

int exact_diff(sbmbp_dist *d, double *out) {
    CHK(sbmbp_shard_msgdiff_partial(d->eng));
    CHK(reduce_red(d, 0, 1));
    HIPCHK(hipMemcpyAsync(out, d->d_red, 8, hipMemcpyDeviceToHost, d->s_compute));
    HIPCHK(hipStreamSynchronize(d->s_compute));
    return SBMBP_OK;
}

// converge (bp.cpp:386-415) over all shards. Every decision (2-step hints arming the exact criterion, the stop flag) is taken
// by k_finalize from the SAME gathered rows on every rank, so all ranks queue and stop identically without talking about it.
int run(sbmbp_dist *d, double crit, u32 max_sweeps, double damping, int *niter, double *last) {
    if (!d->have_params || !d->have_state) { set_error("set_params and an initial state must precede converge"); return SBMBP_ERR_STATE; }
    const bool psi_ok = psi_form_allowed(d, damping);
    const bool first_from_psi = psi_ok && sbmbp_shard_query(d->eng, 3) == 1;
    const bool first_explicit = !d->consistent && !first_from_psi;
    CHK(sbmbp_set_schedule(d->eng, d->field_mix, 1));
    CHK(begin_run(d, crit, psi_ok, psi_ok && !first_explicit));
    u32 done = 0;
    sbmbp_conv_state cs{0.0, -1, 0, 0, 1, 0, 0, -1};
    // batch sizes follow the decay of the reported difference (as run_sweeps of the single engine does): identical on every
    // rank, because the state they are computed from is
    const u32 batch_max = std::max<u32>(1, d->check_every);
    u32 next_batch = batch_max;
    double prev_md = -1.0;
    int prev_idx = 0;
    auto plan_next = [&](const sbmbp_conv_state &st) {
        if (crit > 0 && prev_md > 0 && st.maxdiff > 0 && st.maxdiff < prev_md && st.sweep_idx > prev_idx) {
            const double rate = std::pow(st.maxdiff / prev_md, 1.0 / double(st.sweep_idx - prev_idx));
            const double need = st.maxdiff > crit ? std::ceil(std::log(crit / st.maxdiff) / std::log(rate)) : 1.0;
            const double ahead = double(done) - double(st.sweep_idx);
            next_batch = u32(std::min<double>(batch_max, std::max(1.0, need - ahead)));
        } else {
            next_batch = batch_max;
        }
        if (st.maxdiff > 0) { prev_md = st.maxdiff; prev_idx = st.sweep_idx; }
    };
    bool form_psi = psi_ok;  // adaptive relaxation can ask for damping in the middle of a run (sbmbp_conv_state::pause): every
                             // rank sees the same state, so all switch to the message-gather form on the same sweep
    u32 psi_count = 0;
    auto queue_batch = [&](int slot) -> int {
        const u32 batch = std::min(next_batch, max_sweeps - done);
        for (u32 b = 0; b < batch; ++b) {
            const u32 j = done + b;
            if (form_psi && !(j == 0 && first_explicit)) CHK(queue_sweep_psi(d, j));
            else CHK(queue_sweep_explicit(d, j, damping, form_psi));
        }
        done += batch;
        return sbmbp_shard_state_record(d->eng, slot);
    };
    while (done < max_sweeps) {
        const u32 start = done;
        CHK(queue_batch(0));
        for (int k = 0;; ++k) {
            const bool more = done < max_sweeps;
            if (more) CHK(queue_batch((k + 1) & 1));
            CHK(sbmbp_shard_state_wait(d->eng, k & 1, &cs));
            plan_next(cs);
            if (cs.stop || !more) {
                if (more) CHK(sbmbp_shard_state_wait(d->eng, (k + 1) & 1, &cs));  // drain the batch queued ahead (no-ops after a stop)
                break;
            }
        }
        if (form_psi) psi_count += u32(cs.sweep_idx) - start - ((first_explicit && start == 0 && cs.sweep_idx > 0) ? 1 : 0);
        if (!(cs.stop && cs.pause)) break;
        done = u32(cs.sweep_idx);
        form_psi = false;
        CHK(sbmbp_shard_resume(d->eng));
        next_batch = batch_max;
        prev_md = -1.0;
    }
    HIPCHK(hipStreamSynchronize(d->s_comm));
    for (auto sx : d->s_aux) HIPCHK(hipStreamSynchronize(sx));
    HIPCHK(hipStreamSynchronize(d->s_compute));
    sbmbp_conv_state poll;
    CHK(sbmbp_shard_poll(d->eng, &poll));  // also collects the kernel timing events of the engine
    if (d->timing) CHK(collect_phases(d));
    const u32 executed = u32(cs.sweep_idx);
    CHK(sbmbp_shard_commit(d->eng, executed));
    double exact = cs.maxdiff;
    if (executed > 0 && last != nullptr && !cs.last_exact) CHK(exact_diff(d, &exact));
    d->total_sweeps += executed;
    d->psi_sweeps += psi_count;
    if (executed > 0)
        d->consistent = damping == 1.0 && cs.ar_generic_level < 2 /* levels 0 and 1 leave the damping at 1 */ && sbmbp_shard_query(d->eng, 0) == 1 && sbmbp_shard_query(d->eng, 1) == 0;
    if (niter) *niter = cs.conv_iter;
    if (last) *last = exact;
    return SBMBP_OK;
}

// everything the reductions read beyond the owned rows: halo marginals in the table, the incoming messages of the cut
// edges behind the own records, and the exact global field of the current marginals
int refresh_for_reductions(sbmbp_dist *d) {
    if (!d->have_params || !d->have_state) { set_error("engine has no parameters or no state"); return SBMBP_ERR_STATE; }
    CHK(refresh_marginal_halo(d, 0));
    CHK(refresh_message_halo(d, 0));
    CHK(sbmbp_shard_set_incoming(d->eng, 1));
    CHK(sbmbp_shard_begin(d->eng, -1.0, 0));  // parameter block in sync with the host mirror
    CHK(sbmbp_shard_field_partial(d->eng, 0));
    CHK(gather_red(d, 1));
    CHK(sbmbp_shard_finalize(d->eng, 1, u32(d->world), 0));
    return SBMBP_OK;
}

int read_red(sbmbp_dist *d, double *out, size_t n) {
    HIPCHK(hipMemcpyAsync(out, d->d_red, n * 8, hipMemcpyDeviceToHost, d->s_compute));
    HIPCHK(hipStreamSynchronize(d->s_compute));
    return SBMBP_OK;
}

// {f_site, f_edge, f_nonedge, e_site, e_edge, e_nonedge}   (bp.cpp:744-758)
int fe_terms(sbmbp_dist *d, bool want_entropy, double out[6]) {
    CHK(refresh_for_reductions(d));
    CHK(sbmbp_shard_fe_partial(d->eng, want_entropy ? 1 : 0));
    CHK(reduce_red(d, 5, 0));
    double fe[4];
    CHK(sbmbp_shard_fe_finish(d->eng, fe));
    double ne[2] = {0.0, 0.0};
    const u32 N = d->plan.n_global;
    if (d->dc == 0 && N <= 32768) {
        // small graphs: the reference's O(N^2) loop exactly, as the single engine does: every rank gets the marginals of all
        // vertices (own rows summed into a global table) and sums its own rows against them
        if (!d->d_psi_all) CHK(dalloc(&d->d_psi_all, size_t(N) * d->Q));
        HIPCHK(hipMemsetAsync(d->d_psi_all, 0, size_t(N) * d->Q * 8, d->s_compute));
        HIPCHK(hipMemcpyAsync(d->d_psi_all + size_t(d->plan.row0) * d->Q, d->d_psi[sbmbp_shard_read_buffer(d->eng, 0)],
                              size_t(d->plan.n_own) * d->Q * 8, hipMemcpyDeviceToDevice, d->s_compute));
        CHK(comm_allreduce(d->comm, d->d_psi_all, size_t(N) * d->Q, 0, d->s_compute));
        CHK(sbmbp_shard_nonedge_exact_partial(d->eng, d->d_psi_all, want_entropy ? 1 : 0));
        CHK(reduce_red(d, 4, 0));
        double v[4];
        CHK(read_red(d, v, 4));
        ne[0] = (v[0] - v[2]) / (2.0 * N);
        ne[1] = (v[1] - v[3]) / (2.0 * N);
    } else {
        u32 n = 0;
        int order = 0;
        CHK(sbmbp_shard_nonedge_partial(d->eng, want_entropy ? 1 : 0, &n, &order));
        if (n) CHK(reduce_red(d, n, 0));
        CHK(sbmbp_shard_nonedge_finish(d->eng, want_entropy ? 1 : 0, order, ne));
    }
    out[0] = fe[0]; out[1] = fe[1]; out[2] = ne[0];
    out[3] = fe[2]; out[4] = fe[3]; out[5] = ne[1];
    return SBMBP_OK;
}

int row_sums(sbmbp_dist *d, std::vector<double> &out) {
    const u32 Q = d->Q, T = 2 * Q + Q * Q;
    CHK(sbmbp_shard_rowsums_partial(d->eng));
    CHK(reduce_red(d, T, 0));
    out.resize(T);
    return read_red(d, out.data(), T);
}

int overlap_of(sbmbp_dist *d, double *ov, double *Cout) {
    if (!d->have_state) { set_error("engine has no state"); return SBMBP_ERR_STATE; }
    const u32 Q = d->Q;
    std::vector<double> rs;
    CHK(row_sums(d, rs));
    const double *C = rs.data() + 2 * Q;
    if (Cout) std::copy(C, C + Q * Q, Cout);
    if (ov) {
        std::vector<u32> perm(Q);
        std::iota(perm.begin(), perm.end(), 0u);
        double best = -1.0;
        do {  // compute_overlap (bp.cpp:775-811): all Q! permutations for Q <= 8, the identity alone above (:784-790)
            double s = 0.0;
            for (u32 a = 0; a < Q; ++a) s += C[a * Q + perm[a]];
            s /= double(d->plan.n_global);
            if (s > best) best = s;
            if (Q > 8) break;
        } while (std::next_permutation(perm.begin(), perm.end()));
        *ov = best;
    }
    return SBMBP_OK;
}

int em_expect(sbmbp_dist *d, double *na_e, double *nna_e, double *cab_e) {
    CHK(refresh_for_reductions(d));
    u32 n = 0;
    CHK(sbmbp_shard_em_partial(d->eng, &n));
    CHK(reduce_red(d, n, 0));
    return sbmbp_shard_em_finish(d->eng, na_e, nna_e, cab_e);
}

int apply_params(sbmbp_dist *d, const double *cab, const u32 *na, double beta) {
    d->cab.assign(cab, cab + size_t(d->Q) * d->Q);
    d->na.assign(na, na + d->Q);
    d->beta = beta;
    d->have_params = true;
    d->consistent = false;  // the reconstruction psi / (W^T m) needs the W the marginals were formed with
    return sbmbp_set_params(d->eng, cab, na, beta);
}

}  // namespace

extern "C" {

int sbmbp_dist_create(sbmbp_dist_t **out, sbmbp_comm_t *comm, const sbmbp_graph_t *g, uint32_t Q, uint32_t dc, int device,
                      uint32_t n_chunks) {
    if (!out || !comm || !g) return arg_error(__func__, __LINE__);
    if (Q < 2 || Q > 16) { set_error("the multi-GPU driver handles Q in [2, 16]"); return SBMBP_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device visible; the engine has no CPU fallback"); return SBMBP_ERR_NODEVICE; }
    if (device >= 0) HIPCHK(hipSetDevice(device));
    // (a failure half way frees what exists so far: sbmbp_dist_destroy copes with a partly built object)
    std::unique_ptr<sbmbp_dist, void (*)(sbmbp_dist *)> d(new sbmbp_dist(), sbmbp_dist_destroy);
    HIPCHK(hipGetDevice(&d->device));
    d->comm = comm;
    d->graph = g;
    d->rank = comm->rank;
    d->world = comm->world;
    d->Q = Q;
    d->dc = dc;
    const char *env_c = std::getenv("SBMBP_SHARD_CHUNKS");  // tuning knob of the compute/exchange overlap
    if (n_chunks == 0) n_chunks = d->world == 1 ? 1 : (env_c ? std::max(1, std::atoi(env_c)) : 4);
    CHK(build_plan(*g, d->world, d->rank, n_chunks, d->plan));
    const shard_plan &P = d->plan;
    const char *env_h = std::getenv("SBMBP_HALO_COMPRESS");  // Q-1 components of a marginal on the wire (default) or all Q
    d->ncomp = (env_h && std::string(env_h) == "0") ? Q : Q - 1;
    const size_t n_tab = size_t(P.n_own) + P.n_halo;
    for (int t = 0; t < 2; ++t) {
        CHK(dalloc(&d->d_psi[t], n_tab * Q));
        HIPCHK(hipMemset(d->d_psi[t], 0, std::max<size_t>(1, n_tab * Q) * 8));
        CHK(dalloc(&d->d_recvbuf[t], size_t(P.n_halo) * d->ncomp));
        HIPCHK(hipMemset(d->d_recvbuf[t], 0, std::max<size_t>(1, size_t(P.n_halo) * d->ncomp) * 8));
    }
    const size_t red_cap = std::max<size_t>(8192, SBMBP_RED_GATHER_OFFSET + size_t(d->world) * SBMBP_FOLD_ROWS * (16 + 1));
    CHK(dalloc(&d->d_red, red_cap));
    HIPCHK(hipMemset(d->d_red, 0, red_cap * 8));
    CHK(dalloc(&d->d_sendbuf, P.send_idx_chunked.size() * d->ncomp));
    CHK(dalloc(&d->d_send_idx, P.send_idx_chunked.size()));
    CHK(dalloc(&d->d_stage_to_halo, P.n_halo));
    CHK(dalloc(&d->d_msg_send, size_t(P.n_halo_msgs) * (Q - 1)));
    CHK(dalloc(&d->d_msg_send_edge, P.n_halo_msgs));
    if (!P.send_idx_chunked.empty()) HIPCHK(hipMemcpy(d->d_send_idx, P.send_idx_chunked.data(), P.send_idx_chunked.size() * 4, hipMemcpyHostToDevice));
    if (P.n_halo) {
        std::vector<u32> ident(P.n_halo);
        std::iota(ident.begin(), ident.end(), 0u);
        HIPCHK(hipMemcpy(d->d_stage_to_halo, ident.data(), size_t(P.n_halo) * 4, hipMemcpyHostToDevice));
    }
    if (P.n_halo_msgs) HIPCHK(hipMemcpy(d->d_msg_send_edge, P.msg_send_edge.data(), P.n_halo_msgs * 4, hipMemcpyHostToDevice));
    sbmbp_shard_desc desc;
    std::memset(&desc, 0, sizeof desc);
    desc.n_global = P.n_global;
    desc.n_own = P.n_own;
    desc.n_halo = P.n_halo;
    desc.row0 = P.row0;
    desc.n_edges = P.n_edges;
    desc.edge0 = P.edge0;
    static const u32 none32 = 0;  // an empty shard still HAS these arrays (a null pointer means "not provided")
    desc.row_ptr = P.row_ptr.data();
    desc.nbr_local = P.nbr_local.empty() ? &none32 : P.nbr_local.data();
    desc.psi_buf0 = d->d_psi[0];
    desc.psi_buf1 = d->d_psi[1];
    desc.red_buf = d->d_red;
    desc.n_chunks = P.n_chunks;
    desc.chunk_row = P.chunk_row.data();
    desc.rev_local = P.rev_local.empty() ? &none32 : P.rev_local.data();
    desc.n_halo_msgs = P.n_halo_msgs;
    desc.table_deg = P.table_deg.empty() ? &none32 : P.table_deg.data();
    CHK(sbmbp_shard_create(&d->eng, &desc, Q, dc, d->device));
    HIPCHK(hipStreamCreateWithFlags(&d->s_compute, hipStreamNonBlocking));
    // Stream priorities (opt-in: SBMBP_SHARD_PRIO=1): the even chunks' stream and the exchange stream above the odd chunks'
    // stream. Two equal streams sweep chunks 0 and 1 side by side and finish both at the half-way mark, so half of the sweep's
    // halo leaves only at the end; with the even stream preferred the chunks complete one after the other (0, 2, 1, 3 at four
    // chunks) and the exchange kernels never queue behind a sweep. Alone on the GPU (null transport) it costs 6 - 10 % of the
    // sweep (2 chunks 0.450 -> 0.502 ms, 4 chunks 0.490 -> 0.518 ms per rank of the 8-rank C3 plan), so whether the earlier
    // exchanges pay for that is for the links to say: bench.py --gpus N tries both.
    int prio_least = 0, prio_greatest = 0;
    const bool prio = std::getenv("SBMBP_SHARD_PRIO") && std::atoi(std::getenv("SBMBP_SHARD_PRIO")) != 0 &&
                      hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_least != prio_greatest;
    if (prio) {
        HIPCHK(hipStreamCreateWithPriority(&d->s_comm, hipStreamNonBlocking, prio_greatest));
        HIPCHK(hipStreamCreateWithPriority(&d->s_aux[0], hipStreamNonBlocking, prio_greatest));
        HIPCHK(hipStreamCreateWithPriority(&d->s_aux[1], hipStreamNonBlocking, prio_least));
    } else {
        HIPCHK(hipStreamCreateWithFlags(&d->s_comm, hipStreamNonBlocking));
        for (auto &sx : d->s_aux) HIPCHK(hipStreamCreateWithFlags(&sx, hipStreamNonBlocking));
    }
    HIPCHK(hipEventCreateWithFlags(&d->ev_start, hipEventDisableTiming));
    if (const char *ts = std::getenv("SBMBP_SHARD_STREAMS")) d->two_streams = std::atoi(ts) > 1;  // A/B: 1 = all chunks on the compute stream
    CHK(sbmbp_set_stream(d->eng, d->s_compute));
    CHK(sbmbp_shard_set_io(d->eng, P.snd_ptr.data(), P.snd_slot.data(), d->d_sendbuf, d->d_recvbuf[0], d->d_recvbuf[1], d->ncomp));
    d->ev_chunk.resize(P.n_chunks);
    d->ev_xchg.resize(P.n_chunks);
    for (u32 c = 0; c < P.n_chunks; ++c) {
        HIPCHK(hipEventCreateWithFlags(&d->ev_chunk[c], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&d->ev_xchg[c], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&d->ev_a, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&d->ev_b, hipEventDisableTiming));
    if (comm->kind == 1) {  // the in-process transport signals between the ranks' streams with events of this device
        local_group::slot &me = comm->grp->slots[comm->rank];
        if (!me.ready) HIPCHK(hipEventCreateWithFlags(&me.ready, hipEventDisableTiming));
        if (!me.done) HIPCHK(hipEventCreateWithFlags(&me.done, hipEventDisableTiming));
    }
    // the plan's big host arrays are on the device now
    shard_plan &Pm = d->plan;
    std::vector<u32>().swap(Pm.nbr_local);
    std::vector<u32>().swap(Pm.rev_local);
    std::vector<u32>().swap(Pm.msg_send_edge);
    std::vector<u32>().swap(Pm.snd_slot);
    std::vector<u32>().swap(Pm.table_deg);
    *out = d.release();
    return SBMBP_OK;
}

void sbmbp_dist_destroy(sbmbp_dist_t *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->s_comm) (void)hipStreamSynchronize(d->s_comm);
    for (auto sx : d->s_aux) if (sx) (void)hipStreamSynchronize(sx);
    if (d->s_compute) (void)hipStreamSynchronize(d->s_compute);
    if (d->eng) { (void)sbmbp_set_stream(d->eng, nullptr); sbmbp_destroy(d->eng); }
    void *ptrs[] = {d->d_psi[0], d->d_psi[1], d->d_red, d->d_sendbuf, d->d_recvbuf[0], d->d_recvbuf[1], d->d_msg_send, d->d_psi_all,
                    d->d_send_idx, d->d_stage_to_halo, d->d_msg_send_edge};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto ev : d->ev_chunk) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : d->ev_xchg) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : d->phase_ev) if (ev) (void)hipEventDestroy(ev);
    if (d->ev_a) (void)hipEventDestroy(d->ev_a);
    if (d->ev_b) (void)hipEventDestroy(d->ev_b);
    if (d->ev_start) (void)hipEventDestroy(d->ev_start);
    for (auto sx : d->s_aux) if (sx) (void)hipStreamDestroy(sx);
    if (d->s_comm) (void)hipStreamDestroy(d->s_comm);
    if (d->s_compute) (void)hipStreamDestroy(d->s_compute);
    delete d;
}

int sbmbp_dist_info(const sbmbp_dist_t *d, sbmbp_dist_info_t *out) {
    if (!d || !out) return arg_error(__func__, __LINE__);
    const shard_plan &P = d->plan;
    out->rank = d->rank;
    out->world = d->world;
    out->n_global = P.n_global;
    out->row0 = P.row0;
    out->n_own = P.n_own;
    out->n_halo = P.n_halo;
    out->n_chunks = P.n_chunks;
    out->halo_components = d->ncomp;
    out->n_edges = P.n_edges;
    out->e2_global = P.e2_global;
    out->n_halo_msgs = P.n_halo_msgs;
    u64 tot = 0, mx = 0;
    for (u64 x : P.send_counts) { tot += x; mx = std::max(mx, x); }
    out->sent_rows_per_sweep = tot;
    out->busiest_peer_rows = mx;
    return SBMBP_OK;
}

int sbmbp_dist_peer_rows(const sbmbp_dist_t *d, uint64_t *send_rows, uint64_t *recv_rows) {
    if (!d) return arg_error(__func__, __LINE__);
    if (send_rows) std::copy(d->plan.send_counts.begin(), d->plan.send_counts.end(), send_rows);
    if (recv_rows) std::copy(d->plan.recv_counts.begin(), d->plan.recv_counts.end(), recv_rows);
    return SBMBP_OK;
}

// init_messages (bp.cpp:101-217): the state of the WHOLE graph is drawn from the reference's std::mt19937 stream by every
// rank (same seed, same order) and each keeps the rows and out-messages it owns
int sbmbp_dist_init_messages(sbmbp_dist_t *d, uint32_t flag, const int32_t *conf, const uint32_t *true_conf, uint32_t seed,
                             int conditional) {
    if (!d || !true_conf) return arg_error(__func__, __LINE__);
    const sbmbp_graph_t *g = d->graph;
    if (flag >= 4) { set_error("bp_messages_init_flag must be < 4"); return SBMBP_ERR_ARG; }  // assert at bp.cpp:106
    if (flag != 0 && !conf) { set_error("init flag != 0 needs a conf vector"); return SBMBP_ERR_ARG; }
    device_guard guard(d->device);
    const shard_plan &P = d->plan;
    const u32 Q = d->Q;
    std::vector<u32> rp32(size_t(g->n) + 1);
    for (size_t i = 0; i <= g->n; ++i) rp32[i] = u32(g->row_ptr[i]);
    std::vector<double> psi_own(size_t(P.n_own) * Q), msg_own(size_t(P.n_edges) * Q);
    const u32 lo = P.row0, hi = P.row0 + P.n_own;
    sbmbp::state_sink sink;
    sink.put = [&](u32 a, u32 b, const double *prow, const double *mrow) {
        const u32 x = std::max(a, lo), y = std::min(b, hi);
        if (x >= y) return;
        std::memcpy(psi_own.data() + size_t(x - lo) * Q, prow + size_t(x - a) * Q, size_t(y - x) * Q * 8);
        const u64 k0 = rp32[x], k1 = rp32[y], ka = rp32[a];
        if (k1 > k0) std::memcpy(msg_own.data() + (k0 - P.edge0) * Q, mrow + (k0 - ka) * Q, (k1 - k0) * Q * 8);
    };
    sbmbp::init_state_host(g->n, rp32.data(), g->e2(), Q, flag, conf, seed, nullptr, nullptr, &sink);
    CHK(sbmbp_set_state(d->eng, psi_own.data(), msg_own.data()));
    bool any_clamp = false;
    if (conditional && flag != 0 && conf)
        for (u32 i = 0; i < g->n; ++i) {
            if (conf[i] < -1 || conf[i] >= int32_t(Q)) { set_error("conf entry out of range"); return SBMBP_ERR_ARG; }
            if (conf[i] != -1) any_clamp = true;
        }
    for (u32 i = 0; i < g->n; ++i) if (true_conf[i] >= Q) { set_error("true_conf entry out of range"); return SBMBP_ERR_ARG; }
    CHK(sbmbp_shard_set_labels(d->eng, conf ? conf + lo : nullptr, true_conf + lo, flag, conditional, any_clamp ? 1 : 0));
    d->have_state = true;
    d->consistent = false;
    return SBMBP_OK;
}

int sbmbp_dist_init_messages_device(sbmbp_dist_t *d, uint64_t seed, const uint32_t *true_conf) {
    if (!d || !true_conf) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    CHK(sbmbp_init_messages_device(d->eng, seed, true_conf + d->plan.row0));
    d->have_state = true;
    d->consistent = false;
    return SBMBP_OK;
}

int sbmbp_dist_set_state(sbmbp_dist_t *d, const double *psi_own, const double *msg_own) {
    if (!d) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    CHK(sbmbp_set_state(d->eng, psi_own, msg_own));
    if (psi_own && (msg_own || d->plan.n_edges == 0)) d->have_state = true;
    d->consistent = false;
    return SBMBP_OK;
}

int sbmbp_dist_get_state(sbmbp_dist_t *d, double *psi_own, double *msg_own) {
    if (!d) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    HIPCHK(hipStreamSynchronize(d->s_comm));
    return sbmbp_get_state(d->eng, psi_own, msg_own);
}

int sbmbp_dist_gather_marginals(sbmbp_dist_t *d, double *psi_all) {
    if (!d || !psi_all) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    const u32 N = d->plan.n_global, Q = d->Q;
    double *tab = nullptr;
    CHK(dalloc(&tab, size_t(N) * Q));
    hipError_t err = hipMemsetAsync(tab, 0, size_t(N) * Q * 8, d->s_compute);
    if (err == hipSuccess)
        err = hipMemcpyAsync(tab + size_t(d->plan.row0) * Q, d->d_psi[sbmbp_shard_read_buffer(d->eng, 0)], size_t(d->plan.n_own) * Q * 8,
                             hipMemcpyDeviceToDevice, d->s_compute);
    int rc = err == hipSuccess ? comm_allreduce(d->comm, tab, size_t(N) * Q, 0, d->s_compute) : SBMBP_ERR_HIP;
    if (rc == SBMBP_OK) {
        err = hipMemcpyAsync(psi_all, tab, size_t(N) * Q * 8, hipMemcpyDeviceToHost, d->s_compute);
        if (err == hipSuccess) err = hipStreamSynchronize(d->s_compute);
        if (err != hipSuccess) rc = SBMBP_ERR_HIP;
    }
    (void)hipFree(tab);
    return rc;
}

int sbmbp_dist_set_params(sbmbp_dist_t *d, const double *cab, const uint32_t *na, double beta) {
    if (!d || !cab || !na) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    return apply_params(d, cab, na, beta);
}

int sbmbp_dist_get_params(sbmbp_dist_t *d, double *cab, uint32_t *na) {
    if (!d || !d->have_params) return SBMBP_ERR_STATE;
    if (cab) std::copy(d->cab.begin(), d->cab.end(), cab);
    if (na) std::copy(d->na.begin(), d->na.end(), na);
    return SBMBP_OK;
}

int sbmbp_dist_set_schedule(sbmbp_dist_t *d, double field_mix, uint32_t check_every) {
    if (!d || !(field_mix > 0.0) || field_mix > 1.0 || check_every < 1) return arg_error(__func__, __LINE__);
    d->field_mix = field_mix;
    d->check_every = check_every;
    return SBMBP_OK;
}

int sbmbp_dist_set_learning_schedule(sbmbp_dist_t *d, double field_mix, double snap) {
    if (!d || !(field_mix > 0.0) || field_mix > 1.0 || !(snap >= 0.0)) return arg_error(__func__, __LINE__);
    d->learn_field_mix = field_mix;
    d->learn_snap = snap;
    return SBMBP_OK;
}

int sbmbp_dist_set_gather_mode(sbmbp_dist_t *d, int mode) {
    if (!d || mode < 0 || mode > 1) return arg_error(__func__, __LINE__);
    d->gather_mode = mode;
    return SBMBP_OK;
}

int sbmbp_dist_set_auto_relax(sbmbp_dist_t *d, int on) {
    if (!d) return arg_error(__func__, __LINE__);
    return sbmbp_set_auto_relax(d->eng, on);
}
int sbmbp_dist_get_relaxation(const sbmbp_dist_t *d, int *field_level, int *generic_level) {
    if (!d) return arg_error(__func__, __LINE__);
    return sbmbp_get_relaxation(d->eng, field_level, generic_level, nullptr, nullptr);
}

int sbmbp_dist_converge(sbmbp_dist_t *d, double crit, uint32_t max_sweeps, double damping, int *niter, double *last) {
    if (!d) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    return run(d, crit, max_sweeps, damping, niter, last);
}

int sbmbp_dist_sweep(sbmbp_dist_t *d, double damping, uint32_t n_sweeps, double *last) {
    if (!d) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    const u32 keep = d->check_every;
    d->check_every = std::max<u32>(keep, 64);  // no convergence test: sync rarely
    const int r = run(d, -1.0, n_sweeps, damping, nullptr, last);
    d->check_every = keep;
    return r;
}

int sbmbp_dist_free_energy(sbmbp_dist_t *d, double *f, double *parts) {
    if (!d) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    double t[6];
    CHK(fe_terms(d, false, t));
    if (parts) { parts[0] = t[0]; parts[1] = t[1]; parts[2] = t[2]; }
    if (f) *f = -t[0] + t[1] + t[2];
    return SBMBP_OK;
}

int sbmbp_dist_entropy(sbmbp_dist_t *d, double *ent, double *parts) {
    if (!d) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    if (d->dc != 0) {  // the reference evaluates 0/0 in e_site for deg_corr_flag != 0 (bp.cpp:550-556)
        const double nan = std::nan("");
        if (parts) { parts[0] = nan; parts[1] = nan; parts[2] = 0.0; }
        if (ent) *ent = nan;
        return SBMBP_OK;
    }
    double t[6];
    CHK(fe_terms(d, true, t));
    if (parts) { parts[0] = t[3]; parts[1] = t[4]; parts[2] = t[5]; }
    if (ent) *ent = -t[3] + t[4] - t[5];
    return SBMBP_OK;
}

int sbmbp_dist_em_expectations(sbmbp_dist_t *d, double *na_e, double *nna_e, double *cab_e) {
    if (!d) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    return em_expect(d, na_e, nna_e, cab_e);
}

int sbmbp_dist_confusion(sbmbp_dist_t *d, double *C) {
    if (!d || !C) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    return overlap_of(d, nullptr, C);
}

int sbmbp_dist_overlap(sbmbp_dist_t *d, double *ov) {
    if (!d || !ov) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    return overlap_of(d, ov, nullptr);
}

int sbmbp_dist_inference(sbmbp_dist_t *d, float conv_crit, uint32_t time_conv, float dumping_rate, sbmbp_infer_result *out) {
    if (!d || !out) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    // belief_propagation::inference (bp.cpp:77-99); crit and damping arrive as float, compared as double (:406)
    CHK(run(d, double(conv_crit), time_conv, double(dumping_rate), &out->niter, &out->last_maxdiff));
    double t[6];
    CHK(fe_terms(d, d->dc == 0, t));
    out->free_energy = -t[0] + t[1] + t[2];
    out->entropy = d->dc == 0 ? -t[3] + t[4] - t[5] : std::nan("");
    return overlap_of(d, &out->overlap, nullptr);
}

int sbmbp_dist_learning(sbmbp_dist_t *d, float learning_conv_crit, uint32_t learning_max_time, float learning_rate, float dumping_rate,
                        sbmbp_learn_result *out) {
    if (!d || !out) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    if (!d->have_params || !d->have_state) { set_error("set_params and an initial state must precede learning"); return SBMBP_ERR_STATE; }
    const u32 Q = d->Q, N = d->plan.n_global;
    std::vector<double> na_e(Q), nna_e(Q), cab_e(size_t(Q) * Q);
    double fold = 0.0, fdiff = 1.0;
    // the two rules of the synchronous EM loop (sbmbp_set_learning_schedule), as sbmbp_learning
    const double keep_mix = d->field_mix;
    d->field_mix = std::min(d->field_mix, d->learn_field_mix);
    struct restore { sbmbp_dist *d; double v; ~restore() { d->field_mix = v; } } restore_mix{d, keep_mix};
    out->em_steps = 0;
    out->status = 0;
    const u64 sweeps0 = d->total_sweeps;
    for (u32 t = 0; t < learning_max_time; ++t) {  // belief_propagation::learning (bp.cpp:27-47)
        if (fdiff < learning_conv_crit) learning_conv_crit = float(double(learning_conv_crit) * 0.1);
        int niter;
        double last;
        CHK(run(d, double(learning_conv_crit), learning_max_time, double(dumping_rate), &niter, &last));
        CHK(em_expect(d, na_e.data(), nna_e.data(), cab_e.data()));
        double tt[6];
        CHK(fe_terms(d, false, tt));
        const double fnew = -tt[0] + tt[1] + tt[2];
        fdiff = std::fabs(fnew - fold);
        fold = fnew;
        if (std::isnan(fold) || std::isinf(fold)) { out->status = 2; break; }
        if (fdiff < learning_conv_crit) { out->status = 1; break; }
        std::vector<u32> na(d->na);  // learning_step (bp.cpp:53-75)
        u32 rest = N;
        const double snap = std::min(d->learn_snap * double(N) * double(learning_conv_crit), 0.01);
        for (u32 i = 0; i + 1 < Q; ++i) {
            na[i] = unsigned(int(learning_rate * na_e[i] + (1.0 - learning_rate) * na[i] + snap));
            rest -= na[i];
        }
        na[Q - 1] = rest;
        std::vector<double> cab(d->cab);
        for (size_t a = 0; a < size_t(Q) * Q; ++a) cab[a] = learning_rate * cab_e[a] + (1.0 - learning_rate) * cab[a];
        CHK(apply_params(d, cab.data(), na.data(), d->beta));
        out->em_steps++;
    }
    out->free_energy = fold;
    out->total_sweeps = d->total_sweeps - sweeps0;
    return overlap_of(d, &out->overlap, nullptr);
}

int sbmbp_dist_get_stats(sbmbp_dist_t *d, sbmbp_stats *out) {
    if (!d || !out) return arg_error(__func__, __LINE__);
    device_guard guard(d->device);
    CHK(sbmbp_get_stats(d->eng, out));
    out->sweeps = d->total_sweeps;
    out->edge_msg_updates = d->total_sweeps * d->plan.e2_global;
    out->psi_form_sweeps = d->psi_sweeps;
    return SBMBP_OK;
}

int sbmbp_dist_reset_stats(sbmbp_dist_t *d) {
    if (!d) return arg_error(__func__, __LINE__);
    d->total_sweeps = d->psi_sweeps = 0;
    d->phase_ms[0] = d->phase_ms[1] = d->phase_ms[2] = 0.0;
    d->phase_n = 0;
    return sbmbp_reset_stats(d->eng);
}

int sbmbp_dist_set_timing(sbmbp_dist_t *d, int on) {
    if (!d) return arg_error(__func__, __LINE__);
    d->timing = on != 0;
    return sbmbp_set_timing(d->eng, on);
}

int sbmbp_dist_phase_times(sbmbp_dist_t *d, double ms_per_sweep[3], uint64_t *n_sweeps) {
    if (!d || !ms_per_sweep) return arg_error(__func__, __LINE__);
    const double n = d->phase_n ? double(d->phase_n) : 1.0;
    for (int i = 0; i < 3; ++i) ms_per_sweep[i] = d->phase_ms[i] / n;
    if (n_sweeps) *n_sweeps = d->phase_n;
    return SBMBP_OK;
}

// plan of rank `rank` of `world` without a device (tests, dry runs): counts only
int sbmbp_plan_summary(const sbmbp_graph_t *g, int world, int rank, uint32_t n_chunks, sbmbp_dist_info_t *info, uint64_t *send_rows,
                       uint64_t *recv_rows, uint64_t *msg_rows) {
    if (!g || !info) return arg_error(__func__, __LINE__);
    shard_plan P;
    CHK(build_plan(*g, world, rank, n_chunks ? n_chunks : 1, P));
    info->rank = rank;
    info->world = world;
    info->n_global = P.n_global;
    info->row0 = P.row0;
    info->n_own = P.n_own;
    info->n_halo = P.n_halo;
    info->n_chunks = P.n_chunks;
    info->halo_components = 0;
    info->n_edges = P.n_edges;
    info->e2_global = P.e2_global;
    info->n_halo_msgs = P.n_halo_msgs;
    u64 tot = 0, mx = 0;
    for (u64 x : P.send_counts) { tot += x; mx = std::max(mx, x); }
    info->sent_rows_per_sweep = tot;
    info->busiest_peer_rows = mx;
    if (send_rows) std::copy(P.send_counts.begin(), P.send_counts.end(), send_rows);
    if (recv_rows) std::copy(P.recv_counts.begin(), P.recv_counts.end(), recv_rows);
    if (msg_rows) std::copy(P.msg_counts.begin(), P.msg_counts.end(), msg_rows);
    return SBMBP_OK;
}

// the whole plan of one rank, for tests against the Python restatement (arrays may be NULL; sizes from sbmbp_plan_summary)
int sbmbp_plan_arrays(const sbmbp_graph_t *g, int world, int rank, uint32_t n_chunks, uint32_t *nbr_local, uint32_t *halo_global,
                      uint32_t *chunk_row, uint64_t *send_counts_cp, uint64_t *recv_counts_cp, uint32_t *send_idx_chunked,
                      uint32_t *snd_ptr, uint32_t *snd_slot, uint32_t *rev_local, uint32_t *msg_send_edge) {
    if (!g) return arg_error(__func__, __LINE__);
    shard_plan P;
    CHK(build_plan(*g, world, rank, n_chunks ? n_chunks : 1, P));
    auto cp = [](auto &v, auto *dst) { if (dst) std::copy(v.begin(), v.end(), dst); };
    cp(P.nbr_local, nbr_local);
    cp(P.halo_global, halo_global);
    cp(P.chunk_row, chunk_row);
    cp(P.send_counts_cp, send_counts_cp);
    cp(P.recv_counts_cp, recv_counts_cp);
    cp(P.send_idx_chunked, send_idx_chunked);
    cp(P.snd_ptr, snd_ptr);
    cp(P.snd_slot, snd_slot);
    cp(P.rev_local, rev_local);
    cp(P.msg_send_edge, msg_send_edge);
    return SBMBP_OK;
}

}  // extern "C"
