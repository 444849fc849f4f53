// bin/bp — drop-in command line of the reference (src/main.cpp) on the MI355X engine.
// Same long/short flags, multitoken semantics, stdout/stderr lines and return codes as
// main.cpp:85-365, without Boost; all BP work goes through the C ABI of include/sbmbp.h.
// Host code is C++14. Flags the reference parses but never reads (main.cpp:126-135)
// are accepted and ignored. Extensions (default off, stdout unchanged): --precision, --device,
// --gather, --field_mix, --check_every, --metrics_json, and --gpus N: the graph is sharded by vertex range over N GPUs of
// this node, one host thread per GPU driving the C++ multi-GPU driver (sbmbp_dist_*, RCCL over xGMI); with fewer devices
// than ranks the ranks share devices over the in-process transport (a rehearsal, not a speed-up).
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/sbmbp.h"

namespace {

struct opt_spec { const char *lng; char sht; int arity; /* 0 switch, 1 single, 2 multitoken */ };
const opt_spec OPTS[] = {
    {"edge_list_path", 'l', 1}, {"n", 'n', 2}, {"beta", 'b', 1}, {"mb_rand", 0, 0}, {"mb_n", 0, 0}, {"mb", 0, 2},
    {"mb_path", 0, 1}, {"epsilon_c", 0, 2}, {"bp_messages_init_flag", 'i', 1}, {"beliefs_path", 0, 1},
    {"true_conf_path", 0, 1}, {"deg_corr_flag", 0, 1}, {"learning_rate", 'r', 1}, {"dumping_rate", 'R', 1},
    {"bp_conv_crit", 'e', 1}, {"learning_conv_crit", 'E', 1}, {"time_conv", 't', 1}, {"probabilities", 'P', 2},
    {"fixed_nodes", 'f', 2}, {"cab_rand", 0, 0}, {"cab_ppm", 0, 0}, {"cab_ec", 0, 0}, {"cab_file", 0, 1},
    {"pa", 0, 2}, {"cab", 0, 2}, {"if_output_marginals", 0, 0}, {"mode", 'm', 1}, {"seed", 'd', 1}, {"help", 'h', 0},
    // extensions
    {"precision", 0, 1}, {"device", 0, 1}, {"gather", 0, 1}, {"field_mix", 0, 1}, {"check_every", 0, 1}, {"metrics_json", 0, 1},
    {"gpus", 0, 1}, {"transport", 0, 1},
};

const opt_spec *find_long(const std::string &name) {
    for (const auto &o : OPTS) if (name == o.lng) return &o;
    return nullptr;
}
const opt_spec *find_short(char c) {
    for (const auto &o : OPTS) if (o.sht && o.sht == c) return &o;
    return nullptr;
}
bool looks_numeric(const std::string &s) {
    if (s.empty()) return false;
    char *end = nullptr;
    std::strtod(s.c_str(), &end);
    return end && *end == '\0';
}

struct cmdline {
    std::map<std::string, std::vector<std::string>> v;
    std::string error;
    int count(const std::string &k) const { return v.count(k) ? 1 : 0; }
    const std::vector<std::string> &get(const std::string &k) const { static std::vector<std::string> none; auto it = v.find(k); return it == v.end() ? none : it->second; }
};

cmdline parse(int argc, char const *argv[]) {
    cmdline c;
    int i = 1;
    while (i < argc) {
        std::string tok = argv[i++];
        const opt_spec *o = nullptr;
        std::string inline_val;
        bool has_inline = false;
        if (tok.size() > 2 && tok[0] == '-' && tok[1] == '-') {
            std::string name = tok.substr(2);
            auto eq = name.find('=');
            if (eq != std::string::npos) { inline_val = name.substr(eq + 1); name = name.substr(0, eq); has_inline = true; }
            o = find_long(name);
            if (!o) { c.error = "unrecognised option '" + tok + "'"; return c; }
        } else if (tok.size() >= 2 && tok[0] == '-' && !looks_numeric(tok)) {
            o = find_short(tok[1]);
            if (!o) { c.error = "unrecognised option '" + tok + "'"; return c; }
            if (tok.size() > 2) { inline_val = tok.substr(2); has_inline = true; }
        } else {
            c.error = "too many positional options have been specified on the command line";
            return c;
        }
        auto &dst = c.v[o->lng];
        if (o->arity == 0) {
            if (has_inline) { c.error = std::string("option '--") + o->lng + "' does not take any arguments"; return c; }
            continue;
        }
        if (has_inline) dst.push_back(inline_val);
        if (o->arity == 1) {
            if (!has_inline) {
                if (i >= argc) { c.error = std::string("the required argument for option '--") + o->lng + "' is missing"; return c; }
                dst.push_back(argv[i++]);
            }
        } else {  // multitoken: consume until the next option (tokens that parse as numbers are values)
            while (i < argc) {
                std::string nx = argv[i];
                if (nx.size() >= 2 && nx[0] == '-' && !looks_numeric(nx)) break;
                dst.push_back(nx);
                ++i;
            }
            if (dst.empty()) { c.error = std::string("the required argument for option '--") + o->lng + "' is missing"; return c; }
        }
    }
    return c;
}

template <typename T> bool to_vec(const std::vector<std::string> &in, std::vector<T> &out) {
    out.clear();
    for (const auto &s : in) {
        char *end = nullptr;
        double d = std::strtod(s.c_str(), &end);
        if (!end || *end != '\0') return false;
        out.push_back(T(d));
    }
    return true;
}

void usage(const char *argv0) {
    std::clog << "BP algorithms for the SBM (final output only)\n";
    std::clog << "Usage:\n  " << argv0 << " [--option_1=value] [--option_s2=value] ...\n";
    std::clog << "Options:\n"
                 "  -l [ --edge_list_path ] arg           Path to the input edgelist file.\n"
                 "  -n [ --n ] arg                        Block sizes vector.\n"
                 "  -b [ --beta ] arg (=1)                beta, the inverse temperature\n"
                 "  --mb_rand                             Randomize initial block memberships.\n"
                 "  --mb_n                                Initialize membership from n [DEFAULT].\n"
                 "  --mb arg                              Directly initialize membership from input vector.\n"
                 "  --mb_path arg                         use an external file to define the memberships.\n"
                 "  --epsilon_c arg                       Assign epsilon and c to define cab and pa [DEFAULT].\n"
                 "  -i [ --bp_messages_init_flag ] arg (=0) flag to initialize BP: 0 random, 1 partly planted,\n"
                 "                                        2 planted with noise, 3 fixed planted.\n"
                 "  --beliefs_path arg                    Path to planted membership.\n"
                 "  --true_conf_path arg                  Path to true membership.\n"
                 "  --deg_corr_flag arg (=0)              0 no degree correction, 1 degree correction, 2 variant.\n"
                 "  -r [ --learning_rate ] arg (=0.2)     learning_rate, from 0.0 to 1.0.\n"
                 "  -R [ --dumping_rate ] arg (=1)        dumping_rate, from 0.0 to 1.0 (1 = no dumping).\n"
                 "  -e [ --bp_conv_crit ] arg (=5e-06)    convergence criterium of BP.\n"
                 "  -E [ --learning_conv_crit ] arg (=1e-06) convergence criterium of learning.\n"
                 "  -t [ --time_conv ] arg (=100)         maximum time for BP to converge.\n"
                 "  -P [ --probabilities ] arg            (accepted, unused)\n"
                 "  -f [ --fixed_nodes ] arg              Fixed nodes with known labels.\n"
                 "  --cab_rand --cab_ppm --cab_ec --cab_file arg   (accepted, unused)\n"
                 "  --pa arg                              pa vector.\n"
                 "  --cab arg                             cab vector (upper triangle, row-major).\n"
                 "  --if_output_marginals                 whether output marginals in the infer mode\n"
                 "  -m [ --mode ] arg                     Mode for the algorithm; valid values: infer | learn.\n"
                 "  -d [ --seed ] arg                     Seed of the pseudo random number generator (mt19937).\n"
                 "  -h [ --help ]                         Produce this help message.\n"
                 "MI355X engine extensions:\n"
                 "  --precision arg (=6)  --device arg (=0)  --gather auto|messages  --field_mix arg (=1)\n"
                 "  --check_every arg (=8)  --metrics_json path\n"
                 "  --gpus arg (=1)       shard the graph by vertex range over this many GPUs (one host thread each, RCCL)\n"
                 "  --transport rccl|local  (default rccl; local = ranks may share devices, rehearsal)\n";
}

bool read_column(const std::string &path, std::vector<long long> &out) {  // load_beliefs/load_confs (graph_utilities.cpp:8-40)
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    std::string line;
    long long v = 0;
    while (std::getline(f, line)) {
        std::stringstream ls(line);
        ls >> v;  // a malformed line keeps the previous value, as the reference's stream read does
        out.push_back(v);
    }
    return true;
}

double signed_nan_like_reference(double x) { return std::isnan(x) ? -std::fabs(x) : x; }  // the reference prints "-nan" (0/0 on x86)

int fail(int code) {
    std::clog << "bp: " << sbmbp_strerror(code) << ": " << sbmbp_last_error() << "\n";
    return 1;
}

}  // namespace

int main(int argc, char const *argv[]) {
    // the host driver of the target pool only supports dmabuf IPC (RCCL between devices fails without it); an exported value wins
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    cmdline var_map = parse(argc, argv);
    if (!var_map.error.empty()) {  // boost::program_options throws: uncaught -> terminate; we report and fail
        std::clog << "bp: " << var_map.error << "\n";
        return 1;
    }
    if (var_map.count("help") > 0 || argc == 1) {  // main.cpp:154-160
        usage(argv[0]);
        return 0;
    }
    if (var_map.count("edge_list_path") == 0) { std::clog << "edge_list_path is required (-e flag)\n"; return 1; }  // :162-165
    if (var_map.count("mode") == 0) { std::clog << "mode is required (-m flag)\n"; return 1; }                      // :167-170
    if (var_map.count("n") == 0) { std::clog << "n is required (-n flag)\n"; return 1; }                            // :172-175

    std::string memberships_status;
    if (var_map.count("mb_n") + var_map.count("mb") + var_map.count("mb_path") > 1) {  // :178-184
        std::clog << "Error! Please just select one option to assign the membership vector.\n";
        return 1;
    } else if (var_map.count("mb_n") + var_map.count("mb") + var_map.count("mb_path") == 0) {
        memberships_status = "from_n";
    }
    if (var_map.count("mb_rand") > 0) { /* shuffles a membership vector BP never reads */ }
    else if (var_map.count("mb_n") > 0) memberships_status = "from_n";
    else if (var_map.count("mb") > 0) memberships_status = "direct";
    else if (var_map.count("mb_path") > 0) memberships_status = "from_file";

    std::string bm_params_string;
    if (var_map.count("epsilon_c") + (var_map.count("pa") * var_map.count("cab")) > 1) {  // :196-206
        std::clog << "Error! Please just choose one way to initialize the pa/cab parameter.\n";
        return 1;
    } else if (var_map.count("epsilon_c") == 0 && (var_map.count("pa") + var_map.count("cab")) < 2) {
        std::clog << "Error! Please just input both pa/cab parameters.\n";
        return 1;
    } else if (var_map.count("epsilon_c") > 0) {
        bm_params_string = "cab_ec";
    } else {
        bm_params_string = "cab_direct";
    }

    auto num = [&](const char *k, double dflt) { return var_map.count(k) ? std::strtod(var_map.get(k)[0].c_str(), nullptr) : dflt; };
    unsigned bp_messages_init_flag = unsigned(num("bp_messages_init_flag", 0));
    if (bp_messages_init_flag != 0 && var_map.count("fixed_nodes") == 0) {  // :208-219
        if (var_map.count("beliefs_path") == 0) {
            std::clog << "Error! Please assign the file path of the initial belief of node membership.\n";
            return 1;
        }
    } else if (var_map.count("fixed_nodes") > 0) {
        std::clog << "Randomly assign initial messages, except certain fixed nodes.\n";
    } else {
        std::clog << "Randomly assign initial messages!\n";
    }
    const bool if_output_marginals = var_map.count("if_output_marginals") > 0;

    unsigned seed;
    if (var_map.count("seed") == 0) seed = (unsigned)std::chrono::high_resolution_clock::now().time_since_epoch().count();  // :230-233
    else seed = unsigned(num("seed", 0));

    std::vector<unsigned> n, mb, fixed_nodes;
    std::vector<double> epsilon_c, pa, cab;
    if (!to_vec(var_map.get("n"), n) || !to_vec(var_map.get("mb"), mb) || !to_vec(var_map.get("fixed_nodes"), fixed_nodes) ||
        !to_vec(var_map.get("epsilon_c"), epsilon_c) || !to_vec(var_map.get("pa"), pa) || !to_vec(var_map.get("cab"), cab)) {
        std::clog << "bp: the argument for a vector option is invalid\n";
        return 1;
    }
    const unsigned Q = unsigned(n.size());
    unsigned N = 0;
    for (auto x : n) N += x;
    std::vector<uint32_t> memberships_init;
    if (memberships_status == "from_n") {  // :240-252
        memberships_init.resize(N, 0);
        unsigned shift = 0;
        for (unsigned r = 0; r < Q; ++r) { for (unsigned i = 0; i < n[r]; ++i) memberships_init[shift + i] = r; shift += n[r]; }
    } else if (memberships_status == "direct") {  // :253-266
        if (mb.size() != N) { std::clog << "Error! Size of assigned membership vector does not fit the number of nodes assigned by n.\n"; return 1; }
        memberships_init.assign(mb.begin(), mb.end());
    }  // from_file: unimplemented in the reference (empty vector, :267-269)
    if (Q < 2 || Q > SBMBP_MAX_Q) { std::clog << "bp: the number of blocks must be between 2 and " << SBMBP_MAX_Q << "\n"; return 1; }
    if (bm_params_string == "cab_ec" && epsilon_c.size() != 2) { std::clog << "bp: --epsilon_c needs two values\n"; return 1; }
    if (bm_params_string == "cab_direct" && (pa.size() != Q || cab.size() != size_t(Q) * (Q + 1) / 2)) {
        std::clog << "bp: --pa needs Q values and --cab the Q(Q+1)/2 upper-triangle values\n";
        return 1;
    }

    const std::string mode = var_map.get("mode")[0];
    const unsigned deg_corr_flag = unsigned(num("deg_corr_flag", 0));
    const double beta = num("beta", 1.0);
    const float learning_rate = float(num("learning_rate", 0.2)), dumping_rate = float(num("dumping_rate", 1.0));
    const float bp_conv_crit = float(num("bp_conv_crit", 5.0e-6)), learning_conv_crit = float(num("learning_conv_crit", 1.0e-6));
    const unsigned time_conv = unsigned(num("time_conv", 100));

    // SBMBP_HOST_TIMING=1: wall time of each stage on stderr (measurement aid; stdout unchanged)
    const bool stage_timing = std::getenv("SBMBP_HOST_TIMING") != nullptr;
    auto stage_t = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (stage_timing) std::fprintf(stderr, "[sbmbp cli ] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - stage_t).count());
        stage_t = now;
    };

    // ---- graph (main.cpp:277-281) ---------------------------------------------------------------------
    sbmbp_graph_t *graph = nullptr;
    int rc = sbmbp_graph_load_edgelist(&graph, var_map.get("edge_list_path")[0].c_str(), N);
    if (rc != SBMBP_OK) return fail(rc);  // deviation: the reference continues with an empty graph
    if (sbmbp_graph_num_vertices(graph) != N) {
        std::clog << "bp: the edge list names vertex ids >= sum(n) = " << N << "\n";  // the reference indexes out of range here (B14)
        return 1;
    }

    std::vector<uint32_t> true_conf;  // :283-293
    if (var_map.count("true_conf_path") == 0) {
        std::clog << "Warning! Assign true conf using ordered node membership.\n";
        true_conf = memberships_init;
    } else {
        std::vector<long long> col;
        if (!read_column(var_map.get("true_conf_path")[0], col)) {
            std::clog << "Warning! Reading true_conf_path error. Assign true conf using ordered node membership.\n";
            true_conf = memberships_init;
        } else {
            true_conf.assign(col.begin(), col.end());
        }
    }
    if (true_conf.size() != N) { std::clog << "bp: the true configuration needs " << N << " entries\n"; return 1; }

    if (mode != "infer" && mode != "learn") return 0;  // the reference silently does nothing (:361-365)
    stage("read + index graph");

    std::vector<int32_t> beliefs;  // :325-336
    if (var_map.count("beliefs_path")) {
        std::vector<long long> col;
        if (read_column(var_map.get("beliefs_path")[0], col)) beliefs.assign(col.begin(), col.end());
    }
    if (var_map.count("fixed_nodes") > 0) {
        beliefs.resize(true_conf.size(), -1);
        for (auto vtx : fixed_nodes) if (vtx < beliefs.size()) beliefs[vtx] = int32_t(true_conf[vtx]);
    }
    if (bp_messages_init_flag != 0 && beliefs.size() != N) { std::clog << "bp: the beliefs vector needs " << N << " entries (-1 = unknown)\n"; return 1; }
    std::vector<double> cab_full(size_t(Q) * Q);
    std::vector<uint32_t> na(Q);
    if (bm_params_string == "cab_ec") rc = sbmbp_param_from_epsilon_c(N, Q, epsilon_c[0], epsilon_c[1], cab_full.data(), na.data());
    else rc = sbmbp_param_from_direct(N, Q, pa.data(), cab.data(), cab_full.data(), na.data());
    if (rc != SBMBP_OK) return fail(rc);
    std::cout << std::setprecision(int(num("precision", 6)));
    std::clog << std::setprecision(int(num("precision", 6)));

    const int n_gpus = int(num("gpus", 1));
    if (n_gpus > 1) {
        // ---- multi-GPU: one host thread per rank, all ranks make the same calls (include/sbmbp.h, "Multi-GPU") ----------
        const int n_dev = sbmbp_device_count();
        if (n_dev <= 0) return fail(SBMBP_ERR_NODEVICE);
        bool local = var_map.count("transport") && var_map.get("transport")[0] == "local";
        if (n_dev < n_gpus && !local) {
            std::clog << "bp: --gpus " << n_gpus << " on " << n_dev << " device(s): ranks share devices over the in-process transport (rehearsal)\n";
            local = true;
        }
        std::vector<sbmbp_comm_t *> comms(n_gpus, nullptr);
        unsigned char comm_id[SBMBP_COMM_ID_BYTES];
        if (local) rc = sbmbp_comm_init_local(comms.data(), n_gpus);
        else rc = sbmbp_comm_unique_id(comm_id);
        if (rc != SBMBP_OK) return fail(rc);
        std::vector<int> rcs(n_gpus, SBMBP_OK);
        std::vector<std::string> errs(n_gpus);
        sbmbp_infer_result ires{};
        sbmbp_learn_result lres{};
        std::vector<double> psi_all;
        std::vector<double> cab_out(cab_full);
        std::vector<uint32_t> na_out(na);
        sbmbp_stats st0{};
        const auto t0 = std::chrono::steady_clock::now();
        // A rank can fail OUTSIDE a collective (out of memory in sbmbp_dist_create, a bad initial state): on RCCL its peers
        // would queue collectives that spin on it and then sit in hipEventSynchronize for ever. Two guards: (1) the rank
        // threads AGREE on the set-up before the first collective is queued - a failed rank makes all of them leave; (2) a
        // rank that fails later aborts EVERY communicator of the run, not only its own (ncclCommAbort is safe from another
        // thread), so the kernels of the healthy ranks end and their host threads return an error.
        std::mutex mu;
        std::condition_variable cv;
        int arrived = 0, generation = 0;
        bool any_failed = false;
        auto agree = [&](bool ok) {  // all rank threads meet here; false when any of them has failed
            std::unique_lock<std::mutex> lk(mu);
            if (!ok) any_failed = true;
            const int gen = generation;
            if (++arrived == n_gpus) { arrived = 0; ++generation; cv.notify_all(); }
            else cv.wait(lk, [&] { return generation != gen; });
            return !any_failed;
        };
        auto abort_all = [&]() {
            std::lock_guard<std::mutex> lk(mu);
            for (auto *c : comms) if (c) sbmbp_comm_abort(c);
        };
        const char *inject = std::getenv("SBMBP_INJECT_RANK_FAIL");  // tests: "<rank>:setup" or "<rank>:run"
        auto rank_main = [&](int r) {
            const int dev = r % n_dev;
            sbmbp_dist_t *d = nullptr;
            int e = SBMBP_OK;
            auto step = [&](int code) { if (e == SBMBP_OK && code != SBMBP_OK) { e = code; errs[r] = sbmbp_last_error(); } return e == SBMBP_OK; };
            auto injected = [&](const char *where) {
                return inject && std::atoi(inject) == r && std::strstr(inject, where) != nullptr;
            };
            if (!local) {
                sbmbp_comm_t *c = nullptr;
                step(sbmbp_comm_init_rank(&c, comm_id, n_gpus, r, dev));
                std::lock_guard<std::mutex> lk(mu);
                comms[r] = c;
            }
            if (e == SBMBP_OK) step(sbmbp_dist_create(&d, comms[r], graph, Q, deg_corr_flag, dev, 0));
            if (e == SBMBP_OK && injected("setup")) { e = SBMBP_ERR_NOMEM; errs[r] = "injected failure (SBMBP_INJECT_RANK_FAIL)"; }
            if (e == SBMBP_OK) step(sbmbp_dist_init_messages(d, bp_messages_init_flag, beliefs.size() == N ? beliefs.data() : nullptr, true_conf.data(),
                                                             seed, mode == "learn" ? 0 : 1));
            if (e == SBMBP_OK) step(sbmbp_dist_set_params(d, cab_full.data(), na.data(), beta));
            if (e == SBMBP_OK) step(sbmbp_dist_set_schedule(d, num("field_mix", 1.0), unsigned(num("check_every", 8))));
            if (e == SBMBP_OK && var_map.count("gather") && var_map.get("gather")[0] == "messages") step(sbmbp_dist_set_gather_mode(d, 1));
            if (!agree(e == SBMBP_OK)) {  // some rank failed during set-up: nobody queues a collective
                if (e == SBMBP_OK) { e = SBMBP_ERR_COMM; errs[r] = "another rank failed during set-up"; }
                if (d) sbmbp_dist_destroy(d);
                rcs[r] = e;
                return;
            }
            if (injected("run")) { e = SBMBP_ERR_NOMEM; errs[r] = "injected failure (SBMBP_INJECT_RANK_FAIL)"; }
            if (e == SBMBP_OK && mode == "infer") {
                sbmbp_infer_result res{};
                if (step(sbmbp_dist_inference(d, bp_conv_crit, time_conv, dumping_rate, &res)) && r == 0) ires = res;
                if (e == SBMBP_OK && if_output_marginals) {
                    std::vector<double> all(size_t(N) * Q);
                    if (step(sbmbp_dist_gather_marginals(d, all.data())) && r == 0) psi_all.swap(all);
                }
            } else if (e == SBMBP_OK) {
                sbmbp_learn_result res{};
                if (step(sbmbp_dist_learning(d, learning_conv_crit, time_conv, learning_rate, dumping_rate, &res)) && r == 0) {
                    lres = res;
                    step(sbmbp_dist_get_params(d, cab_out.data(), na_out.data()));
                }
            }
            if (e == SBMBP_OK && r == 0) step(sbmbp_dist_get_stats(d, &st0));
            if (e != SBMBP_OK) abort_all();  // nobody waits for a rank that gave up: every communicator of the run ends
            if (d) sbmbp_dist_destroy(d);
            rcs[r] = e;
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < n_gpus; ++r) threads.emplace_back(rank_main, r);
        for (auto &t : threads) t.join();
        for (auto *c : comms) if (c) sbmbp_comm_destroy(c);
        for (int r = 0; r < n_gpus; ++r)
            if (rcs[r] != SBMBP_OK && rcs[r] != SBMBP_ERR_COMM) { std::clog << "bp: rank " << r << ": " << sbmbp_strerror(rcs[r]) << ": " << errs[r] << "\n"; return 1; }
        for (int r = 0; r < n_gpus; ++r)
            if (rcs[r] != SBMBP_OK) { std::clog << "bp: rank " << r << ": " << sbmbp_strerror(rcs[r]) << ": " << errs[r] << "\n"; return 1; }
        if (mode == "infer") {
            std::cout << signed_nan_like_reference(ires.entropy) << " " << signed_nan_like_reference(ires.free_energy) << " " << ires.overlap << " "
                      << ires.niter << " \n";
            if (if_output_marginals) {
                for (unsigned v = 0; v < N; ++v) {
                    for (unsigned q = 0; q < Q; ++q) std::cout << psi_all[size_t(v) * Q + q] << " ";
                    std::cout << "\n";
                }
                for (unsigned v = 0; v < N; ++v) {
                    double h = 0.0;
                    for (unsigned q = 0; q < Q; ++q) { const double p = psi_all[size_t(v) * Q + q]; if (p > 0) h -= p * std::log(p); }
                    std::clog << "Node-" << v << "; margEntropy H(v) is " << h << "\n";
                }
            }
        } else {
            if (lres.status == 2) std::clog << "Bethe energy is calculated as nan.\n";
            if (lres.status == 1) std::clog << "Algorithm stop because of fdiff < learning_conv_crit. [which is good]\n";
            for (unsigned q = 0; q < Q; ++q) std::cout << double(na_out[q]) / N << " ";
            std::cout << "\n";
            for (unsigned r = 0; r < Q; ++r) {
                for (unsigned s2 = 0; s2 < Q; ++s2) std::cout << cab_out[size_t(r) * Q + s2] << " ";
                std::cout << "\n";
            }
            std::clog << "overlap:" << lres.overlap << "\n";
        }
        stage(mode == "infer" ? "inference (all ranks)" : "learning (all ranks)");
        if (var_map.count("metrics_json")) {
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::ofstream mj(var_map.get("metrics_json")[0].c_str());
            mj << std::setprecision(12) << "{\"gpus\":" << n_gpus << ",\"transport\":\"" << (local ? "local" : "rccl") << "\",\"sweeps\":" << st0.sweeps
               << ",\"edge_msg_updates\":" << st0.edge_msg_updates << ",\"marginal_gather_sweeps\":" << st0.psi_form_sweeps
               << ",\"run_seconds\":" << secs << "}\n";
        }
        sbmbp_graph_destroy(graph);
        return 0;
    }

    sbmbp_engine_t *eng = nullptr;
    rc = sbmbp_create(&eng, graph, Q, deg_corr_flag, int(num("device", 0)));
    if (rc != SBMBP_OK) return fail(rc);
    stage("create engine (device)");

    rc = sbmbp_init_messages(eng, bp_messages_init_flag, beliefs.size() == N ? beliefs.data() : nullptr, true_conf.data(), seed,
                             mode == "learn" ? 0 : 1);  // bp_basic for learn, bp_conditional otherwise (:318-323)
    if (rc != SBMBP_OK) return fail(rc);
    stage("initial state + upload");

    if ((rc = sbmbp_set_params(eng, cab_full.data(), na.data(), beta)) != SBMBP_OK) return fail(rc);
    if ((rc = sbmbp_set_schedule(eng, num("field_mix", 1.0), unsigned(num("check_every", 8)))) != SBMBP_OK) return fail(rc);
    if (var_map.count("gather") && var_map.get("gather")[0] == "messages") sbmbp_set_gather_mode(eng, 1);

    const auto t0 = std::chrono::steady_clock::now();
    if (mode == "infer") {  // belief_propagation::inference (bp.cpp:77-99)
        sbmbp_infer_result res;
        if ((rc = sbmbp_inference(eng, bp_conv_crit, time_conv, dumping_rate, &res)) != SBMBP_OK) return fail(rc);
        std::cout << signed_nan_like_reference(res.entropy) << " " << signed_nan_like_reference(res.free_energy) << " " << res.overlap << " "
                  << res.niter << " \n";
        if (if_output_marginals) {
            std::vector<double> psi(size_t(N) * Q);
            if ((rc = sbmbp_get_state(eng, psi.data(), nullptr)) != SBMBP_OK) return fail(rc);
            for (unsigned v = 0; v < N; ++v) {  // output_mat (output_functions.h:7-15)
                for (unsigned q = 0; q < Q; ++q) std::cout << psi[size_t(v) * Q + q] << " ";
                std::cout << "\n";
            }
            for (unsigned v = 0; v < N; ++v) {  // bp.cpp:95-97 with entropy() of :4-12
                double h = 0.0;
                for (unsigned q = 0; q < Q; ++q) { const double p = psi[size_t(v) * Q + q]; if (p > 0) h -= p * std::log(p); }
                std::clog << "Node-" << v << "; margEntropy H(v) is " << h << "\n";
            }
        }
    } else {  // belief_propagation::learning (bp.cpp:14-51)
        sbmbp_learn_result res;
        if ((rc = sbmbp_learning(eng, learning_conv_crit, time_conv, learning_rate, dumping_rate, &res)) != SBMBP_OK) return fail(rc);
        if (res.status == 2) std::clog << "Bethe energy is calculated as nan.\n";
        if (res.status == 1) std::clog << "Algorithm stop because of fdiff < learning_conv_crit. [which is good]\n";
        if ((rc = sbmbp_get_params(eng, cab_full.data(), na.data())) != SBMBP_OK) return fail(rc);
        for (unsigned q = 0; q < Q; ++q) std::cout << double(na[q]) / N << " ";  // output_vec(eta_)
        std::cout << "\n";
        for (unsigned r = 0; r < Q; ++r) {  // output_mat(cab_)
            for (unsigned s = 0; s < Q; ++s) std::cout << cab_full[size_t(r) * Q + s] << " ";
            std::cout << "\n";
        }
        std::clog << "overlap:" << res.overlap << "\n";
    }
    stage(mode == "infer" ? "inference" : "learning");
    if (var_map.count("metrics_json")) {
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        sbmbp_stats st;
        sbmbp_get_stats(eng, &st);
        std::ofstream mj(var_map.get("metrics_json")[0].c_str());
        // where the adaptive relaxation of the synchronous schedule ended: [0, -1, mix, 1] = plain sweeps
        int ar_f = 0, ar_g = -1;
        double ar_mix = 1.0, ar_damp = 1.0;
        (void)sbmbp_get_relaxation(eng, &ar_f, &ar_g, &ar_mix, &ar_damp);
        mj << std::setprecision(12) << "{\"sweeps\":" << st.sweeps << ",\"edge_msg_updates\":" << st.edge_msg_updates
           << ",\"marginal_gather_sweeps\":" << st.psi_form_sweeps << ",\"run_seconds\":" << secs
           << ",\"bytes_per_sweep\":" << st.bytes_per_sweep << ",\"device_bytes\":" << st.device_bytes
           << ",\"relaxation\":[" << ar_f << "," << ar_g << "," << ar_mix << "," << ar_damp << "]}\n";
    }
    sbmbp_destroy(eng);
    sbmbp_graph_destroy(graph);
    return 0;
}
