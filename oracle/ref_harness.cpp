// TEST INFRASTRUCTURE — not product code.
//
// Boost-free driver for the *untouched* reference sources under /root/reference/src
// (belief_propagation.cpp, blockmodel.cpp, graph_utilities.cpp, output_functions.cpp).
// It replays what the reference's own src/main.cpp:236-365 does (main.cpp itself needs
// Boost.ProgramOptions, which this image lacks) and prints results as one JSON object at
// 17 significant digits. The reference objects are compiled by oracle/Makefile into
// oracle/_ref/ (git-ignored); no reference source is copied into this repository.
//
// Uses: (1) generating the golden fixtures under tests/golden/ (oracle/make_golden.py);
//       (2) the "reference" CPU baseline of bench.py (converge() only — the reference's
//           compute_free_energy()/compute_entropy() are O(N^2), belief_propagation.cpp:675-741).
//
// Access to protected/private members (mmap_, real_psi_, cab_expect_, compute_f_site(), ...)
// is obtained by redefining the access keywords for the two reference headers only; the
// Itanium ABI does not encode access in mangled names, so the untouched objects link as is.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#define private public
#define protected public
#include "belief_propagation.h"
#include "blockmodel.h"
#undef private
#undef protected
#include "graph_utilities.h"

namespace {

using clk = std::chrono::steady_clock;
double secs(clk::time_point a, clk::time_point b) {
    return std::chrono::duration<double>(b - a).count();
}

std::vector<double> parse_doubles(const std::string &s) {
    std::vector<double> v;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) if (!tok.empty()) v.push_back(std::strtod(tok.c_str(), nullptr));
    return v;
}

std::vector<unsigned> parse_uints(const std::string &s) {
    std::vector<unsigned> v;
    for (double d : parse_doubles(s)) v.push_back(unsigned(d));
    return v;
}

struct args_t {
    std::map<std::string, std::string> kv;
    std::string get(const std::string &k, const std::string &dflt = "") const {
        auto it = kv.find(k);
        return it == kv.end() ? dflt : it->second;
    }
    bool has(const std::string &k) const { return kv.count(k) > 0; }
    double num(const std::string &k, double dflt) const {
        return has(k) ? std::strtod(get(k).c_str(), nullptr) : dflt;
    }
};

void json_vec(std::ostream &os, const std::vector<double> &v) {
    os << "[";
    for (size_t i = 0; i < v.size(); ++i) {
        if (i) os << ",";
        if (std::isnan(v[i])) os << "\"nan\"";
        else if (std::isinf(v[i])) os << (v[i] > 0 ? "\"inf\"" : "\"-inf\"");
        else os << v[i];
    }
    os << "]";
}

void json_num(std::ostream &os, double v) {
    if (std::isnan(v)) os << "\"nan\"";
    else if (std::isinf(v)) os << (v > 0 ? "\"inf\"" : "\"-inf\"");
    else os << v;
}

bool load_edges_bin(edge_list_t &el, const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    f.seekg(0, std::ios::end);
    size_t bytes = size_t(f.tellg());
    f.seekg(0);
    std::vector<uint32_t> raw(bytes / 4);
    f.read(reinterpret_cast<char *>(raw.data()), std::streamsize(raw.size() * 4));
    el.clear();
    el.reserve(raw.size() / 2);
    for (size_t i = 0; i + 1 < raw.size(); i += 2) el.push_back(std::make_pair(raw[i], raw[i + 1]));
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        std::cerr << "usage: bp_ref <infer|learn|converge|node_update|em_expect|rng> key=value ...\n"
                     "  keys: l=<edgelist|.bin> n=a,b,.. pa=.. cab=<upper triangle> | eps=.. c=..\n"
                     "        d=<seed> dc=<0|1|2> beta= e=<crit> E=<learn crit> t=<max sweeps> R=<damp> r=<lr>\n"
                     "        i=<init flag> beliefs=<path> dump=psi,msg nodes=a,b,.. quiet=1\n";
        return 2;
    }
    std::string cmd = argv[1];
    args_t a;
    for (int i = 2; i < argc; ++i) {
        std::string s = argv[i];
        auto p = s.find('=');
        if (p == std::string::npos) a.kv[s] = "1";
        else a.kv[s.substr(0, p)] = s.substr(p + 1);
    }
    std::cout << std::setprecision(17);

    if (cmd == "rng") {  // Appendix E of SURVEY.md: first draws of the shared engine
        std::mt19937 engine(unsigned(a.num("d", 0)));
        std::uniform_real_distribution<> u(0, 1);
        std::vector<double> v;
        for (int i = 0; i < 8; ++i) v.push_back(u(engine));
        std::cout << "{\"seed\":" << unsigned(a.num("d", 0)) << ",\"doubles\":";
        json_vec(std::cout, v);
        std::cout << "}\n";
        return 0;
    }

    // ---- main.cpp:236-296 replay -------------------------------------------------------
    unsigned seed = unsigned(a.num("d", 0));
    std::mt19937 engine(seed);
    uint_vec_t n = parse_uints(a.get("n"));
    unsigned Q = unsigned(n.size()), N = 0;
    for (auto x : n) N += x;
    uint_vec_t memberships_init(N, 0);
    {
        unsigned shift = 0;
        for (unsigned r = 0; r < Q; ++r) {
            for (unsigned i = 0; i < n[r]; ++i) memberships_init[shift + i] = r;
            shift += n[r];
        }
    }
    auto t0 = clk::now();
    edge_list_t edge_list;
    std::string path = a.get("l");
    if (path.size() > 4 && path.substr(path.size() - 4) == ".bin") load_edges_bin(edge_list, path);
    else load_edge_list(edge_list, path);
    adj_list_t adj_list = edge_to_adj(edge_list, N);
    edge_list.clear();
    auto t1 = clk::now();
    uint_vec_t true_conf = memberships_init;
    if (a.has("true_conf")) load_confs(true_conf, a.get("true_conf"));
    unsigned dc = unsigned(a.num("dc", 0));
    blockmodel_t blockmodel(memberships_init, Q, unsigned(adj_list.size()), dc, &adj_list);

    std::string mode = a.get("mode", cmd == "learn" ? "learn" : "infer");
    std::unique_ptr<belief_propagation> alg;
    if (mode == "learn") alg.reset(new bp_basic());
    else alg.reset(new bp_conditional());

    int_vec_t beliefs;
    load_beliefs(beliefs, a.get("beliefs"));
    unsigned init_flag = unsigned(a.num("i", 0));
    if (a.has("quiet")) std::clog.setstate(std::ios::failbit);  // hub path prints a line per call (bp.cpp:816)

    alg->init_messages(blockmodel, init_flag, beliefs, true_conf, engine);
    alg->init_special_needs(false);
    alg->set_beta(a.num("beta", 1.0));
    auto t2 = clk::now();

    bp_blockmodel_state state;
    if (a.has("eps")) state = bp_param_from_epsilon_c(blockmodel, a.num("eps", 0.1), a.num("c", 3.0));
    else state = bp_param_from_direct(blockmodel, parse_doubles(a.get("pa")), parse_doubles(a.get("cab")));

    float crit = float(a.num("e", 5.0e-6));
    float lcrit = float(a.num("E", 1.0e-6));
    unsigned tmax = unsigned(a.num("t", 100));
    float damp = float(a.num("R", 1.0));
    float lr = float(a.num("r", 0.2));
    std::string dump = a.get("dump");

    size_t E2 = 0;
    for (auto &s : adj_list) E2 += s.size();

    auto dump_state = [&](std::ostream &os) {
        if (dump.find("psi") != std::string::npos) {
            std::vector<double> flat;
            for (unsigned i = 0; i < N; ++i) for (unsigned q = 0; q < Q; ++q) flat.push_back(alg->real_psi_[i][q]);
            os << ",\"psi\":";
            json_vec(os, flat);
        }
        if (dump.find("msg") != std::string::npos) {
            // in-ordered, exactly as the reference stores it: mmap_[i][l][q] = message into i from its l-th neighbour
            std::vector<double> flat;
            for (unsigned i = 0; i < N; ++i)
                for (size_t l = 0; l < alg->mmap_[i].size(); ++l)
                    for (unsigned q = 0; q < Q; ++q) flat.push_back(alg->mmap_[i][l][q]);
            os << ",\"msg_in\":";
            json_vec(os, flat);
        }
    };

    std::ostringstream js;
    js << std::setprecision(17);
    js << "{\"cmd\":\"" << cmd << "\",\"N\":" << N << ",\"Q\":" << Q << ",\"E2\":" << E2 << ",\"seed\":" << seed
       << ",\"dc\":" << dc << ",\"load_s\":" << secs(t0, t1) << ",\"init_s\":" << secs(t1, t2);
    {
        std::vector<double> cabflat, naflat;
        for (unsigned q = 0; q < Q; ++q) {
            naflat.push_back(state.na[q]);
            for (unsigned t = 0; t < Q; ++t) cabflat.push_back(state.cab[q][t]);
        }
        js << ",\"cab\":";
        json_vec(js, cabflat);
        js << ",\"na\":";
        json_vec(js, naflat);
    }

    if (cmd == "infer") {
        // == belief_propagation::inference (bp.cpp:77-99) with the phases timed separately
        alg->expand_bp_params(state);
        auto c0 = clk::now();
        int niter = alg->converge(crit, tmax, damp, engine);
        auto c1 = clk::now();
        js << ",\"niter\":" << niter << ",\"converge_s\":" << secs(c0, c1);
        if (!a.has("skip_fe")) {
            double fs = alg->compute_f_site(), fe = alg->compute_f_edge(), fn = alg->compute_f_non_edge();
            double f = alg->compute_free_energy();
            auto c2 = clk::now();
            double es = alg->compute_entropy_site(), ee = alg->compute_entropy_edge(), en = alg->compute_entropy_non_edge();
            double e = alg->compute_entropy();
            auto c3 = clk::now();
            double ov = alg->compute_overlap();
            js << ",\"f_site\":"; json_num(js, fs);
            js << ",\"f_edge\":"; json_num(js, fe);
            js << ",\"f_nonedge\":"; json_num(js, fn);
            js << ",\"f\":"; json_num(js, f);
            js << ",\"e_site\":"; json_num(js, es);
            js << ",\"e_edge\":"; json_num(js, ee);
            js << ",\"e_nonedge\":"; json_num(js, en);
            js << ",\"e\":"; json_num(js, e);
            js << ",\"overlap\":"; json_num(js, ov);
            js << ",\"fe_s\":" << secs(c1, c2) / 2 << ",\"entropy_s\":" << secs(c2, c3) / 2;
            std::vector<double> h(alg->h_.begin(), alg->h_.end());
            js << ",\"h\":"; json_vec(js, h);
        } else {
            double ov = alg->compute_overlap();
            js << ",\"overlap\":"; json_num(js, ov);
        }
        dump_state(js);
    } else if (cmd == "converge") {
        // CPU-baseline mode: time converge() only; tmax sweeps at an unreachable criterion when e=0
        alg->expand_bp_params(state);
        auto c0 = clk::now();
        int niter = alg->converge(crit, tmax, damp, engine);
        auto c1 = clk::now();
        unsigned sweeps = niter < 0 ? tmax : unsigned(niter + 1);
        js << ",\"niter\":" << niter << ",\"sweeps\":" << sweeps << ",\"converge_s\":" << secs(c0, c1)
           << ",\"edge_msg_per_s\":" << double(sweeps) * double(E2) / secs(c0, c1);
        dump_state(js);
    } else if (cmd == "learn") {
        // belief_propagation::learning prints eta and cab on std::cout (bp.cpp:48-49): capture them
        std::ostringstream cap;
        cap << std::setprecision(17);
        auto *old = std::cout.rdbuf(cap.rdbuf());
        auto c0 = clk::now();
        alg->learning(blockmodel, state, lcrit, tmax, lr, damp, engine);
        auto c1 = clk::now();
        std::cout.rdbuf(old);
        std::vector<double> eta(alg->eta_.begin(), alg->eta_.end()), cabflat, nae(alg->na_expect_.begin(), alg->na_expect_.end()), cabe;
        std::vector<double> na(alg->na_.begin(), alg->na_.end());
        for (unsigned q = 0; q < Q; ++q)
            for (unsigned t = 0; t < Q; ++t) {
                cabflat.push_back(alg->cab_[q][t]);
                cabe.push_back(alg->cab_expect_[q][t]);
            }
        std::string text = cap.str();
        std::string esc;
        for (char ch : text) { if (ch == '\n') esc += "\\n"; else esc += ch; }
        js << ",\"learn_s\":" << secs(c0, c1) << ",\"stdout\":\"" << esc << "\"";
        js << ",\"eta\":"; json_vec(js, eta);
        js << ",\"na_final\":"; json_vec(js, na);
        js << ",\"cab_final\":"; json_vec(js, cabflat);
        js << ",\"na_expect\":"; json_vec(js, nae);
        js << ",\"cab_expect\":"; json_vec(js, cabe);
        js << ",\"overlap\":"; json_num(js, alg->compute_overlap());
        dump_state(js);
    } else if (cmd == "em_expect") {
        // converge tightly, then one EM expectation on the fixed point (bp.cpp:428-440, 892-989)
        alg->expand_bp_params(state);
        int niter = alg->converge(crit, tmax, damp, engine);
        alg->compute_na_expect();
        alg->compute_cab_expect();
        std::vector<double> nae(alg->na_expect_.begin(), alg->na_expect_.end()), nnae(alg->nna_expect_.begin(), alg->nna_expect_.end()), cabe;
        for (unsigned q = 0; q < Q; ++q) for (unsigned t = 0; t < Q; ++t) cabe.push_back(alg->cab_expect_[q][t]);
        js << ",\"niter\":" << niter;
        js << ",\"na_expect\":"; json_vec(js, nae);
        js << ",\"nna_expect\":"; json_vec(js, nnae);
        js << ",\"cab_expect\":"; json_vec(js, cabe);
        dump_state(js);
    } else if (cmd == "node_update") {
        // known-answer test of one node update from the seeded initial state (no schedule, no RNG):
        // init_h(), then bp_iter_update_psi / _large_degree on the listed nodes in order.
        alg->expand_bp_params(state);
        alg->init_h();
        std::vector<double> h0(alg->h_.begin(), alg->h_.end());
        js << ",\"h0\":"; json_vec(js, h0);
        uint_vec_t nodes = parse_uints(a.get("nodes"));
        std::vector<double> diffs;
        bool force_large = a.has("large");
        for (auto i : nodes) {
            double d = (force_large || adj_list[i].size() >= alg->LARGE_DEGREE)
                           ? alg->bp_iter_update_psi_large_degree(i, damp)
                           : alg->bp_iter_update_psi(i, damp);
            diffs.push_back(d);
        }
        js << ",\"nodes\":"; { std::vector<double> nd(nodes.begin(), nodes.end()); json_vec(js, nd); }
        js << ",\"diffs\":"; json_vec(js, diffs);
        std::vector<double> h1(alg->h_.begin(), alg->h_.end());
        js << ",\"h1\":"; json_vec(js, h1);
        // the updated nodes' marginals and the messages they emitted (stored in the neighbours' slots)
        std::vector<double> psis, outs;
        for (auto i : nodes) {
            for (unsigned q = 0; q < Q; ++q) psis.push_back(alg->real_psi_[i][q]);
            for (size_t l = 0; l < alg->graph_neis_[i].size(); ++l) {
                unsigned i2 = alg->graph_neis_[i][l], l2 = alg->graph_neis_inv_[i][l];
                for (unsigned q = 0; q < Q; ++q) outs.push_back(alg->mmap_[i2][l2][q]);
            }
        }
        js << ",\"psi_nodes\":"; json_vec(js, psis);
        js << ",\"out_msgs\":"; json_vec(js, outs);
        dump_state(js);
    } else if (cmd == "init") {
        dump_state(js);
    } else {
        std::cerr << "unknown command " << cmd << "\n";
        return 2;
    }
    js << "}";
    std::cout << js.str() << "\n";
    return 0;
}
