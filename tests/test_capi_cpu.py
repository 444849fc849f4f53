"""CPU tier: the C-ABI library loads and exports every symbol include/sbmbp.h declares, and the
host-side logic behind it (CSR builder, edge-list parser, parameter constructors) agrees with
the oracle. No compute entry point is called here (no GPU in this tier)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLD, ROOT, args_of, golden, gpath


@pytest.fixture(scope="module")
def S():
    import sbm_bp_amd as S
    S.build_all()
    S.load_library()
    return S


def test_every_declared_symbol_is_exported_and_bound(S):
    from sbm_bp_amd.capi import SYMBOLS
    hdr = open(os.path.join(ROOT, "include", "sbmbp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sbmbp_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(SYMBOLS), (declared ^ set(SYMBOLS))
    raw = C.CDLL(S.lib_path())
    for name in declared:
        assert getattr(raw, name) is not None


def test_header_is_plain_c():
    import shutil
    import subprocess
    import tempfile
    cc = shutil.which("gcc")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "sbmbp.h"\nint main(void){return sizeof(sbmbp_stats) > 0 ? 0 : 1;}\n')
        subprocess.check_call([cc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o",
                               os.path.join(d, "t.o")])


def test_error_strings_and_no_device_failure(S):
    lib = S.load_library()
    assert lib.sbmbp_strerror(0) == b"ok"
    assert b"GPU" in lib.sbmbp_strerror(-3)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    g = S.load_edge_list(gpath("c1_dataset.edgelist"), 1000)
    bp = S.bp_conditional()
    with pytest.raises(S.SbmbpError) as ei:  # the product path fails loudly, it never falls back to a CPU path
        bp.init_messages(S.blockmodel_t(g, 2, 0), 0, None, np.repeat([0, 1], 500), 0)
    assert ei.value.code == -3


def test_csr_builder_equals_oracle_on_dataset(S, orc):
    g = S.load_edge_list(gpath("c1_dataset.edgelist"), 1000)
    og = orc.Graph.from_edgelist(gpath("c1_dataset.edgelist"), 1000)
    rp, nbr, rev = g.csr()
    assert (g.N, g.E2, g.max_degree) == (1000, 2996, 10)
    assert (rp == og.row_ptr).all() and (nbr == og.nbr).all() and (rev == og.rev).all()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_csr_builder_random_multigraph(S, orc, seed):
    rng = np.random.default_rng(seed)
    N = 300
    pairs = rng.integers(0, N, size=(2000, 2)).astype(np.uint32)  # duplicates, reversed duplicates, self-loops
    g = S.Graph.from_edges(pairs, N)
    og = orc.Graph.from_edges(pairs, N)
    rp, nbr, rev = g.csr()
    assert (rp == og.row_ptr).all() and (nbr == og.nbr).all() and (rev == og.rev).all()
    assert (rev[rev] == np.arange(g.E2)).all()
    g2 = S.Graph.from_csr(rp, nbr)  # adopting a CSR recomputes the same reverse index
    assert (g2.csr()[2] == rev).all()
    with pytest.raises(S.SbmbpError):
        bad = nbr.copy()
        bad[0] = (bad[0] + 1) % N
        S.Graph.from_csr(rp, bad)


def test_empty_and_ragged_graphs(S, orc):
    g = S.Graph.from_edges(np.zeros((0, 2), dtype=np.uint32), 5)
    assert (g.N, g.E2, g.max_degree) == (5, 0, 0)
    g = S.Graph.from_edges([[0, 9]], 3)  # ids beyond N grow the graph (graph_utilities.cpp:65-72)
    assert g.N == 10 and g.E2 == 2
    star = np.array([[0, i] for i in range(1, 2000)], dtype=np.uint32)  # one hub, many leaves
    g = S.Graph.from_edges(star, 2000)
    assert g.max_degree == 1999 and g.E2 == 2 * 1999


def test_edgelist_parser_errors(S, tmp_path):
    p = tmp_path / "ok.edgelist"
    p.write_text("0 1\n\n2 3 0.5\n 4\t5 \r\n")
    g = S.load_edge_list(str(p), 0)
    assert g.N == 6 and g.E2 == 6
    for bad in ("0 1\n7\n", "0 x\n", "a b\n"):
        q = tmp_path / "bad.edgelist"
        q.write_text(bad)
        with pytest.raises(S.SbmbpError):
            S.load_edge_list(str(q), 0)
    with pytest.raises(S.SbmbpError) as ei:
        S.load_edge_list(str(tmp_path / "missing.edgelist"), 10)
    assert ei.value.code == -5  # documented deviation: the reference yields an empty graph


def test_edgelist_parser_many_pieces(S, tmp_path):
    """a file large enough to be cut into several pieces parsed concurrently: same pairs in file order, line number
    of the first error counted across pieces, last line without a newline"""
    rng = np.random.default_rng(3)
    pairs = rng.integers(0, 200000, size=(1_200_000, 2), dtype=np.uint32)
    p = tmp_path / "big.edgelist"
    text = "\n".join("%d %d" % (a, b) for a, b in pairs)
    assert len(text) > (12 << 20)
    p.write_text(text)  # no trailing newline
    g = S.load_edge_list(str(p), 200000)
    ref = S.Graph.from_edges(pairs, 200000)
    assert all(np.array_equal(x, y) for x, y in zip(g.csr(), ref.csr()))
    lines = text.split("\n")
    lines[1_100_000] = "17"  # line 1100001 holds one id
    lines[1_150_000] = "x y"
    p.write_text("\n".join(lines) + "\n")
    with pytest.raises(S.SbmbpError) as ei:
        S.load_edge_list(str(p), 200000)
    assert "line 1100001" in str(ei.value) and "only one id" in str(ei.value)


@pytest.mark.parametrize("Q,flag", [(2, 0), (4, 0), (3, 1), (4, 2), (5, 3), (8, 0)])
def test_host_initial_state_is_the_mt19937_stream(S, orc, Q, flag):
    """sbmbp_host_init_state (block MT19937, rows filled concurrently) == the oracle's init_messages through
    std::mt19937 + std::uniform_real_distribution (belief_propagation.cpp:101-217), bit for bit"""
    rng = np.random.default_rng(Q * 10 + flag)
    N = 3000
    pairs = rng.integers(0, N, size=(9000, 2), dtype=np.uint32)
    pairs = np.concatenate([pairs, np.array([[7, i] for i in range(8, 2500)], dtype=np.uint32)])  # one long row
    g, og = S.Graph.from_edges(pairs, N), orc.Graph.from_edges(pairs, N)
    conf = None
    if flag:
        conf = rng.integers(-1, Q, size=N).astype(np.int32)
        conf[7] = -1 if flag == 1 else 1
    for seed in (0, 5, 4294967295):
        bp = orc.OracleBP(og, Q, 0)
        bp.init_messages(flag, conf, np.zeros(N, dtype=np.uint32), orc.Rng(seed))
        psi_o, msg_o = bp.get_state()
        psi, msg = g.initial_state(Q, flag, conf, seed)
        assert np.array_equal(psi, psi_o) and np.array_equal(msg, msg_o)


def test_host_initial_state_across_slabs(S, orc):
    """more than one slab of raw words (the generator thread runs ahead of the rows being filled)"""
    N, Q = 1_500_000, 8
    rng = np.random.default_rng(1)
    pairs = rng.integers(0, N, size=(2_600_000, 2), dtype=np.uint32)
    g, og = S.Graph.from_edges(pairs, N), orc.Graph.from_edges(pairs, N)
    assert (N + g.E2) * 2 * Q > 3 * (32 << 20)
    bp = orc.OracleBP(og, Q, 0)
    bp.init_messages(0, None, np.zeros(N, dtype=np.uint32), orc.Rng(42))
    psi_o, msg_o = bp.get_state()
    psi, msg = g.initial_state(Q, 0, None, 42)
    assert np.array_equal(psi, psi_o) and np.array_equal(msg, msg_o)


def test_param_constructors_equal_oracle_and_golden(S, orc):
    for name in ("c1_matched_tight_seed0", "q4_tight_seed0", "q4_epsc_default_seed0", "hub_dc0_tight_seed0", "q10_tight_seed1"):
        gd = golden(name)
        a, r = args_of(gd), gd["result"]
        g = S.load_edge_list(a["path"], a["N"])
        bm = S.blockmodel_t(g, a["Q"], a["dc"])
        st = S.bp_param_from_epsilon_c(bm, a["eps"], a["c"]) if "eps" in a else S.bp_param_from_direct(bm, a["pa"], a["cab_upper"])
        assert list(st.cab.ravel()) == r["cab"] and list(st.na) == r["na"]
    bm = S.blockmodel_t(S.Graph.from_edges([[0, 1]], 1001), 3, 0)  # truncation quirk B7: na does not sum to N
    assert list(S.bp_param_from_epsilon_c(bm, 0.1, 3.0).na) == [333, 333, 333]
    st = S.bp_param_from_epsilon_c(bm, -1.0, 3.0)  # epsilon < 0: fully disassortative (blockmodel.cpp:251-254)
    assert st.cab[0, 0] == 0 and abs(st.cab[0, 1] - 4.5) < 1e-15


def test_synthetic_generator(S):
    from sbm_bp_amd import synth
    pairs, cin, cout = synth.planted_partition(40000, 4, 10.0, 0.1, 7)
    assert abs(cin - 40 / 1.3) < 1e-12 and abs(cout - 0.1 * cin) < 1e-12
    assert (pairs[:, 0] < pairs[:, 1]).all()
    mean_deg = 2 * len(pairs) / 40000
    assert abs(mean_deg - 10.0) < 0.2
    tc = synth.true_conf(40000, 4)
    same = (tc[pairs[:, 0]] == tc[pairs[:, 1]]).mean()
    assert abs(same - cin / (cin + 3 * cout)) < 0.01


def test_communicator_handles_and_no_device_errors():
    """the host side of the multi-GPU layer without a GPU: in-process and null communicators are plain host objects; creating
    a rank's engine without a device fails loudly (no CPU fallback), as does an RCCL communicator"""
    import ctypes as C
    import torch
    import sbm_bp_amd as S
    from sbm_bp_amd.capi import COMM_ID_BYTES
    lib = S.load_library()
    arr = (C.c_void_p * 3)()
    assert lib.sbmbp_comm_init_local(arr, 3) == 0
    assert [lib.sbmbp_comm_rank(arr[r]) for r in range(3)] == [0, 1, 2] and lib.sbmbp_comm_size(arr[1]) == 3
    assert lib.sbmbp_comm_transport(arr[0]) == b"local"
    lib.sbmbp_comm_abort(arr[2])  # a rank gives up: the group is marked failed, nobody would block on it
    h = C.c_void_p()
    assert lib.sbmbp_comm_init_null(C.byref(h), 8, 5) == 0 and lib.sbmbp_comm_rank(h) == 5 and lib.sbmbp_comm_transport(h) == b"null"
    assert lib.sbmbp_comm_init_null(C.byref(C.c_void_p()), 4, 4) != 0  # rank outside the communicator
    if not torch.cuda.is_available():
        g = S.Graph.from_edges(np.array([[0, 1], [1, 2], [2, 3]], dtype=np.uint32), 4)
        d = C.c_void_p()
        rc = lib.sbmbp_dist_create(C.byref(d), h, g._h, 2, 0, 0, 0)
        assert rc == -3 and b"no CPU fallback" in lib.sbmbp_last_error()
        assert lib.sbmbp_device_count() == 0
        c2 = C.c_void_p()
        assert lib.sbmbp_comm_init_rank(C.byref(c2), bytes(COMM_ID_BYTES), 1, 0, 0) == -3
    lib.sbmbp_comm_destroy(h)
    for r in range(3):
        lib.sbmbp_comm_destroy(arr[r])


def test_host_code_under_asan_ubsan(tmp_path):
    """the host side of the engine (csrc/host_graph.cpp) and the CPU restatement (oracle/bp_oracle.cpp)
    compiled with -fsanitize=address,undefined and driven through their edge cases by tests/sanitize/host_sanitize.cpp (ragged
    and missing edge lists, duplicates / self-loops / ids at the bound, every init flag against the oracle bit for bit - also
    through the streaming sink -, both schedules, reductions, learning, a row of 699 edges). CPU build only: GPU
    AddressSanitizer is not available on this pool."""
    import shutil
    import subprocess
    from conftest import ROOT, gpath
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "host_sanitize"
    src = [os.path.join(ROOT, "tests", "sanitize", "host_sanitize.cpp"), os.path.join(ROOT, "sbm-bp_amd", "csrc", "host_graph.cpp"),
           os.path.join(ROOT, "oracle", "bp_oracle.cpp")]
    subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-mavx2", "-pthread",
                    "-o", str(exe)] + src, check=True, timeout=600)
    pr = subprocess.run([str(exe), gpath("c1_dataset.edgelist")], capture_output=True, text=True, timeout=600,
                        env=dict(os.environ, TMPDIR=str(tmp_path), UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert pr.returncode == 0 and "host_sanitize ok" in pr.stdout, (pr.stdout[-500:], pr.stderr[-3000:])
    assert "runtime error" not in pr.stderr and "AddressSanitizer" not in pr.stderr and "LeakSanitizer" not in pr.stderr
