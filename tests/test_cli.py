"""bin/bp: the process-level drop-in boundary (flags, stdout/stderr lines and return codes of
the reference's main.cpp:85-365). CPU tier: argument handling; GPU tier: the README commands."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden, gpath

BP = os.path.join(ROOT, "bin", "bp")
DS = gpath("c1_dataset.edgelist")


@pytest.fixture(scope="module", autouse=True)
def built():
    import sbm_bp_amd as S
    S.build_all()
    assert os.path.exists(BP)


def run(*args, env=None):
    p = subprocess.run([BP] + [str(a) for a in args], capture_output=True, text=True, timeout=300,
                       env=None if env is None else dict(os.environ, **env))
    return p.returncode, p.stdout, p.stderr


def test_help_and_required_flags():
    rc, out, err = run()
    assert rc == 0 and "BP algorithms for the SBM (final output only)" in err and out == ""  # main.cpp:154-160
    rc, out, err = run("-h")
    assert rc == 0 and "--edge_list_path" in err
    assert run("-m", "infer", "-n", 500, 500)[::2] == (1, "edge_list_path is required (-e flag)\n")  # :162-165 (sic)
    assert run("-l", DS, "-n", 500, 500)[::2] == (1, "mode is required (-m flag)\n")
    assert run("-l", DS, "-m", "infer")[::2] == (1, "n is required (-n flag)\n")


def test_parameter_selection_errors():
    base = ["-l", DS, "-n", 500, 500, "-m", "infer"]
    assert run(*base)[::2] == (1, "Error! Please just input both pa/cab parameters.\n")  # :199-201
    assert run(*base, "--pa", 0.5, 0.5)[::2] == (1, "Error! Please just input both pa/cab parameters.\n")
    rc, _, err = run(*base, "--epsilon_c", 0.1, 3, "--pa", 0.5, 0.5, "--cab", 1, 2, 3)
    assert (rc, err) == (1, "Error! Please just choose one way to initialize the pa/cab parameter.\n")  # :196-198
    rc, _, err = run(*base, "--epsilon_c", 0.1, 3, "-i", 1)
    assert (rc, err) == (1, "Error! Please assign the file path of the initial belief of node membership.\n")  # :208-213
    rc, _, err = run(*base, "--mb_n", "--mb", 0, 1, "--epsilon_c", 0.1, 3)
    assert (rc, err) == (1, "Error! Please just select one option to assign the membership vector.\n")  # :178-180
    rc, _, err = run("--no_such_flag")
    assert rc == 1 and "unrecognised option" in err


def test_option_syntax_variants_and_silent_modes():
    # --opt=value, short options, negative multitoken values, an unknown --mode does nothing and returns 0 (:361-365)
    rc, out, err = run("--edge_list_path=" + DS, "-n", 500, 500, "--epsilon_c", -1, 3, "--mode=neither", "-d0", "-t", 10)
    assert rc == 0 and out == ""
    assert err == "Randomly assign initial messages!\nWarning! Assign true conf using ordered node membership.\n"
    rc, out, err = run("-l", DS, "-n", 500, 500, "--epsilon_c", 0.1, 3, "-m", "neither", "-f", 1, 2, 3)
    assert rc == 0 and err.startswith("Randomly assign initial messages, except certain fixed nodes.\n")  # :214-215


def test_missing_edgelist_fails_loudly():
    rc, out, err = run("-l", "/nonexistent/file", "-n", 500, 500, "--epsilon_c", 0.1, 3, "-m", "infer")
    assert rc == 1 and "cannot open edge list" in err  # documented deviation from SURVEY B14


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc, out, err = run("-l", DS, "-n", 500, 500, "--epsilon_c", 0.1, 3, "-m", "infer")
    assert rc == 1 and out == "" and "no CPU fallback" in err


@pytest.mark.gpu
def test_readme_infer_command():
    """README.md:36 — reference stdout `2.99556 -0.143476 0.5 32` (niter is schedule dependent)"""
    rc, out, err = run("-l", DS, "-n", 500, 500, "--pa", 0.5, 0.5, "--cab", 3.63, 2.36, 3.63, "-t", 1000, "-i", 0, "-m", "infer", "-d", 0)
    assert rc == 0
    assert err == "Randomly assign initial messages!\nWarning! Assign true conf using ordered node membership.\n"
    assert out.endswith(" \n") and out.count("\n") == 1  # trailing space before the newline (bp.cpp:88)
    tok = out.split()
    assert tok[0] == "2.99556" and tok[1] == "-0.143476" and abs(float(tok[2]) - 0.5) < 1e-5 and 5 <= int(tok[3]) <= 60


@pytest.mark.gpu
def test_readme_learn_command():
    """README.md:41 — reference stdout `0.5 0.5` / `3.63024 2.36016` / `2.36016 3.63024`, stderr `overlap:0.5`"""
    g = golden("c1_readme_learn_seed0")["result"]
    rc, out, err = run("-l", DS, "-n", 500, 500, "--pa", 0.5, 0.5, "--cab", 3.63, 2.36, 3.63, "-t", 1000, "-i", 0, "-m", "learn", "-d", 0)
    assert rc == 0
    lines = out.split("\n")
    assert len(lines) == 4 and lines[3] == "" and lines[0].endswith(" ")
    # the reference's stdout, byte for byte (c1_readme_learn_seed0.json: stdout)
    assert lines[0] == "0.5 0.5 " and lines[1] == "3.63024 2.36016 " and lines[2] == "2.36016 3.63024 "
    assert [float(x) for x in lines[1].split()] == [float("%g" % x) for x in g["cab_final"][:2]]
    assert "Algorithm stop because of fdiff < learning_conv_crit. [which is good]\n" in err
    ov = [l for l in err.split("\n") if l.startswith("overlap:")]
    assert len(ov) == 1 and ov[0] == "overlap:0.5"


@pytest.mark.gpu
def test_matched_parameters_precision_and_marginals(tmp_path):
    g = golden("c1_matched_tight_seed0")["result"]
    mj = tmp_path / "m.json"
    rc, out, err = run("-l", DS, "-n", 500, 500, "--epsilon_c", 0.1, 3.0, "-t", 2000, "-m", "infer", "-d", 0, "-e", 1e-13,
                       "--precision", 15, "--if_output_marginals", "--metrics_json", mj)
    assert rc == 0
    lines = out.split("\n")
    e, f, ov, niter = lines[0].split()
    assert abs(float(f) - g["f"]) < 1e-9 and abs(float(e) - g["e"]) < 1e-9 and abs(float(ov) - g["overlap"]) < 1e-9
    psi = np.array([[float(x) for x in l.split()] for l in lines[1:1001]])
    ref = np.array(g["psi"]).reshape(1000, 2)
    assert min(np.abs(psi - ref).max(), np.abs(psi[:, ::-1] - ref).max()) < 1e-9
    assert err.count("margEntropy H(v) is") == 1000  # per-node entropies on clog (bp.cpp:95-97)
    m = json.load(open(mj))
    assert m["sweeps"] == int(niter) + 1 and m["marginal_gather_sweeps"] == m["sweeps"] - 1
    assert m["relaxation"][:2] == [0, -1]  # plain synchronous sweeps: the adaptive relaxation never stepped in


@pytest.mark.gpu
def test_degree_corrected_run_prints_minus_nan_entropy():
    g = golden("c1_dc1_default_seed0")["result"]
    rc, out, err = run("-l", DS, "-n", 500, 500, "--pa", 0.5, 0.5, "--cab", 0.60606060606060608, 0.060606060606060608,
                       0.60606060606060608, "-t", 1000, "-m", "infer", "-d", 0, "--deg_corr_flag", 1)
    assert rc == 0
    tok = out.split()
    assert tok[0] == "-nan"  # the reference prints -nan for deg_corr_flag != 0
    assert abs(float(tok[1]) - g["f"]) < 1e-5 and abs(float(tok[2]) - g["overlap"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [[], ["--gpus", 3]])
def test_reference_flags_only_converge_on_the_oscillating_hub_instance(gpus, tmp_path):
    """plain SBM on a power-law graph (fixture hub_dc0_tight_seed0: the reference converges after 139 sweeps): with nothing but
    the reference's own flags bin/bp prints the reference's line - free energy, entropy and overlap to 1e-9, niter of the same
    size - on one GPU and sharded. (Round 2 printed niter = -1 here: plain synchronous sweeps oscillate.)"""
    g = golden("hub_dc0_tight_seed0")["result"]
    rc, out, err = run("-l", gpath("hub_n600.edgelist"), "-n", 200, 200, 200, "--pa", 0.3333333333333333, 0.3333333333333333,
                       0.3333333333333333, "--cab", 9, 1.5, 1.5, 9, 1.5, 9, "-t", 2000, "-e", 1e-12, "-m", "infer", "-d", 0,
                       "--precision", 15, "--metrics_json", tmp_path / "m.json", *gpus)
    assert rc == 0, err
    e, f, ov, niter = out.split("\n")[0].split()
    assert abs(int(niter) - g["niter"]) <= 10  # reference 139, synchronous with the field relaxed on the device 142
    if not gpus:  # the metrics say what the device did about the oscillation: field level 1 (field_mix 0.25), no damping
        m = json.load(open(tmp_path / "m.json"))
        assert m["relaxation"][:2] == [1, -1] and m["relaxation"][2] == 0.25 and m["relaxation"][3] == 1.0
    assert abs(float(f) - g["f"]) < 1e-9 and abs(float(e) - g["e"]) < 1e-8 and abs(float(ov) - g["overlap"]) < 1e-9


@pytest.mark.gpu
def test_more_than_sixteen_blocks(tmp_path, orc):
    """the reference has no cap on the number of blocks (main.cpp:271); bin/bp takes up to 64 (matrix-core kernels above 16):
    -m infer at Q = 32 prints the oracle's synchronous numbers, -m learn runs; --gpus says what it does not do yet"""
    from sbm_bp_amd import synth
    N, Q = 1600, 32
    pairs, cin, cout = synth.planted_partition(N, Q, 12.0, 0.02, 21)
    path = tmp_path / "q32.edgelist"
    np.savetxt(path, pairs, fmt="%d")
    sizes = synth.group_sizes(N, Q)
    rc, out, err = run("-l", path, "-n", *sizes, "--epsilon_c", 0.02, 12.0, "-t", 1500, "-e", 1e-10, "-m", "infer", "-d", 0, "--precision", 15)
    assert rc == 0, err
    e, f, ov, niter = out.split("\n")[0].split()
    og = orc.Graph.from_edges(pairs, N)
    ob = orc.OracleBP(og, Q, 0)
    ob.init_messages(0, None, synth.true_conf(N, Q), orc.Rng(0))
    cab, na = orc.param_from_epsilon_c(N, Q, 0.02, 12.0)
    ob.set_params(cab, na, 1.0)
    ob.set_msg_form(True)
    it, _ = ob.converge_sync(1e-10, 1500, 1.0)
    assert it >= 0 and int(niter) == it
    fo, _ = ob.free_energy(0)
    eo, _ = ob.entropy(0)
    assert abs(float(f) - fo) < 1e-9 * abs(fo) and abs(float(e) - eo) < 1e-8 * abs(eo) and abs(float(ov) - ob.overlap()) < 1e-9
    rc, out, err = run("-l", path, "-n", *sizes, "--epsilon_c", 0.02, 12.0, "-m", "infer", "--gpus", 2)
    assert rc == 1 and "Q in [2, 16]" in err
    rc, out, err = run("-l", path, "-n", *sizes, "--epsilon_c", 0.02, 12.0, "-m", "learn", "-t", 30, "-d", 0)
    assert rc == 0 and len(out.split()) == Q + Q * Q, err  # group fractions, then the learned cab (tests/test_gpu_wide.py pins the numbers)
    rc, out, err = run("-l", path, "-n", *([25] * 65), "--epsilon_c", 0.02, 12.0, "-m", "infer")
    assert rc == 1 and "between 2 and 64" in err


MATCHED = ["-l", DS, "-n", 500, 500, "--pa", 0.5, 0.5, "--cab", 5.4545454545454541, 0.54545454545454541, 5.4545454545454541,
           "-t", 5000, "-m", "infer", "-d", 0, "--precision", 15]


def _line(out):
    e, f, ov, niter = out.split("\n")[0].split()
    return float(e), float(f), float(ov), int(niter)


@pytest.mark.gpu
def test_clamped_beliefs_flag_i1():
    """-i 1 --beliefs_path: planted rows are clamped (bp_conditional), message-gather kernel"""
    g = golden("c1_planted_i1_seed0")["result"]
    rc, out, err = run(*MATCHED, "-e", 1e-13, "-i", 1, "--beliefs_path", gpath("c1_beliefs.txt"))
    assert rc == 0 and "Randomly assign" not in err
    e, f, ov, niter = _line(out)
    assert abs(f - g["f"]) < 1e-9 and abs(e - g["e"]) < 1e-9 and abs(ov - g["overlap"]) < 1e-9 and niter >= 0


@pytest.mark.gpu
def test_damping_flag_R():
    g = golden("c1_matched_damped_seed0")["result"]
    rc, out, err = run(*MATCHED, "-e", 1e-12, "-R", 0.5)
    assert rc == 0
    e, f, ov, niter = _line(out)
    assert abs(f - g["f"]) < 1e-9 and abs(ov - g["overlap"]) < 1e-9 and niter >= 0


@pytest.mark.gpu
def test_beta_flag():
    g = golden("c1_matched_beta08_seed0")["result"]
    rc, out, err = run(*MATCHED, "-e", 1e-12, "-b", 0.8)
    assert rc == 0
    e, f, ov, niter = _line(out)
    assert abs(f - g["f"]) < 1e-9 and abs(e - g["e"]) < 1e-9 and abs(ov - g["overlap"]) < 1e-9


@pytest.mark.gpu
def test_deg_corr_flag_2():
    g = golden("c1_dc2_tight_seed0")["result"]
    rc, out, err = run("-l", DS, "-n", 500, 500, "--pa", 0.5, 0.5, "--cab", 0.60606060606060608, 0.060606060606060608,
                       0.60606060606060608, "-t", 5000, "-m", "infer", "-d", 0, "-e", 1e-13, "--deg_corr_flag", 2, "--precision", 15)
    assert rc == 0
    tok = out.split()
    assert tok[0] == "-nan" and abs(float(tok[1]) - g["f"]) < 1e-9 * abs(g["f"]) and abs(float(tok[2]) - g["overlap"]) < 1e-9


@pytest.mark.gpu
def test_fixed_nodes_with_i0_have_no_effect_and_true_conf_file(tmp_path):
    """-i 0 -f ...: the planted vector is never stored, so the result equals the plain run;
    --true_conf_path with swapped labels leaves the permutation-maximised overlap unchanged"""
    g = golden("c1_matched_tight_seed0")["result"]
    tc = tmp_path / "tc.txt"
    tc.write_text("\n".join(["1"] * 500 + ["0"] * 500) + "\n")
    rc, out, err = run(*MATCHED, "-e", 1e-13, "-f", 1, 2, 3, "--true_conf_path", tc)
    assert rc == 0 and err.startswith("Randomly assign initial messages, except certain fixed nodes.\n")
    assert "Warning! Assign true conf" not in err
    e, f, ov, niter = _line(out)
    assert abs(f - g["f"]) < 1e-9 and abs(ov - g["overlap"]) < 1e-9


def test_vertex_id_beyond_n_is_rejected(tmp_path):
    """ids >= sum(n) make the reference index out of range; here it is an error before any GPU work"""
    p = tmp_path / "g.edgelist"
    p.write_text("0 1\n2 12\n")
    rc, out, err = run("-l", p, "-n", 5, 5, "--epsilon_c", 0.1, 3, "-m", "infer")
    assert rc == 1 and out == "" and "vertex ids >= sum(n) = 10" in err


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-R", 0.5], ["-i", 1, "--beliefs_path", gpath("c1_beliefs.txt")]])
def test_gpus_flag_shards_the_run_and_prints_the_same_lines(extra):
    """--gpus N: one host thread per rank over the C++ multi-GPU driver. On a one-GPU box the ranks share the device
    (in-process transport; a note on stderr says so); stdout is the single-GPU run's, digit for digit at 12 places."""
    args = MATCHED[:-2] + ["--precision", 12, "-e", 1e-12] + extra
    rc1, out1, err1 = run(*args)
    rc3, out3, err3 = run(*args, "--gpus", 3)
    assert rc1 == 0 and rc3 == 0, err3
    assert "ranks share devices over the in-process transport" in err3 or "--gpus" not in err3
    e1, f1, o1, n1 = _line(out1)
    e3, f3, o3, n3 = _line(out3)
    assert n1 == n3 and abs(f1 - f3) < 1e-11 and abs(e1 - e3) < 1e-11 and abs(o1 - o3) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["1:setup", "2:run"])
def test_gpus_flag_a_failing_rank_ends_the_run(where):
    """a rank that fails outside a collective - during set-up, or after the others have started their sweeps - must end the
    whole run with its error, not leave the other rank threads waiting for it: set-up is agreed on before the first
    collective, and a later failure aborts EVERY communicator of the run"""
    import time
    t0 = time.monotonic()
    rc, out, err = run(*MATCHED, "--gpus", 3, env={"SBMBP_INJECT_RANK_FAIL": where, "SBMBP_LOCAL_TIMEOUT_S": "20"})
    assert rc == 1 and out == "" and ("rank %s" % where[0]) in err and "injected failure" in err, err
    assert time.monotonic() - t0 < 120


@pytest.mark.gpu
def test_gpus_flag_learn_and_marginals():
    rc1, out1, _ = run("-l", DS, "-n", 500, 500, "--pa", 0.5, 0.5, "--cab", 5, 1, 5, "-t", 1000, "-i", 0, "-m", "learn", "-d", 0, "--precision", 9)
    rc2, out2, err2 = run("-l", DS, "-n", 500, 500, "--pa", 0.5, 0.5, "--cab", 5, 1, 5, "-t", 1000, "-i", 0, "-m", "learn", "-d", 0, "--precision", 9,
                          "--gpus", 2)
    assert rc1 == 0 and rc2 == 0, err2
    a = np.array([float(x) for x in out1.split()])
    b = np.array([float(x) for x in out2.split()])
    assert a.shape == b.shape == (6,) and np.abs(a - b).max() < 1e-7 and "overlap:" in err2
    rc, out, err = run(*MATCHED, "-e", 1e-13, "--if_output_marginals", "--gpus", 4)
    assert rc == 0
    lines = out.split("\n")
    g = golden("c1_matched_tight_seed0")["result"]
    psi = np.array([[float(x) for x in l.split()] for l in lines[1:1001]])
    ref = np.array(g["psi"]).reshape(1000, 2)
    assert min(np.abs(psi - ref).max(), np.abs(psi[:, ::-1] - ref).max()) < 1e-9 and err.count("margEntropy H(v) is") == 1000
