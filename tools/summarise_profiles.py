"""Collects what tools/profile_round.sh produced under gpurun_out/profile_<round>/ into profiles/ (committed)."""
import csv, glob, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
PMC_ONLY = len(sys.argv) > 2 and sys.argv[2] == "pmc"  # only the counter summary (profile_round.sh runs this before its benches)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha
OUT = os.path.join(ROOT, "gpurun_out", "profile_" + R)
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)
for wl in (() if PMC_ONLY else ("C3", "C2", "C4", "C5")):
    st = glob.glob(os.path.join(OUT, "stats_" + wl, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(PROF, "%s_%s_kernel_stats.csv" % (R, wl.lower())))
    b = os.path.join(OUT, "bench_%s.json" % wl)
    if os.path.exists(b):
        line = [l for l in open(b) if l.startswith("{")]
        if line:
            open(os.path.join(PROF, "%s_%s_bench.json" % (R, wl.lower())), "w").write(line[-1])
b = os.path.join(OUT, "bench_C3_messages.json")
if os.path.exists(b) and not PMC_ONLY:
    line = [l for l in open(b) if l.startswith("{")]
    if line:
        open(os.path.join(PROF, "%s_c3_bench_message_gather.json" % R), "w").write(line[-1])
# the kernel sources the round was measured on, recorded by profile_round.sh on the GPU box (never recomputed here: a
# summary made after an edit would otherwise tie old counters to new sources)
try:
    measured_sha = open(os.path.join(OUT, "source_sha.txt")).read().strip()
except OSError:
    measured_sha = None
if measured_sha and measured_sha != source_sha():
    print("NOTE: gpurun_out/profile_%s was measured on sources %s, the tree is now %s" % (R, measured_sha, source_sha()))
counters = {}
for d in sorted(glob.glob(os.path.join(OUT, "pmc_*"))) if measured_sha else []:
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_sweep_psi" not in row.get("Kernel_Name", ""):
                continue
            counters.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
if counters:
    try:
        bench = json.loads(open(os.path.join(PROF, "%s_c3_bench.json" % R)).read())
        kernel, e2 = bench["roofline"]["kernel"], bench["config"]["E2"]
    except Exception:
        kernel, e2 = "k_sweep_psi<4>", 100015584  # bench.py's default workload (seeded synthetic graph)
    out = {"kernel": kernel, "workload": "C3 N=1e7 Q=4 c=10 (E2=%d)" % e2,
           "source_sha": measured_sha,
           "command": "rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline (one pass per counter; tools/profile_round.sh)",
           "counters": {k: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for k, v in counters.items()}}
    json.dump(out, open(os.path.join(PROF, "%s_c3_pmc_k_sweep_psi.json" % R), "w"), indent=1)
    print(json.dumps(out["counters"]))
if PMC_ONLY:
    sys.exit(0)
for f in glob.glob(os.path.join(OUT, "budget_c3_w8_c*.json")):
    shutil.copy(f, os.path.join(PROF, "%s_%s" % (R, os.path.basename(f))))
for name, dst in (("bench_rehearsal3.json", "%s_small_bench_3ranks_rehearsal.json"), ("bench_C3_rccl1.json", "%s_c3_bench_rccl_1rank.json")):
    b = os.path.join(OUT, name)
    if os.path.exists(b):
        line = [l for l in open(b) if l.startswith("{")]
        if line:
            open(os.path.join(PROF, dst % R), "w").write(line[-1])
print("profiles/:", sorted(os.listdir(PROF)))
