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
for wl in (() if PMC_ONLY else ("C3", "C2", "C4", "C5", "Q12", "Q16", "Q32", "Q48", "Q64")):
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
import subprocess
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_*" % R))):
    wl = os.path.basename(d).split("_")[-1]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarise_pmc.py"), R, wl], check=False)
if PMC_ONLY:
    sys.exit(0)
for f in glob.glob(os.path.join(OUT, "budget_c3_w8_c*.json")):
    shutil.copy(f, os.path.join(PROF, "%s_%s" % (R, os.path.basename(f))))
for name, dst in (("infer_phases_C3.log", "%s_c3_infer_phases.log"), ("learn_C5.log", "%s_c5_learn_phases.log")):
    if os.path.exists(os.path.join(OUT, name)):
        shutil.copy(os.path.join(OUT, name), os.path.join(PROF, dst % R))
st = glob.glob(os.path.join(OUT, "stats_infer_C3", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(PROF, "%s_c3_infer_phases_kernel_stats.csv" % R))
for name, dst in (("bench_rehearsal3.json", "%s_small_bench_3ranks_rehearsal.json"), ("bench_C3_rccl1.json", "%s_c3_bench_rccl_1rank.json"),
                  ("bench_C3_200steps.json", "%s_c3_bench_200steps.json")):
    b = os.path.join(OUT, name)
    if os.path.exists(b):
        line = [l for l in open(b) if l.startswith("{")]
        if line:
            open(os.path.join(PROF, dst % R), "w").write(line[-1])
print("profiles/:", sorted(os.listdir(PROF)))
