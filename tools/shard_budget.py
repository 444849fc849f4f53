"""Measurement aid: what ONE shard of the 8-rank C3 plan executes per sweep (kernel time by kind), for a given number of
chunks, from a rocprofv3 kernel trace of tools/check_shards_full_size.py. usage (on the GPU box):
  SBMBP_SHARD_CHUNKS=k rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/check_shards_full_size.py C3 8 6
  python3 tools/shard_budget.py DIR 8 8      # world, sweeps timed + warm-up"""
import csv, glob, sys
d, world, sweeps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the world-k run is the second half of the trace: take kernels launched after the last k_init_random pair
names = [r["Kernel_Name"] for r in rows]
last_init = max(i for i, n in enumerate(names) if "k_init_random" in n)
tot = {}
for r in rows[last_init + 1:]:
    n = r["Kernel_Name"]
    key = next((k for k in ("k_sweep_psi", "k_pack_rows", "k_unpack_rows", "k_fold_stage", "k_finalize", "copyBuffer") if k in n), None)
    if key:
        tot[key] = tot.get(key, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
per = {k: v / (world * sweeps) for k, v in tot.items()}
print(" | ".join("%s %.3f ms" % (k, v) for k, v in per.items()), "| kernels per shard-sweep (without copies): %.3f ms" % sum(v for k, v in per.items() if k != "copyBuffer"))
