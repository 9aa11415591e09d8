"""SQ_VALU_MFMA_BUSY_CYCLES per kernel family / (kernel duration x 2.4 GHz x 1024 SIMDs) = matrix-pipe utilisation.
Usage: python tools/reduce_mfma_util.py gpurun_out/prof_mfma_util/pmc/m_counter_collection.csv > profiles/r01_mfma_util.json"""
import csv, json, sys, collections
busy = collections.Counter(); dur = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES": continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    busy[k] += float(r["Counter_Value"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
CLK, SIMDS = 2.4e9, 256 * 4
rows = {}
tb = td = 0.0
for k, b in busy.most_common():
    if b <= 0: continue
    u = b / (dur[k] * 1e-9 * CLK * SIMDS)
    rows[k] = {"launches": n[k], "ms": round(dur[k] / 1e6, 2), "mfma_busy_frac": round(u, 3)}
    tb += b; td += dur[k]
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES, CORRIF_SERIAL=1 python bench.py --steps 1 --warmup 1 (2 steps in the run); "
                 "busy cycles / (kernel duration x 2.4 GHz x 1024 SIMDs); durations under the counter pass are a few % longer than un-instrumented",
       "all_mfma_kernels_busy_frac": round(tb / (td * 1e-9 * CLK * SIMDS), 3), "per_kernel": rows}
print(json.dumps(out, indent=1))
