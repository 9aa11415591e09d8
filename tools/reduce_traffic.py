"""Reduce the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh to HBM bytes per step, per kernel family.
FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section); units are KiB.
Usage: python tools/reduce_traffic.py gpurun_out/prof_<tag> <steps_in_run> > profiles/<tag>_mfma_traffic.json"""
import csv, json, sys, collections
d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
MFMA = ("gemm_fwd_kernel", "gemm_sk_kernel", "gemm_sk_fixup", "wgrad_kernel", "wgrad_split_kernel", "conv3_patch", "smalln_fwd", "smallm_wgrad", "conv1x1_small", "flash_fwd", "flash_bwd", "stem_fwd", "stem_wgrad")
def load(path, name, scale):
    fam = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        fam[k] += float(r["Counter_Value"]) * 1024.0 * scale; n[k] += 1
    return fam, n
fe, nf = load(d + "/fetch/f_counter_collection.csv", "FETCH_SIZE", 2.0)
wr, _ = load(d + "/write/w_counter_collection.csv", "WRITE_SIZE", 1.0)
is_m = lambda k: any(m in k for m in MFMA) and "final" not in k
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), CORRIF_SERIAL=1 python bench.py --steps 1 --warmup 1; "
              "FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md); bytes per step = run total / %g" % steps,
    "mfma_family_fetch_bytes_per_step": sum(v for k, v in fe.items() if is_m(k)) / steps,
    "mfma_family_write_bytes_per_step": sum(v for k, v in wr.items() if is_m(k)) / steps,
    "mfma_launches_per_step": sum(v for k, v in nf.items() if is_m(k)) / steps,
    "all_kernels_hbm_bytes_per_step": (sum(fe.values()) + sum(wr.values())) / steps,
    "per_kernel_GB_per_step": {k: round((fe[k] + wr.get(k, 0.0)) / steps / 1e9, 2) for k, _ in (fe + wr).most_common(25)},
}
out["mfma_family_hbm_bytes_per_step"] = out["mfma_family_fetch_bytes_per_step"] + out["mfma_family_write_bytes_per_step"]
# achieved HBM-side bandwidth of the HBM-bound kernel families: bytes of the counter passes / the kernel time of the un-instrumented
# serial kernel-stats run of the same command (third argument, optional), against 8 TB/s
if len(sys.argv) > 3:
    t = {}
    nsteps = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
    for r in csv.DictReader(open(sys.argv[3])):
        t[r["Name"].split("(")[0].replace("void ", "")] = float(r["TotalDurationNs"]) / nsteps
    bw = {}
    for k, _ in (fe + wr).most_common(40):
        if is_m(k) or k not in t or t[k] <= 0: continue
        gbps = (fe[k] + wr.get(k, 0.0)) / steps / t[k]
        bw[k] = {"GB_per_step": round((fe[k] + wr.get(k, 0.0)) / steps / 1e9, 2), "ms_per_step": round(t[k] / 1e6, 2), "GBps": round(gbps, 0),
                 "frac_of_8TBps": round(gbps / 8000.0, 3)}
    out["hbm_bound_kernels"] = bw
# fingerprint of the kernel sources these counters were taken with: bench.py only quotes the traffic while it still matches
import glob, hashlib, os
h = hashlib.sha1()
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "corrifnet*_amd", "csrc", "*"))):
    h.update(open(f, "rb").read())
out["csrc_sha1"] = h.hexdigest()
print(json.dumps(out, indent=1))
