"""Occupancy timeline from a rocprofv3 kernel trace: how much of the last step has an MFMA kernel running, only
non-MFMA kernels running, or nothing running.  Usage: python tools/timeline.py <kernel_trace.csv> [step_ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])) for r in rows]
ev.sort()
# one whole step = the interval between the last two loss kernels (backward of step k + forward of step k+1); the optional second
# argument (ms, counted back from the end of the trace) is the fall-back when the trace holds fewer than two of them
marks = sorted(e[0] for e in ev if "bce_kernel" in e[2])
if len(marks) >= 2:
    t0, t_end = marks[-2], marks[-1]
    span = float(t_end - t0)
else:
    t_end = max(e[1] for e in ev)
    span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 345e6
    t0 = t_end - span
ev = [e for e in ev if e[0] < t_end]
print("window: %.1f ms" % (span / 1e6))
ev = [e for e in ev if e[1] > t0]
def is_mfma(n): return any(k in n for k in ("gemm_fwd_kernel", "gemm_sk_kernel", "gemm_sk_fixup", "wgrad_kernel", "wgrad_split_kernel", "conv3_patch", "smalln_fwd", "smallm_wgrad", "conv1x1_small", "flash_fwd", "flash_bwd", "stem_fwd", "stem_wgrad"))
pts = []
for s, e, n, q in ev:
    s = max(s, t0)
    pts.append((s, 1, is_mfma(n))); pts.append((e, -1, is_mfma(n)))
pts.sort()
nm = no = 0; last = t0; acc = collections.Counter(); conc = collections.Counter()
for t, d, m in pts:
    dt = t - last
    if dt > 0:
        acc["mfma" if nm else ("other_only" if no else "idle")] += dt
        conc[(nm, no)] += dt
    last = t
    if m: nm += d
    else: no += d
tot = sum(acc.values())
for k, v in acc.items(): print("%-12s %8.1f ms  %.1f%%" % (k, v / 1e6, 100 * v / tot))
print("concurrency (n_mfma, n_other) -> ms")
for k, v in sorted(conc.items(), key=lambda kv: -kv[1])[:12]: print("  ", k, "%.1f" % (v / 1e6))
# phases: bucket into 20 slices, list top kernels by busy time in each
NB = 23
for b in range(NB):
    lo, hi = t0 + span * b / NB, t0 + span * (b + 1) / NB
    c = collections.Counter()
    for s, e, n, q in ev:
        o = min(e, hi) - max(s, lo)
        if o > 0: c[n.split("(")[0][:40]] += o
    top = ", ".join("%s %.1f" % (k.replace("void ", ""), v / 1e6) for k, v in c.most_common(4))
    print("[%5.0f-%5.0f ms] busy-sum %.1f : %s" % ((lo - t0) / 1e6, (hi - t0) / 1e6, sum(c.values()) / 1e6, top))

# ---- which kernels run while NO MFMA kernel is resident (the exposed non-MFMA time), and when
iv = sorted((max(s, t0), e) for s, e, n, q in ev if is_mfma(n))
merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
gaps = []
prev = t0
for s, e in merged:
    if s > prev: gaps.append((prev, s))
    prev = max(prev, e)
if prev < t_end: gaps.append((prev, t_end))
c = collections.Counter()
for s, e, n, q in ev:
    if is_mfma(n): continue
    for gs, ge in gaps:
        o = min(e, ge) - max(s, gs)
        if o > 0: c[n.split("(")[0][:48].replace("void ", "")] += o
print("exposed (no MFMA kernel resident): %.1f ms in %d gaps; kernels running then (busy ms):" % (sum(g[1] - g[0] for g in gaps) / 1e6, len(gaps)))
for k, v in c.most_common(14): print("   %-50s %.2f" % (k, v / 1e6))
big = sorted(gaps, key=lambda g: g[0] - g[1])[:12]
print("largest gaps (start ms, length ms):", [(round((g[0] - t0) / 1e6, 1), round((g[1] - g[0]) / 1e6, 2)) for g in sorted(big)])
