"""Two-level stream fork inside a HIP-graph capture, reduced to raw streams / events and trivial kernels (one case per process).

    python -X faulthandler tools/probe/graph_fork2.py raw|alloc|origin|origin_lane|join1|onelane|model

raw    : capture stream C forks lane streams L0, L1; each lane forks its own side stream S; S joins L through four events (the decoder's
         skip branch pattern), L joins C.  Pre-allocated tensors, torch element-wise kernels only.
         ROCm 7.2 / torch 2.10: SEGFAULT inside hipStreamEndCapture (gpurun_out/r3a_graph_raw.log) - a runtime bug, no product code involved.
alloc  : the same topology with tensors allocated (and dropped) inside the capture on the forked streams (same segfault).
origin : S forks from the ORIGIN stream C instead of from its lane (no second fork level); it still joins its lane L.
origin_lane: S waits for the origin's event FIRST and then for an event of its lane (what Decoder_fuse.forward does).
join1  : nested fork as in `raw`, but S joins L through ONE event (its last) instead of four.
onelane: nested fork as in `raw` with a single lane.
model  : the product's own two-lane forward (MMVit4 with decoder_split = 2, batch 4) captured by train.GraphedForward.
Prints one line per completed phase, so the last line before a crash names the phase.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402,F401
import torch  # noqa: E402

dev = torch.device("cuda:0")
case = sys.argv[1]


def say(*a):
    print("[graph_fork2:%s]" % case, *a, flush=True)


def topology(alloc, origin_fork=False, one_join=False, lanes=2, then_lane=False):
    C = torch.cuda.Stream()
    L = [torch.cuda.Stream() for _ in range(lanes)]
    S = [torch.cuda.Stream() for _ in range(lanes)]
    keep = []
    bufs = [[torch.zeros(1 << 20, device=dev) for _ in range(6)] for _ in range(lanes)]

    def body():
        cur = torch.cuda.current_stream()
        outs = []
        for k in range(lanes):
            e = torch.cuda.Event(); keep.append(e); e.record(cur); L[k].wait_event(e)
            if origin_fork:
                S[k].wait_event(e)
            with torch.cuda.stream(L[k]):
                if not origin_fork or then_lane:
                    e2 = torch.cuda.Event(); keep.append(e2); e2.record(L[k]); S[k].wait_event(e2)
                evs = []
                with torch.cuda.stream(S[k]):
                    parts = []
                    for l in range(4):
                        t = (torch.empty(1 << 20, device=dev).fill_(l + 1.0) if alloc else bufs[k][l].fill_(l + 1.0))
                        parts.append(t)
                        ev = torch.cuda.Event(); keep.append(ev); ev.record(S[k]); evs.append(ev)
                y = (torch.zeros(1 << 20, device=dev) if alloc else bufs[k][4].zero_())
                for l in range(4):
                    if not one_join:
                        L[k].wait_event(evs[l])
                    elif l == 0:
                        L[k].wait_event(evs[3])
                    y = y + parts[l] if alloc else y.add_(parts[l])
                outs.append(y)
                del parts
        for k in range(lanes):
            e = torch.cuda.Event(); keep.append(e); e.record(L[k]); cur.wait_event(e)
        return outs[0] + outs[-1]

    with torch.cuda.stream(C):
        ref = body().clone()
        torch.cuda.synchronize()
        say("eager ok", float(ref[0]))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=C):
            out = body()
        say("capture ended")
        g.replay()
        torch.cuda.synchronize()
        say("replay ok", bool(torch.equal(out, ref)))
        again = body()
        torch.cuda.synchronize()
        say("eager after capture ok", bool(torch.equal(again, ref)))


if case in ("raw", "alloc", "origin", "origin_lane", "join1", "onelane"):
    topology(case == "alloc", origin_fork=case.startswith("origin"), one_join=case == "join1", lanes=1 if case == "onelane" else 2,
             then_lane=case == "origin_lane")
else:
    import mmvit4
    import train
    torch.manual_seed(0)
    model = mmvit4.MMVit4().to(dev).eval()
    model.decoder_split = 2
    mmvit4.CAPTURE_LANES = True
    x, _ = helpers.make_inputs(4, 3, 64, 64)
    x = x.to(dev)
    with torch.no_grad():
        ref = model(x).clone()
    torch.cuda.synchronize()
    say("eager ok")
    g = train.GraphedForward(model, x)
    say("capture ended")
    o = g(x)
    torch.cuda.synchronize()
    say("replay ok", bool(torch.equal(o, ref)))
    with torch.no_grad():
        o2 = model(x)
    torch.cuda.synchronize()
    say("eager after capture ok", bool(torch.equal(o2, ref)))
