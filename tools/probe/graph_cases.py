"""HIP-graph capture cases, one per process (a crash in one does not hide the others): python tools/probe/graph_cases.py A|B|C"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, train
case = sys.argv[1]
dev = "cuda:0"
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).eval()
def run(B, split):
    model.decoder_split = split
    x, _ = helpers.make_inputs(B, 3, 64, 64); x = x.to(dev)
    with torch.no_grad():
        ref = model(x).clone()
    g = train.GraphedForward(model, x)
    ok = torch.equal(g(x), ref)
    with torch.no_grad():
        ok = ok and torch.equal(model(x), ref)
    torch.cuda.synchronize()
    return g, ok
if case == "A":      # B=4, lanes off, first capture
    g, ok = run(4, 0); print("A B=4 no lanes:", ok, flush=True)
elif case == "B":    # two multi-stream captures at B=1
    g1, ok1 = run(1, 2); g2, ok2 = run(1, 2); print("B two captures B=1:", ok1, ok2, flush=True)
elif case == "C":    # B=4 with two lanes, first capture
    g, ok = run(4, 2); print("C B=4 two lanes:", ok, flush=True)
elif case == "D":    # B=1 capture then B=4 without lanes
    g1, ok1 = run(1, 2); g2, ok2 = run(4, 0); print("D B=1 then B=4 no lanes:", ok1, ok2, flush=True)
