"""Census of the stock ATen device work left on one training step (GPU box): torch.profiler over one fwd+loss+bwd at the bench
workload, grouped by operator and input shapes.  `python tools/aten_ops.py [B]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import mmvit4  # noqa: E402
import ops  # noqa: E402
from data_parallel import GradAllReducer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).train()
reducer = GradAllReducer(model) if os.environ.get("NO_REDUCER") != "1" else None
x, mask = helpers.make_inputs(B, 4, 224, 224)
x, mask = x.to(dev), mask.to(dev)


def step():
    if reducer:
        reducer.zero_grad()
    else:
        model.zero_grad(set_to_none=True)
    loss = ops.bce_with_logits_mean(model(x), mask)
    loss.backward()
    if reducer:
        reducer.finish()


step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 and e.self_device_time_total <= 0:
        continue
    if e.self_device_time_total <= 0:
        continue
    st = ""
    for fr in (e.stack or []):
        if "corrifnet" in fr or "data_parallel" in fr or "bench" in fr or "tools/" in fr:
            st = fr.split("/")[-1][:60]
            break
    key = (e.name, str(e.input_shapes)[:70], st)
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += e.self_device_time_total
tot = 0
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%5d x %-22s %9.1f us  %-70s %s" % (v[0], k[0], v[1], k[1], k[2]))
    tot += v[0]
print("total aten device ops with self device time:", sum(v[0] for v in rows.values()))
