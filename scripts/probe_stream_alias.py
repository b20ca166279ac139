"""Does a torch.cuda.Stream() that aliases the graph-capture stream (torch hands streams out of a pool of 32 per device,
round-robin) break a capture that forks to it and joins back?  (diagnosis of a capture_end segfault, round 3)"""
import torch
a = torch.ones(1 << 20, device="cuda")
g0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g0):
    b = a * 2
cap = torch.cuda.graph.default_capture_stream
print("capture stream", hex(cap.cuda_stream))
made = []
alias = None
for i in range(70):
    s = torch.cuda.Stream()
    made.append(s)
    if s.cuda_stream == cap.cuda_stream:
        alias = s
        print("aliased after", i + 1, "Stream() constructions")
        break
print("distinct handles among", len(made), "streams:", len({s.cuda_stream for s in made}))
if alias is None:
    raise SystemExit("no alias found")
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    x = a * 2
    alias.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(alias):
        y = a * 3
    torch.cuda.current_stream().wait_stream(alias)
    z = x + y
print("captured")
g.replay(); torch.cuda.synchronize(); print("replayed", float(z[0]))
