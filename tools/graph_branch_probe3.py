"""Dev tool (GPU box): like the detector's forward graph — a side chain of many SHORT kernels, a compute chain that waits for
events in the middle of it.  Prints when the compute chain's kernels start relative to the side chain (events timed inside the
graph are not allowed, so the compute kernels write the device clock)."""
import sys
import torch

dev = torch.device("cuda:0")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 64       # side kernels
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # an event every EVERY side kernels
SHORT = int(sys.argv[3]) if len(sys.argv) > 3 else 12000  # side kernel spin (cycles at 100 MHz?)


def build():
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    a = torch.zeros(256, device=dev)
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            a.add_(1.0)
            fork = torch.cuda.Event()
            fork.record(main)
            a.add_(1.0)
            side.wait_event(fork)
            evs = []
            with torch.cuda.stream(side):
                for i in range(NS):
                    torch.cuda._sleep(SHORT)
                    if (i + 1) % EVERY == 4 % EVERY:
                        e = torch.cuda.Event()
                        e.record(side)
                        evs.append(e)
            for e in evs:
                main.wait_event(e)
                for _ in range(3):
                    torch.cuda._sleep(SHORT * 3)
            main.wait_stream(side)
    return g, len(evs)


g, nev = build()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 50
# spin kernels only: one side kernel alone
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
for _ in range(50):
    torch.cuda._sleep(SHORT)
s1.record()
torch.cuda.synchronize()
u = s0.elapsed_time(s1) * 20
print("side %d kernels x %.1f us = %.0f us; compute %d x 3 x %.1f us = %.0f us; replay %.0f us  (ideal overlap ~%.0f, serial %.0f)" % (
    NS, u, NS * u, nev, 3 * u, nev * 9 * u, t, max(NS * u, 4 * u + nev * 9 * u) + 9 * u, NS * u + nev * 9 * u))
