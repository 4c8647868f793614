"""Dev tool (GPU box): when does hipGraph replay start a branch that depends, node by node, on another branch?
Captures  main: A | side: S1..S8 (each records an event) | main: M_i waits for S_i's event  in two capture ORDERS —
all of the side stream first (what spx.prebuild does) or interleaved — and prints the replay timeline from
`rocprofv3 --kernel-trace` (run this under it; the kernels are torch's _sleep spin kernels and fills of distinct sizes).

rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/graph_branch_probe.py
"""
import sys
import torch

dev = torch.device("cuda:0")
N = 8
CY = 200000          # ~100 us of spinning per side-stream node


def build(interleaved):
    main = torch.cuda.Stream()
    side = torch.cuda.Stream()
    bufs = [torch.zeros(1024 * (i + 1), device=dev) for i in range(2 * N + 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main):
            bufs[0].add_(1.0)                                  # A
            side.wait_stream(main)
            evs = []

            def s_node(i):
                with torch.cuda.stream(side):
                    torch.cuda._sleep(CY)
                    bufs[1 + i].add_(1.0)                      # S_i marker (distinct size)
                    e = torch.cuda.Event()
                    e.record(side)
                    evs.append(e)

            def m_node(i):
                main.wait_event(evs[i])
                torch.cuda._sleep(CY // 4)
                bufs[1 + N + i].add_(1.0)                      # M_i marker

            if interleaved:
                for i in range(N):
                    s_node(i)
                    m_node(i)
            else:
                for i in range(N):
                    s_node(i)
                for i in range(N):
                    m_node(i)
            main.wait_stream(side)
    return g, bufs


for mode in (False, True):
    g, bufs = build(mode)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("capture order %-12s: %.1f us per replay" % ("interleaved" if mode else "side first", e0.elapsed_time(e1) * 100), flush=True)
