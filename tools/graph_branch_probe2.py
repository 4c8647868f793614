"""Dev tool (GPU box): two hipGraphs on two streams tied by EXTERNAL events (event record / wait nodes) — does the compute
graph follow the index graph node by node?  Side graph: S1..S40 (20 us each), an external event after each; main graph: M_i
(5 us) waits for S_i's event.  One graph with an internal fork for comparison."""
import torch

dev = torch.device("cuda:0")
N = 40
CY = 40000


def one_graph():
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    b = torch.zeros(1024, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            b.add_(1.0)
            side.wait_stream(main)
            evs = []
            with torch.cuda.stream(side):
                for i in range(N):
                    torch.cuda._sleep(CY)
                    e = torch.cuda.Event()
                    e.record(side)
                    evs.append(e)
            for i in range(N):
                main.wait_event(evs[i])
                torch.cuda._sleep(CY // 4)
            main.wait_stream(side)
    return lambda: g.replay()


def two_graphs():
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event(external=True) for _ in range(N)]
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(ga, stream=side):
            for i in range(N):
                torch.cuda._sleep(CY)
                evs[i].record(side)
    with torch.cuda.stream(main):
        with torch.cuda.graph(gb, stream=main):
            for i in range(N):
                main.wait_event(evs[i])
                torch.cuda._sleep(CY // 4)

    def run():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        main.wait_stream(cur)
        with torch.cuda.stream(side):
            ga.replay()
        with torch.cuda.stream(main):
            gb.replay()
        cur.wait_stream(side)
        cur.wait_stream(main)
    return run


for name, mk in (("one graph, internal fork", one_graph), ("two graphs, external events", two_graphs)):
    try:
        f = mk()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("%-30s %.1f us per replay (side chain alone ~%d x S)" % (name, e0.elapsed_time(e1) * 100, N), flush=True)
    except Exception as ex:  # noqa: BLE001
        print(name, "FAILED:", repr(ex)[:300], flush=True)
