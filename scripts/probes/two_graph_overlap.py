"""Do two captured HIP graphs on two streams overlap on this runtime, and what does the event hand-shake of the trainer's geometry
prefetch cost?  A = a long chain of matrix-core kernels (dense step stand-in), B = a chain of small latency-bound kernels
(geometry stand-in).  Variants: A alone; A then B on one stream; B on a second stream with no synchronisation (legal here: B is
independent); B with the trainer's per-step event pattern; the same with persistent events."""
import os
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402


def main():
    dev = "cuda"
    x = torch.randn(8192, 1024, device=dev)
    w = torch.randn(1024, 1024, device=dev) * 0.03
    small = torch.randn(32, 1024, 3, device=dev)
    outA, outB = [None], [None]

    def work_a():
        y = x
        for _ in range(40):
            y = torch.relu(y @ w)
        outA[0] = y

    def work_b():
        z = small
        for _ in range(60):  # ~5 us launches in a dependent chain on a few workgroups
            z = z * 1.0001 + 0.1
        outB[0] = z

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    graphs = {}
    for name, fn, st in (("A", work_a, s1), ("B", work_b, s2)):
        with torch.cuda.stream(st):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
        graphs[name] = g
    torch.cuda.synchronize()
    pack_g, pack_d = torch.zeros(2 << 20, device=dev), torch.zeros(2 << 20, device=dev)
    ev_done, ev_cons, ev_end = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
    ev_end.record(s1)

    def run(variant, n=40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if variant == "A":
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
            elif variant == "A+B serial":
                with torch.cuda.stream(s1):
                    graphs["B"].replay()
                    graphs["A"].replay()
            elif variant == "B on S2, no sync":
                with torch.cuda.stream(s2):
                    graphs["B"].replay()
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
            elif variant in ("events (new each step)", "events (persistent)"):
                with torch.cuda.stream(s1):
                    s1.wait_event(ev_done)
                    pack_d.copy_(pack_g, non_blocking=True)
                    cons = torch.cuda.Event() if variant.startswith("events (new") else ev_cons
                    cons.record(s1)
                with torch.cuda.stream(s2):
                    s2.wait_event(cons)
                    graphs["B"].replay()
                    ev_done.record(s2)
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
            elif variant == "only S1 waits B's event":
                with torch.cuda.stream(s1):
                    s1.wait_event(ev_done)
                    pack_d.copy_(pack_g, non_blocking=True)
                with torch.cuda.stream(s2):
                    graphs["B"].replay()
                    ev_done.record(s2)
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
            elif variant == "only S2 waits S1's event":
                with torch.cuda.stream(s1):
                    pack_d.copy_(pack_g, non_blocking=True)
                    ev_cons.record(s1)
                with torch.cuda.stream(s2):
                    s2.wait_event(ev_cons)
                    graphs["B"].replay()
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
            elif variant == "copy only, no events":
                with torch.cuda.stream(s1):
                    pack_d.copy_(pack_g, non_blocking=True)
                with torch.cuda.stream(s2):
                    graphs["B"].replay()
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
            elif variant == "S1 records, nobody waits":
                with torch.cuda.stream(s1):
                    pack_d.copy_(pack_g, non_blocking=True)
                    ev_cons.record(s1)
                with torch.cuda.stream(s2):
                    graphs["B"].replay()
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
            elif variant == "S2 waits the END of the previous step (+ S1 waits B)":
                with torch.cuda.stream(s1):
                    s1.wait_event(ev_done)
                    pack_d.copy_(pack_g, non_blocking=True)
                with torch.cuda.stream(s2):
                    s2.wait_event(ev_end)        # recorded after the previous step's A
                    graphs["B"].replay()
                    ev_done.record(s2)
                with torch.cuda.stream(s1):
                    graphs["A"].replay()
                    ev_end.record(s1)
            elif variant == "events, A launched first":
                with torch.cuda.stream(s1):
                    s1.wait_event(ev_done)
                    pack_d.copy_(pack_g, non_blocking=True)
                    ev_cons.record(s1)
                    graphs["A"].replay()
                with torch.cuda.stream(s2):
                    s2.wait_event(ev_cons)
                    graphs["B"].replay()
                    ev_done.record(s2)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    ev_done.record(s2)
    for v in ("A", "A+B serial", "B on S2, no sync", "copy only, no events", "only S1 waits B's event", "only S2 waits S1's event", "S1 records, nobody waits",
              "S2 waits the END of the previous step (+ S1 waits B)",
              "events (new each step)", "events (persistent)", "events, A launched first"):
        run(v, 5)
        print("%-55s %.3f ms / step" % (v, run(v)))


if __name__ == "__main__":
    main()
