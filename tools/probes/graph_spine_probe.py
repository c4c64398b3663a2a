#!/usr/bin/env python3
"""The shape of an HRModule's backward in a captured graph: a SPINE of short kernels on the capture stream (the fan-in sums of the
branches, one after the other), each of which releases one branch chain - chain 0 continues on the capture stream right behind the
spine, chains 1..B-1 run on side streams that wait for an event recorded behind "their" spine kernel.  Do the side chains start
when their spine kernel is done (concurrent: ~spine + one chain) or behind chain 0 (serial)?  Variants: chain 0 on a side stream
too (the capture stream only carries the spine); every chain released by ONE event recorded behind the whole spine (a single fork
point)."""
import sys

import torch


def timed(fn, dev, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    k, branches = 8, 4
    t1 = timed(lambda: torch.cuda._sleep(cycles), dev, 50)
    print(f"spin kernel {cycles} cycles: {t1:.1f} us; spine kernels are a quarter of that")
    for chain0_on_side, one_event in ((False, False), (True, False), (False, True), (True, True)):
        main_s = torch.cuda.Stream(device=dev)
        side = [torch.cuda.Stream(device=dev) for _ in range(branches)]

        def body():
            cur = torch.cuda.current_stream(dev)
            evs = []
            for b in range(branches - 1, -1, -1):  # spine: fan-in of branch 3, 2, 1, 0
                torch.cuda._sleep(cycles // 4)
                ev = torch.cuda.Event()
                ev.record(cur)
                evs.append((b, ev))
            if one_event:  # every chain is released by the END of the spine: one fork point
                evs = [(b, evs[-1][1]) for b, _ in evs]
            for b, ev in evs:
                if b == 0 and not chain0_on_side:
                    continue
                s = side[b]
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    for _ in range(k):
                        torch.cuda._sleep(cycles)
            if not chain0_on_side:
                for _ in range(k):
                    torch.cuda._sleep(cycles)
            for b, _ in evs:
                if b == 0 and not chain0_on_side:
                    continue
                cur.wait_stream(side[b])
            torch.cuda._sleep(cycles // 4)

        with torch.cuda.stream(main_s):
            body()
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_s):
                body()
            tg = timed(g.replay, dev)
            te = timed(body, dev)
        print(f"{'ONE event behind the spine' if one_event else 'an event per spine kernel '}, chain 0 on {'a side stream' if chain0_on_side else 'the capture stream'}: graph {tg:7.1f} us, eager {te:7.1f} us   "
              f"(concurrent ~ {t1 * (k + 1.25):.0f}, serial ~ {t1 * (branches * k + 1.25):.0f})")


if __name__ == "__main__":
    main()
