#!/usr/bin/env python3
"""Do the parallel branches of a captured hipGraph run concurrently on this runtime?

A graph of B independent chains (one stream each, forked from / joined to the capture stream) of K spin kernels of ONE workgroup
(torch.cuda._sleep: ~T us each).  Concurrent replay: ~K T per replay; chains one after the other: ~B K T.  Variants: the chain on the
capture stream issued first or last; more chains than hardware queues; the same chains launched WITHOUT a graph on plain streams."""
import sys
import time

import torch


def build(branches, k, cycles, main_first, dev):
    main = torch.cuda.Stream(device=dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(branches - 1)]

    def body():
        cur = torch.cuda.current_stream(dev)
        torch.cuda._sleep(cycles // 4)  # the fork point
        if main_first:
            for _ in range(k):
                torch.cuda._sleep(cycles)
        for s in side:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(k):
                    torch.cuda._sleep(cycles)
        if not main_first:
            for _ in range(k):
                torch.cuda._sleep(cycles)
        for s in side:
            cur.wait_stream(s)
        torch.cuda._sleep(cycles // 4)  # the join
    return main, body


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
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    k = 4
    # one kernel alone
    t0 = timed(lambda: torch.cuda._sleep(cycles), dev, 50)
    print(f"one spin kernel of {cycles} cycles: {t0:.1f} us per launch (back to back on one stream)")
    for branches in (1, 2, 4, 8):
        for main_first in (True, False):
            main_s, body = build(branches, k, cycles, main_first, dev)
            with torch.cuda.stream(main_s):
                body()  # warm
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=main_s):
                    body()
                tg = timed(g.replay, dev)
                te = timed(body, dev)
            print(f"{branches} chains x {k} kernels, capture-stream chain {'first' if main_first else 'last '}: graph replay {tg:7.1f} us, "
                  f"eager streams {te:7.1f} us   (concurrent ~ {k * t0 + t0 / 2:.0f}, serial ~ {branches * k * t0 + t0 / 2:.0f})")


if __name__ == "__main__":
    main()
