"""GPU busy / idle analysis of a rocprofv3 kernel trace (csv): the last complete step = the kernels between the last two
launches of a delimiting kernel (the optimizer update by default).  Prints wall time, the union of kernel intervals (busy), time with >= 2 kernels running,
and the largest idle gaps with the kernels around them."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
qs = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])): (r.get("Queue_Id"), r.get("Stream_Id")) for r in rows}
ks.sort()
# step delimiter: the optimizer update kernel (two launches per training step: decay / no-decay groups), or - third argument - a
# kernel that runs once per step (e.g. "decode_kernel" for the inference plan)
marker = sys.argv[3].lower() if len(sys.argv) > 3 else "adamw"
per_step = 2 if marker == "adamw" else 1
upd = [i for i, k in enumerate(ks) if marker in k[2].lower()]
if len(upd) < 2 * per_step:
    sys.exit("not enough step-delimiting launches in the trace")
a, b = upd[-1 - per_step], upd[-1]
step = ks[a + 1:b + 1]
t0, t1 = step[0][0], max(k[1] for k in step)
print(f"kernels in the step: {len(step)}  wall {(t1 - t0) / 1e6:.3f} ms  sum of durations {sum(k[1] - k[0] for k in step) / 1e6:.3f} ms")
ev = []
for s, e, _ in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = multi = 0
depth = 0
last = ev[0][0]
gaps = []
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    if depth == 0 and t > last: gaps.append((t - last, last, t))
    depth += d
    last = t
print(f"busy {busy / 1e6:.3f} ms  idle {(t1 - t0 - busy) / 1e6:.3f} ms  >=2 kernels {multi / 1e6:.3f} ms")
hist = {}
for g, *_ in gaps:
    b_ = "<1us" if g < 1000 else "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else ">=10us"
    h = hist.setdefault(b_, [0, 0]); h[0] += 1; h[1] += g
print("idle gaps:", {k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
gaps.sort(reverse=True)
for g, s, e in gaps[:25]:
    before = max((k for k in step if k[1] <= s), key=lambda k: k[1], default=None)
    after = min((k for k in step if k[0] >= e), key=lambda k: k[0], default=None)
    print(f"gap {g / 1e3:7.1f} us at +{(s - t0) / 1e6:6.2f} ms  after {before[2][:60] if before else None}  before {after[2][:60] if after else None}")

# exclusive time: intervals with exactly ONE kernel running, by kernel name (the serial part of the timeline)
import collections
evs = []
for i, (s_, e_, _) in enumerate(step):
    evs.append((s_, 1, i)); evs.append((e_, -1, i))
evs.sort()
running = set()
excl = collections.Counter()
cnt = collections.Counter()
last = evs[0][0]
for t, d, i in evs:
    if len(running) == 1:
        (j,) = running
        excl[step[j][2]] += t - last
    if d == 1: running.add(i)
    else: running.discard(i)
    last = t
for _, _, nm in step: cnt[nm] += 1
tot = sum(excl.values())
print(f"exclusive (one kernel running) {tot / 1e6:.3f} ms:")
for nm, v in excl.most_common(22):
    durs = [e_ - s_ for s_, e_, n_ in step if n_ == nm]
    print(f"  {v / 1e6:6.3f} ms  x{cnt[nm]:4d}  avg {sum(durs) / len(durs) / 1e3:6.1f} us  {nm[:100]}")

# hardware queues / streams the step's kernels ran on
qc = collections.Counter(qs[(s_, e_)] for s_, e_, _ in step)
print("(queue, stream) -> kernels:", dict(qc))

# compact timeline of the step for offline inspection: start_us,dur_us,queue,kernel
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        for s_, e_, nm in step:
            fh.write(f"{(s_ - t0) / 1e3:.1f},{(e_ - s_) / 1e3:.1f},{qs[(s_, e_)][0]},{nm[:70]}\n")
