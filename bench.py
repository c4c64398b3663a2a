"""Benchmark of the hot path: HRNet-W32 256x192 top-down inference (BASELINE.json metric / configs[2]).

    python bench.py --gpus N --steps K --warmup W
        N > 1: one rank per GPU over RCCL.  Either the driver launches the ranks itself (torch.distributed.run: RANK /
        WORLD_SIZE in the environment), or - when no WORLD_SIZE is set - this process starts them as children through
        torch.distributed.run BEFORE it makes any GPU call and exits with their status.

A step = one pass of the hot path over one per-GPU batch of synthetic 256x192 crops that is already
resident in HBM: HRNet-W32 backbone + HRNetHead (direct fp32-MFMA conv plan) + TopDownHeatMapDecoder
(arg-max + +-0.25 shift + back-projection).  Crops shard over ranks with no data-path collective
(weak scaling); value = images all ranks processed / max-over-ranks wall time.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     - the dominant kernel (the conv instantiation with the largest share of step time):
                 algorithmic FLOP per launch / its average launch duration, measured live with HIP
                 events on the launch stream; peak = fp32 matrix-core rate of MI355X_MICROARCH.md.
  cpu_baseline - the CPU oracle (torch-CPU fp32 restatement; the literal MindSpore-CPU path cannot run:
                 MindSpore is not installable here or on the GPU box) timed per SURVEY.md 8(d) on the host cores
                 (3 warm-up + 10 timed iterations at N=1 and N=32, threads = the process's CPU affinity), plus the reference's
                 own single-core numpy target loop next to mp_gaussian_target; rank 0 at N=1 only.
  extra_workloads (N=1, after the headline's timed region) - short legs, each in a child process with its own timed region and
                 roofline: HRNet-W32 under amp O2, BASELINE configs[4] (W48 384x288 UDP/DARK + flip test, fp16), configs[1]
                 (SimpleBaseline-R50), and the configs[3] training step in fp32 and in the reference's amp-O2 recipe.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_FP16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense" (--amp O2 runs only)
PEAK_HBM_GBPS = 8000.0  # same guide, "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured copy)
F16_VARIANT_TEMPLATE = {0: "3,2,4,1", 1: "3,4,4,1", 2: "3,3,4,1", 3: "3,2,2,2", 4: "3,1,2,2", 5: "3,2,4,1", 6: "3,4,4,1",
                        7: "3,3,4,1", 8: "3,2,2,2", 9: "3,1,2,2"}
VARIANT_TEMPLATE = {0: "3,2,4,1", 1: "3,4,4,1", 2: "3,3,4,1", 3: "3,2,2,2", 4: "3,1,2,2", 5: "3,2,2,2", 6: "3,2,4,1",
                    7: "3,1,2,2"}  # PS,CS,WAVES_P,WAVES_C


# name -> (backbone, head, image HxW, decoder kwargs, flip test, description)
WORKLOADS = {
    "hrnet_w32": ("hrnet_w32", "hrnet_head", (256, 192), dict(shift_coordinate=True), False,
                  "configs[2]: HRNet-W32 256x192 inference, 1xMI355X per rank, multi-branch conv + fuse layers on fp32 MFMA, "
                  "HRNetHead, arg-max+shift decode"),
    "simplebaseline_r50": ("resnet50", "simple_baseline_head", (256, 192), dict(shift_coordinate=True), False,
                           "configs[1]: SimpleBaseline ResNet-50 256x192 inference, HIP deconv head (4 sub-pixel phase convs) + "
                           "arg-max+shift decode"),
    "hrnet_w48_384_udp_flip": ("hrnet_w48", "hrnet_head", (384, 288),
                               dict(use_udp=True, dark_udp_refine=True, kernel_size=17), True,
                               "configs[4] shape in fp32: HRNet-W48 384x288, flip-test aggregation fused with UDP/DARK decode "
                               "(two forwards per crop)"),
}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def kernel_name(info):
    """Template head <KS,S,PS,CS,WAVES_P,WAVES_C> of the instantiation; '/occ3' marks the light build (rocprof shows it as
    the trailing template argument OCC = 3)."""
    if info["kind_id"] == 9:  # fp32 Winograd F(2x2,3x3): <NI staging units per thread, QROW = 16-byte epilogue (full 48-tile bands,
        # TW % 4 == 0), GROUP = image-grouped bands (W % 4 != 0), TEAMS> - the rocprofv3 template list
        group = info["w"] % 4 != 0
        tw = info["w"] // 2
        tr = min(48 // tw, info["h"] // 2)
        qrow = not group and tr * tw == 48 and tw % 4 == 0 and info["h"] % (2 * tr) == 0
        teams = max(1, info.get("cout_tile", 32) // 32)  # two four-wave teams: a 64-channel cout tile per workgroup
        return f"conv_wino_f32_kernel<{info.get('light', 0)},{'true' if qrow else 'false'},{'true' if group else 'false'},{teams}>"
    if info["kind_id"] == 12:  # fp32 first conv as a streaming kernel (stem_f32.hip)
        return "stem_conv_f32_kernel"
    if info["kind_id"] == 11:  # first conv straight from the fp32 image (stem_f16.hip)
        return "stem_conv_f16_kernel"
    if info["kind_id"] == 13:  # fp32 expand + reduce chain of stage 1, persistent weight-stationary launch (pwchain_f32.hip)
        form, v = info.get("light", 4), info.get("variant", 0)  # entry_info: light = the form (waves / tile), variant = down-sample + 2 * no reduce
        if form == 4:
            return f"expand_reduce_f32_kernel<{'true' if v & 1 else 'false'},{'false' if v & 2 else 'true'}>"
        return f"expand_reduce_f32_w8_kernel<{'false' if v & 2 else 'true'},{2 if form == 8 else 1}>"
    if info["kind_id"] == 10:  # expand conv of a Bottleneck + reduce conv of the next one in one launch (pwchain_f16.hip)
        return {1: "expand_reduce_f16_kernel<64,256,64,true,false>", 2: "expand_reduce_f16_kernel<64,256,64,false,true>"}.get(
            info.get("variant"), "expand_reduce_f16_kernel<64,256,64,false,false>")  # dual 1x1 / down-sample inside / identity chain
    if info["kind_id"] == 8:  # fused fp16 BasicBlock: the <5,3> or <6,5> pixel-tile build ("variant" = 1 for the small one)
        return {0: "basicblock_f16_kernel<6,5>", 1: "basicblock_f16_kernel<5,3>", 2: "basicblock_f16_v2_kernel<8,4,3>",
                4: "basicblock_f16_c64_kernel<2,8,6>", 5: "basicblock_f16_c64_kernel<4,4,3>"}[info["variant"]]  # 4 / 5: 64 / 128 channels
    if info["kind_id"] == 3:
        v = info["variant"]
        if 37 <= v < 45:  # weight-stationary persistent kernel <k-steps, cout tiles per wave, pixel-splitting waves, pixel tiles, waves/SIMD
            # bound> (csrc/conv_f16_ws.hip kWsShapes; rocprof lists two more arguments: statistics mode, residual)
            nq, csw, wp, ps, occ = {37: (1, 2, 4, 3, 2), 38: (2, 3, 4, 5, 1), 39: (2, 4, 4, 3, 1), 40: (2, 2, 4, 3, 1), 41: (3, 2, 4, 5, 1),
                                    42: (3, 3, 4, 3, 1), 43: (4, 2, 2, 3, 1), 44: (4, 2, 1, 6, 1)}[v]
            return f"conv_f16_ws_kernel<{nq},{csw},{wp},{ps},{occ}>"
        if v >= 25:  # weights-in-registers kernel <KS, pixel tiles, cout tiles per wave, waves/SIMD bound>
            ps, csw, wp, occ = {45: (7, 3, 1, 1), 46: (4, 3, 1, 1), 47: (5, 4, 1, 1), 25: (6, 2, 1, 1), 26: (3, 2, 1, 2), 27: (6, 1, 1, 2), 28: (6, 3, 1, 1), 29: (3, 3, 1, 1), 30: (3, 4, 1, 1),
                                31: (6, 2, 2, 2), 32: (6, 3, 2, 1), 33: (3, 2, 2, 2), 34: (6, 2, 4, 2), 35: (6, 3, 4, 1), 36: (3, 2, 4, 2)}[v]
            return f"conv_f16_wreg_kernel<{info['ks']},{info['stride']},{ps},{csw},{wp},{occ}>"  # = the rocprof template list
        if v == 24:  # 32 couts x 384 pixels, single-chunk build
            return f"conv_f16_kernel<{info['ks']},{info['stride']},6,2,4,1>"
        if v >= 20:  # 16-cout tiles: regular, light, multi-tile (2 / 1 workgroups per CU)
            head = f"<{info['ks']},{info['stride']},3,1,4,1>"
            return (f"conv_f16_kernel{head}" + ("/occ3" if v == 21 else "")) if v < 22 else f"conv_f16_mt_kernel{head}/occ{24 - v}"
        if v >= 10:
            return f"conv_f16_mt_kernel<{info['ks']},{info['stride']},{F16_VARIANT_TEMPLATE[v % 5]}>/occ{1 if v >= 15 else 2}"
        return f"conv_f16_kernel<{info['ks']},{info['stride']},{F16_VARIANT_TEMPLATE[v]}>" + ("/occ3" if info.get("light") else "")
    if info["variant"] in (11, 12):  # K-split kernel for small problems (conv_small_f32.hip): <KS, S, 16-pixel tiles per workgroup>
        return f"conv_small_f32_kernel<{info['ks']},{info['stride']},{info.get('pixel_tile', 16) // 16}>"
    if info["variant"] == 10:  # blocked-GEMM kernel <GATHER (stride 2 / 2x2 phases; entry_info: light), 32-column blocks per wave>
        return f"conv1x1_f32_gemm_kernel<{'true' if info.get('light', 0) else 'false'},{info.get('pixel_tile', 128) // 64},{info.get('cout_tile', 128) // 64}>"
    if info["variant"] == 8:  # streaming 1x1 kernel <Cin / 4, cout blocks per wave> (entry_info: light = Cin / 4, images_per_tile = CBW)
        return f"conv1x1_f32_stream_kernel<{info.get('light', 0)},{info.get('images_per_tile', 0)}>"
    return f"conv_mfma_kernel<{info['ks']},{info['stride']},{VARIANT_TEMPLATE[info['variant']]}>" + ("/occ3" if info.get("light") else "")


def time_plan_entries(plan, reps):
    """Average device time of every plan entry, HIP events on the current (launch) stream."""
    n = len(plan)
    out = []
    stream = torch.cuda.current_stream()
    for i in range(n):
        plan.run_range(i, 1)  # warm
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            plan.run_range(i, 1)
        e1.record(stream)
        e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e-3 / reps)
    return out


def pmc_traffic(kernel, family=""):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary OF THE SAME WORKLOAD FAMILY
    (profiles/<tag>_pmc_traffic.json with <tag> ending in `_o2` for the fp16 plans, `_sb` for SimpleBaseline, neither for the fp32
    HRNet headline: the same instantiation runs other shapes in another network; bench.py cannot collect PMC counters itself).
    None when no such summary names the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        tag = os.path.basename(path)[:-len("_pmc_traffic.json")]
        fam = "_o2" if tag.endswith("_o2") else "_sb" if tag.endswith("_sb") else ""
        if fam != family:
            continue
        try:
            with open(path) as f:
                k = json.load(f)["kernels"].get(kernel)
            if k:
                return k["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def roofline_report(plan, reps=5, layers_csv="", workload=""):
    peak = PEAK_FP16_MFMA_TFLOPS if plan.half else PEAK_FP32_MFMA_TFLOPS
    per_entry = time_plan_entries(plan, reps)
    if layers_csv:
        with open(layers_csv, "w") as f:
            f.write("index,kind,kernel,us,tflops,n,cin,cout,k,stride,h,w,workgroups,lds_bytes,cin_chunk,images_per_tile,rows_per_tile\n")
            for i, t in enumerate(per_entry):
                e = plan.entry_info(i)
                name = kernel_name(e) if e["kind_id"] in (0, 3, 8, 9, 10, 11, 12) else e["kind"]
                tf = 2.0 * e.get("macs", 0) / t / 1e12 if t > 0 else 0.0
                f.write(f"{i},{e['kind']},\"{name}\",{t * 1e6:.2f},{tf:.2f},{e.get('n', '')},{e.get('cin', e.get('c', ''))},"
                        f"{e.get('cout', '')},{e.get('k', '')},{e.get('stride', '')},{e.get('h', '')},{e.get('w', '')},"
                        f"{e['workgroups']},{e['lds_bytes']},{e['cin_chunk']},{e['images_per_tile']},{e['rows_per_tile']}\n")
    # Winograd F(2x2,3x3) (plan entry kind 9) executes 16 of the 36 multiplies of the direct form: its matrix-pipe work is
    # 2 * MACs * 16 / 36.  `achieved` stays ALGORITHMIC FLOP over the measured time; `peak` is the ceiling of THAT
    # quantity for the kernel's algorithm (157.3 TFLOP/s x 2.25 for a Winograd kernel), so `frac` = executed MFMA FLOP/s over the
    # MFMA peak - a utilisation that cannot exceed 1.  `mfma_tflops` is the executed rate itself.
    WINO_RATIO = 36.0 / 16.0
    groups = {}
    for i, t in enumerate(per_entry):
        info = plan.entry_info(i)
        if info["kind_id"] not in (0, 3, 8, 9, 10, 11, 12):
            continue
        g = groups.setdefault(kernel_name(info), dict(time=0.0, flops=0.0, exec_flops=0.0, launches=0, wino=info["kind_id"] == 9))
        g["time"] += t
        g["flops"] += 2.0 * info["macs"]
        g["exec_flops"] += 2.0 * info["macs"] / (WINO_RATIO if info["kind_id"] == 9 else 1.0)
        g["launches"] += 1
    dom = max(groups, key=lambda k: groups[k]["time"])
    g = groups[dom]
    fam_t = sum(v["time"] for v in groups.values())
    fam_f = sum(v["flops"] for v in groups.values())
    fam_x = sum(v["exec_flops"] for v in groups.values())
    achieved = g["flops"] / g["time"] / 1e12
    executed = g["exec_flops"] / g["time"] / 1e12
    kernel_peak = peak * (WINO_RATIO if g["wino"] else 1.0)
    traffic, traffic_source = pmc_traffic(dom, "_o2" if plan.half else "_sb" if workload.startswith("simplebaseline") else "" if workload == "hrnet_w32" else "-")
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": round(kernel_peak, 1), "unit": "TFLOP/s",
        "frac": round(achieved / kernel_peak, 4),
        "algorithm": "winograd_f2x2_3x3" if g["wino"] else "direct",
        "effective_tflops": round(achieved, 2), "mfma_tflops": round(executed, 2), "mfma_peak": peak,
        "peak_note": ("achieved = algorithmic FLOP (2 x MACs of the convolution) / measured time; peak = fp32 MFMA peak x 2.25, the "
                      "ceiling of that quantity for F(2x2,3x3) (16 of 36 multiplies executed); frac = mfma_tflops / mfma_peak"
                      if g["wino"] else "achieved = algorithmic = executed FLOP; peak = dense MFMA peak of the dtype"),
        # HBM bytes per launch of the dominant kernel (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes; the committed
        # summary named in traffic_source - bench.py cannot collect PMC counters itself); null when no summary names the kernel
        "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_source,
        "timing": "HIP events around each launch replayed alone on one stream (mp_plan_run_range); the timed region of "
                  "`value` overlaps the four HRNet branch lanes, which stretches individual kernels but shortens the step",
        "kernel": dom, "launches_per_step": g["launches"],
        "flop_per_launch": round(g["flops"] / g["launches"]), "avg_launch_us": round(g["time"] / g["launches"] * 1e6, 2),
        "share_of_conv_time": round(g["time"] / fam_t, 3),
        # all conv launches of the step: `achieved` algorithmic, `frac` = EXECUTED matrix-pipe FLOP/s over the MFMA peak
        "all_conv_launches": {"launches_per_step": sum(v["launches"] for v in groups.values()),
                              "achieved": round(fam_f / fam_t / 1e12, 2), "mfma_tflops": round(fam_x / fam_t / 1e12, 2),
                              "frac": round(fam_x / fam_t / 1e12 / peak, 4),
                              "effective_frac": round(fam_f / fam_t / 1e12 / peak, 4),
                              "sum_launch_ms": round(fam_t * 1e3, 3)},
        "per_kernel": {k: {"launches": v["launches"], "ms": round(v["time"] * 1e3, 3),
                           "tflops": round(v["flops"] / v["time"] / 1e12, 2),
                           **({"mfma_tflops": round(v["exec_flops"] / v["time"] / 1e12, 2)} if v["wino"] else {})}
                       for k, v in sorted(groups.items())},
    }


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Threads the CPU oracle may use: the process's CPU affinity, limited by the cgroup CPU quota of the container (a GPU box
    shows every core of the host in the affinity mask - 256 on the EPYC 9575F boxes - but grants one GPU's share of them; running
    256 threads inside a 16-core quota took minutes per forward).  Returns (threads, description)."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = float(f1.read()), float(f2.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    cores = affinity
    note = f"len(os.sched_getaffinity(0))={affinity}"
    if quota is not None:
        cores = max(1, min(affinity, int(round(quota))))
        note += f", cgroup cpu quota={quota:.1f}"
    elif affinity > 32:
        cores = 16  # no visible quota on a many-core host: the pool documents a 16-core share per GPU
        note += ", no cgroup quota visible: capped to the pool's documented 16-core share per GPU"
    env = os.environ.get("MINDPOSE_BENCH_CPU_THREADS")
    if env:
        cores = max(1, int(env))
        note += f", MINDPOSE_BENCH_CPU_THREADS={cores}"
    return cores, note


def graph_time(fn, dev, reps=50, warm=5):
    """Device seconds per call of ``fn``: ``reps`` calls captured once and replayed as ONE hipGraph between two events (a Python ->
    ctypes call costs the host ~9 us, more than the HBM-bound kernels of the path run: back-to-back calls would time the host)."""
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(reps):
                fn()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        e1.synchronize()
    torch.cuda.current_stream(dev).wait_stream(side)
    return e0.elapsed_time(e1) * 1e-3 / reps


def cpu_baseline(state_dict, mp, dev):
    """the CPU oracle (same graph, torch-CPU fp32) on the host cores of this box - threads = the CPU affinity
    of the process, 3 warm-up + 10 timed iterations at N=1 and N=32 (HRNet-W32 256x192 forward + decode) - and the reference's
    own target loop (numpy, 17 joints x N samples, ONE core: topdown_transform.py:346-369) beside mp_gaussian_target."""
    from oracle import decoder as od
    from oracle import nets as onets
    from oracle import target as otarget
    cores, cores_note = host_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    budget_left = [45.0]  # hard bound on the HRNet leg; the protocol's 13 iterations fit it on the 16-core share of a GPU box

    def time_oracle(sd, backbone, head, what):
        out = {}
        for batch in (1, 32):
            x = torch.randn(batch, 3, 256, 192, generator=g)
            center = np.full((batch, 2), [96.0, 128.0], dtype=np.float32)
            scale = np.full((batch, 2), [0.96, 1.28], dtype=np.float32)
            score = np.ones(batch, dtype=np.float32)

            def one():
                hm = onets.net_forward(sd, x, backbone, head).numpy()
                od.decode(hm, center, scale, score, shift_coord=True)

            t0 = time.perf_counter()
            one()
            first = time.perf_counter() - t0
            n_warm = 3 if first * 12 <= budget_left[0] else 1  # a host far slower than expected: keep the leg bounded, say so below
            for _ in range(n_warm - 1):
                one()
            warm = (time.perf_counter() - t0) / n_warm
            iters = 10 if warm * 10 <= budget_left[0] else max(1, int(budget_left[0] / max(warm, 1e-3)))
            t0 = time.perf_counter()
            for _ in range(iters):
                one()
            dt = time.perf_counter() - t0
            budget_left[0] = max(budget_left[0] - dt - n_warm * warm, 5.0)
            out[batch] = dict(images_per_s=round(batch * iters / dt, 2), ms_per_iter=round(dt / iters * 1e3, 2), warmup_iters=n_warm,
                              timed_iters=iters)
            log(f"cpu_baseline: {what} N={batch}: {out[batch]['images_per_s']} img/s on {cores} threads ({iters} timed iterations)")
        return out

    rates = time_oracle({k: v.detach().cpu() for k, v in state_dict.items()}, "hrnet_w32", "hrnet_head", "HRNet-W32")
    # BASELINE.json configs[0] / [1]: SimpleBaseline-R50 256x192, batch 1 (the reference's CPU-runnable case) and batch 32, beside
    # the `config1_simplebaseline...` GPU leg; its own time budget so a slow host cannot starve it
    budget_left[0] = 25.0
    sb_net = mp.init_synthetic(mp.create_network("resnet50", "simple_baseline_head"), seed=0)
    sb_rates = time_oracle({k: v.detach().cpu() for k, v in sb_net.state_dict().items()}, "resnet50", "simple_baseline_head",
                           "SimpleBaseline-R50")
    del sb_net
    # the reference's target generation IS numpy on one core: time the pinned restatement next to the HIP kernel
    torch.set_num_threads(1)
    rng = np.random.default_rng(0)
    n_t = 128
    kp = np.empty((n_t, 17, 3), dtype=np.float32)
    kp[..., 0] = rng.uniform(-20, 212, size=(n_t, 17))
    kp[..., 1] = rng.uniform(-20, 276, size=(n_t, 17))
    kp[..., 2] = (rng.uniform(size=(n_t, 17)) < 0.7)
    otarget.generate_target(kp[:8], (192, 256), (48, 64), sigma=2.0)
    t0 = time.perf_counter()
    otarget.generate_target(kp, (192, 256), (48, 64), sigma=2.0)
    cpu_t = time.perf_counter() - t0
    torch.set_num_threads(cores)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    kpd = torch.from_numpy(kp).to(dev)
    gpu_t = graph_time(lambda: tgt(kpd), dev, reps=20, warm=2)  # device time (the host mirror's call costs the host more)
    target_bytes = n_t * 17 * 64 * 48 * 4
    return {"value": rates[32]["images_per_s"], "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"SURVEY 8(d) protocol: 3 warm-up + {rates[32]['timed_iters']} timed iterations of batch 32 (and batch 1) HRNet-W32 "
                      f"256x192 forward+decode, torch-CPU fp32 oracle (MindSpore-CPU reference path not installable); "
                      f"threads={cores} ({cores_note}), torch.get_num_threads()={torch.get_num_threads()}, "
                      f"os.cpu_count()={os.cpu_count()}, cpu='{_cpu_model()}'",
            "batch_1": rates[1], "batch_32": rates[32],
            "simplebaseline_r50": {"what": "BASELINE.json configs[0] (batch 1) / configs[1] (batch 32): SimpleBaseline ResNet-50 256x192 "
                                           "forward+decode, same oracle / threads / protocol", "batch_1": sb_rates[1], "batch_32": sb_rates[32]},
            "target_generation": {"cpu_samples_per_s": round(n_t / cpu_t, 1), "cpu_cores": 1,
                                  "cpu_what": "numpy restatement of TopDownGenerateTarget._encoding (17 joints x 128 samples, one core; "
                                              "pinned bit-exact to the reference's own output)",
                                  "gpu_samples_per_s": round(n_t / gpu_t, 1), "gpu_us_per_batch_128": round(gpu_t * 1e6, 2),
                                  "gpu_write_GBps": round(target_bytes / gpu_t / 1e9, 1),
                                  "gpu_what": "mp_gaussian_target, 128 x 17 x 64x48 fp32 heat-maps written once (HBM-bound, 8 TB/s peak)"}}


def hbm_ops_report(mp, dev, n=128, n_sets=12):
    """The HBM-bound rows of the path at N = 128, 17 x 64x48 fp32 heat-maps: device time per launch (HIP
    events around ONE replay of a hipGraph holding 48 back-to-back calls of the C-ABI entry on preallocated buffers - the host mirror's
    wrappers allocate their outputs per call and would time the host), ALGORITHMIC bytes per launch (each tensor read /
    written once) and the fraction of the 8 TB/s HBM peak they amount to.  Two timings per op: `hbm_resident` rotates the calls over
    ``n_sets`` distinct buffer sets (12 x 27 - 80 MB = 320 - 960 MB, more than the 256 MB memory-side cache holds, so every launch
    streams from / to HBM: THE roofline number), `cache_resident` replays the same 27 - 80 MB every launch (what a decode right
    behind the network's last kernel sees)."""
    from mindpose_amd import _lib
    from mindpose_amd.engine.inferencer.topdown_inferencer import COCO_FLIP_INDEX
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(7)
    k, h, w = 17, 64, 48
    hm_bytes = n * k * h * w * 4
    hms = [torch.rand(n, k, h, w, generator=g).to(dev) for _ in range(n_sets)]
    hfs = [torch.rand(n, k, h, w, generator=g).to(dev) for _ in range(n_sets)]
    tgts = [torch.rand(n, k, h, w, generator=g).to(dev) for _ in range(n_sets)]
    grads = [torch.empty_like(hms[0]) for _ in range(n_sets)]
    wgt = (torch.rand(n, k, generator=g) < 0.7).float().to(dev)
    center = (torch.rand(n, 2, generator=g) * 400).to(dev)
    scale = (torch.rand(n, 2, generator=g) * 2.7 + 0.3).to(dev)
    score = torch.rand(n, generator=g).to(dev)
    kp = torch.empty(n, k, 3)
    kp[..., 0] = torch.rand(n, k, generator=g) * 232 - 20
    kp[..., 1] = torch.rand(n, k, generator=g) * 296 - 20
    kp[..., 2] = (torch.rand(n, k, generator=g) < 0.7).float()
    kp = kp.to(dev)
    flip_index = torch.as_tensor(np.array(COCO_FLIP_INDEX), dtype=torch.int32, device=dev)
    dec_dark = mp.create_decoder("topdown_heatmap", use_udp=True, dark_udp_refine=True, kernel_size=11).to(dev)
    dec_shift = mp.create_decoder("topdown_heatmap", shift_coordinate=True).to(dev)
    preds = torch.empty(n, k, 3, device=dev)
    boxes = torch.empty(n, 6, device=dev)
    argmax = torch.empty(n, k, device=dev, dtype=torch.int32)
    tweight = torch.empty(n, k, device=dev)
    loss = torch.empty(1, device=dev)
    go = torch.ones(1, device=dev)
    ws_bytes = lib.mp_joints_mse_workspace_bytes(n, k)
    ws = torch.empty(ws_bytes // 4, device=dev)
    tgt_gen = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    patch = torch.from_numpy(tgt_gen._gaussian_patch()).to(dev)  # the constants exactly as TopDownGenerateTarget.generate passes them
    side = patch.shape[0]
    fsx, fsy = tgt_gen._feat_stride()

    blur = {id(d): d._blur_on(dev) for d in (dec_dark, dec_shift)}  # (cached by the decoder; fetched outside the captures)

    def decode(dec):
        return lambda i, st: lib.mp_decode_topdown(_lib.ptr(hms[i]), _lib.ptr(center), _lib.ptr(scale), _lib.ptr(score), _lib.ptr(preds), _lib.ptr(boxes),
                                                   _lib.ptr(argmax), n, k, h, w, dec.refine_mode, int(dec.use_udp), int(dec.to_original),
                                                   float(dec.pixel_std), _lib.ptr(blur[id(dec)]), int(dec.kernel_size), st)

    def flip_decode(i, st):
        return lib.mp_flip_aggregate_decode(_lib.ptr(hms[i]), _lib.ptr(hfs[i]), _lib.ptr(flip_index), 1, None, _lib.ptr(center), _lib.ptr(scale),
                                            _lib.ptr(score), _lib.ptr(preds), _lib.ptr(boxes), _lib.ptr(argmax), n, k, h, w,
                                            dec_shift.refine_mode, int(dec_shift.use_udp), int(dec_shift.to_original),
                                            float(dec_shift.pixel_std), _lib.ptr(blur[id(dec_shift)]), int(dec_shift.kernel_size), st)

    def mse_fwd(i, st):
        return lib.mp_joints_mse_fwd(_lib.ptr(hms[i]), _lib.ptr(tgts[i]), _lib.ptr(wgt), _lib.ptr(loss), _lib.ptr(ws), ws_bytes, n, k, h * w, st)

    def mse_bwd(i, st):
        return lib.mp_joints_mse_bwd(_lib.ptr(hms[i]), _lib.ptr(tgts[i]), _lib.ptr(wgt), _lib.ptr(go), _lib.ptr(grads[i]), n, k, h * w, st)

    def target(i, st):
        return lib.mp_gaussian_target(_lib.ptr(kp), _lib.ptr(patch), int(side), None, _lib.ptr(tgts[i]), _lib.ptr(tweight), n, k, h, w,
                                      fsx, fsy, 2.0, 0, st)

    def timed(fn, reps=48):
        out = {}
        for mode, sets in (("hbm_resident", n_sets), ("cache_resident", 1)):
            counter = [0]

            def call():
                _lib.check(fn(counter[0] % sets, _lib.stream()), "hbm_ops")
                counter[0] += 1
            out[mode] = graph_time(call, dev, reps=reps)
        return out

    rows = {
        "decode_argmax_shift": (timed(decode(dec_shift)), hm_bytes,
                                "mp_decode_topdown (decode_kernel<false>): arg-max + quarter-pixel shift + transform, a11 / a12 / a14"),
        "decode_dark_udp": (timed(decode(dec_dark)), hm_bytes,
                            "mp_decode_topdown (decode_kernel<true>): arg-max + DARK / UDP refinement (11x11 blur of the 3x3 neighbourhood), a13"),
        "flip_aggregate_decode": (timed(flip_decode), 2 * hm_bytes,
                                  "mp_flip_aggregate_decode: flip back + one-pixel shift + average + decode in one kernel, a15"),
        "joints_mse_fwd": (timed(mse_fwd), 2 * hm_bytes, "mp_joints_mse_fwd (mse_row + mse_final): weighted squared error, fixed-order reduction, a10"),
        "joints_mse_bwd": (timed(mse_bwd), 3 * hm_bytes, "mp_joints_mse_bwd: 2 w (p - t) / (N K H W), a10"),
    }
    rows["gaussian_target"] = (timed(target), hm_bytes, "mp_gaussian_target: 128 x 17 x 64x48 fp32 heat-maps written once (zero planes + one 13x13 stamp each), a8")
    return {"batch": n, "heatmaps": f"{k}x{h}x{w} fp32", "hbm_peak_GBps": 8000.0, "buffer_sets": n_sets,
            "timing": "HIP events around one replay of a hipGraph holding 48 back-to-back C-ABI calls (5 warm-ups; a Python call costs the "
                      "host more than these kernels run); bytes = algorithmic (each tensor once).  us / GBps / frac_of_hbm_peak = the "
                      f"HBM-resident figure (calls rotate over {n_sets} buffer sets, more than the 256 MB memory-side cache holds); "
                      "cache_resident_* = the same buffers every launch",
            "ops": {name: {"us": round(t["hbm_resident"] * 1e6, 2), "algorithmic_bytes": b, "GBps": round(b / t["hbm_resident"] / 1e9, 1),
                           "frac_of_hbm_peak": round(b / t["hbm_resident"] / 8e12, 4),
                           "cache_resident_us": round(t["cache_resident"] * 1e6, 2),
                           "cache_resident_frac": round(b / t["cache_resident"] / 8e12, 4), "what": what}
                    for name, (t, b, what) in rows.items()}}


class CallTimer:
    """HIP events around every call of the named C-ABI entries (recorded on the launch stream, read once at the end): the
    per-kernel-family device time of an eager training step.  The entries are wrapped on the loaded library object, which is
    what the host mirror calls through, and restored afterwards."""

    CONV = ("mp_f16_conv2d_fwd", "mp_f16_conv2d_fwd_stats", "mp_conv2d_fwd_variant", "mp_conv2d_fwd", "mp_conv2d_winograd_fwd")
    WGRAD = ("mp_f16_conv_wgrad", "mp_f16_conv_wgrad_grouped", "mp_conv_wgrad")
    BN = ("mp_f16_bn_train_fwd", "mp_f16_bn_train_bwd", "mp_f16_bn_train_fwd_stats", "mp_f16_bn_train_bwd_stats", "mp_bn_train_fwd",
          "mp_bn_train_bwd_acc", "mp_bn_train_bwd")
    OTHER = ("mp_f16_fuse_upsample_sum", "mp_f16_fuse_upsample_sum_bwd", "mp_fuse_upsample_sum", "mp_fuse_upsample_sum_bwd",
             "mp_f16_to_c8", "mp_f16_from_c8", "mp_f16_pack_weight_batch", "mp_conv_pack_weight_batch", "mp_f16_pack_weight",
             "mp_conv_pack_weight", "mp_joints_mse_fwd", "mp_joints_mse_bwd", "mp_gaussian_target", "mp_adamw_step_scaled",
             "mp_grad_finite_check", "mp_sum_tensors", "mp_f16_sum_tensors_stats", "mp_f16_fuse_sum_bwd_term_stats", "mp_f16_bn_train_finalize")

    def __init__(self, lib):
        self.lib, self.records, self._orig = lib, [], {}

    def __enter__(self):
        for name in self.CONV + self.WGRAD + self.BN + self.OTHER:
            orig = getattr(self.lib, name)
            self._orig[name] = orig
            setattr(self.lib, name, self._wrap(name, orig))
        return self

    def __exit__(self, *exc):
        for name, orig in self._orig.items():
            setattr(self.lib, name, orig)

    def _wrap(self, name, orig):
        def call(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig(*args)
            e1.record()
            self.records.append((name, self._describe(name, args), e0, e1))
            return rc
        return call

    @staticmethod
    def _describe(name, args):
        if name in CallTimer.CONV + CallTimer.WGRAD:
            d = args[0]._obj  # ctypes.byref(mp_conv_desc)
            flops = 2.0 * d.n * d.conv_h * d.conv_w * d.cout * d.cin * d.kh * d.kw
            if name == "mp_f16_conv_wgrad_grouped":
                flops *= args[4]  # n_jobs layers of this shape in one launch pair
            variant = args[1] if name in ("mp_f16_conv2d_fwd", "mp_f16_conv2d_fwd_stats", "mp_conv2d_fwd_variant") else (
                9 if name == "mp_conv2d_winograd_fwd" else None)
            return dict(flops=flops, shape=f"{d.kh}x{d.kw} s{d.stride} {d.cin}->{d.cout} @{d.h}x{d.w} N={d.n}", variant=variant,
                        ks=d.kh, stride=d.stride)
        if name in CallTimer.BN:
            half = "f16" in name
            if name == "mp_f16_bn_train_bwd_stats":  # apply pass only: read g, z; write dz (g, z, gamma, mean, invstd, dz, ... n, c, hw at 10..12)
                n, c, hw = args[10:13]
                return dict(bytes=3.0 * n * c * hw * 2, shape=f"C={c} HW={hw} N={n}")
            fwd = name.endswith("fwd") or name.endswith("fwd_stats")
            # positional layout of the BatchNorm entries (include/mindpose_hip.h): fwd (z,g,b,res,y,...,n,c,hw at 9..11);
            # bwd (dy,z,y,g,mean,invstd,dz,dres,...) with n,c,hw after the pointer block
            if fwd:
                n, c, hw = args[9:12]
                tensors = 2 + (args[3] is not None)            # read z, write y (+ read res)
            else:
                k = 13 if name == "mp_f16_bn_train_bwd" else (12 if name == "mp_bn_train_bwd_acc" else 10)
                n, c, hw = args[k:k + 3]
                dres = args[8] if name == "mp_f16_bn_train_bwd" else args[7]
                tensors = 3 + (args[2] is not None) + (dres is not None)   # read dy, z (, y); write dz (+ dres)
            return dict(bytes=float(tensors) * n * c * hw * (2 if half else 4), shape=f"C={c} HW={hw} N={n}")
        return {}

    def summary(self):
        torch.cuda.synchronize()
        fam = {}
        for name, info, e0, e1 in self.records:
            t = e0.elapsed_time(e1) * 1e-3
            key = name
            f = fam.setdefault(key, dict(time=0.0, launches=0, flops=0.0, bytes=0.0, shapes={}))
            f["time"] += t
            f["launches"] += 1
            f["flops"] += info.get("flops", 0.0)
            f["bytes"] += info.get("bytes", 0.0)
            if "shape" in info:
                sh = f["shapes"].setdefault((info["shape"], info.get("variant"), info.get("ks"), info.get("stride")),
                                            dict(time=0.0, launches=0, flops=0.0, bytes=0.0))
                sh["time"] += t
                sh["launches"] += 1
                sh["flops"] += info.get("flops", 0.0)
                sh["bytes"] += info.get("bytes", 0.0)
        return fam


def f16_kernel_for(ks, stride, variant):
    return kernel_name(dict(kind_id=3, ks=ks, stride=stride, variant=variant, light=5 <= variant <= 9)) if variant is not None and variant >= 0 else "library heuristic"


def f32_kernel_for(ks, stride, variant):
    if variant == 9:
        return "conv_wino_f32_kernel"
    if variant == 8:
        return "conv1x1_f32_stream_kernel"
    if variant == 10:
        return "conv1x1_f32_gemm_kernel"
    light = variant in (0, 5, 7)  # conv_mfma.h variant_light
    return kernel_name(dict(kind_id=0, ks=ks, stride=stride, variant=variant, light=light)) if variant is not None and variant >= 0 else "library heuristic"


def train_roofline(eager_step, half):
    """Roofline of the training step's dominant kernel family: one eager step (same kernels as the graph replay, one stream) with
    HIP events around every C-ABI call.  Conv / weight-gradient entries are priced in algorithmic FLOP (2 x MACs), the BatchNorm
    passes in algorithmic bytes (each operand tensor once)."""
    from mindpose_amd import _lib
    lib = _lib.load()
    eager_step()  # warm (allocator, tuner)
    with CallTimer(lib) as ct:
        eager_step()
        fam = ct.summary()
    total = sum(f["time"] for f in fam.values())
    dump = os.environ.get("MINDPOSE_BENCH_TRAIN_SHAPES")
    if dump:  # per (entry, shape) table of the instrumented step, for kernel work: entry,shape,variant,launches,total_us,avg_us
        with open(dump, "w") as fh:
            fh.write("entry,shape,variant,launches,total_us,avg_us,tflops,GBps\n")
            for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["time"]):
                for (shape, variant, _, _), sh in sorted(f["shapes"].items(), key=lambda kv: -kv[1]["time"]):
                    tf = sh["flops"] / sh["time"] / 1e12 if sh["time"] > 0 else 0.0
                    fh.write(f"{name},\"{shape}\",{variant},{sh['launches']},{sh['time'] * 1e6:.1f},"
                             f"{sh['time'] / sh['launches'] * 1e6:.2f},{tf:.1f},{sh['bytes'] / sh['time'] / 1e9 if sh['time'] > 0 else 0.0:.0f}\n")
    mfma_peak = PEAK_FP16_MFMA_TFLOPS if half else PEAK_FP32_MFMA_TFLOPS
    dom = max(fam, key=lambda k: fam[k]["time"])
    d = fam[dom]
    per_entry = {}
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["time"]):
        e = {"launches": f["launches"], "ms": round(f["time"] * 1e3, 3), "share": round(f["time"] / total, 3)}
        if f["flops"]:
            e["tflops"] = round(f["flops"] / f["time"] / 1e12, 2)
            if k == "mp_conv2d_winograd_fwd":  # algorithmic FLOP above; the matrix pipe executes 16 / 36 of them
                e["mfma_tflops"] = round(f["flops"] / f["time"] / 1e12 * 16.0 / 36.0, 2)
        if f["bytes"]:
            e["GBps"] = round(f["bytes"] / f["time"] / 1e9, 1)
        per_entry[k] = e
    out = {"timing": "HIP events around each C-ABI call of one eager step on one stream (the timed region replays the same kernels "
                     "from a hipGraph with the HRModule branches on side streams)",
           "entry": dom, "launches_per_step": d["launches"], "share_of_step_kernel_time": round(d["time"] / total, 3),
           "sum_kernel_ms": round(total * 1e3, 3), "per_entry": per_entry, "traffic": None}
    if d["flops"]:
        shapes = d["shapes"]
        top = max(shapes, key=lambda k: shapes[k]["time"])
        name = (f16_kernel_for if "f16" in dom else f32_kernel_for)(top[2], top[3], top[1] if top[1] is None or top[1] >= 0 else None) if dom in CallTimer.CONV else (
            f"conv_wgrad_f16_dma_k{top[2]}s{top[3]}" if "f16" in dom else f"conv_wgrad_pipe_kernel<{top[2]},{top[3]}>")
        t = shapes[top]
        ach = t["flops"] / t["time"] / 1e12
        out.update({"bound": "mfma", "achieved": round(ach, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(ach / mfma_peak, 4),
                    "kernel": name, "shape": top[0], "kernel_launches_per_step": t["launches"],
                    "flop_per_launch": round(t["flops"] / t["launches"]), "avg_launch_us": round(t["time"] / t["launches"] * 1e6, 2),
                    "all_launches_of_entry": {"achieved": round(d["flops"] / d["time"] / 1e12, 2),
                                              "frac": round(d["flops"] / d["time"] / 1e12 / mfma_peak, 4)}})
    else:
        ach = d["bytes"] / d["time"] / 1e9
        out.update({"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBPS, 4),
                    "kernel": dom, "bytes_per_launch": round(d["bytes"] / d["launches"]),
                    "avg_launch_us": round(d["time"] / d["launches"] * 1e6, 2)})
    return out


def train_step_hbm(roofline, step_s, n, backbone, half):
    """The WHOLE step against the HBM roof.  The per-kernel view above prices the dominant conv against the matrix pipe; the step as a
    whole moves ~88 GB through HBM at N = 128 (rocprofv3 FETCH_SIZE / WRITE_SIZE over every launch of a step: profiles/
    r05_a_train_o2_pmc_traffic.json "step"), i.e. it is bandwidth-bound as a sum - this block divides those bytes by the TIMED step.
    The counters exist for the amp-O2 HRNet-W32 step at N = 128 only (bench.py cannot collect PMC counters itself): None otherwise."""
    if not (half and n == 128 and backbone == "hrnet_w32"):
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_train_o2_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                st = json.load(f).get("step")
            if st:
                gb = st["hbm_GB_per_step"]
                return {"bound": "hbm", "peak_GBps": PEAK_HBM_GBPS, "ms_per_step": round(step_s * 1e3, 3), "pmc_GB_per_step": gb,
                        "achieved_GBps": round(gb / step_s, 1), "frac": round(gb / step_s / PEAK_HBM_GBPS, 4),
                        "hbm_floor_ms_at_6300_GBps": round(gb / 6300.0 * 1e3, 2), "pmc_sum_kernel_ms_per_step": st["sum_kernel_ms_per_step"],
                        "pmc_source": os.path.relpath(path, ROOT),
                        "what": "HBM bytes of every launch of one step (PMC, measured once per round) / the step time measured here"}
        except (OSError, ValueError, KeyError):
            continue
    return None


def train_bench(args, mp, dev, dist, world, rank):
    """configs[3]: HRNet-W32 256x192 training step, data parallel - Gaussian targets on the device, forward with
    batch-statistics BatchNorm, JointsMSELoss, backward (MFMA dgrad/wgrad), RCCL gradient mean, AdamWeightDecay.
    Extra measurement (the contract metric is inference)."""
    from mindpose_amd.utils import AdamWeightDecay, DynamicLossScaleManager
    n = args.batch
    bb, hd = ("resnet50", "simple_baseline_head") if args.workload == "simplebaseline_r50_train" else ("hrnet_w32", "hrnet_head")
    net = mp.init_synthetic(mp.create_network(bb, hd), seed=0).to(dev).train()
    scaler = None
    if args.amp != "O0":  # the reference's recipe: amp O2 + DynamicLossScaleManager (tools/train.py:170-181)
        mp.models.auto_mixed_precision(net, args.amp)
        scaler = DynamicLossScaleManager()
    nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
    # The step (forward + loss + backward) is captured into a hipGraph on every rank: amp O2 is launch-bound from Python
    # (eager ~2450 img/s) and inside the graph the HRModule branches run on side streams.  Under DP the bucket all-reduces of
    # the gradient arena are issued right after the replay (RCCL, asynchronous) and the update kernel folds the 1/world in.
    # MINDPOSE_TRAIN_GRAPH=0: the eager step with the bucket all-reduces launched from backward hooks (overlapped).
    graphed = os.environ.get("MINDPOSE_TRAIN_GRAPH", "1") != "0"
    opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=not graphed)
    tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    image = torch.randn(n, 3, 256, 192, generator=gen).to(dev)
    kp = torch.empty(n, 17, 3)
    kp[..., 0] = torch.rand(n, 17, generator=gen) * 232 - 20
    kp[..., 1] = torch.rand(n, 17, generator=gen) * 296 - 20
    kp[..., 2] = (torch.rand(n, 17, generator=gen) < 0.7).float()
    kp = kp.to(dev)

    gstep = None
    if graphed:  # forward + loss + backward captured once into a hipGraph, replayed per step (utils/graph_step.py)
        from mindpose_amd.utils import GraphedTrainStep
        t0, w0 = tgt(kp)
        gstep = mp.models.tune_on_rank0_first(lambda: GraphedTrainStep(nwl, opt, (image, t0, w0), loss_scale_manager=scaler))

    def eager_step(update=True):
        opt.zero_grad()
        target, weight = tgt(kp)
        loss = nwl(image, target, weight)
        if scaler is not None:
            scaler.scale(loss).backward()
        else:
            loss.backward()
        if update:
            opt.step(loss_scale_manager=scaler)
        return loss

    def step():
        if gstep is not None:
            target, weight = tgt(kp)
            return gstep(image, target, weight)
        return eager_step()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    log(f"rank {rank}: training step built ({'hipGraph' if graphed else 'eager'}), warming up")
    for _ in range(args.warmup):
        step()
    sync_all()
    log("timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss.detach())
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    roofline = None
    if world == 1 and not args.no_roofline:  # one rank only: an extra instrumented step on rank 0 alone would strand its collectives
        roofline = train_roofline(lambda: eager_step(update=False), half=scaler is not None)
        roofline["step_hbm"] = train_step_hbm(roofline, elapsed / args.steps, n, bb, half=scaler is not None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        gflop = 45.9 if bb == "hrnet_w32" else None  # ~3x the forward's 15.29 GFLOP
        print(json.dumps({
            "metric": f"images/sec, {args.workload} 256x192 training step (targets+fwd+loss+bwd+grad mean+AdamWeightDecay)",
            "value": round(world * n * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if scaler is None else "f16", "data": "synthetic",
            "config": {"workload": ("configs[3] in fp32" if scaler is None else "configs[3] as the reference trains it (amp O2: fp16 "
                                    "matrix-core convs / activations, fp32 statistics + master weights, dynamic loss scale)") +
                                   f": {bb} + {hd} 256x192 training, DP, Gaussian targets + JointsMSE + bucketed RCCL gradient "
                                   "mean + AdamWeightDecay",
                       "per_gpu_batch": n, "global_batch": n * world, "final_loss": final_loss,
                       "gflop_per_image": gflop,
                       "step_tflops": None if gflop is None else round(gflop * world * n * args.steps / elapsed / 1e3, 2),
                       "step": ("one hipGraph replay (forward+loss+backward), then bucket all-reduces + overflow check + update"
                                if graphed else "eager autograd, bucket all-reduces overlapped with backward"),
                       "loss_scale": None if scaler is None else scaler.loss_scale,
                       "skipped_steps": None if scaler is None else scaler.skipped_steps},
            "roofline": roofline, "cpu_baseline": None}))


def dp_leg_report(step, opt, dist, world, rank, dev, per_gpu_batch, steps, warmup, sync_device=None, graphed=None):
    """The data-parallel training leg of a multi-rank run, on ALL ranks: `warmup` + `steps` calls of `step()` (forward + loss +
    backward + bucketed gradient all-reduce + update), timed between barriers with the MAX over ranks like the headline; the
    collectives' share from device events around the optimizer's wait for them (`opt.time_comm`).  Everything that touches the
    model sits in `step` / `opt`, so the CPU rehearsal (tests/test_sharding_cpu.py, gloo) drives the same reporting code."""
    def sync_all():
        if sync_device is not None:
            sync_device()
        dist.barrier()
        if sync_device is not None:
            sync_device()

    opt.time_comm = True
    for _ in range(warmup):
        step()
    sync_all()
    opt.comm_events.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    comm_ms = [e0.elapsed_time(e1) for e0, e1 in opt.comm_events]
    opt.time_comm = False
    grads = opt.grads
    overlap = {}
    if graphed is not None:  # GraphedTrainStep: the backward pass as several hipGraphs, finished buckets released between them
        issue = graphed.issue_ms[-steps:]
        overlap = {"backward_segments": graphed.segments, "buckets_released_per_segment": [len(r) for r in graphed.bucket_schedule],
                   "allreduce_issue_ms": round(sum(issue) / len(issue), 3) if issue else None,
                   "allreduce_issue_what": "host time per step spent handing finished buckets to the all-reduce between the segments "
                                           "(the collectives then run on the communication stream under the next segment's kernels)"}
    return {**overlap, "value": round(world * per_gpu_batch * steps / elapsed, 2), "unit": "images/s", "ms_per_step": round(elapsed / steps * 1e3, 3),
            "steps": steps, "warmup": warmup, "per_gpu_batch": per_gpu_batch, "global_batch": per_gpu_batch * world,
            "allreduce_ms_per_step": round(sum(comm_ms) / len(comm_ms), 3) if comm_ms else None,
            "allreduce_what": "EXPOSED wait: device time between 'all gradients in the arena' and 'bucket all-reduces landed' on the "
                              "compute stream (events around GradientAverager.finish(); rank 0); buckets released between the backward "
                              "segments have been running since",
            "rccl_nranks": grads.comm_ranks(), "transport": "native (mp_comm_*, RCCL bound by the library)" if grads.native is not None
            else f"torch.distributed ({dist.get_backend()})",
            "gradient_bytes": int(grads.arena.numel() * 4), "buckets": len(grads.buckets), "mean": grads.mean,
            "n_gpus_seen": torch.cuda.device_count()}


def all_ranks_ok(dist, dev, ok: bool) -> bool:
    """MIN over ranks of a success flag: a rank that failed on its own (e.g. out of memory while building) must not leave the others
    blocked inside the leg's collectives - every rank learns of the failure here and all skip the leg together."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def dp_train_leg(args, mp, dev, dist, world, rank):
    """configs[3] under data parallelism on every rank of a multi-rank run: HRNet-W32 amp-O2 training step as
    one hipGraph replay + bucketed RCCL gradient mean + AdamWeightDecay, so that a scaling run exercises the collective.

    Built in two phases, each followed by `all_ranks_ok`: (A) model, data, optimizer arena - no collective inside; (B) the
    communicator and the captured step (`tune_on_rank0_first`: one broadcast, reached by every rank whatever its build did).  A rank
    that fails in either phase makes ALL ranks return an error entry instead of entering the timed collectives alone."""
    from mindpose_amd.utils import AdamWeightDecay, DynamicLossScaleManager, GraphedTrainStep
    n = int(os.environ.get("MINDPOSE_BENCH_DP_BATCH", args.batch))
    err = None
    try:
        net = mp.init_synthetic(mp.create_network("hrnet_w32", "hrnet_head"), seed=0).to(dev).train()
        mp.models.auto_mixed_precision(net, "O2")
        scaler = DynamicLossScaleManager()
        nwl = mp.create_network_with_loss(net, mp.create_loss("joint_mse", use_target_weight=True), has_extra_inputs=True)
        tgt = mp.TopDownGenerateTarget(config=dict(image_size=[192, 256], heatmap_size=[48, 64]), sigma=2.0)
        gen = torch.Generator(device="cpu").manual_seed(2000 + rank)
        image = torch.randn(n, 3, 256, 192, generator=gen).to(dev)
        kp = torch.empty(n, 17, 3)
        kp[..., 0] = torch.rand(n, 17, generator=gen) * 232 - 20
        kp[..., 1] = torch.rand(n, 17, generator=gen) * 296 - 20
        kp[..., 2] = (torch.rand(n, 17, generator=gen) < 0.7).float()
        kp = kp.to(dev)
        t0, w0 = tgt(kp)
    except Exception as exc:
        err = f"build phase A, rank {rank}: {type(exc).__name__}: {exc}"
    if not all_ranks_ok(dist, dev, err is None):
        return {"error": err or "another rank failed in build phase A"}
    opt = gstep = None
    try:
        opt = AdamWeightDecay(net, lr=1e-3, weight_decay=0.05, filter_bias_and_bn=True, overlap=False)
        # the tuner never communicates: rank 0 builds (and tunes) first, its choices reach the other ranks in ONE broadcast, then they
        # build on cache hits - the warm-up passes and the capture inside GraphedTrainStep contain no collective
        gstep = mp.models.tune_on_rank0_first(lambda: GraphedTrainStep(nwl, opt, (image, t0, w0), loss_scale_manager=scaler))
    except Exception as exc:
        err = f"build phase B, rank {rank}: {type(exc).__name__}: {exc}"
    if not all_ranks_ok(dist, dev, err is None):
        if opt is not None:
            opt.close()
        return {"error": err or "another rank failed in build phase B"}

    def step():
        target, weight = tgt(kp)
        return gstep(image, target, weight)

    log(f"rank {rank}: DP training leg built (hipGraph step, {len(opt.grads.buckets)} gradient buckets)")
    rep = dp_leg_report(step, opt, dist, world, rank, dev, n, steps=max(3, min(args.steps, 10)), warmup=max(1, min(args.warmup, 3)),
                        sync_device=torch.cuda.synchronize, graphed=gstep)
    rep.update({"workload": "configs[3]: hrnet_w32 + hrnet_head 256x192 amp-O2 training, data parallel: one hipGraph replay per rank, "
                            "bucketed RCCL all-reduce of the 114 MB gradient arena overlapped with the backward segments (1/world folded into the "
                            "update), AdamWeightDecay",
                "dtype": "f16", "loss_scale": scaler.loss_scale, "skipped_steps": scaler.skipped_steps})
    opt.close()
    return rep


# name -> bench.py arguments of the extra legs run after the headline (N=1 only); >= 20 timed steps after 5 warm-ups each
# (5 steps were within the +-3 % box noise), the whole default run stays under ~2 minutes
EXTRA_LEGS = {
    "hrnet_w32_256x192_infer_ampO2": ["--workload", "hrnet_w32", "--amp", "O2", "--batch", "128", "--steps", "30", "--warmup", "5"],
    "config5_hrnet_w48_384x288_udp_dark_flip_ampO2": ["--workload", "hrnet_w48_384_udp_flip", "--amp", "O2", "--batch", "64",
                                                      "--steps", "20", "--warmup", "5"],
    "config1_simplebaseline_r50_256x192_infer_f32": ["--workload", "simplebaseline_r50", "--batch", "128", "--steps", "20", "--warmup", "5"],
    "config3_hrnet_w32_train_f32": ["--workload", "hrnet_w32_train", "--batch", "128", "--steps", "20", "--warmup", "5"],
    "config3_hrnet_w32_train_ampO2": ["--workload", "hrnet_w32_train", "--amp", "O2", "--batch", "128", "--steps", "20", "--warmup", "5"],
    # (the reference recipe is 128 per device - hrnet_w32_ascend.yaml:19; the same step at 256: every launch carries twice the work)
    "config3_hrnet_w32_train_ampO2_n256": ["--workload", "hrnet_w32_train", "--amp", "O2", "--batch", "256", "--steps", "12", "--warmup", "3",
                                           "--no-roofline"],
    # what the input pipeline delivers next to what the training step consumes (run_extra_legs adds the ratio)
    "loader_coco_topdown_train_pipeline": ["--workload", "loader_coco_topdown", "--batch", "128", "--steps", "24", "--warmup", "1"],
    # SURVEY 8(d) configs 2 / 3, BASELINE.md 3: the batch sweep N in {1, 32, 256} beside the N = 128 lines above (one child per
    # precision / model; every N is its own tuned plan, 20 timed steps after 5 warm-ups)
    "batch_sweep_hrnet_w32_f32": ["--workload", "hrnet_w32", "--sweep", "1,32,256", "--steps", "20", "--warmup", "5", "--no-roofline"],
    "batch_sweep_hrnet_w32_ampO2": ["--workload", "hrnet_w32", "--amp", "O2", "--sweep", "1,32,256", "--steps", "20", "--warmup", "5",
                                    "--no-roofline"],
    "batch_sweep_simplebaseline_r50_f32": ["--workload", "simplebaseline_r50", "--sweep", "1,32,256", "--steps", "20", "--warmup", "5",
                                           "--no-roofline"],
}


def run_extra_legs(selected=None, timeout_s=240):
    """Each leg = this script in a fresh child process (own timed region, own roofline), started AFTER the headline's timed
    region; a leg that fails or exceeds its limit is reported as such and never takes the headline line down."""
    out = {}
    env = dict(os.environ)
    env.setdefault("MINDPOSE_TUNE_CACHE", os.path.join("/tmp", f"mindpose_tune_{os.getuid()}.json"))
    for name, leg_args in EXTRA_LEGS.items():
        if selected and name not in selected:
            continue
        t0 = time.perf_counter()
        log(f"extra leg {name}: {' '.join(leg_args)}")
        try:
            # the leg's progress lines go straight to this process's stderr (a silent parent looks hung to the GPU box's watchdog)
            proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--leg", *leg_args], env=env,
                                  stdout=subprocess.PIPE, stderr=None, text=True, timeout=timeout_s)
            line = next((ln for ln in reversed(proc.stdout.splitlines()) if ln.startswith("{")), None)
            if proc.returncode != 0 or line is None:
                out[name] = {"error": f"rc {proc.returncode}", "stdout_tail": proc.stdout[-400:]}
                continue
            r = json.loads(line)
            if "sweep" in r:
                out[name] = {"sweep": r["sweep"], "unit": r["unit"], "dtype": r["dtype"], "steps": r["steps"], "warmup": r["warmup"],
                             "workload": r["config"]["workload"], "leg_wall_s": round(time.perf_counter() - t0, 1)}
                continue
            rl = r.get("roofline") or {}
            out[name] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                         "warmup": r["warmup"], "dtype": r["dtype"], "per_gpu_batch": r["config"].get("per_gpu_batch"),
                         "workload": r["config"]["workload"],
                         "roofline": {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "algorithm", "mfma_tflops",
                                                             "kernel", "avg_launch_us", "launches_per_step", "all_conv_launches",
                                                             "all_launches_of_entry", "per_entry", "share_of_step_kernel_time", "step_hbm")
                                      if k in rl},
                         "leg_wall_s": round(time.perf_counter() - t0, 1)}
            for k in ("gflop_per_image", "step_tflops", "final_loss", "step", "loss_scale", "skipped_steps", "host_threads", "cpu", "prefetch_2",
                      "synchronous", "what"):
                if k in r["config"]:
                    out[name][k] = r["config"][k]
            if name.startswith("loader_") and "value" in out.get("config3_hrnet_w32_train_ampO2", {}):
                out[name]["ratio_to_training_consumption"] = round(r["value"] / out["config3_hrnet_w32_train_ampO2"]["value"], 3)
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timeout after {timeout_s}s"}
        except (ValueError, KeyError) as exc:
            out[name] = {"error": f"unparsable leg output: {exc}"}
    return out


def sweep_bench(args, mp, net, eval_net, dev, size, workload_desc, world, rank):
    """`--sweep 1,32,256`: the same inference step at several per-GPU batch sizes, one tuned launch plan each (no flip test)."""
    ih, iw = size
    rows = {}
    for n in [int(v) for v in args.sweep.split(",") if v]:
        gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
        image = net.input_buffer((n, 3, ih, iw), dev)
        image.copy_(torch.randn(n, 3, ih, iw, generator=gen))
        center = (torch.rand(n, 2, generator=gen) * 400).to(dev)
        scale = (torch.rand(n, 2, generator=gen) * 2.7 + 0.3).to(dev)
        score = torch.rand(n, generator=gen).to(dev)
        plan_len = len(net.get_plan((n, 3, ih, iw), dev))
        for _ in range(args.warmup):
            eval_net(image, center, scale, score)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eval_net(image, center, scale, score)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows[str(n)] = {"images_per_s": round(n * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 3), "launches": plan_len}
        log(f"sweep N={n}: {rows[str(n)]['images_per_s']} img/s ({rows[str(n)]['ms_per_step']} ms/step)")
    if rank == 0:
        print(json.dumps({"metric": f"images/sec, {args.workload} top-down inference (backbone+head+decode), batch sweep", "sweep": rows,
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dtype": "f32" if args.amp == "O0" else "f16", "data": "synthetic", "config": {"workload": workload_desc}}))


def loader_bench(args, mp, dev):
    """SURVEY 8(f) N2 / data_factory.py:59-151: the training input pipeline on its own - synthetic COCO-format JPEGs (640x480) on
    local disk -> create_dataset (shuffled records) -> create_pipeline(batch 128, is_train=True: box -> centre / scale, random flip,
    half-body, random scale / rotation on the host in sample order; decode in a thread pool; ONE pinned upload, ONE mp_warp_affine
    and ONE mp_gaussian_target launch per batch), batches prepared ahead on a side stream.  Reports what the loader ALONE delivers
    (the consumer only waits for each batch) with and without the prefetch thread."""
    import tempfile
    from PIL import Image
    n_img, per_img, batch = 384, 2, args.batch
    rng = np.random.RandomState(3)
    tmp = tempfile.mkdtemp(prefix="mindpose_loader_")
    images, anns = [], []
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float32)
    for i in range(n_img):
        # smooth structure + mild noise: JPEG entropy (decode cost) in the range of photographs, not of white noise
        base = np.stack([127 + 90 * np.sin(xx / rng.uniform(20, 90) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(20, 90)) for _ in range(3)], axis=2)
        im = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
        Image.fromarray(im).save(os.path.join(tmp, f"{i:06d}.jpg"), quality=90)
        images.append(dict(id=i + 1, file_name=f"{i:06d}.jpg", width=640, height=480))
        for j in range(per_img):
            x0, y0 = rng.uniform(20, 300), rng.uniform(20, 200)
            bw, bh = rng.uniform(120, 300), rng.uniform(150, 260)
            kp = np.concatenate([rng.uniform([x0, y0], [x0 + bw, y0 + bh], (17, 2)), rng.randint(1, 3, (17, 1))], axis=1)
            anns.append(dict(id=len(anns) + 1, image_id=i + 1, category_id=1, iscrowd=0, bbox=[x0, y0, bw, bh], area=float(bw * bh),
                             num_keypoints=17, keypoints=kp.reshape(-1).tolist()))
    ann = os.path.join(tmp, "train.json")
    with open(ann, "w") as f:
        json.dump(dict(images=images, annotations=anns, categories=[dict(id=1, name="person")]), f)
    cfg = dict(image_size=[192, 256], heatmap_size=[48, 64], pixel_std=200.0, scale_padding=1.25, upper_body_ids=list(range(11)),
               flip_pairs=[[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]], det_bbox_thr=0.0)
    names = ["topdown_box_to_center_scale", {"topdown_horizontal_random_flip": {"flip_prob": 0.5}}, "topdown_halfbody_transform",
             "topdown_randomscale_rotation", "topdown_affine", {"topdown_generate_target": {"sigma": 2.0}}]
    threads = host_cores()[0]
    ds = mp.create_dataset(tmp, ann, is_train=True, config=cfg)
    out = {}
    for label, prefetch in (("prefetch_2", 2), ("synchronous", 0)):
        pipe = mp.create_pipeline(ds, names, batch_size=batch, is_train=True, num_workers=threads, config=cfg, prefetch=prefetch)
        np.random.seed(5)
        seen, t0, first = 0, None, None
        for epoch in range(64):
            for b in pipe:
                torch.cuda.current_stream().synchronize()  # the consumer: takes delivery of the batch, nothing else
                if t0 is None:
                    t0, first = time.perf_counter(), b  # (the first batch pays thread start-up and the kernels' first launch)
                    continue
                seen += 1
            if seen >= args.steps:
                break
        dt = time.perf_counter() - t0
        out[label] = {"images_per_s": round(seen * batch / dt, 1), "ms_per_batch": round(dt / seen * 1e3, 2), "batches": seen}
        log(f"loader {label}: {out[label]}")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    best = out["prefetch_2"]
    print(json.dumps({
        "metric": "images/sec, training input pipeline alone (decode + host transforms + GPU crop / normalise / target)", "value": best["images_per_s"],
        "unit": "images/s", "n_gpus": 1, "steps": best["batches"], "warmup": 1, "ms_per_step": best["ms_per_batch"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"SURVEY 8(f) N2: create_dataset('coco_topdown') + create_pipeline(batch {batch}, is_train=True) over {n_img} synthetic "
                               f"640x480 JPEGs ({n_img * per_img} person records), the recipe's transform list, 256x192 crops + 17x64x48 targets",
                   "per_gpu_batch": batch, "host_threads": threads, "cpu": _cpu_model(), "prefetch_2": out["prefetch_2"], "synchronous": out["synchronous"],
                   "what": "the consumer only waits for each batch: the loader's own rate.  decode = PIL in a thread pool (releases the GIL); "
                           "the per-sample geometry / random draws run on one thread in sample order (the reference maps them over "
                           "num_parallel_workers processes: data_factory.py:116-151)"},
        "roofline": None, "cpu_baseline": None}))


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run as ONE child process tree.
    Nothing in this process has touched the GPU yet (importing torch does not), so no GPU-initialised process is replaced."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"no WORLD_SIZE in the environment: launching {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="crops per GPU per step (reference per-device batch_size)")
    ap.add_argument("--workload", default="hrnet_w32", choices=list(WORKLOADS) + ["hrnet_w32_train", "simplebaseline_r50_train", "loader_coco_topdown"],
                    help="hrnet_w32 = BASELINE.json metric / configs[2] (default); the others are extra measurements")
    ap.add_argument("--layers", default="", help="write a per-launch timing table (CSV) to this path")
    ap.add_argument("--amp", default="O0", choices=["O0", "O2"],
                    help="O0 = fp32 (the reference's eval precision, the headline); O2 = fp16 matrix-core kernels (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_workloads legs (N=1 headline runs only)")
    ap.add_argument("--extra", default="", help="comma-separated subset of the extra legs to run (default: all)")
    ap.add_argument("--leg", action="store_true", help="this process IS an extra leg: no CPU baseline, no further legs")
    ap.add_argument("--sweep", default="", help="comma-separated per-GPU batch sizes: time each (its own plan) and report them all")
    ap.add_argument("--hbm-ops", action="store_true", help="print only the hbm_ops block (decode / loss / target kernels against the HBM roofline)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # rehearsal on a one-GPU box (never the measured configuration): every rank on device 0, gloo for the barrier / MAX
    rehearsal = os.environ.get("MINDPOSE_BENCH_SHARED_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL; only the timing barrier / MAX uses it

    import mindpose_amd as mp

    torch.manual_seed(0)
    if args.hbm_ops:
        print(json.dumps(hbm_ops_report(mp, dev)))
        return
    if args.workload == "loader_coco_topdown":
        return loader_bench(args, mp, dev)
    if args.workload in ("hrnet_w32_train", "simplebaseline_r50_train"):
        return train_bench(args, mp, dev, dist, world, rank)
    backbone, head, (ih, iw), dec_kw, flip, workload_desc = WORKLOADS[args.workload]
    net = mp.init_synthetic(mp.create_network(backbone, head), seed=0).to(dev).eval()
    if args.amp != "O0":
        mp.models.auto_mixed_precision(net, args.amp)
        workload_desc += f" [amp {args.amp}: fp16 MFMA kernels, fp32 accumulate]"
    decoder = mp.create_decoder("topdown_heatmap", **dec_kw).to(dev)
    eval_net = mp.create_eval_network(net, decoder, output_raw=True)
    multi_run = None
    if flip:
        from mindpose_amd.engine.inferencer.topdown_inferencer import COCO_FLIP_INDEX, _MultiRunNet
        multi_run = _MultiRunNet(eval_net, decoder, np.array(COCO_FLIP_INDEX), shift_heatmap=False).to(dev)

    if args.sweep:
        return sweep_bench(args, mp, net, eval_net, dev, (ih, iw), workload_desc, world, rank)
    n = args.batch
    # the flip test runs ONE forward of the 2N batch [crops | mirrors] (layers.py forward_flip_pair; MINDPOSE_FLIP_BATCHED=0: two of N)
    from mindpose_amd.models.layers import flip_pair_batched
    n_plan = 2 * n if flip and args.amp != "O0" and flip_pair_batched() else n  # (fp32 keeps the two forwards: topdown_inferencer.py)
    passes = 2 if flip and n_plan == n else 1
    # synthetic crops written straight into the plan's resident input buffer (inputs in HBM before timing)
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    if n_plan != n:
        image = net.input_buffer((n_plan, 3, ih, iw), dev)[:n]  # the crops' half of the 2N plan's input; the mirrors are written per step
    else:
        image = net.input_buffer((n, 3, ih, iw), dev)
    image.copy_(torch.randn(n, 3, ih, iw, generator=gen))
    if flip and n_plan == n:
        image = image.clone()  # the two-forward flip test runs both through the same plan input buffer
    center = (torch.rand(n, 2, generator=gen) * 400).to(dev)
    scale = (torch.rand(n, 2, generator=gen) * 2.7 + 0.3).to(dev)
    score = torch.rand(n, generator=gen).to(dev)

    def step():
        if multi_run is not None:
            return multi_run(image, center, scale, score)
        return eval_net(image, center, scale, score)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    plan_len = mp.models.tune_on_rank0_first(lambda: len(net.get_plan((n_plan, 3, ih, iw), dev)))  # rank 0 tunes, one broadcast, the rest replay
    log(f"rank {rank}: plan built ({plan_len} launches), warming up")
    for _ in range(args.warmup):
        step()
    sync_all()
    log("timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    result = None
    if rank == 0:
        plan = net.get_plan((n_plan, 3, ih, iw), dev)
        result = {
            "metric": ("images/sec at 256x192, HRNet-W32 top-down inference (backbone+head+decode)" if args.workload == "hrnet_w32"
                       else f"images/sec, {args.workload} top-down inference (backbone+head+decode)"),
            "value": round(world * n * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.amp == "O0" else "f16", "data": "synthetic",
            "config": {"workload": workload_desc,
                       "per_gpu_batch": n, "global_batch": n * world, "image": f"{ih}x{iw}", "heatmap": f"{ih // 4}x{iw // 4}x17",
                       "sharding": "crops split over ranks, no data-path collective",
                       "gflop_per_image": round(2e-9 * plan.total_macs / n * passes, 3),
                       "launches_per_step": (sum(1 for e in plan.layer_info if e["kind"] != "barrier") + 1) * passes,
                       "flip_test": (f"one forward of {n_plan} = [{n} crops | {n} mirrors]" if n_plan != n else f"two forwards of {n}") if flip
                       else None,
                       "execution_lanes": 4 if any(e["kind"] == "barrier" for e in plan.layer_info) else 1},
        }
        if not args.no_roofline:
            result["roofline"] = roofline_report(plan, layers_csv=args.layers, workload=args.workload)
            log("roofline done")
        headline = world == 1 and args.workload == "hrnet_w32" and args.amp == "O0" and not args.leg
        if headline:
            result["hbm_ops"] = hbm_ops_report(mp, dev)
        if headline and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(net.state_dict(), mp, dev)
        if headline and not args.no_extra:
            result["extra_workloads"] = run_extra_legs([x for x in args.extra.split(",") if x] or None)
        result["n_gpus_seen"] = torch.cuda.device_count()
    if dist is not None and args.workload == "hrnet_w32" and args.amp == "O0" and not args.no_extra:
        # a multi-rank run also exercises the path's one collective: the DP training leg on ALL ranks, after the headline's timed
        # region (the inference path itself has no data-path collective)
        try:
            leg = dp_train_leg(args, mp, dev, dist, world, rank)
        except Exception as exc:  # never take the headline line down; build failures of ONE rank are agreed on inside the leg
            leg = {"error": f"{type(exc).__name__}: {exc}"}  # (all_ranks_ok), so no rank enters the timed collectives alone
        if rank == 0:
            result.setdefault("extra_workloads", {})["config3_train_ampO2_dp"] = leg
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
