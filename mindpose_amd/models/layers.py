"""Parameter holders and the launch-plan builder shared by the backbones and heads.

The modules only OWN parameters (named exactly like the reference's MindSpore cells so a mindpose
checkpoint maps 1:1); all arithmetic happens in ``libmindpose_hip.so``.  A network forward is
recorded once per input shape into a native launch plan (``mp_plan_*``) and replayed by one C call.
"""
import ctypes
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib

BN_EPS = 1e-5  # mindspore.nn.BatchNorm2d default eps
F16_VARIANTS = 48  # csrc/conv_f16.h F_COUNT: tile shapes the fp16 autotuner times per launch shape
F16_WS_BASE = 37   # csrc/conv_f16.h F_WS_BASE: first weight-stationary persistent shape (conv_f16_ws.hip)


class Conv2d(nn.Module):
    """Parameter holder for ``mindspore.nn.Conv2d`` (weight [Cout,Cin,k,k], optional bias)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 has_bias: bool = False) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if has_bias else None


class Conv2dTranspose(nn.Module):
    """Parameter holder for ``mindspore.nn.Conv2dTranspose(k=4, s=2, pad_mode="pad", padding=1)``
    (weight [Cin,Cout,4,4], no bias) - simple_baseline_head.py:80-88."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 4) -> None:
        super().__init__()
        if kernel_size != 4:
            raise ValueError("Invalid deconv_kernel.")  # only the k=4 / padding=1 recipe is implemented natively
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 4, 4))


class BatchNorm2d(nn.Module):
    """Parameter holder for ``mindspore.nn.BatchNorm2d`` (gamma, beta, moving_mean, moving_variance)."""

    def __init__(self, num_features: int) -> None:
        super().__init__()
        self.num_features = num_features
        self.gamma = nn.Parameter(torch.ones(num_features))
        self.beta = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("moving_mean", torch.zeros(num_features))
        self.register_buffer("moving_variance", torch.ones(num_features))

    def folded(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Eval-mode affine: y = x*scale + shift."""
        scale = self.gamma.detach().float() / torch.sqrt(self.moving_variance.float() + BN_EPS)
        shift = self.beta.detach().float() - self.moving_mean.float() * scale
        return scale.contiguous(), shift.contiguous()


_TUNE_CACHE: Dict[Tuple, int] = {}
_TUNE_FILE_LOADED = False


def _tune_file() -> Optional[str]:
    """MINDPOSE_TUNE_CACHE=<path>: persist the autotuner's choices (JSON, keyed by launch shape) so that a later process
    - a profiling run, a production worker - replays them without timing trial launches."""
    return os.environ.get("MINDPOSE_TUNE_CACHE") or None


_BUILD_ID = None


def _tune_stamp() -> str:
    """Variant indices only mean something for one BUILD of the kernels: the stamp carries the library's version string and a
    digest of the shared object itself (a rebuilt kernel invalidates the persisted choices without a hand-bumped version)."""
    global _BUILD_ID
    if _BUILD_ID is None:
        import hashlib
        h = hashlib.sha1()
        try:
            with open(_lib.LIB_PATH, "rb") as f:
                for block in iter(lambda: f.read(1 << 20), b""):
                    h.update(block)
            _BUILD_ID = h.hexdigest()[:16]
        except OSError:
            _BUILD_ID = "unknown"
    return f"{_lib.load().mp_version().decode()}|{_BUILD_ID}"


def _tune_load() -> None:
    global _TUNE_FILE_LOADED
    path = _tune_file()
    if _TUNE_FILE_LOADED or not path:
        return
    _TUNE_FILE_LOADED = True
    try:
        import json
        with open(path) as f:
            doc = json.load(f)
        if doc.get("stamp") != _tune_stamp():  # written by another library version: variant indices may have moved
            return
        for k, v in doc.get("choices", {}).items():
            _TUNE_CACHE.setdefault(k, int(v))
    except (OSError, ValueError, AttributeError):
        pass


def _dist_rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def share_tuner_choices(group=None) -> int:
    """THE collective of the tuner, at a point the CALLER chooses: rank 0's whole choice table is broadcast once
    (``broadcast_object_list``) and every other rank adopts it, so that the ranks of one job run the same numeric form of every
    layer (the fp32 candidates differ numerically: Winograd vs direct, one GEMM launch vs four phase convs).  Every rank of
    ``group`` must call it, at the same point of its program - e.g. right after rank 0's warm-up (`tune_on_rank0_first`).  The
    tuner itself never communicates: a plan that only one rank builds (EvalCallback's rank-0 evaluation, a no-grad probe) can
    therefore never strand or cross-match a collective.  Returns the number of choices adopted (0 on rank 0 / one rank)."""
    if _dist_rank_world()[1] == 1:
        return 0
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)  # ranks OF THE GROUP: a sub-group without global rank 0 has its own sender
    if world <= 1:
        return 0
    box = [{k: v for k, v in _TUNE_CACHE.items() if isinstance(k, str)} if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if rank == 0:
        return 0
    _TUNE_CACHE.update(box[0])
    return len(box[0])


def tune_on_rank0_first(build, group=None):
    """``build()`` - anything that records plans / runs warm-up passes and contains NO collective - on rank 0 first (it tunes),
    then `share_tuner_choices`, then on the other ranks (every shape is a cache hit: nothing is timed twice, all ranks run rank 0's
    forms).  One rank: just ``build()``.  Rank 0 reaches the broadcast even when its ``build()`` raised (the exception is re-raised
    behind it), so a failure on rank 0 cannot leave the other ranks waiting."""
    if _dist_rank_world()[1] == 1:
        return build()
    import torch.distributed as dist
    rank = dist.get_rank(group)
    out, failure = None, None
    if rank == 0:
        try:
            out = build()
        except Exception as exc:  # noqa: BLE001 - re-raised below, after the collective every rank is waiting in
            failure = exc
    share_tuner_choices(group)
    if failure is not None:
        raise failure
    return out if rank == 0 else build()


def _tune_save() -> None:
    """Whole-file replace through a temporary (a torn file would silently drop the cache); rank 0 is the only writer of a job."""
    path = _tune_file()
    if not path or _dist_rank_world()[0] != 0:
        return
    try:
        import json
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump({"stamp": _tune_stamp(), "choices": {k: v for k, v in _TUNE_CACHE.items() if isinstance(k, str)}}, f)
        os.replace(tmp, path)
    except OSError:
        pass


class ActC8:
    """fp16 activation in the channel-blocked layout of the fp16 matrix-core path: physical tensor
    ``[N][ceil(C/8)][H][W][8]`` halfs (padding channels zero), ``shape`` = the logical NCHW shape."""

    def __init__(self, n: int, c: int, h: int, w: int, device) -> None:
        self.shape = torch.Size((n, c, h, w))
        self.c8_tensor = torch.zeros(n, (c + 7) // 8, h, w, 8, device=device, dtype=torch.float16)
        self.device = self.c8_tensor.device

    def data_ptr(self) -> int:
        return self.c8_tensor.data_ptr()

    def to_nchw(self) -> torch.Tensor:
        """fp32 NCHW copy (tests / debugging)."""
        n, c, h, w = self.shape
        return self.c8_tensor.permute(0, 1, 4, 2, 3).reshape(n, -1, h, w)[:, :c].float().contiguous()


# launches below this many multiply-accumulates keep the library's heuristic (MINDPOSE_TUNE_MIN_MACS overrides).  Round 4 tuned from
# 2^26 up - which left EVERY layer of a one-crop forward (28 M MACs per 32-channel conv at N = 1) on the heuristic: a top-down
# pipeline serves a handful of crops per frame, and there the tile choice decides whether a launch covers 8 or 64 CUs
_TUNE_MIN_MACS = int(os.environ.get("MINDPOSE_TUNE_MIN_MACS", str(1 << 22)))


def _autotune(key, macs, n_variants, launch) -> int:
    """Time ``launch(v)`` for every tile variant (HIP events, best of two groups of 5 launches; MP_ERR_UNSUPPORTED = variant not
    available, any other error code raises) and cache the winner per launch shape; -1 = library heuristic when tuning is off (MINDPOSE_AUTOTUNE=0) or
    pointless (tiny layers)."""
    if os.environ.get("MINDPOSE_AUTOTUNE", "1") == "0":
        return -1
    _tune_load()
    key = repr(key)
    hit = _TUNE_CACHE.get(key)
    if hit is not None:
        return hit
    best, best_t = -1, None
    if macs >= _TUNE_MIN_MACS:  # a miss is timed on whichever rank meets it - no communication here (share_tuner_choices)
        for v in range(n_variants):
            rc = launch(v)
            if rc == -3:  # MP_ERR_UNSUPPORTED: this variant does not serve the shape
                continue
            _lib.check(rc, f"tuner trial launch, variant {v}, {key}")  # any other code (a HIP error) must not silently drop a candidate
            t = None
            for _ in range(2):  # best of two groups of five: one group of three mis-ranked close candidates run to run (+-1.5 %)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    launch(v)
                e1.record()
                e1.synchronize()
                dt = e0.elapsed_time(e1)
                t = dt if t is None or dt < t else t
            if best_t is None or t < best_t:
                best, best_t = v, t
            log = os.environ.get("MINDPOSE_TUNE_LOG")  # per-candidate timings (ms per 5 launches), for kernel work
            if log:
                with open(log, "a") as fh:
                    fh.write(f"{key}\t{v}\t{t:.4f}\n")
    _TUNE_CACHE[key] = best
    if macs >= _TUNE_MIN_MACS:
        _tune_save()
    return best


BLOCK_ROWS = (0, 4, 2, 1)  # band heights the fused-BasicBlock tuner times (0 = the tallest that fits)
F32_VARIANTS = 9   # direct MFMA tile variants 0..7 (csrc/conv_mfma.h ConvVariant) + 8 = the streaming 1x1 kernel (conv_pw_f32.hip)
F32_WINOGRAD = 9   # the tuner's index of the Winograd F(2x2,3x3) form (csrc/conv_wino_f32.hip)
F32_GEMM = 10      # the blocked-GEMM 1x1 kernel (csrc/conv_gemm_f32.hip; conv_api.hip kGemm)
F32_SMALL = 11     # the K-split kernel for small problems - a handful of crops (csrc/conv_small_f32.hip; conv_api.hip kSmall)
F32_SMALL_WIDE = 12  # ... with 48 / 64 pixels per workgroup (a few dozen crops: the weights of a workgroup serve more pixels)


def winograd_enabled() -> bool:
    """``MINDPOSE_WINOGRAD=0`` keeps every fp32 3x3 convolution on the direct kernel (bit-identical to round 1's results)."""
    return os.environ.get("MINDPOSE_WINOGRAD", "1") != "0"


def tune_conv_variant(lib, d, x, packed, scale, shift, res1, res2, out, half: bool = False, packed_u=None, stats=None) -> int:
    """Pick the tile variant for one conv launch shape by timing the candidates on the layer's real buffers.  Results
    are cached per shape, so a network's ~40 distinct shapes are tuned once per process.  ``packed_u`` (fp32 only): the
    Winograd-transformed weights; the Winograd form then competes as index ``F32_WINOGRAD``.  ``stats`` (fp16 training):
    ``dict(mode, z, y, relu)`` - the launch is the one with BatchNorm statistics in its epilogue (mp_f16_conv2d_fwd_stats: other
    register budgets, two more tensor reads in mode 2), timed as such and cached under its own key; with ``pre = dict(scale, shift, y,
    relu)`` the launch also applies the BatchNorm of the layer below on its operand (candidates: mp_f16_conv_pre_supported)."""
    key = tuple(getattr(d, f) for f, _ in d._fields_) + (res1 is not None, res2 is not None, str(out.device), half)
    if packed_u is not None:
        key += ("wino",)
    if stats is not None:
        key += ("stats", int(stats["mode"]), int(bool(stats.get("relu"))))
        if stats.get("pre") is not None:  # BatchNorm apply of the layer below on the operand: its own candidate set, its own key
            key += ("pre",)
    macs = d.n * d.conv_h * d.conv_w * d.cout * d.cin * d.kh * d.kw
    stream = _lib.stream()
    # in-place accumulation (out aliases res1) must not be disturbed by trial launches: tune into a scratch copy
    alias = res1 is not None and res1.data_ptr() == out.data_ptr()
    trial_out = out
    if alias:
        trial_out = torch.empty_like(out) if torch.is_tensor(out) else ActC8(*out.shape, out.device)
    fn = lib.mp_f16_conv2d_fwd if half else lib.mp_conv2d_fwd_variant

    stats_buf = {}
    # a statistics build that leaves more than 512 partial slots per channel costs its consumer an extra fold launch (~5 us): such
    # variants compete only when no variant of the shape stays within 512
    slot_cap = [512]
    with_pre = stats is not None and stats.get("pre") is not None
    if stats is not None and not any(0 < lib.mp_f16_conv_stats_parts(ctypes.byref(d), v) <= 512 for v in range(F16_VARIANTS)
                                     if not with_pre or lib.mp_f16_conv_pre_supported(ctypes.byref(d), v)):
        slot_cap[0] = 1 << 30

    no_small = os.environ.get("MINDPOSE_F32_SMALL", "1") == "0"  # before / after evidence: the candidate set without the small-problem kernel
    no_ws = half and os.environ.get("MINDPOSE_F16_WS", "1") == "0"  # before / after evidence: the round-3 candidate set

    def launch(v):
        if no_ws and v >= F16_WS_BASE:  # (the round-4 weights-in-registers shapes 45.. included)
            return -3
        if stats is not None:
            n_parts = lib.mp_f16_conv_stats_parts(ctypes.byref(d), v)
            if n_parts <= 0 or n_parts > slot_cap[0]:
                return -3
            need = (d.cout + 7) // 8 * n_parts * 16
            if stats_buf.get("n", 0) < need:
                stats_buf["t"], stats_buf["n"] = torch.empty(need, device=out.device, dtype=torch.float32), need
            st = _lib.ConvStats(mode=int(stats["mode"]), relu=int(bool(stats.get("relu"))), partials=stats_buf["t"].data_ptr(),
                                partials_bytes=need * 4, z=_lib.ptr(stats.get("z")), y=_lib.ptr(stats.get("y")) if stats.get("relu") else None)
            pre = stats.get("pre")
            if pre is not None:
                if not lib.mp_f16_conv_pre_supported(ctypes.byref(d), v):
                    return -3
                st.pre_scale, st.pre_shift, st.pre_out, st.pre_relu = _lib.ptr(pre["scale"]), _lib.ptr(pre["shift"]), _lib.ptr(pre["y"]), int(pre["relu"])
            return lib.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift),
                                               _lib.ptr(res1), _lib.ptr(trial_out), ctypes.byref(st), stream)
        if not half and v in (F32_SMALL, F32_SMALL_WIDE) and no_small:
            return -3
        if not half and v == F32_WINOGRAD:
            if packed_u is None:
                return -3  # MP_ERR_UNSUPPORTED: no Winograd form of this layer
            return lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(packed_u), _lib.ptr(scale), _lib.ptr(shift),
                                              _lib.ptr(res1), _lib.ptr(res2), _lib.ptr(trial_out), stream)
        return fn(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res1),
                  _lib.ptr(res2), _lib.ptr(trial_out), stream)

    return _autotune(key, macs, F16_VARIANTS if half else F32_SMALL_WIDE + 1, launch)


class Plan:
    """A recorded forward: native ``mp_plan`` + the tensors it points into."""

    def __init__(self, device: torch.device, half: bool = False) -> None:
        self.lib = _lib.load()
        self.device = device
        self.half = half  # amp O2/O3: fp16 matrix-core kernels over channel-blocked fp16 activations
        self.lanes = os.environ.get("MINDPOSE_PLAN_LANES", "1") != "0"
        self.handle = ctypes.c_void_p(self.lib.mp_plan_create())
        if not self.handle:
            raise _lib.MindposeHipError("mp_plan_create failed")
        self.keep: List[torch.Tensor] = []  # every buffer the plan references
        self.input: Optional[torch.Tensor] = None
        self.output: Optional[torch.Tensor] = None
        self.layer_info: List[Dict] = []  # per entry: kind, shapes, MACs (for the roofline report)
        self._packed: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._folded: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._graph = None
        self._graph_ok = True
        self._runs = 0
        self._marks: List[Tuple[int, object]] = []  # (entry index, lane | "barrier"): the lane structure, for the captured replay
        self._multi = False
        self._lane_streams: List[torch.cuda.Stream] = []

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.mp_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def alloc(self, *shape: int):
        """Activation buffer of the plan's compute type (fp32 NCHW tensor, or ActC8 in fp16 mode)."""
        if self.half:
            t = ActC8(*shape, self.device)
            self.keep.append(t.c8_tensor)
            return t
        return self.alloc_f32(*shape)

    def alloc_f32(self, *shape: int) -> torch.Tensor:
        t = torch.empty(shape, device=self.device, dtype=torch.float32)
        self.keep.append(t)
        return t

    def to_c8(self, x: torch.Tensor) -> "ActC8":
        """NCHW fp32 -> channel-blocked fp16 (first entry of an fp16 plan)."""
        n, c, h, w = x.shape
        out = self.alloc(n, c, h, w)
        _lib.check(self.lib.mp_plan_add_layout_f16(self.handle, 1, _lib.ptr(x), _lib.ptr(out), n, c, h, w), "mp_plan_add_layout_f16")
        self.layer_info.append(dict(kind="to_c8", n=n, c=c, h=h, w=w, macs=0))
        return out

    def enter(self, x):
        """Switch to the plan's compute type: in an fp16 plan an fp32 NCHW tensor is converted to channel-blocked fp16 here
        (HRNet: the input image; ResNet: after the fp32 7x7 stem and max-pool); otherwise ``x`` is returned unchanged."""
        if self.half and not isinstance(x, ActC8):
            return self.to_c8(x)
        return x

    def from_c8(self, x: "ActC8") -> torch.Tensor:
        """channel-blocked fp16 -> NCHW fp32 (the network output handed to the decoder / loss)."""
        n, c, h, w = x.shape
        out = self.alloc_f32(n, c, h, w)
        _lib.check(self.lib.mp_plan_add_layout_f16(self.handle, 0, _lib.ptr(x), _lib.ptr(out), n, c, h, w), "mp_plan_add_layout_f16")
        self.layer_info.append(dict(kind="from_c8", n=n, c=c, h=h, w=w, macs=0))
        return out

    # -- execution lanes ----------------------------------------------------------------------
    def set_lane(self, lane: int) -> None:
        """Entries recorded from now on replay on execution lane ``lane`` (0 = the caller's stream, 1..3 = side streams of
        the plan): independent sub-graphs overlap on the chip.  No-op when lanes are disabled (MINDPOSE_PLAN_LANES=0)."""
        if self.lanes:
            _lib.check(self.lib.mp_plan_set_lane(self.handle, int(lane) % 4), "mp_plan_set_lane")
            self._marks.append((int(self.lib.mp_plan_size(self.handle)), int(lane) % 4))
            if lane:
                # The native multi-lane replay (plain HIP streams and events inside mp_plan_run, all-to-all waits at a barrier)
                # cannot be captured: ROCm 7.2's capture_end faults on it.  The captured replay is driven from here instead
                # (_replay_lanes): the lanes as torch streams, every barrier a STAR - side lanes join the origin stream and fork
                # from it again - which is the only fork / join shape capture survives here (the captured training step has no
                # other).  Inside a hipGraph a cross-lane dependency costs ~10 us instead of the ~30 us of an event wait between
                # two queues (tools/timeline.sh on the inference plan: 12 barriers per HRNet-W32 forward): O2 inference +10 %, fp32 +2 %.
                # MINDPOSE_PLAN_GRAPH_LANES=0: keep replaying through the native call.
                self._multi = True
                if os.environ.get("MINDPOSE_PLAN_GRAPH_LANES", "1") == "0":
                    self._graph_ok = False

    def barrier(self) -> None:
        """Every lane waits for everything recorded so far on every other lane."""
        if self.lanes:
            self._marks.append((int(self.lib.mp_plan_size(self.handle)), "barrier"))
            _lib.check(self.lib.mp_plan_add_barrier(self.handle), "mp_plan_add_barrier")
            self.layer_info.append(dict(kind="barrier", macs=0))

    def run(self) -> None:
        """Replay the recorded forward: as one hipGraph launch once captured (MINDPOSE_HIP_GRAPH=0 disables it),
        otherwise as one native call that issues the launches back to back."""
        if self._graph is not None:
            self._graph.replay()
            return
        _lib.check(self.lib.mp_plan_run(self.handle, _lib.stream()), "mp_plan_run")
        self._runs += 1
        if self._runs == 2 and self._graph_ok and os.environ.get("MINDPOSE_HIP_GRAPH", "1") != "0":
            self._capture()

    def _capture(self) -> None:
        # the plan has run twice (kernel attributes set, nothing allocates): capture the launch sequence
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                if self._multi:
                    self._replay_lanes()
                else:
                    _lib.check(self.lib.mp_plan_run(self.handle, _lib.stream()), "mp_plan_run (capture)")
            self._graph = graph
        except Exception as exc:  # capture unsupported in this context: keep direct launches (same kernels)
            self._graph_ok = False
            import warnings
            warnings.warn(f"hipGraph capture of the launch plan failed, using direct launches: {exc}")

    def _replay_lanes(self) -> None:
        """The recorded forward with lane l on torch stream l (0 = the current stream): what mp_plan_run does natively, in a
        form stream capture accepts.  Barrier entries become all-to-all wait_stream calls among the lanes in use."""
        cur = torch.cuda.current_stream(self.device)
        lanes_used = sorted({m for _, m in self._marks if m != "barrier"} | {0})
        while len(self._lane_streams) < max(lanes_used):
            self._lane_streams.append(torch.cuda.Stream(device=self.device))
        streams = [cur] + self._lane_streams
        for l in lanes_used[1:]:
            streams[l].wait_stream(cur)
        pos, lane = 0, 0
        for idx, what in self._marks + [(len(self), None)]:
            if idx > pos:
                with torch.cuda.stream(streams[lane]):
                    _lib.check(self.lib.mp_plan_run_range(self.handle, pos, idx - pos, _lib.stream()), "mp_plan_run_range")
                pos = idx
            if what == "barrier":
                pos = idx + 1  # the barrier entry itself launches nothing
                # star-shaped: every side lane joins the origin stream, then forks from it again.  Side-to-side waits (a stream
                # joining another FORKED stream) make ROCm 7.2's capture_end fault - the training step's captures never have them
                for l in lanes_used[1:]:
                    cur.wait_stream(streams[l])
                for l in lanes_used[1:]:
                    streams[l].wait_stream(cur)
            elif what is not None:
                lane = what
        for l in lanes_used[1:]:
            cur.wait_stream(streams[l])

    def run_range(self, first: int, count: int) -> None:
        _lib.check(self.lib.mp_plan_run_range(self.handle, first, count, _lib.stream()), "mp_plan_run_range")

    def entry_info(self, index: int) -> Dict[str, int]:
        """Launch geometry the native side chose for entry ``index`` (+ the python-side shape record)."""
        buf = (ctypes.c_int64 * 12)()
        _lib.check(self.lib.mp_plan_entry_info(self.handle, index, buf), "mp_plan_entry_info")
        keys = ["kind_id", "ks", "stride", "variant", "workgroups", "lds_bytes", "cout_tile", "pixel_tile", "cin_chunk",
                "images_per_tile", "rows_per_tile", "light"]
        info = dict(zip(keys, [int(v) for v in buf]))
        info.update(self.layer_info[index])
        return info

    def __len__(self) -> int:
        return int(self.lib.mp_plan_size(self.handle))

    @property
    def total_macs(self) -> int:
        return sum(e.get("macs", 0) for e in self.layer_info)

    # -- weights ------------------------------------------------------------------------------
    def _pack(self, weight: torch.Tensor, cout: int, cin: int, k: int, transposed: bool, py: int, px: int,
              half: bool = False) -> torch.Tensor:
        key = (id(weight), py if transposed else -1, px if transposed else -1, half)
        if key in self._packed:
            return self._packed[key]
        w = weight.detach().to(self.device, torch.float32).contiguous()
        if half:
            nbytes = self.lib.mp_f16_packed_weight_bytes(cout, cin, k, k)
            packed = torch.empty(nbytes // 2, device=self.device, dtype=torch.float16)
            _lib.check(self.lib.mp_f16_pack_weight(_lib.ptr(w), _lib.ptr(packed), cout, cin, k, k, int(transposed), py, px,
                                                   _lib.stream()), "mp_f16_pack_weight")
            self.keep += [w, packed]
            self._packed[key] = packed
            return packed
        nbytes = self.lib.mp_conv_packed_weight_bytes(cout, cin, k, k)
        packed = torch.empty(nbytes // 4, device=self.device, dtype=torch.float32)
        _lib.check(self.lib.mp_conv_pack_weight(_lib.ptr(w), _lib.ptr(packed), cout, cin, k, k, int(transposed), py, px,
                                                _lib.stream()), "mp_conv_pack_weight")
        self.keep += [w, packed]
        self._packed[key] = packed
        return packed

    def _pack_winograd(self, weight: torch.Tensor, cout: int, cin: int) -> torch.Tensor:
        """U = G w G^T of a 3x3 weight, [Cin_pad4][Cout_pad16][16] fp32 (mp_conv_winograd_pack_weight)."""
        key = (id(weight), "wino")
        if key in self._packed:
            return self._packed[key]
        w = weight.detach().to(self.device, torch.float32).contiguous()
        packed = torch.empty(self.lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, device=self.device, dtype=torch.float32)
        _lib.check(self.lib.mp_conv_winograd_pack_weight(_lib.ptr(w), _lib.ptr(packed), cout, cin, _lib.stream()),
                   "mp_conv_winograd_pack_weight")
        self.keep += [w, packed]
        self._packed[key] = packed
        return packed

    def _affine(self, cout: int, bn: Optional[BatchNorm2d], bias: Optional[torch.Tensor], half: bool = False):
        # keyed on both parameter holders and the width: a conv with neither BatchNorm nor bias must not share the
        # ones / zeros pair of another width (the kernel reads cout entries)
        key = (id(bn) if bn is not None else None, id(bias) if bias is not None else None, cout, half)
        if key in self._folded:
            return self._folded[key]
        if bn is not None:
            scale, shift = bn.folded()
            scale, shift = scale.to(self.device), shift.to(self.device)
        else:
            scale = torch.ones(cout, device=self.device)
            shift = bias.detach().float().to(self.device).contiguous() if bias is not None else torch.zeros(cout, device=self.device)
        if half:  # the fp16 kernel reads 4 couts per lane: arrays padded to Cout_pad16 with zeros
            pad = (-cout) % 16
            scale = torch.cat([scale, scale.new_zeros(pad)]).contiguous()
            shift = torch.cat([shift, shift.new_zeros(pad)]).contiguous()
        self.keep += [scale, shift]
        self._folded[key] = (scale, shift)
        return scale, shift

    # -- ops ----------------------------------------------------------------------------------
    def conv(self, x: torch.Tensor, conv: Conv2d, bn: Optional[BatchNorm2d] = None, relu: bool = False,
             res1: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, upsample: int = 1) -> torch.Tensor:
        """conv (+BN or bias) (+res1) (+res2) (+ReLU); ``upsample`` = nearest factor applied while storing
        (``out`` / ``res*`` then live at the up-sampled resolution)."""
        n, cin, h, w = x.shape
        k, s, pad = conv.kernel_size, conv.stride, conv.padding
        if cin != conv.in_channels:
            raise ValueError(f"conv expects {conv.in_channels} input channels, got {cin}")
        half = isinstance(x, ActC8)  # the kernel family follows the activation: fp32 NCHW tensors take the fp32 kernels
        ho = (h + 2 * pad - k) // s + 1
        wo = (w + 2 * pad - k) // s + 1
        oh, ow = ho * upsample, wo * upsample
        if out is None:
            out = self.alloc(n, conv.out_channels, oh, ow) if half else self.alloc_f32(n, conv.out_channels, oh, ow)
        if tuple(out.shape) != (n, conv.out_channels, oh, ow):
            raise ValueError(f"bad out shape {tuple(out.shape)}")
        for r in (res1, res2):
            if r is not None and tuple(r.shape) != tuple(out.shape):
                raise ValueError(f"residual shape {tuple(r.shape)} != out shape {tuple(out.shape)}")
        packed = self._pack(conv.weight, conv.out_channels, cin, k, False, 0, 0, half)
        scale, shift = self._affine(conv.out_channels, bn, conv.bias, half)
        d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=conv.out_channels, kh=k, kw=k, stride=s, pad_top=pad,
                          pad_left=pad, conv_h=ho, conv_w=wo, out_h=oh, out_w=ow, out_mul=upsample, out_rep=upsample,
                          out_off_y=0, out_off_x=0, relu=int(relu), flags=0)
        if half and upsample != 1:
            raise NotImplementedError("fp16 plans add up-sampled terms with fuse_sum, not through the conv epilogue")
        packed_u = None
        if (not half and winograd_enabled() and os.environ.get("MINDPOSE_AUTOTUNE", "1") != "0"
                and self.lib.mp_conv_winograd_supported(ctypes.byref(d)) == 0):
            packed_u = self._pack_winograd(conv.weight, conv.out_channels, cin)
        variant = tune_conv_variant(self.lib, d, x, packed, scale, shift, res1, res2, out, half=half, packed_u=packed_u)
        if not half and variant == F32_WINOGRAD:
            _lib.check(self.lib.mp_plan_add_conv_winograd(self.handle, ctypes.byref(d), _lib.ptr(x), _lib.ptr(packed_u), _lib.ptr(scale),
                                                          _lib.ptr(shift), _lib.ptr(res1), _lib.ptr(res2), _lib.ptr(out)),
                       "mp_plan_add_conv_winograd")
            self.layer_info.append(dict(kind="conv_winograd", k=k, stride=s, cin=cin, cout=conv.out_channels, h=h, w=w, n=n,
                                        macs=n * ho * wo * conv.out_channels * cin * k * k))
            return out
        add = self.lib.mp_plan_add_conv_f16 if half else self.lib.mp_plan_add_conv_variant
        _lib.check(add(self.handle, ctypes.byref(d), variant, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift),
                       _lib.ptr(res1), _lib.ptr(res2), _lib.ptr(out)), "mp_plan_add_conv")
        self.layer_info.append(dict(kind="conv_f16" if half else "conv", k=k, stride=s, cin=cin, cout=conv.out_channels,
                                    h=h, w=w, n=n, macs=n * ho * wo * conv.out_channels * cin * k * k))
        return out

    def fuses_basic_block(self, x: torch.Tensor, conv1: Conv2d, conv2: Conv2d) -> bool:
        """The fused fp16 BasicBlock kernels cover the 32-channel branch (four 8-channel blocks) and, round 4, the 64-channel branch
        and 128-channel branches (3x3 stride 1 both convs; the library says whether the map fits a band: ``mp_f16_basicblock_supported``);
        ``MINDPOSE_FUSE_BLOCK=0`` keeps the two-launch path, ``MINDPOSE_FUSE_BLOCK64=0`` for the 64 / 128-channel blocks only."""
        if not isinstance(x, ActC8) or os.environ.get("MINDPOSE_FUSE_BLOCK", "1") == "0":
            return False
        n, c, h, w = x.shape
        if not all(cv.in_channels == c and cv.out_channels == c and cv.kernel_size == 3 and cv.stride == 1 and cv.padding == 1
                   and cv.bias is None for cv in (conv1, conv2)):
            return False
        if c in (64, 128) and os.environ.get("MINDPOSE_FUSE_BLOCK64", "1") == "0":
            return False
        return (24 < c <= 32 or c in (64, 128)) and self.lib.mp_f16_basicblock_supported(n, c, h, w) == 1

    def basic_block(self, x: torch.Tensor, conv1: Conv2d, bn1: BatchNorm2d, conv2: Conv2d, bn2: BatchNorm2d) -> torch.Tensor:
        """relu(bn2(conv2(relu(bn1(conv1 x)))) + x) in ONE launch (mp_f16_basicblock_fwd): the intermediate tensor stays in LDS;
        bit-identical to the two conv launches."""
        n, c, h, w = x.shape
        out = self.alloc(n, c, h, w)
        p1, p2 = (self._pack(cv.weight, c, c, 3, False, 0, 0, True) for cv in (conv1, conv2))
        (s1, b1), (s2, b2) = self._affine(c, bn1, None, True), self._affine(c, bn2, None, True)
        # rows per workgroup band: 0 = the tallest band that fits (fewest halo rows: right when N x bands fills the chip); a handful
        # of crops leaves 2 - 8 workgroups per launch, and shorter bands trade recomputed halo rows for parallelism - timed per shape
        stream = _lib.stream()

        def launch(v):
            return self.lib.mp_f16_basicblock_fwd(_lib.ptr(x), _lib.ptr(p1), _lib.ptr(s1), _lib.ptr(b1), _lib.ptr(p2), _lib.ptr(s2), _lib.ptr(b2),
                                                  _lib.ptr(out), n, c, h, w, BLOCK_ROWS[v], stream)
        pick = _autotune(("basicblock_f16", n, c, h, w, str(out.device)), 2 * n * h * w * c * c * 9, len(BLOCK_ROWS), launch)
        rows = BLOCK_ROWS[pick] if pick >= 0 else 0
        _lib.check(self.lib.mp_plan_add_basicblock_f16(self.handle, _lib.ptr(x), _lib.ptr(p1), _lib.ptr(s1), _lib.ptr(b1),
                                                       _lib.ptr(p2), _lib.ptr(s2), _lib.ptr(b2), _lib.ptr(out), n, c, h, w, rows),
                   "mp_plan_add_basicblock_f16")
        self.layer_info.append(dict(kind="basicblock_f16", k=3, stride=1, cin=c, cout=c, h=h, w=w, n=n,
                                    macs=2 * n * h * w * c * c * 9))
        return out

    def fuses_dual_pw(self, x: torch.Tensor, conv_a: Conv2d, conv_b: Conv2d) -> bool:
        """Two 1x1 convs on ONE 64-channel input as one launch (mp_f16_dual_pw_fwd): the down-sample conv (64 -> 256) and the reduce conv
        (64 -> 64) of stage 1's first Bottleneck.  fp16 plans, stride 1, no bias, pixel count a multiple of 64; switched with the chain
        launch (``MINDPOSE_FUSE_PWCHAIN``)."""
        if not isinstance(x, ActC8) or os.environ.get("MINDPOSE_FUSE_PWCHAIN", "1") == "0":
            return False
        n, c, h, w = x.shape
        ok = lambda cv, co: (cv.in_channels == 64 and cv.out_channels == co and cv.kernel_size == 1 and cv.stride == 1  # noqa: E731
                             and cv.padding == 0 and cv.bias is None)
        return c == 64 and ok(conv_a, 256) and ok(conv_b, 64) and (h * w) % 64 == 0 and n * 32 * h * w * 16 < 0x7FFFFFF0

    def dual_pw(self, x: torch.Tensor, conv_a: Conv2d, bn_a: BatchNorm2d, relu_a: bool, conv_b: Conv2d, bn_b: BatchNorm2d,
                relu_b: bool) -> Tuple[torch.Tensor, torch.Tensor]:
        """(act_a(bn_a(conv_a x)), act_b(bn_b(conv_b x))) in one launch, x read once; bit-identical to the two conv launches."""
        n, c, h, w = x.shape
        ya, zb = self.alloc(n, conv_a.out_channels, h, w), self.alloc(n, conv_b.out_channels, h, w)
        pa = self._pack(conv_a.weight, conv_a.out_channels, c, 1, False, 0, 0, True)
        pb = self._pack(conv_b.weight, conv_b.out_channels, c, 1, False, 0, 0, True)
        (sa, ba), (sb, bb) = self._affine(conv_a.out_channels, bn_a, None, True), self._affine(conv_b.out_channels, bn_b, None, True)
        _lib.check(self.lib.mp_plan_add_dual_pw_f16(self.handle, _lib.ptr(x), _lib.ptr(pa), _lib.ptr(sa), _lib.ptr(ba), int(relu_a), _lib.ptr(pb),
                                                    _lib.ptr(sb), _lib.ptr(bb), int(relu_b), _lib.ptr(ya), _lib.ptr(zb), n, c,
                                                    conv_a.out_channels, conv_b.out_channels, h, w), "mp_plan_add_dual_pw_f16")
        self.layer_info.append(dict(kind="pwchain_f16", k=1, stride=1, cin=c, cout=conv_a.out_channels, h=h, w=w, n=n,
                                    macs=n * h * w * c * (conv_a.out_channels + conv_b.out_channels)))
        return ya, zb

    def fuses_stem(self, x: torch.Tensor, conv: Conv2d) -> bool:
        """Can the first conv run as the dedicated streaming kernel with (tap, channel) as its k axis?  fp16 plans: mp_f16_stem_conv_fwd
        reads the fp32 NCHW image itself (no layout pass, the 27 real k positions of a 3-channel 3x3 conv in ONE k-step); fp32 plans:
        mp_stem_conv_fwd (seven k-steps of 4).  ``MINDPOSE_FUSE_STEM=0`` keeps (layout pass +) the general conv."""
        if isinstance(x, ActC8) or os.environ.get("MINDPOSE_FUSE_STEM", "1") == "0":
            return False
        n, c, h, w = x.shape
        lds = (6 if self.half else 12) * (w + 4) * 17 + 16  # the 17 staged rows of the three planes (fp16 / fp32)
        return (c == 3 and conv.in_channels == 3 and conv.out_channels == 64 and conv.kernel_size == 3 and conv.stride == 2
                and conv.padding == 1 and conv.bias is None and h % 2 == 0 and w % 32 == 0 and x.dtype == torch.float32
                and lds <= 64 * 1024 and n * 8 * (h // 2) * (w // 2) * 16 < 0x7FFFFFF0)

    def stem(self, x: torch.Tensor, conv: Conv2d, bn: BatchNorm2d):
        """relu(bn(conv x)) of the network's first conv (hrnet.py:377-385) in one launch of the dedicated kernel: fp16 plans - from the
        fp32 image to the channel-blocked fp16 activation; fp32 plans - NCHW to NCHW."""
        n, _, h, w = x.shape
        wt = conv.weight.detach().to(self.device, torch.float32).contiguous()
        self.keep.append(wt)
        if not self.half:  # fp32 plans: the same (tap, channel) k axis on the fp32 matrix cores, NCHW in and out (mp_stem_conv_fwd)
            out = self.alloc_f32(n, 64, h // 2, w // 2)
            scale, shift = self._affine(64, bn, None, False)
            _lib.check(self.lib.mp_plan_add_stem_conv(self.handle, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(out),
                                                      n, h, w), "mp_plan_add_stem_conv")
            self.layer_info.append(dict(kind="stem_f32", k=3, stride=2, cin=3, cout=64, h=h, w=w, n=n, macs=n * (h // 2) * (w // 2) * 64 * 27))
            return out
        out = self.alloc(n, 64, h // 2, w // 2)
        scale, shift = self._affine(64, bn, None, True)
        _lib.check(self.lib.mp_plan_add_stem_conv_f16(self.handle, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(out),
                                                      n, h, w), "mp_plan_add_stem_conv_f16")
        self.layer_info.append(dict(kind="stem_f16", k=3, stride=2, cin=3, cout=64, h=h, w=w, n=n, macs=n * (h // 2) * (w // 2) * 64 * 27))
        return out

    def fuses_expand_reduce(self, mid: torch.Tensor, res: torch.Tensor, conv3: Conv2d, conv1_next: Conv2d) -> bool:
        """Can the expand conv of a Bottleneck and the reduce conv of the next one run as ONE launch (fp16 plans:
        mp_f16_expand_reduce_fwd, bit-identical to the two launches; fp32 plans: mp_expand_reduce_fwd, the persistent weight-stationary
        form - same values up to the association of the k sums)?  The 64 -> 256 -> 64 widths of HRNet's stage 1, 1x1 stride-1 convs
        without bias, maps whose pixel count is a multiple of 64.  ``MINDPOSE_FUSE_PWCHAIN=0`` (fp16) / ``MINDPOSE_FUSE_PWCHAIN32=0``
        (fp32) keep the two launches."""
        half = isinstance(mid, ActC8)
        if half != isinstance(res, ActC8) or os.environ.get("MINDPOSE_FUSE_PWCHAIN" if half else "MINDPOSE_FUSE_PWCHAIN32", "1") == "0":
            return False
        n, cm, h, w = mid.shape
        if not (cm == 64 and tuple(res.shape) == (n, 256, h, w) and self._pw(conv3, 64, 256) and self._pw(conv1_next, 256, 64)
                and (h * w) % 64 == 0):
            return False
        return n * 32 * h * w * 16 < 0x7FFFFFF0 if half else n * 256 * h * w * 4 < 0x7FFFFFF0

    @staticmethod
    def _pw(cv: Conv2d, ci: int, co: int) -> bool:
        return (cv.in_channels == ci and cv.out_channels == co and cv.kernel_size == 1 and cv.stride == 1 and cv.padding == 0
                and cv.bias is None)

    def fuses_ds_expand_reduce(self, x0: torch.Tensor, ds_conv: Conv2d, conv3: Conv2d, conv1_next: Conv2d) -> bool:
        """Can the FIRST Bottleneck's down-sample conv (hrnet.py:74-81, 64 -> 256 on the block's input) be computed inside the expand +
        reduce chain launch, instead of being written and read back as the residual tensor by a launch of its own?
        ``MINDPOSE_FUSE_PWCHAIN32_DS=0`` (fp32) / ``MINDPOSE_FUSE_PWCHAIN_DS=0`` (fp16: the dual 1x1 launch + the identity chain instead)
        keep the separate launch."""
        half = isinstance(x0, ActC8)
        if half and (os.environ.get("MINDPOSE_FUSE_PWCHAIN", "1") == "0" or os.environ.get("MINDPOSE_FUSE_PWCHAIN_DS", "1") == "0"):
            return False
        if not half and (os.environ.get("MINDPOSE_FUSE_PWCHAIN32", "1") == "0" or os.environ.get("MINDPOSE_FUSE_PWCHAIN32_DS", "1") == "0"):
            return False
        n, c, h, w = x0.shape
        if not (c == 64 and self._pw(ds_conv, 64, 256) and self._pw(conv3, 64, 256) and self._pw(conv1_next, 256, 64) and (h * w) % 64 == 0):
            return False
        return n * 32 * h * w * 16 < 0x7FFFFFF0 if half else n * 256 * h * w * 4 < 0x7FFFFFF0

    def fuses_expand_only(self, mid: torch.Tensor, res: torch.Tensor, conv3: Conv2d) -> bool:
        """fp32 plans: the LAST Bottleneck's expand conv + identity through the persistent weight-stationary kernel (no reduce conv
        follows).  ``MINDPOSE_FUSE_PWCHAIN32_LAST=0`` keeps the tuned general conv."""
        if isinstance(mid, ActC8) or isinstance(res, ActC8) or os.environ.get("MINDPOSE_FUSE_PWCHAIN32", "1") == "0":
            return False
        if os.environ.get("MINDPOSE_FUSE_PWCHAIN32_LAST", "1") == "0":
            return False
        n, cm, h, w = mid.shape
        return cm == 64 and tuple(res.shape) == (n, 256, h, w) and self._pw(conv3, 64, 256) and (h * w) % 64 == 0 and n * 256 * h * w * 4 < 0x7FFFFFF0

    def expand_reduce(self, mid: torch.Tensor, res: Optional[torch.Tensor], conv3: Conv2d, bn3: BatchNorm2d, conv1_next: Optional[Conv2d],
                      bn1_next: Optional[BatchNorm2d], ds=None):
        """(y, z) = (relu(bn3(conv3 mid) + res), relu(bn1'(conv1' y))) in one launch: y never comes back from HBM for the second conv
        (hrnet.py:107-146).  fp16: bit-identical to the two conv launches; fp32: equal up to the association of the k sums.
        fp32 only: ``ds = (x0, conv, bn)`` - the residual is the block's down-sample conv of its input x0, computed in the launch
        (``res`` None); ``conv1_next`` None - the expand conv alone, returns y."""
        n, cm, h, w = mid.shape
        ce = conv3.out_channels
        cr = conv1_next.out_channels if conv1_next is not None else 0
        macs = n * h * w * (cm * ce + ce * cr + (cm * ce if ds is not None else 0))
        if not isinstance(mid, ActC8):
            y = self.alloc_f32(n, ce, h, w)
            z = self.alloc_f32(n, cr, h, w) if conv1_next is not None else None
            p3 = self._pack(conv3.weight, ce, cm, 1, False, 0, 0)
            s3, b3 = self._affine(ce, bn3, None)
            p1 = s1 = b1 = pd = sd = bd = x0 = None
            if conv1_next is not None:
                p1 = self._pack(conv1_next.weight, cr, ce, 1, False, 0, 0)
                s1, b1 = self._affine(cr, bn1_next, None)
            if ds is not None:
                x0, ds_conv, ds_bn = ds
                pd = self._pack(ds_conv.weight, ce, cm, 1, False, 0, 0)
                sd, bd = self._affine(ce, ds_bn, None)
            _lib.check(self.lib.mp_plan_add_expand_reduce(self.handle, _lib.ptr(mid), _lib.ptr(res), _lib.ptr(x0), _lib.ptr(pd), _lib.ptr(sd),
                                                          _lib.ptr(bd), _lib.ptr(p3), _lib.ptr(s3), _lib.ptr(b3), _lib.ptr(p1), _lib.ptr(s1), _lib.ptr(b1),
                                                          _lib.ptr(y), _lib.ptr(z), n, cm, ce, cr if cr else 64, h, w),
                       "mp_plan_add_expand_reduce")
            self.layer_info.append(dict(kind="pwchain_f32", k=1, stride=1, cin=cm, cout=ce, h=h, w=w, n=n, macs=macs))
            return (y, z) if conv1_next is not None else y
        y, z = self.alloc(n, ce, h, w), self.alloc(n, cr, h, w)
        p3, p1 = self._pack(conv3.weight, ce, cm, 1, False, 0, 0, True), self._pack(conv1_next.weight, cr, ce, 1, False, 0, 0, True)
        (s3, b3), (s1, b1) = self._affine(ce, bn3, None, True), self._affine(cr, bn1_next, None, True)
        if ds is not None:
            x0, ds_conv, ds_bn = ds
            pd = self._pack(ds_conv.weight, ce, cm, 1, False, 0, 0, True)
            sd, bd = self._affine(ce, ds_bn, None, True)
            _lib.check(self.lib.mp_plan_add_ds_expand_reduce_f16(self.handle, _lib.ptr(mid), _lib.ptr(x0), _lib.ptr(pd), _lib.ptr(sd), _lib.ptr(bd),
                                                                 _lib.ptr(p3), _lib.ptr(s3), _lib.ptr(b3), 1, _lib.ptr(p1), _lib.ptr(s1), _lib.ptr(b1), 1,
                                                                 _lib.ptr(y), _lib.ptr(z), n, cm, ce, cr, h, w),
                       "mp_plan_add_ds_expand_reduce_f16")
            self.layer_info.append(dict(kind="pwchain_f16", k=1, stride=1, cin=cm, cout=ce, h=h, w=w, n=n, macs=macs))
            return y, z
        _lib.check(self.lib.mp_plan_add_expand_reduce_f16(self.handle, _lib.ptr(mid), _lib.ptr(res), _lib.ptr(p3), _lib.ptr(s3), _lib.ptr(b3), 1,
                                                          _lib.ptr(p1), _lib.ptr(s1), _lib.ptr(b1), 1, _lib.ptr(y), _lib.ptr(z), n, cm, ce, cr, h, w),
                   "mp_plan_add_expand_reduce_f16")
        self.layer_info.append(dict(kind="pwchain_f16", k=1, stride=1, cin=cm, cout=ce, h=h, w=w, n=n, macs=macs))
        return y, z

    def deconv4x4s2(self, x: torch.Tensor, deconv: Conv2dTranspose, bn: BatchNorm2d, relu: bool = True) -> torch.Tensor:
        """Conv2dTranspose(k=4, s=2, p=1) + BN + ReLU as four 2x2 sub-pixel phase convolutions - four launches with a tuned form each,
        or (fp32, where the tuner finds it faster: the small maps of the head) all four phases in ONE launch of the blocked-GEMM
        kernel (mp_plan_add_deconv4x4s2_gemm)."""
        half = isinstance(x, ActC8)
        n, cin, h, w = x.shape
        cout = deconv.out_channels
        out = self.alloc(n, cout, 2 * h, 2 * w) if half else self.alloc_f32(n, cout, 2 * h, 2 * w)
        scale, shift = self._affine(cout, bn, None, half)

        def desc(py, px):
            return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=2, kw=2, stride=1, pad_top=1 - py, pad_left=1 - px, conv_h=h,
                                 conv_w=w, out_h=2 * h, out_w=2 * w, out_mul=2, out_rep=1, out_off_y=py, out_off_x=px, relu=int(relu),
                                 flags=0)

        phases = [(py, px) for py in (0, 1) for px in (0, 1)]
        macs = n * h * w * cout * cin * 4
        if half:
            for py, px in phases:
                packed = self._pack(deconv.weight, cout, cin, 2, True, py, px, True)
                d = desc(py, px)
                v = tune_conv_variant(self.lib, d, x, packed, scale, shift, None, None, out, half=True)
                _lib.check(self.lib.mp_plan_add_conv_f16(self.handle, ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed),
                                                         _lib.ptr(scale), _lib.ptr(shift), None, None, _lib.ptr(out)),
                           "mp_plan_add_conv_f16(deconv phase)")
                self.layer_info.append(dict(kind="deconv_phase_f16", k=2, stride=1, cin=cin, cout=cout, h=h, w=w, n=n, macs=macs))
            return out
        # fp32: the four phase packings back to back in one buffer (the one-launch form reads them as slices)
        key = (id(deconv.weight), "deconv4")
        if key not in self._packed:
            wsrc = deconv.weight.detach().to(self.device, torch.float32).contiguous()
            per = self.lib.mp_conv_packed_weight_bytes(cout, cin, 2, 2) // 4
            buf = torch.empty(4 * per, device=self.device, dtype=torch.float32)
            for i, (py, px) in enumerate(phases):
                _lib.check(self.lib.mp_conv_pack_weight(_lib.ptr(wsrc), _lib.ptr(buf[i * per:(i + 1) * per]), cout, cin, 2, 2, 1, py, px,
                                                        _lib.stream()), "mp_conv_pack_weight(deconv phase)")
            self.keep += [wsrc, buf]
            self._packed[key] = buf
        buf = self._packed[key]
        per = buf.numel() // 4
        slices = [buf[i * per:(i + 1) * per] for i in range(4)]
        descs = [desc(py, px) for py, px in phases]
        variants = [tune_conv_variant(self.lib, d, x, pk, scale, shift, None, None, out) for d, pk in zip(descs, slices)]
        fused = False
        if os.environ.get("MINDPOSE_AUTOTUNE", "1") != "0" and self.lib.mp_deconv4x4s2_gemm_supported(ctypes.byref(descs[0])) == 0:
            stream = _lib.stream()

            def launch(form):
                if form == 1:
                    return self.lib.mp_deconv4x4s2_gemm_fwd(ctypes.byref(descs[0]), _lib.ptr(x), _lib.ptr(buf), _lib.ptr(scale),
                                                            _lib.ptr(shift), _lib.ptr(out), stream)
                rc = 0
                for d, pk, v in zip(descs, slices, variants):
                    rc = rc or self.lib.mp_conv2d_fwd_variant(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(pk), _lib.ptr(scale),
                                                              _lib.ptr(shift), None, None, _lib.ptr(out), stream)
                return rc

            tkey = ("deconv4x4s2",) + tuple(getattr(descs[0], f) for f, _ in descs[0]._fields_) + (str(out.device),)
            fused = _autotune(tkey, 4 * macs, 2, launch) == 1
        if fused:
            _lib.check(self.lib.mp_plan_add_deconv4x4s2_gemm(self.handle, ctypes.byref(descs[0]), _lib.ptr(x), _lib.ptr(buf),
                                                             _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(out)),
                       "mp_plan_add_deconv4x4s2_gemm")
            self.layer_info.append(dict(kind="deconv_gemm", k=2, stride=1, cin=cin, cout=cout, h=h, w=w, n=n, macs=4 * macs))
            return out
        for d, pk, v in zip(descs, slices, variants):
            _lib.check(self.lib.mp_plan_add_conv_variant(self.handle, ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(pk),
                                                         _lib.ptr(scale), _lib.ptr(shift), None, None, _lib.ptr(out)),
                       "mp_plan_add_conv_variant(deconv phase)")
            self.layer_info.append(dict(kind="deconv_phase", k=2, stride=1, cin=cin, cout=cout, h=h, w=w, n=n, macs=macs))
        return out

    def fuse_sum(self, base: torch.Tensor, terms, out: torch.Tensor, relu: bool = True) -> torch.Tensor:
        """out = act(((base + up(t1)) + up(t2)) + up(t3)); ``terms`` = [(low-res tensor, integer scale), ...] (1-3)."""
        n, c, h, w = base.shape
        if not 1 <= len(terms) <= 3:
            raise ValueError("fuse_sum takes 1 to 3 up-sampled terms")
        args = []
        for t, sc in list(terms) + [(None, 1)] * (3 - len(terms)):
            if t is not None and tuple(t.shape) != (n, c, h // sc, w // sc):
                raise ValueError(f"fuse term shape {tuple(t.shape)} does not match {(n, c, h // sc, w // sc)}")
            args += [_lib.ptr(t), int(sc)]
        add = self.lib.mp_plan_add_fuse_sum_f16 if isinstance(base, ActC8) else self.lib.mp_plan_add_fuse_sum
        _lib.check(add(self.handle, _lib.ptr(base), *args, _lib.ptr(out), n, c, h, w, int(relu)), "mp_plan_add_fuse_sum")
        self.layer_info.append(dict(kind="fuse_sum", n=n, c=c, h=h, w=w, terms=len(terms), macs=0))
        return out

    def maxpool3x3s2_same(self, x: torch.Tensor) -> torch.Tensor:
        if isinstance(x, ActC8):
            raise NotImplementedError("max-pool runs on fp32 NCHW activations (the ResNet stem stays fp32 under amp O2)")
        n, c, h, w = x.shape
        out = self.alloc_f32(n, c, (h + 1) // 2, (w + 1) // 2)
        _lib.check(self.lib.mp_plan_add_maxpool(self.handle, _lib.ptr(x), _lib.ptr(out), n, c, h, w), "mp_plan_add_maxpool")
        self.layer_info.append(dict(kind="maxpool", n=n, c=c, h=h, w=w, macs=0))
        return out


class PlannedModule(nn.Module):
    """Base of every module whose forward is a recorded HIP launch plan.

    Sub-classes implement ``emit(plan, x) -> out`` (record launches, return the output buffer).
    ``forward`` copies the batch into the plan's static input buffer (skipped when the caller already
    wrote into ``input_buffer(shape)``), replays the plan and returns the plan's output buffer
    (a view that the next call overwrites - clone it if it must outlive the next forward).
    """

    def __init__(self) -> None:
        super().__init__()
        self._plans: Dict[Tuple, Plan] = {}
        self.training = False  # like mindspore.nn.Cell: inference mode until .train() / set_train(True)
        self.amp_level = "O0"  # see auto_mixed_precision()

    def set_train(self, mode: bool = True):
        """mindspore.nn.Cell.set_train alias."""
        return self.train(mode)

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        """Training-mode forward (batch-statistics BatchNorm, autograd through the HIP backward kernels)."""
        raise NotImplementedError("this module has no training path yet")

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("Child class must implement this method.")

    def invalidate_plans(self) -> None:
        """Call after parameters change (checkpoint load, optimizer step): packed weights are re-built."""
        for m in self.modules():
            if isinstance(m, PlannedModule):
                m._plans.clear()

    def _load_from_state_dict(self, *args, **kwargs):
        self._plans.clear()
        return super()._load_from_state_dict(*args, **kwargs)

    def get_plan(self, shape, device) -> Plan:
        half = self.amp_level in ("O2", "O3")
        key = (tuple(shape), str(device), half)
        plan = self._plans.get(key)
        if plan is None:
            if device.type != "cuda":
                raise _lib.MindposeHipError(
                    "mindpose_amd networks run on the MI355X HIP path only (no CPU fallback): move the module and "
                    "its inputs to a CUDA device")
            with torch.no_grad():
                plan = Plan(device, half=half)
                plan.input = plan.alloc_f32(*shape)
                out = self.emit(plan, plan.input)  # backbones switch to the fp16 layout themselves (plan.enter)
                plan.output = plan.from_c8(out) if isinstance(out, ActC8) else out
            self._plans[key] = plan
        return plan

    def input_buffer(self, shape, device=None) -> torch.Tensor:
        device = device or next(self.parameters()).device
        return self.get_plan(shape, torch.device(device)).input

    def forward(self, x: torch.Tensor, flip_width: bool = False) -> torch.Tensor:
        """``flip_width``: run the network on the horizontally mirrored batch (the flip test's second run,
        topdown_inferencer.py:168-170): the mirror is written straight into the plan's input buffer by ``mp_flip_width``."""
        x = _lib.require_cuda_f32(x, "input")
        if self.training:
            if flip_width:
                raise ValueError("flip_width is an inference-time option")
            self._plans.clear()  # parameters are about to change: packed weights of recorded plans go stale
            return self.train_forward(x)
        plan = self.get_plan(x.shape, x.device)
        if flip_width:
            if x.data_ptr() == plan.input.data_ptr():
                x = x.clone()  # the caller wrote the batch into the plan's own input buffer: mirror from a copy
            n, c, h, w = x.shape
            _lib.check(plan.lib.mp_flip_width(_lib.ptr(x), _lib.ptr(plan.input), n, c, h, w, _lib.stream()), "mp_flip_width")
        elif x.data_ptr() != plan.input.data_ptr():
            plan.input.copy_(x)
        plan.run()
        return plan.output

    def forward_flip_pair(self, x: torch.Tensor) -> torch.Tensor:
        """The flip test's two runs (topdown_inferencer.py:168-170: ``net(img)``, ``net(flip_W(img))``) as ONE forward of the batch
        [x | mirror(x)]: ``mp_flip_width`` writes the mirrored crops straight into the second half of a 2N plan's input buffer.
        Returns the [2N, K, h, w] output (first N = the crops, last N = their mirrors).  Every sample's arithmetic is what the
        N-crop plan does (inference BatchNorm is per sample); the persistent kernels amortise their weight prologue over twice
        the tiles and the step has half the launches."""
        x = _lib.require_cuda_f32(x, "input")
        if self.training:
            raise ValueError("the flip test is an inference-time path")
        n, c, h, w = x.shape
        plan = self.get_plan((2 * n, c, h, w), x.device)
        first, second = plan.input[:n], plan.input[n:]
        if x.data_ptr() != first.data_ptr():
            first.copy_(x)
        _lib.check(plan.lib.mp_flip_width(_lib.ptr(first), _lib.ptr(second), n, c, h, w, _lib.stream()), "mp_flip_width")
        plan.run()
        return plan.output


def flip_pair_batched() -> bool:
    """``MINDPOSE_FLIP_BATCHED=0``: the flip test runs two forwards of N crops through one plan (rounds 1 - 3) instead of ONE
    forward of the 2N-crop batch [crops | mirrored crops]."""
    return os.environ.get("MINDPOSE_FLIP_BATCHED", "1") != "0"


def auto_mixed_precision(network: nn.Module, amp_level: str = "O0") -> nn.Module:
    """``mindspore.amp.auto_mixed_precision(network, amp_level)`` for the planned networks (what
    ``mindspore.Model(amp_level=...)`` applies in the reference's tools/train.py:176-181).

    O0: fp32 everywhere (fp32 MFMA kernels).  O2 / O3: inference runs the fp16 matrix-core kernels - fp16 conv operands
    and activations, fp32 accumulation, BatchNorm folded in fp32 - and hands fp32 heat maps to the decoder.  (O1 is not
    offered: its white-list casts are a graph rewrite of the MindSpore cells with no counterpart here.)"""
    if amp_level not in ("O0", "O2", "O3"):
        raise ValueError(f"amp_level must be one of O0, O2, O3, got {amp_level!r}")
    for m in network.modules():
        if isinstance(m, PlannedModule):
            m.amp_level = amp_level
            m._plans.clear()
    return network
