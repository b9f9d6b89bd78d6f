"""ctypes binding of libxdemhip.so (the C-ABI declared in include/xdemhip.h).

There is deliberately NO fallback: if the HIP library has not been built, or no GPU context can be
created, the package raises -- a silent CPU path would void every parity claim.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
import weakref

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libxdemhip.so")

OK = 0
F32, F64 = 0, 1
HOST, DEVICE = 0, 1

_lib = None
_lock = threading.Lock()


# Conventions of the two un-vendored third-party engines that nothing readable offline pins (DESIGN.md sections 2-3): built-in
# defaults, overridden by xdem_amd/thirdparty_decision.json where oracle/pin_thirdparty.py could decide them from the packages' own
# outputs.  What the best available evidence favours for each is stated in DESIGN.md; none of it is a measurement made here.
THIRDPARTY_DEFAULTS = {"nk_nan_rule": 0, "vario_edge": 0, "vario_diff": 0}
_THIRDPARTY_RANGE = {"nk_nan_rule": (0, 3), "vario_edge": (0, 1), "vario_diff": (0, 1)}


def thirdparty_decision() -> dict:
    """The decided conventions (subset of THIRDPARTY_DEFAULTS' keys) from thirdparty_decision.json next to this file; {} if absent."""
    import json

    path = os.environ.get("XDEM_THIRDPARTY_DECISION") or os.path.join(HERE, "thirdparty_decision.json")   # (the variable: tests)
    if not os.path.exists(path):
        return {}
    d = json.load(open(path))
    out = {}
    for k, (lo, hi) in _THIRDPARTY_RANGE.items():
        if k in d:
            v = int(d[k])
            if not lo <= v <= hi:
                raise ValueError(f"{path}: {k} = {v} outside {lo}..{hi}")
            out[k] = v
    return out


class XdemHipError(RuntimeError):
    """Raised for any non-zero status from libxdemhip.so."""


def host_library(required: bool = True):
    """The library for its HOST-side entry points (csrc/hostprep.hip: no context, no GPU needed); None if it is missing and not required."""
    try:
        return lib()
    except XdemHipError:
        if required:
            raise
        return None


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library with argtypes declared."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise XdemHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). xdem_amd has no CPU fallback."
            )
        # torch bundles its own libamdhip64: import it first so that both share one HIP runtime in-process
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for host-buffer use
            pass
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        c_ctx = ctypes.c_void_p
        L.xdemhip_version.restype = ctypes.c_int
        L.xdemhip_create.argtypes = [ctypes.c_int, ctypes.POINTER(c_ctx)]
        L.xdemhip_destroy.argtypes = [c_ctx]
        L.xdemhip_destroy.restype = None
        L.xdemhip_last_error.argtypes = [c_ctx]
        L.xdemhip_last_error.restype = ctypes.c_char_p
        L.xdemhip_set_stream.argtypes = [c_ctx, ctypes.c_void_p]
        L.xdemhip_synchronize.argtypes = [c_ctx]
        L.xdemhip_device_alloc.argtypes = [c_ctx, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
        L.xdemhip_device_free.argtypes = [c_ctx, ctypes.c_void_p]
        L.xdemhip_set_option.argtypes = [c_ctx, ctypes.c_char_p, ctypes.c_int]
        L.xdemhip_set_test_switch.argtypes = [c_ctx, ctypes.c_char_p, ctypes.c_int]
        L.xdemhip_set_allreduce.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_void_p]
        L.xdemhip_set_allreduce_device.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_void_p]
        L.xdemhip_reduction_calls.argtypes = [c_ctx, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.xdemhip_set_rank.argtypes = [c_ctx, ctypes.c_int, ctypes.c_int]
        L.xdemhip_last_kernel_ms.argtypes = [c_ctx, ctypes.POINTER(ctypes.c_float)]
        L.xdemhip_clock_probe.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.xdemhip_terrain.argtypes = [
            c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
        ]
        c_i64p, c_dp = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)
        L.xdemhip_fractal_constants.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), c_dp, c_dp, c_dp]
        L.xdemhip_texture_shading.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.xdemhip_nk_create.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), c_i64p]
        L.xdemhip_nk_create_block.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), c_i64p]
        L.xdemhip_nk_step.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_int, c_dp, c_i64p, c_dp, c_dp, c_dp, c_i64p, c_dp]
        L.xdemhip_nk_step_fit.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                          c_dp, c_i64p, c_dp, c_dp, c_dp]
        L.xdemhip_nk_set_bin_edges.argtypes = [ctypes.c_void_p, c_dp, ctypes.c_int, ctypes.c_int]
        L.xdemhip_nk_get_aux.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.xdemhip_nk_destroy.argtypes = [ctypes.c_void_p]
        L.xdemhip_nk_destroy.restype = None
        L.xdemhip_binned_median.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                            c_dp, c_i64p, c_dp]
        L.xdemhip_mean_filter_nan.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.xdemhip_binstats_bin_numbers.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.xdemhip_convolution.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_dp, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.xdemhip_perbin_lookup.argtypes = [c_ctx, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int64,
                                            ctypes.POINTER(ctypes.c_int), c_dp, c_dp, c_dp, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p,
                                            c_i64p, ctypes.c_int]
        L.xdemhip_shift_bilinear.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
        L.xdemhip_set_allreduce.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_void_p]
        L.xdemhip_nk_set_rows.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, c_i64p]
        L.xdemhip_nk_set_statistic.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.xdemhip_nk_route_counts.argtypes = [ctypes.c_void_p, c_i64p, c_i64p]
        L.xdemhip_host_ring_sample.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_int64, c_i64p, c_i64p,
                                               ctypes.c_int, c_dp, c_dp, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int, c_i64p, c_i64p]
        L.xdemhip_host_gather_points.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_int, c_i64p, c_i64p,
                                                 ctypes.c_int, c_dp, c_dp, ctypes.c_void_p, c_dp, c_dp, ctypes.c_void_p]
        L.xdemhip_nk_step_values.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                             c_dp, c_i64p, c_dp, c_dp, c_dp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.xdemhip_nk_predict_counts.argtypes = [ctypes.c_void_p, c_i64p, c_i64p, c_i64p]
        L.xdemhip_nk_subsample.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, c_i64p]
        L.xdemhip_host_count_finite.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_i64p, ctypes.c_void_p]
        c_u64p = ctypes.POINTER(ctypes.c_uint64)
        L.xdemhip_pairs_create.argtypes = [c_ctx, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), c_i64p]
        L.xdemhip_pairs_sums.argtypes = [ctypes.c_void_p, ctypes.c_int, c_dp, c_i64p]
        L.xdemhip_pairs_hist.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_u64p, c_u64p]
        L.xdemhip_pairs_succ.argtypes = [ctypes.c_void_p, c_u64p, c_u64p]
        L.xdemhip_pairs_medians.argtypes = [ctypes.c_void_p, c_i64p, c_dp]
        L.xdemhip_pairs_link_sorted.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.xdemhip_pairs_link_shadow.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.xdemhip_pairs_takes_brackets.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.xdemhip_pairs_destroy.argtypes = [ctypes.c_void_p]
        L.xdemhip_pairs_destroy.restype = None
        c_ip = ctypes.POINTER(ctypes.c_int)
        L.xdemhip_binstats_create.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_void_p)]
        L.xdemhip_binstats_add_var.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.xdemhip_binstats_finalize.argtypes = [ctypes.c_void_p, c_i64p, c_dp, c_dp]
        L.xdemhip_binstats_run.argtypes = [ctypes.c_void_p, ctypes.c_int, c_ip, c_dp, c_ip, c_ip, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, c_i64p, c_dp, c_dp]
        L.xdemhip_binstats_destroy.argtypes = [ctypes.c_void_p]
        L.xdemhip_binstats_destroy.restype = None
        L.xdemhip_cov_double_sum.argtypes = [c_ctx, c_dp, c_dp, c_dp, ctypes.c_int64, c_dp, c_dp, c_dp, ctypes.c_int64, ctypes.c_int, c_ip,
                                             c_dp, c_dp, c_dp, c_dp, ctypes.c_int]
        L.xdemhip_nmad.argtypes = [c_ctx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_int, c_dp, c_dp, c_i64p]
        L.xdemhip_interp_grid_linear.argtypes = [c_ctx, ctypes.c_int, c_dp, c_ip, c_dp, ctypes.POINTER(ctypes.c_void_p), c_ip,
                                                 ctypes.c_int64, ctypes.c_double, c_dp, ctypes.c_int]
        _lib = L
        return L


def make_reduce_hook(group="world", device: int | None = None):
    """The Python side of ``xdemhip_set_allreduce``: a callable ``hook(ptr, count, kind, user) -> 0 | 1`` that combines an
    8-byte-element host array in place over the ranks of `group` with torch.distributed.  kind 0 = uint64 sum, 1 = float64
    sum, 2 / 3 = uint64 min / max (include/xdemhip.h XDEMHIP_RED_*)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    pg = None if group == "world" else group
    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.SUM, 2: dist.ReduceOp.MIN, 3: dist.ReduceOp.MAX}

    def hook(ptr, count, kind, user):
        try:
            buf = (ctypes.c_uint64 * count).from_address(ptr)
            a = np.frombuffer(buf, dtype=np.float64 if kind == 1 else np.uint64)
            top = np.uint64(1) << np.uint64(63)
            if kind == 1:
                t = torch.from_numpy(a.copy())
            elif kind == 0:
                t = torch.from_numpy(a.copy().view(np.int64))  # counters: two's-complement addition is exact
            else:
                # min / max of unsigned 64-bit keys travel as int64 with the top bit flipped: an order-preserving map
                # (the all-ones "none" marker of min reductions becomes int64 max by itself)
                t = torch.from_numpy((a ^ top).view(np.int64))
            if dist.get_backend(pg) == "nccl":
                t = t.cuda(device)
            dist.all_reduce(t, op=ops[kind], group=pg)
            r = t.cpu().numpy()
            if kind == 1:
                a[:] = r
            elif kind == 0:
                a[:] = r.view(np.uint64)
            else:
                a[:] = r.view(np.uint64) ^ top
            return 0
        except Exception:  # never propagate a Python exception through the C frame
            import traceback

            traceback.print_exc()
            return 1

    return hook


class _DeviceArray:
    """A device pointer dressed up for ``torch.as_tensor`` (CUDA array interface, zero copy)."""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class _OwnedDeviceArray(_DeviceArray):
    """The same, owning its memory (``xdemhip_device_alloc``): torch keeps the object alive as long as a tensor built on it."""

    def __init__(self, ctx, ptr: int, count: int, typestr: str, nbytes: int = 0, flags: int = 0):
        super().__init__(ptr, count, typestr)
        self._ctx, self._ptr, self._nbytes, self._flags = ctx, ptr, nbytes, flags

    def __del__(self):  # pragma: no cover
        try:
            if self._ptr and getattr(self._ctx, "handle", None):
                self._ctx._give_back(self._ptr, self._nbytes, self._flags)
        except Exception:
            pass
        self._ptr = 0


def make_device_reduce_hook(group="world", device: int | None = None):
    """The Python side of ``xdemhip_set_allreduce_device``: ``hook(device_ptr, count, kind, hip_stream, user) -> 0 | 1`` wraps the
    library's device array as a torch tensor (no copy) and ENQUEUES ``torch.distributed.all_reduce`` (RCCL) with the
    library's stream as the current stream: the collective waits for the work already queued there and the stream waits for
    the collective -- no host synchronisation, no staging.  min / max of unsigned 64-bit keys travel as int64 with the top
    bit flipped (an order-preserving map), flipped on the device before and after."""
    import torch
    import torch.distributed as dist

    pg = None if group == "world" else group
    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.SUM, 2: dist.ReduceOp.MIN, 3: dist.ReduceOp.MAX}
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    top = -(2**63)

    # (a step of a partitioned Nuth-Kaab plan makes ten of these calls on the same few library buffers: the tensor views and the stream
    #  wrapper are kept -- building them costs more host time than enqueueing the collective; a view holds an address, not the memory)
    views, streams = {}, {}

    def hook(ptr, count, kind, stream, user):
        try:
            key = (ptr, count, kind == 1)
            t = views.get(key)
            if t is None:
                if len(views) > 256:
                    views.clear()
                t = views[key] = torch.as_tensor(_DeviceArray(ptr, count, "<f8" if kind == 1 else "<i8"), device=dev)
            s = streams.get(stream)
            if s is None:
                s = streams[stream] = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.stream(s):
                if kind in (2, 3):
                    t.bitwise_xor_(top)
                dist.all_reduce(t, op=ops[kind], group=pg)
                if kind in (2, 3):
                    t.bitwise_xor_(top)
            return 0
        except Exception:  # never propagate a Python exception through the C frame
            import traceback

            traceback.print_exc()
            return 1

    return hook


class Context:
    """One libxdemhip context = one GPU of this process."""

    def __init__(self, device: int = 0) -> None:
        self._L = lib()
        h = ctypes.c_void_p()
        rc = self._L.xdemhip_create(int(device), ctypes.byref(h))
        if rc != OK or not h:
            raise XdemHipError(
                f"xdemhip_create(device={device}) failed with status {rc}: no usable MI355X/HIP device "
                "(xdem_amd has no CPU fallback)"
            )
        self.handle = h
        self.device = int(device)
        self._group = None   # process group of the installed reduction hooks (set_allreduce)
        # Released scattered plane ranges are kept for the next request of the same size (building a range is ~30 us per 8 MiB
        # piece: 0.4 s for the 11 planes of a 16384^2 raster, every call, if nothing were kept) -- at most this many bytes in
        # all, least recently released first out; larger ranges (the 70 GB of the 40000^2 set) go back to the driver at once.
        self._pool: list[tuple[int, int, int]] = []   # (nbytes, flags, ptr)
        self._pool_cap = int(float(os.environ.get("XDEM_PLANE_POOL_GB", "16")) * (1 << 30))
        self._pool_lock = threading.Lock()
        # Options belong to the context, not to a call: engine="numba", mp_config and the float32 shadow of a pair set change one for
        # the duration of their launches and put it back.  Those sequences -- and every terrain launch -- hold this lock, so two
        # threads sharing a context (the process-wide default one, typically) cannot see each other's setting; a context still
        # does one thing at a time (its one stream, its scratch state: include/xdemhip.h).
        self.call_lock = threading.RLock()
        self.options: dict[str, int] = {}  # mirror of the xdemhip_set_option calls made through this object
        # objects that hold library handles created on this context (Nuth-Kaab plans, pair sets, binning plans): closed before the
        # context goes -- a plan destroyed AFTER its context would hand the library a dangling context pointer
        self._dependants: weakref.WeakSet = weakref.WeakSet()
        for k, v in thirdparty_decision().items():   # conventions decided from the third-party packages' own outputs, where recorded
            if v != THIRDPARTY_DEFAULTS[k]:
                self.set_option(k, v)

    def check(self, rc: int) -> None:
        if rc != OK:
            msg = self._L.xdemhip_last_error(self.handle)
            raise XdemHipError(f"libxdemhip status {rc}: {msg.decode() if msg else ''}")

    def set_stream(self, stream_ptr: int | None) -> None:
        """Stream for device-resident calls: a hipStream_t value (0 = HIP's default stream, which is torch's default stream);
        None = the context's private stream."""
        v = ctypes.c_void_p(-1) if stream_ptr is None else ctypes.c_void_p(stream_ptr)
        self.check(self._L.xdemhip_set_stream(self.handle, v))

    def set_allreduce(self, group="world", device_side: bool | None = None) -> None:
        """Install (group given) or remove (group=None) the multi-GPU reduction hooks.  The host hook combines small
        8-byte-element host arrays over the ranks of `group` with torch.distributed; with an RCCL ("nccl") group the
        device-side hook is installed next to it (``device_side`` forces it on / off): the library's per-pass reductions --
        histograms, counters, keys, all in device memory -- are then all-reduced in place on the library's stream with no
        staging and no host synchronisation; gloo groups (CPU tests) stage everything through the host hook."""
        self._group = group   # (remembered so that a caller which installs hooks temporarily can put these back)
        if group is None:
            self._hook = None
            self._dev_hook = None
            self.check(self._L.xdemhip_set_allreduce_device(self.handle, None, None))
            self.check(self._L.xdemhip_set_allreduce(self.handle, None, None))
            self.check(self._L.xdemhip_set_rank(self.handle, 0, 0))
            return
        import torch.distributed as dist

        pg = None if group == "world" else group
        # this process's place in the group: some exchanges carry one slot per rank (xdemhip_set_rank, include/xdemhip.h)
        self.check(self._L.xdemhip_set_rank(self.handle, dist.get_rank(pg), dist.get_world_size(pg)))
        hook = make_reduce_hook(group, self.device)
        CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p)
        self._hook = CB(hook)  # keep alive
        self.check(self._L.xdemhip_set_allreduce(self.handle, ctypes.cast(self._hook, ctypes.c_void_p), None))
        if device_side is None:
            import torch.distributed as dist

            device_side = dist.get_backend(None if group == "world" else group) == "nccl"
        if device_side:
            CBD = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
            self._dev_hook = CBD(make_device_reduce_hook(group, self.device))
            self.check(self._L.xdemhip_set_allreduce_device(self.handle, ctypes.cast(self._dev_hook, ctypes.c_void_p), None))
        else:
            self._dev_hook = None
            self.check(self._L.xdemhip_set_allreduce_device(self.handle, None, None))

    def reduction_calls(self) -> tuple[int, int]:
        """(reductions staged through the host hook, reductions enqueued through the device hook) since the context was created."""
        h, d = ctypes.c_int64(), ctypes.c_int64()
        self.check(self._L.xdemhip_reduction_calls(self.handle, ctypes.byref(h), ctypes.byref(d)))
        return int(h.value), int(d.value)

    def synchronize(self) -> None:
        self.check(self._L.xdemhip_synchronize(self.handle))

    def device_tensor(self, shape, dtype="float32", contiguous: bool = True, recycled: bool = False, chunked: bool = False,
                      scattered: bool = False):
        """A torch tensor over device memory from ``xdemhip_device_alloc`` -- physically contiguous when the driver can provide it
        (``tensor.xdem_contiguous`` tells): the layout the streaming terrain kernel wants for its output planes (include/xdemhip.h).
        The memory is released when the tensor (and every view of it) is gone."""
        import numpy as np
        import torch

        np_dt = np.dtype(dtype)
        count = int(np.prod(shape))
        ptr, got = ctypes.c_void_p(), ctypes.c_int()
        # XDEMHIP_ALLOC_SCATTERED / _CHUNKED / _CONTIGUOUS | _RECYCLED
        flags = 8 if scattered else (4 if chunked else ((1 if contiguous else 0) | (2 if recycled else 0)))
        nbytes = count * np_dt.itemsize
        pooled = self._take_pooled(nbytes, flags)
        if pooled:
            # The previous owner's tensor is gone, but work it queued on ANY stream (the overlap path's side streams, a caller's
            # own streams, RCCL sends of its rows) may still be reading or writing the range: freeing it would have waited for the
            # device (device_free_chunked synchronises before it unmaps), so reusing it waits the same way -- once per reuse,
            # against ~30 us per 8 MiB piece for building a fresh range.  Callers that must never block pass `out=`.
            torch.cuda.synchronize(self.device)
            ptr.value, got.value = pooled, 0
        else:
            rc = self._L.xdemhip_device_alloc(self.handle, nbytes, flags, ctypes.byref(ptr), ctypes.byref(got))
            if rc != OK and self._pool:   # the pool may hold what this request needs: empty it and try once more
                self.release_pool()
                rc = self._L.xdemhip_device_alloc(self.handle, nbytes, flags, ctypes.byref(ptr), ctypes.byref(got))
            self.check(rc)
        owner = _OwnedDeviceArray(self, int(ptr.value), count, np_dt.str, nbytes, flags)
        t = torch.as_tensor(owner, device=torch.device("cuda", self.device)).view(*shape)
        t.xdem_contiguous = bool(got.value)
        return t

    def _take_pooled(self, nbytes: int, flags: int) -> int:
        with self._pool_lock:
            for i, (b, f, p) in enumerate(self._pool):
                if b == nbytes and f == flags:
                    del self._pool[i]
                    return p
        return 0

    def _give_back(self, ptr: int, nbytes: int, flags: int) -> None:
        """A device range nobody references any more: scattered ranges of moderate size wait in the pool, the rest is freed."""
        if flags == 8 and 0 < nbytes <= self._pool_cap:
            evict = []
            with self._pool_lock:
                self._pool.append((nbytes, flags, ptr))
                while sum(b for b, _, _ in self._pool) > self._pool_cap:
                    evict.append(self._pool.pop(0))
            for _, _, p in evict:
                self._L.xdemhip_device_free(self.handle, ctypes.c_void_p(p))
            return
        self._L.xdemhip_device_free(self.handle, ctypes.c_void_p(ptr))

    def release_pool(self) -> None:
        """Hand the pooled plane ranges back to the driver (the analogue of ``torch.cuda.empty_cache()``, and to be called next
        to it: torch's allocator cannot see or reclaim this memory -- up to XDEM_PLANE_POOL_GB, default 16 GiB per context)."""
        with self._pool_lock:
            items, self._pool = self._pool, []
        for _, _, p in items:
            self._L.xdemhip_device_free(self.handle, ctypes.c_void_p(p))

    TEST_SWITCHES = frozenset(("terrain_stream", "terrain_order", "terrain_ring_wait", "terrain_window_lds", "nk_narrow",
                               "vario_grid", "vario_runs", "vario_sort", "terrain_store", "terrain_rows", "terrain_sync", "vario_deff"))

    def set_option(self, name: str, value: int) -> None:
        """Option of the library (``xdemhip_set_option``: the fourteen names of include/xdemhip.h), e.g. ``("selection", 1)`` -- or,
        for the names of include/xdemhip_test.h, a test switch between internal routes (``xdemhip_set_test_switch``)."""
        fn = self._L.xdemhip_set_test_switch if name in self.TEST_SWITCHES else self._L.xdemhip_set_option
        self.check(fn(self.handle, name.encode(), int(value)))
        self.options[name] = int(value)

    @contextlib.contextmanager
    def option_scope(self, name: str, value: int):
        """``with ctx.option_scope("terrain_nonfinite", 1): ...`` -- the option set for the body and restored after it, the whole
        sequence under the context's call lock (see ``call_lock``)."""
        with self.call_lock:
            prev = self.options.get(name, 0)
            if int(value) != prev:
                self.set_option(name, value)
            try:
                yield
            finally:
                if int(value) != prev:
                    self.set_option(name, prev)

    def clock_probe(self, stream, out, sleeps: int = 2000) -> None:
        """Enqueue the shader-clock probe (``xdemhip_clock_probe``) on `stream` (a ``torch.cuda.Stream`` other than the one the
        context launches on); `out` = a device tensor of two int64 / uint64 words: shader-clock ticks, 100 MHz ticks."""
        self.check(self._L.xdemhip_clock_probe(self.handle, ctypes.c_void_p(int(stream.cuda_stream)), int(sleeps), ctypes.c_void_p(int(out.data_ptr()))))

    def last_kernel_ms(self) -> float:
        ms = ctypes.c_float()
        self.check(self._L.xdemhip_last_kernel_ms(self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def adopt(self, obj) -> None:
        """Register an object with a ``close()`` that owns handles created on this context (closed with it, if still alive)."""
        self._dependants.add(obj)

    def close(self) -> None:
        if getattr(self, "handle", None):
            for obj in list(getattr(self, "_dependants", ())):
                try:
                    obj.close()
                except Exception:
                    pass
            self.release_pool()
            self._L.xdemhip_destroy(self.handle)
            self.handle = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_default_ctx: dict[int, Context] = {}
_ctx_lock = threading.Lock()
_tls = threading.local()


class use_context:
    """``with use_context(ctx): ...`` -- calls made by THIS thread inside the block that would take the process-wide default
    context use `ctx` instead.  A context serialises its own work (one stream, one scratch state, one queue of deferred
    result copies) and MUST NOT be driven by two threads at once -- that includes the process-wide default context: threads
    that want to overlap GPU work create one Context each and wrap their calls in ``use_context`` (the library is re-entrant
    per context handle; include/xdemhip.h states the same contract)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def __enter__(self):
        self.prev = getattr(_tls, "ctx", None)
        _tls.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _tls.ctx = self.prev
        return False


def default_context(device: int | None = None) -> Context:
    """Process-wide context for `device` (default: $XDEM_AMD_DEVICE, else LOCAL_RANK, else 0), or the calling thread's
    ``use_context`` override."""
    cur = getattr(_tls, "ctx", None)
    if cur is not None and (device is None or device == cur.device):
        return cur
    if device is None:
        device = int(os.environ.get("XDEM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _default_ctx:
        with _ctx_lock:  # (two threads asking for the first time must not create two contexts; not `_lock`: Context() -> lib() takes that one)
            if device not in _default_ctx:
                _default_ctx[device] = Context(device)
    return _default_ctx[device]
