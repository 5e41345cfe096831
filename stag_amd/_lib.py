"""ctypes binding of libstag_hip.so (include/stag_hip.h) for torch tensors.

The library is the product; there is no CPU or eager-PyTorch fallback behind it:
a missing library, a tensor that is not on a HIP device, or a non-zero return code
raises.  torch is imported first on purpose — the library needs the very HIP
runtime instance torch has loaded (same soname), otherwise torch's streams and
device pointers would mean nothing to it.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# STAG_HIP_SO: A/B tooling only (tools/ab_bench.py, tools/profile_bench.py --lib: a build variant of the same sources)
_SO = os.environ.get("STAG_HIP_SO") or os.path.join(_HERE, "libstag_hip.so")

NOISE_NONE, NOISE_EXPLICIT, NOISE_NORMAL, NOISE_UNIFORM, NOISE_BERNOULLI = range(5)
PARAM_SCALAR, PARAM_PER_CHANNEL, PARAM_PER_EDGE1, PARAM_PER_EDGE = range(4)
REDUCE_SUM, REDUCE_MEAN = 0, 1
GAT_BWD_ROWDOT, GAT_BWD_SOURCE, GAT_BWD_DER = 1, 2, 4   # STAG_GAT_BWD_* (stag_gat_bwd_stages)
HEAVY_LEN = 16   # STAG_HEAVY_LEN (include/stag_hip.h)
XCD_HEADER, XCD_STRIPES, XCD_FINE_MAX = 32, 8, 16   # STAG_XCD_HEADER, STAG_XCD_STRIPES, STAG_XCD_FINE_MAX
# STAG_BLOCK_EDGES / STAG_BLOCK_UNITS of include/stag_hip.h (the environment override pairs with a build variant of the
# library compiled with the same -D values: A/B tooling only)
BLOCK_EDGES, BLOCK_UNITS = int(os.environ.get("STAG_BLOCK_EDGES", "256")), int(os.environ.get("STAG_BLOCK_UNITS", "32"))

_vp = C.c_void_p


class Csr(C.Structure):
    _fields_ = [("n_dst", C.c_int32), ("n_src", C.c_int32), ("n_edges", C.c_int64),
                ("indptr", _vp), ("indices", _vp), ("eid", _vp), ("nidx", _vp)]


class NoiseSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("param_mode", C.c_int32), ("p0", _vp), ("p1", _vp),
                ("p0_scalar", C.c_float), ("p1_scalar", C.c_float),
                ("relu", C.c_int32), ("in_norm", C.c_int32), ("deriv", C.c_int32),
                ("group", C.c_int32), ("seed", C.c_uint64), ("offset", C.c_uint64), ("pos_base", C.c_int64),
                ("chunk_base", C.c_int32), ("p1_log", C.c_int32), ("epoch", C.c_void_p)]


class GatDrop(C.Structure):       # stag_gat_drop
    _fields_ = [("keep_prob", C.c_float), ("seed", C.c_uint64), ("offset", C.c_uint64), ("epoch", C.c_void_p)]


class Plan(C.Structure):
    _fields_ = [("seg_len", C.c_int32), ("n_units", C.c_int32), ("n_long", C.c_int32),
                ("n_seg", C.c_int32), ("units", _vp), ("long_rows", _vp), ("long_seg_ptr", _vp),
                ("seg_counters", _vp), ("workspace", _vp), ("workspace_bytes", C.c_size_t),
                ("n_heavy", C.c_int32), ("n_blocks", C.c_int32), ("block_ptr", _vp),
                ("xcd_order", _vp), ("xcd_stride_heavy", C.c_int32), ("xcd_stride_light", C.c_int32)]


class ConcatJob(C.Structure):      # stag_concat_job: graph._ConcatJobs writes these records with numpy (4 int64 per job)
    _fields_ = [("src", _vp), ("dst", _vp), ("count", C.c_int64), ("add", C.c_int32), ("kind", C.c_int32)]


class StagHipError(RuntimeError):
    pass


_lib = None


def _hip_runtimes_mapped():
    names = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    names.add(line.split()[-1])
    except OSError:
        pass
    return names


def lib():
    """Load libstag_hip.so once; raise if it is missing (build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise StagHipError(
            f"{_SO} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C stag_amd/csrc). "
            "stag_amd has no CPU fallback.")
    _lib = bind(_SO)
    return _lib


def bind(path):
    """A build of the library with every prototype declared (lib() for the shipped one; the A/B tools load variants)."""
    l = C.CDLL(path)
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise StagHipError(f"two HIP runtimes are mapped into this process: {sorted(rts)}")
    ip = C.POINTER(C.c_int32)
    l.stag_abi_version.restype = C.c_int
    l.stag_strerror.restype = C.c_char_p
    l.stag_strerror.argtypes = [C.c_int]
    l.stag_plan_count.argtypes = [_vp, C.c_int32, C.c_int32, ip, ip, ip, ip]
    l.stag_plan_fill.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp]
    l.stag_plan_blocks.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, ip]
    l.stag_plan_workspace_bytes.restype = C.c_size_t
    l.stag_plan_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    l.stag_plan_xcd.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _vp, _vp]
    l.stag_concat_jobs.argtypes = [_vp, _vp, C.c_int32, C.c_int64, _vp]
    l.stag_stripe_locality.argtypes = [_vp, _vp, C.c_int32, C.c_int64, C.POINTER(C.c_int64), _vp, _vp]
    l.stag_plan_blocks_xcd.argtypes = [_vp, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, ip]
    l.stag_plan_xcd_ranges.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp]
    l.stag_plan_blocks_xcd_ranges.argtypes = [_vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, ip]
    l.stag_plan_xcd_device_count_ranges.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp,
                                                    C.c_size_t, _vp]
    l.stag_plan_xcd_ints.restype = C.c_size_t
    l.stag_plan_xcd_ints.argtypes = [C.c_int32, C.c_int32]
    l.stag_plan_xcd_fine.restype = C.c_int32
    l.stag_plan_xcd_fine.argtypes = [C.c_int32]
    l.stag_plan_xcd_device_workspace_bytes.restype = C.c_size_t
    l.stag_plan_xcd_device_workspace_bytes.argtypes = [C.c_int32]
    l.stag_plan_xcd_device_count.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _vp, _vp, C.c_size_t, _vp]
    l.stag_plan_xcd_device_fill.argtypes = [_vp, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_size_t, _vp]
    l.stag_plan_device_workspace_bytes.restype = C.c_size_t
    l.stag_plan_device_workspace_bytes.argtypes = [C.c_int32]
    l.stag_plan_device.argtypes = [_vp, C.c_int32, C.c_int64, C.c_int32, _vp, C.c_int64, _vp, _vp, C.c_int64,
                                   C.POINTER(C.c_int32), _vp, C.c_size_t, _vp]
    l.stag_csr_build_workspace_bytes.restype = C.c_size_t
    l.stag_csr_build_workspace_bytes.argtypes = [C.c_int32, C.c_int64]
    l.stag_csr_build.argtypes = [_vp, _vp, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp, _vp,
                                 C.c_size_t, _vp]
    l.stag_philox_raw.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, _vp, _vp]
    l.stag_normal_tables.argtypes = [_vp, _vp, _vp, _vp]
    l.stag_agg_fwd.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, C.c_int64, C.c_int32,
                               C.POINTER(NoiseSpec), C.c_int32, _vp, _vp, _vp, C.c_int64, _vp, _vp]
    l.stag_noise_materialize.argtypes = [C.POINTER(Csr), C.POINTER(Plan), C.POINTER(NoiseSpec), C.c_int32,
                                         _vp, C.c_int64, _vp, _vp]
    l.stag_agg_bwd_w.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, C.c_int64, _vp, C.c_int64, C.c_int32,
                                 _vp, C.POINTER(NoiseSpec), C.c_int32, _vp, _vp, C.c_int64, _vp]
    l.stag_agg_fwd_mc.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, C.c_int64, C.c_int32,
                                  C.POINTER(NoiseSpec), C.c_int32, C.c_int64, C.c_int32, _vp, _vp, _vp,
                                  C.c_int64, C.c_int64, _vp]
    l.stag_agg_bwd.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, C.c_int64, C.c_int32,
                               C.POINTER(NoiseSpec), _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]
    l.stag_agg_bwd_edge.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, C.c_int64, C.c_int32,
                                    C.POINTER(NoiseSpec), _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp, _vp]
    l.stag_agg_bwd_dp_workspace_bytes.restype = C.c_size_t
    l.stag_agg_bwd_dp_workspace_bytes.argtypes = [C.c_int64, C.c_int32]
    l.stag_agg_bwd_dp.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, C.c_int64, C.c_int32, C.POINTER(NoiseSpec),
                                  _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp, _vp, C.c_size_t, _vp]
    l.stag_amort_workspace_bytes.restype = C.c_size_t
    l.stag_amort_workspace_bytes.argtypes = [C.c_int32]
    l.stag_head_dot_fwd.argtypes = [_vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp]
    l.stag_head_dot_bwd.argtypes = [_vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp,
                                    C.c_int64, _vp, _vp, C.c_size_t, _vp]
    l.stag_node_project_fwd.argtypes = [_vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp]
    l.stag_node_project_bwd.argtypes = [_vp, C.c_int64, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int64,
                                        _vp, _vp, _vp, C.c_size_t, _vp]
    l.stag_edge_mlp_fwd.argtypes = [_vp, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp]
    l.stag_edge_mlp_bwd.argtypes = [_vp, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, _vp,
                                    _vp, _vp, _vp, C.c_size_t, _vp]
    l.stag_normal_kl_fwd.argtypes = [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_size_t, _vp]
    l.stag_normal_kl_bwd.argtypes = [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]
    l.stag_coldot_workspace_bytes.restype = C.c_size_t
    l.stag_coldot_workspace_bytes.argtypes = [C.c_int32]
    l.stag_coldot.argtypes = [_vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp,
                              C.c_size_t, _vp]
    l.stag_segment_reduce.argtypes = [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, C.c_int32, _vp,
                                      C.c_int64, _vp]
    l.stag_gat_workspace_bytes.restype = C.c_size_t
    l.stag_gat_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    l.stag_gat_fwd.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, _vp, _vp, C.c_int32, C.c_int32,
                               C.c_float, C.POINTER(NoiseSpec), _vp, C.POINTER(GatDrop), _vp, _vp, _vp]
    l.stag_gat_attn.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, _vp, C.c_int32, C.c_float,
                                C.POINTER(NoiseSpec), _vp, _vp, _vp, _vp]
    l.stag_gat_bwd_edge.argtypes = [C.POINTER(Csr), C.POINTER(Plan), _vp, _vp, _vp, _vp, _vp, _vp,
                                    C.c_int32, C.c_int32, C.c_float, C.POINTER(NoiseSpec), _vp, _vp,
                                    _vp, _vp, _vp]
    l.stag_gat_bwd_workspace_bytes.restype = C.c_size_t
    l.stag_gat_bwd_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    l.stag_gat_bwd.argtypes = [C.POINTER(Csr), C.POINTER(Plan), C.POINTER(Csr), C.POINTER(Plan), _vp, _vp, _vp,
                               _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_float, C.POINTER(NoiseSpec), _vp,
                               C.POINTER(GatDrop), _vp, _vp, _vp, _vp, _vp, _vp]
    l.stag_gat_bwd_two_pass.argtypes = l.stag_gat_bwd.argtypes
    l.stag_gat_bwd_scratch_bytes.restype = C.c_size_t
    l.stag_gat_bwd_scratch_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32]
    l.stag_comm_unique_id.argtypes = [_vp]
    l.stag_comm_init.argtypes = [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)]
    l.stag_comm_destroy.argtypes = [_vp]
    l.stag_halo_allgather.argtypes = [_vp, _vp, C.c_int64, _vp, _vp]
    l.stag_halo_exchange.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
    l.stag_gat_bwd_dp_workspace_bytes.restype = C.c_size_t
    l.stag_gat_bwd_dp_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    l.stag_gat_bwd_dp.argtypes = [C.POINTER(Csr), C.POINTER(Plan), C.POINTER(Csr), C.POINTER(Plan), _vp, _vp, _vp, _vp, _vp,
                                  _vp, C.c_int32, C.c_int32, C.c_float, C.POINTER(NoiseSpec), _vp, C.POINTER(GatDrop),
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]
    l.stag_halo_exchange_multi.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]
    l.stag_gather_rows.argtypes = [_vp, C.c_int64, _vp, C.c_int64, C.c_int32, _vp, C.c_int64, _vp]
    l.stag_gat_bwd_stages.argtypes = [C.POINTER(Csr), C.POINTER(Plan), C.POINTER(Csr), C.POINTER(Plan), _vp, _vp, _vp,
                                      _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_float, C.POINTER(NoiseSpec), _vp,
                                      C.POINTER(GatDrop), _vp, _vp, _vp, _vp, C.c_int32, _vp]
    if l.stag_abi_version() != 19:
        raise StagHipError("libstag_hip.so ABI version mismatch")
    return l


def check(rc, what):
    if rc != 0:
        raise StagHipError(f"{what}: {lib().stag_strerror(rc).decode()} (rc={rc})")


def require_device(*tensors):
    """Every tensor handed to the library must live on one HIP device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise StagHipError(
                "stag_amd runs on a HIP device only (tensor on %s); there is no CPU path" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise StagHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(device):
    """hipStream_t (as an int) of torch's current stream on `device`; the raw getter skips the
    Stream object (4 us -> 0.3 us of the ~25 us a call costs on the host)."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """`with torch.cuda.device(dev)` only when `dev` is not already current (the common case)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False
