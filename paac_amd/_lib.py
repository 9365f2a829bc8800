"""ctypes binding of libpaac_hip.so (include/paac_hip.h).

The product path is HIP-only: if the library is missing or fails to load this module raises -- there
is no CPU fallback (the CPU restatement lives in oracle/ and is test infrastructure).
"""
import ctypes
import os
from ctypes import POINTER, c_char, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAAC_HIP_LIB") or os.path.join(HERE, "libpaac_hip.so")   # PAAC_HIP_LIB: diagnostic builds

MAX_TENSORS = 12
PROF_FAMILIES = 16
ARCH_NIPS, ARCH_NATURE, ARCH_USER = 0, 1, 2
CLIP_IGNORE, CLIP_GLOBAL = 0, 1


class Layout(ctypes.Structure):
    _fields_ = [("num_tensors", c_int32),
                ("total", c_int64),
                ("total_unpadded", c_int64),
                ("offset", c_int64 * MAX_TENSORS),
                ("size", c_int64 * MAX_TENSORS),
                ("rank", c_int32 * MAX_TENSORS),
                ("shape", (c_int32 * 4) * MAX_TENSORS),
                ("name", (c_char * 32) * MAX_TENSORS)]


class Cfg(ctypes.Structure):
    _fields_ = [("device", c_int32), ("arch", c_int32), ("num_actions", c_int32), ("max_batch", c_int32)]


class Returns(ctypes.Structure):
    """paac_returns (include/paac_hip.h): the rollout records the fused returns + backward entry reads."""
    _fields_ = [("v_boot", c_void_p), ("rewards", c_void_p), ("masks", c_void_p), ("values", c_void_p),
                ("T", c_int32), ("N", c_int32), ("gamma", ctypes.c_double), ("y_out", c_void_p), ("adv_out", c_void_p),
                ("global_step_dev", c_void_p), ("increment", c_int64), ("initial_lr", ctypes.c_double),
                ("lr_annealing_steps", c_int64), ("lr_out_dev", c_void_p), ("tick_dev", c_void_p), ("tick_inc", c_uint64)]


class PaacHipError(RuntimeError):
    pass


_SIGNATURES = {
    "paac_last_error": (c_char_p, []),
    "paac_version": (c_int, []),
    "paac_param_layout": (c_int, [c_int, c_int, POINTER(Layout)]),
    "paac_create": (c_int, [POINTER(Cfg), POINTER(c_void_p)]),
    "paac_destroy": (c_int, [c_void_p]),
    "paac_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "paac_forward_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_uint64, c_void_p, c_uint64,
                                    c_uint32, c_void_p, c_void_p]),
    "paac_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "paac_train_forward_trunk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "paac_keep_next_forward": (c_int, [c_void_p, c_int]),
    "paac_bootstrap_forward_trunk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "paac_loss_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                   c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "paac_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p]),
    "paac_set_managed_weights": (c_int, [c_void_p, c_int]),
    "paac_loss_backward_returns": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Returns), c_int, c_float, c_void_p,
                                           c_void_p, c_int, c_int, c_void_p]),
    "paac_grad_stats": (c_int, [c_void_p, c_void_p, c_void_p]),
    "paac_clip_rmsprop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float,
                                  c_float, c_float, c_float, c_int, c_float, c_void_p, c_void_p]),
    "paac_lr_step": (c_int, [c_void_p, c_int64, c_double, c_int64, c_void_p, c_void_p]),
    "paac_nstep_returns": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p,
                                   c_void_p, c_void_p]),
    "paac_nstep_returns_tick": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p,
                                        c_void_p, c_void_p, c_int64, c_double, c_int64, c_void_p, c_void_p, c_uint64,
                                        c_void_p]),
    "paac_sample_mt_scratch_bytes": (c_int64, [c_int, c_int]),
    "paac_sample_mt": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "paac_sample_philox": (c_int, [c_void_p, c_int, c_int, c_uint64, c_void_p, c_uint64, c_uint32, c_void_p, c_void_p]),
    "paac_counter_add": (c_int, [c_void_p, c_uint64, c_void_p]),
    "paac_preprocess_stack": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "paac_synth_reset": (c_int, [c_uint64, c_uint32, c_int, c_void_p, c_void_p, c_void_p]),
    "paac_synth_step": (c_int, [c_uint64, c_uint32, c_int, c_void_p, c_uint32, c_void_p, c_uint64, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "paac_act_step_mt": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                 c_uint32, c_uint32, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "paac_walk_scratch_bytes": (c_int64, [c_int, c_int]),
    "paac_debug_report_zero": (c_int, [c_int]),
    "paac_sample_mt_synth_step": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_uint64, c_uint32, c_int, c_uint32, c_void_p,
                                          c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "paac_forward_sample_synth_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_uint64, c_void_p,
                                               c_uint64, c_uint32, c_void_p, c_uint64, c_uint32, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "paac_graph_begin": (c_int, [c_void_p]),
    "paac_graph_end": (c_int, [c_void_p, POINTER(c_void_p)]),
    "paac_graph_launch": (c_int, [c_void_p, c_void_p]),
    "paac_graph_destroy": (c_int, [c_void_p]),
    "paac_debug_activation": (c_int64, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "paac_debug_set_tuning": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "paac_debug_get_tuning": (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "paac_debug_clock": (c_int, [c_void_p, c_void_p]),
    "paac_prof_enable": (c_int, [c_void_p, c_int]),
    "paac_prof_read": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_float), c_int]),
    "paac_prof_read_mix": (c_int, [c_void_p, POINTER(c_int32), c_int]),
    "paac_prof_name": (c_char_p, [c_int]),
    "paac_user_arch": (c_int, [POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "paac_user_arch_layers": (c_int, [POINTER(c_int32), POINTER(c_int32)]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


def use_library(path):
    """Choose the library file this process loads (a build for a user architecture, paac_amd.build.build_user_arch).  One
    library per process: it must be chosen before the first call into it."""
    global LIB_PATH
    if _lib is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
        raise PaacHipError("%s is already loaded: a process holds ONE geometry besides Nature -- choose the user "
                           "architecture before anything touches the library" % LIB_PATH)
    LIB_PATH = path


def user_arch():
    """-> (convs [(filters, size, stride), ...], fc width) compiled into the loaded library, or None (stock library)."""
    nconv, filters, fc = c_int32(), (c_int32 * 3)(), c_int32()
    if not load().paac_user_arch(ctypes.byref(nconv), filters, ctypes.byref(fc)):
        return None
    sizes, strides = (c_int32 * 3)(), (c_int32 * 3)()
    load().paac_user_arch_layers(sizes, strides)
    return [(int(filters[i]), int(sizes[i]), int(strides[i])) for i in range(nconv.value)], int(fc.value)


def load():
    """Load libpaac_hip.so (once) and attach the prototypes.  Raises PaacHipError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PaacHipError("%s not found: build it with `python -m paac_amd.build` (hipcc, gfx950). "
                           "paac_amd has no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; the library must bind to THAT runtime instance (device pointers and
    # streams are torch's), so torch is imported first and the loader reuses the already-loaded SONAME.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise PaacHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc is not None and rc < 0:
        msg = load().paac_last_error()
        raise PaacHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
    return rc


def param_layout(arch, num_actions):
    """-> list of dicts (name, shape, offset, size) + totals, straight from the library."""
    lay = Layout()
    check(load().paac_param_layout(int(arch), int(num_actions), ctypes.byref(lay)), "paac_param_layout")
    tensors = []
    for i in range(lay.num_tensors):
        shape = tuple(int(lay.shape[i][d]) for d in range(lay.rank[i]))
        tensors.append(dict(name=bytes(lay.name[i]).split(b"\0", 1)[0].decode(), shape=shape,
                            offset=int(lay.offset[i]), size=int(lay.size[i])))
    return dict(tensors=tensors, total=int(lay.total), total_unpadded=int(lay.total_unpadded))
